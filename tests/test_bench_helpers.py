"""bench.py's host-side pieces that need no GPU: the CPUs the container really grants, the image check against the reference
build's rows, and the rule that attaches a PMC summary to a bench line only for the launch shape it was collected on."""
import importlib.util
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_usable_cpus_is_at_least_one_and_says_how():
    n, how = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1) and ("affinity" in how or "quota" in how)
    from oracle import binding
    assert binding.usable_threads() == n


def test_image_check_compares_whole_rows_bit_for_bit():
    gold = bench.golden_rows("outdoor")
    assert gold is not None and bench.golden_rows("no such view") is None and bench.golden_rows(None) is None
    seeds, rows, want = gold
    assert len(seeds) == 8 and len(rows) == 16 and want.shape == (len(rows), 1920, 3)
    img = np.zeros((1080, 1920, 3), np.float32)
    img[np.asarray(rows)] = want
    ok = bench.compare_golden(img.reshape(-1), gold, 1920)
    assert ok["bit_identical"] and ok["pixels"] == 16 * 1920 and ok["pixels_differing"] == 0 and ok["max_rel_err"] == 0.0
    img[int(rows[2]), 77, 1] = np.nextafter(img[int(rows[2]), 77, 1], np.float32(2))   # one ulp in one channel of one pixel
    bad = bench.compare_golden(img.reshape(-1), gold, 1920)
    assert not bad["bit_identical"] and bad["pixels_differing"] == 1 and 0 < bad["max_rel_err"] < 1e-6


def test_pmc_summary_is_attached_only_to_its_own_launch_shape():
    entries = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["entries"]
    by = {e["config"]: e for e in entries}
    assert set(by) >= {1, 2, 3, 4}
    for c, e in by.items():
        tree, group, bvh, pool = e["kernel_info"]
        info = {"tree": tree, "group": group, "bvh": bool(bvh), "pool": pool}
        got = bench.pmc_entry(c, info, e["passes_per_launch"], e["samples_per_launch"], 0)
        assert got is not None and got["config"] == c and got["limits"]["limiter"]
        assert bench.pmc_entry(c, info, e["passes_per_launch"] + 1, e["samples_per_launch"], 0) is None
        assert bench.pmc_entry(c, info, e["passes_per_launch"], e["samples_per_launch"], 8) is None      # another kernel variant
    # configs 2 and 3 run the same instantiation at the same launch shape: the configuration decides
    assert bench.pmc_entry(3, {"tree": 17, "group": 1, "bvh": False, "pool": by[3]["kernel_info"][3]}, 256, by[2]["samples_per_launch"], 0)["config"] == 3


def _canned_full_line():
    """A full result object of the N = 1 default run as round 5 printed it (20 KB: the line the driver could not parse)."""
    return json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_driver_format.json")).read().strip().splitlines()[-1])


def test_stdout_line_stays_inside_the_drivers_window():
    """The driver keeps about 8 KB of stdout; round 5's line was 20 KB and `BENCH_r05.parsed` is null.  The compact line built
    from that very object is well under 4000 bytes and still carries the contract's keys, `roofline` and `cpu_baseline`."""
    full = _canned_full_line()
    assert len(json.dumps(full)) > 15000
    text = bench.compact_line(full)
    assert len(text.encode()) < bench.LINE_LIMIT == 4000 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k], k
    assert set(line["config"]) == {"workload", "baseline_config", "passes_per_step", "spp_timed", "parallelism"}
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] == full["roofline"]["traffic"] and r["nearest_ceiling"] == {"resource": "valu_issue", "frac": full["roofline"]["nearest_ceiling"]["frac"]}
    c = line["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] == 16 and c["value"] == full["cpu_baseline"]["value"] and c["restatement_value"] and c["sample"]
    assert [o["baseline_config"] for o in line["other_configs"]] == [1, 3, 4] and all(o["bit_identical"] and o["frac"] > 0 for o in line["other_configs"])
    assert line["image_check"] == {"pixels": 30720, "passes": 8, "bit_identical": True} and line["collective"] == {"ranks": 1, "backend": None}
    assert "dropped" not in line and "limits" not in text and "note" not in text


def test_an_eight_gpu_line_stays_inside_the_window_too():
    """The N = 8 shape: per-rank lists, the group check with three transports and their image checks, a failed-RCCL note — all of
    it in the detail object, a summary of it in the line."""
    full = _canned_full_line()
    full.pop("other_configs")
    full.pop("cpu_baseline")
    full["n_gpus"] = 8
    full["rccl_ranks"] = 8
    full["collective"] = {"backend": "nccl", "called_from": "torch.distributed", "ranks": 8, "devices": list(range(8)),
                          "readback_ms": [1.234] * 6, "launcher": "torch.distributed.run", "rccl_failed": "x" * 900, "note": "y" * 300}
    full["per_rank"] = {"kernel_ms": [210.0 + i for i in range(8)], "reduce_ms": [[1.0 + i, 2.0] * 3 for i in range(8)],
                        "device_names": ["AMD Instinct MI355X"], "note": "z" * 300}
    chk = dict(full["image_check"])
    full["group_check"] = {"members": 8, "devices": list(range(8)), "peer_status": [0] + [1] * 7, "transport": {"transport": 1, "name": "rccl-sendrecv", "backend": "rccl", "detail": "d" * 400},
                           "transports_timed": {n: {"gather_ms": 0.5} for n in ("rccl-sendrecv", "rccl-reduce", "peer-copy")},
                           "transports_checked": {n: dict(chk) for n in ("rccl-sendrecv", "rccl-reduce", "peer-copy")},
                           "render_ms": 12.0, "gather_ms": 0.6, "value": 40000.0, "image_check": chk, "ran_in": "r" * 200}
    text = bench.compact_line(full)
    assert len(text.encode()) < 4000
    line = json.loads(text)
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["collective"]["backend"] == "nccl" and len(line["collective"]["rccl_failed"]) <= 120
    assert line["per_rank"] == {"kernel_ms_min": 210.0, "kernel_ms_max": 217.0, "reduce_ms_max": 8.0}
    g = line["group_check"]
    assert g["members"] == 8 and g["transport"] == "rccl-sendrecv" and g["bit_identical"] and g["transports_bit_identical"] == {"rccl-sendrecv": True, "rccl-reduce": True, "peer-copy": True}
    assert "dropped" not in line


def test_a_line_that_would_overflow_gives_up_optional_objects_not_the_contract():
    full = _canned_full_line()
    full["other_configs"] = [dict(o, baseline_config=i) for i in range(150) for o in full["other_configs"][:1]]
    text = bench.compact_line(full)
    line = json.loads(text)
    assert len(text.encode()) < 4000 and "other_configs" in line["dropped"] and line["value"] == full["value"] and line["roofline"]["frac"] and line["cpu_baseline"]["value"]


def test_emit_writes_the_full_object_beside_the_line(tmp_path, capsys):
    full = _canned_full_line()
    p = str(tmp_path / "d.json")
    bench.emit(full, p)
    so = capsys.readouterr()
    lines = [ln for ln in so.out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and so.out.rstrip().endswith(lines[0]) and len(lines[0]) < 4000
    d = json.load(open(p))
    assert d["roofline"]["limits"] and d["other_configs"][0]["roofline"]["limits"] and json.loads(lines[0])["detail"] == "d.json"
    assert "full result object" in so.err
