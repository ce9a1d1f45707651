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
    assert set(by) == {1, 2, 3, 4}
    for c, e in by.items():
        tree, group, bvh, pool = e["kernel_info"]
        info = {"tree": tree, "group": group, "bvh": bool(bvh), "pool": pool}
        got = bench.pmc_entry(c, info, e["passes_per_launch"], e["samples_per_launch"], 0)
        assert got is not None and got["config"] == c and got["limits"]["limiter"]
        assert bench.pmc_entry(c, info, e["passes_per_launch"] + 1, e["samples_per_launch"], 0) is None
        assert bench.pmc_entry(c, info, e["passes_per_launch"], e["samples_per_launch"], 8) is None      # another kernel variant
    # configs 2 and 3 run the same instantiation at the same launch shape: the configuration decides
    assert bench.pmc_entry(3, {"tree": 17, "group": 1, "bvh": False, "pool": 56}, 256, by[2]["samples_per_launch"], 0)["config"] == 3
