#!/usr/bin/env python3
"""bench.py — Msamples/s of the path-tracing hot path on BASELINE.json's workloads.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts one child process per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --config {1,2,3,4}                     (default 2: the headline, the line the driver records)
    python bench.py --group N                              (one process, N GPUs behind one context: what a Chunky JVM binds)

Workloads (`--config`, named in config.workload):
  2  BASELINE configs[2] — the synthetic 32x32-chunk outdoor world at 1920x1080, draw-depth 256, sun + sky, seeds from
     java.util.Random(0): the scene the metric is quoted on.  A step is 256 passes (one launch, the most a launch carries);
     the default 4 steps are the 1024 spp the configuration is quoted at.
  1  configs[1] — the reference's benchmark/OpenCL_test city (tests/golden fixture), 1920x1080, 64 passes per step (4 = its 256 spp).
  3  configs[3] — the indoor emitter room, sun flag 0, 1920x1080, 256 passes per step (the reference's light transport).
  4  configs[4] — 32x32-chunk world + 100 000 world / 5 000 actor triangles at its stated 3840x2160, 16 passes per step.
Scene upload is outside the timed region; the framebuffer lives in HBM.

The timed region follows the reference's loop (OpenClPathTracingRenderer.java:95-184): passes accumulate as a float running
mean; every 1024 spp (its merge interval) the buffer is read back and the mean starts again from zero.  A read-back is the
SAME at every N, so that a scaling curve divides like by like: whatever exchange assembles the image on rank 0 / member 0
(N = 1: none; N > 1: ONE RCCL reduce of the per-rank framebuffers; --group: the library's one RCCL exchange), then rank 0
copies the image into pinned host memory (the reference's clEnqueueReadBuffer, :164-166) — all inside the timed region and
inside `value`; `value_hbm_resident` is the same samples over the timed region minus the read-backs (nothing leaves HBM).
N > 1: one process per GPU, the scene replicated, the image cut into 16x16-pixel blocks dealt round-robin
(chunky_render_set_shard), no collective on the data path.  Total work is fixed as N grows => "scaling": "strong".

STDOUT carries ONE line, last: a summary of the result object under 4000 bytes (compact_line: the contract's keys, roofline,
cpu_baseline, image_check, other_configs in brief; the driver keeps about 8 KB of stdout, and round 5's 20 KB line could not be
parsed).  The FULL object described below goes to bench_detail.json beside this script (--detail PATH) and to stderr.
--config 5 is not a BASELINE configuration: a 128x128-chunk world whose trees do not fit the 256 MiB Infinity Cache, checked
against rows the CPU oracle renders in the same run (how the design degrades beyond the caches).

The full result object carries:
  roofline     — the contract figure: achieved = ALGORITHMIC bytes per sample of the reference's
                 access stream (counted by the CPU oracle on a row-sample of this same view) x samples per launch /
                 mean launch duration from HIP events on the launch stream, against the 8 TB/s HBM peak.  It is a
                 work-rate convention: the scene is cache-resident, so physical HBM traffic is a few % of it.  `traffic`
                 (PMC FETCH_SIZE x2 + WRITE_SIZE per launch), `valu` and `limits` (VALU lane fraction, L1 tag look-ups per
                 cycle, L2 request bandwidth, wait share: the real limiter) come from the committed PMC summary named in
                 `pmc_source` and are attached only when it was collected for the kernel / launch shape of this run.
  other_configs — N = 1, default configuration only: short warm legs (3 steps) of BASELINE configs[1], [3], [4] on the same
                 box in the same run, each with value, launch_ms, image_check against the reference's rows and the contract
                 roofline, so that the driver's record carries every configuration, not only the headline.
  end_to_end   — N = 1 (and --group): the same spp through chunky_render_run_ex with merge interval 1024 — the reference's
                 whole loop incl. the climb to full-size launches, every read-back into host memory and the double-precision
                 merge into Chunky's sample buffer — cold (first run on a fresh target) and warm (second run).
  image_check  — after the timed region every rank renders 4 passes of the same view, the read-back collective runs once
                 more, and rank 0 compares whole image rows with tests/golden/timed_rows.npz (rendered by the REFERENCE
                 build): the first multi-GPU run says by itself whether the reduced image is the reference's.
  group_check  — N > 1: a child process of rank 0 (watchdog: 240 s) then opens ALL N GPUs behind one context
                 (chunky_group_create, the in-process path a JVM binds: RCCL called from C++), renders one step, and reports
                 per-member peer-access status, the transport and the gather milliseconds of every transport (RCCL send/recv,
                 RCCL reduce, peer copies), and the same image check; a failure or a hang is reported in the line, it does
                 not fail the bench.
  per_rank     — N > 1: every rank's kernel milliseconds (HIP events) and the milliseconds of its read-back reduces.
  cpu_baseline — where oracle/_ref travelled with the tree: the reference kernel itself (rayTracer.cl compiled for x86-64 by
                 clang in the build container: kind "reference") timed on this box's host cores on a bounded sample of the same
                 view (rank 0, N = 1), with the C restatement's rate (oracle/port.c, workers pinned) on the same sample beside it;
                 else the restatement alone (kind "port").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
MERGE_INTERVAL = 1024  # OpenClPathTracingRenderer.java:158

LINE_LIMIT = 4000  # bytes of the final stdout line: the driver keeps about 8 KB of stdout, and round 5's 20 KB line was lost to it


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "\u2026"


def compact_line(out):
    """The ONE line the driver parses, from the full result object: the contract's keys and the few figures a reader needs,
    nothing that grows (notes, limits, timelines, per-rank lists, transports: those are in bench_detail.json and on stderr).
    Always shorter than LINE_LIMIT — optional objects are dropped, in order of importance, before that could fail."""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}

    cfg = out.get("config", {})
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    line["metric"] = _short(line["metric"], 160)
    line["config"] = {"workload": _short(cfg.get("workload", ""), 200), **pick(cfg, ("baseline_config", "passes_per_step", "spp_timed")),
                      "parallelism": _short(cfg.get("parallelism", ""), 200)}
    roof = out.get("roofline") or {}
    r = pick(roof, ("bound", "achieved", "peak", "unit", "frac", "traffic", "physical_frac", "physical_bound", "algorithmic_bytes_per_sample",
                    "kernel", "launch_ms", "samples_per_launch"))
    r.setdefault("traffic", None)
    if roof.get("nearest_ceiling"):
        r["nearest_ceiling"] = pick(roof["nearest_ceiling"], ("resource", "frac"))
    if roof.get("pmc_source"):
        r["pmc_source"] = _short(roof["pmc_source"], 80)
    line["roofline"] = r
    cpu = out.get("cpu_baseline")
    if cpu:
        c = pick(cpu, ("value", "unit", "cores", "kind"))
        if isinstance(cpu.get("restatement"), dict):
            c["restatement_value"] = cpu["restatement"].get("value")
        c["sample"] = _short(cpu.get("sample", ""), 150)
        line["cpu_baseline"] = c
    optional = []  # (key, value) in the order they are given up when the line would grow past the limit: last first
    if out.get("image_check"):
        optional.append(("image_check", pick(out["image_check"], ("pixels", "passes", "bit_identical"))))
    optional.append(("rccl_ranks", out.get("rccl_ranks", 0)))
    col = out.get("collective") or {}
    c = pick(col, ("backend", "ranks"))
    c.setdefault("backend", None)
    if col.get("rccl_failed"):
        c["rccl_failed"] = _short(col["rccl_failed"], 120)
    if isinstance(col.get("transport"), dict):
        c["transport"] = col["transport"].get("name")
    optional.append(("collective", c))
    if out.get("other_configs"):
        legs = []
        for leg in out["other_configs"]:
            e = pick(leg, ("baseline_config", "value", "ms_per_step"))
            if leg.get("roofline"):
                e["frac"] = leg["roofline"].get("frac")
            if leg.get("image_check"):
                e["bit_identical"] = leg["image_check"].get("bit_identical")
            if leg.get("error"):
                e["error"] = _short(leg["error"], 100)
            legs.append(e)
        optional.append(("other_configs", legs))
    for k in ("value_hbm_resident", "gpu_over_cpu", "emulated_world"):
        if out.get(k) is not None:
            optional.append((k, out[k]))
    if isinstance(out.get("end_to_end"), dict):
        optional.append(("end_to_end", pick(out["end_to_end"], ("value", "cold_value", "spp", "error"))))
    if out.get("per_rank"):
        km = out["per_rank"].get("kernel_ms") or [0.0]
        rm = [x for xs in out["per_rank"].get("reduce_ms", []) for x in xs] or [0.0]
        optional.append(("per_rank", {"kernel_ms_min": min(km), "kernel_ms_max": max(km), "reduce_ms_max": max(rm)}))
    if isinstance(out.get("group_check"), dict):
        g = out["group_check"]
        e = pick(g, ("members", "value", "render_ms", "gather_ms"))
        if isinstance(g.get("transport"), dict):
            e["transport"] = g["transport"].get("name")
        if isinstance(g.get("image_check"), dict):
            e["bit_identical"] = g["image_check"].get("bit_identical")
        if isinstance(g.get("transports_checked"), dict):
            e["transports_bit_identical"] = {k: v.get("bit_identical") for k, v in g["transports_checked"].items() if isinstance(v, dict)}
        if g.get("error"):
            e["error"] = _short(g["error"], 120)
        optional.append(("group_check", e))
    for k in ("extension_behind_cull", "extension_emitter_nee"):
        if isinstance(out.get(k), dict):
            e = pick(out[k], ("value",))
            if isinstance(out[k].get("image_check"), dict):
                e["bit_identical"] = out[k]["image_check"].get("bit_identical")
            optional.append((k, e))
    if isinstance(out.get("bigworld"), dict):
        optional.append(("bigworld", pick(out["bigworld"], ("value", "bit_identical", "error"))))
    optional.append(("detail", out.get("detail_file", "bench_detail.json")))
    for k, v in optional:
        line[k] = v
    text = json.dumps(line, separators=(",", ":"))
    while len(text.encode()) >= LINE_LIMIT and optional:
        k, _ = optional.pop()
        line.pop(k, None)
        line["dropped"] = line.get("dropped", []) + [k]
        text = json.dumps(line, separators=(",", ":"))
    return text


def emit(out, detail_path=""):
    """Full object -> the detail file (default: bench_detail.json beside this script, + gpurun_out/ when that exists) and stderr;
    compact line -> stdout, LAST."""
    paths = [detail_path] if detail_path else [os.path.join(d, "bench_detail.json") for d in (ROOT, os.path.join(ROOT, "gpurun_out")) if os.path.isdir(d)]
    out["detail_file"] = os.path.basename(paths[0]) if paths else None
    full = json.dumps(out, indent=1)
    for p in paths:
        try:
            with open(p, "w") as f:
                f.write(full + "\n")
        except OSError as e:
            print(f"bench.py: could not write {p}: {e}", file=sys.stderr)
    print("bench.py: full result object (also in bench_detail.json):\n" + json.dumps(out), file=sys.stderr, flush=True)
    text = compact_line(out)
    assert len(text.encode()) < LINE_LIMIT, len(text)
    print(text, flush=True)


def workload(config: int, args):
    """(scene, passes per step, name of its rows in tests/golden/timed_rows.npz, description, spp of the end-to-end leg)."""
    from chunkyclplugin_amd import scenes
    if config == 1:
        from chunkyclplugin_amd import octree2
        sc = octree2.cached_benchmark_scene(1920, 1080)
        return sc, 64, "city", "BASELINE configs[1]: the reference's benchmark/OpenCL_test city (procedural asset pack: block models by name and properties, hashed 16x16 textures), saved camera", 256
    if config == 3:
        sc = scenes.indoor_room(size=64, width=1920, img_height=1080)
        return sc, 256, "indoor", "BASELINE configs[3]: indoor emitter room (64^3), sun flag 0, the reference's light transport", 1024
    if config == 4:
        base = scenes.cached_outdoor_world(chunks=32, height=256)
        sc = scenes.add_entities(base, 100000, seed=11, actor_tris=5000, region=((40, 90, 40), (470, 170, 470))).with_view(3840, 2160)
        return sc, 16, "entities4k", "BASELINE configs[4]: 32x32-chunk world + 100 000 world / 5 000 actor triangles (entity BVHs)", 64
    if config == 5:
        sc = scenes.cached_big_outdoor_world(width=args.width, img_height=args.height)
        return sc, 64, None, ("NOT a BASELINE configuration: 128x128-chunk outdoor world (2048 x 256 x 2048 blocks, depth-11 octree; its re-laid-out "
                              "tree is 367 MB: beyond the 256 MiB Infinity Cache)"), 256
    sc = scenes.cached_outdoor_world(chunks=args.chunks, height=256, width=args.width, img_height=args.height)
    golden = "outdoor" if (args.chunks, args.width, args.height) == (32, 1920, 1080) else None
    return sc, 256, golden, f"BASELINE configs[2]: synthetic {args.chunks}x{args.chunks}-chunk outdoor world", 1024


def usable_cpus():
    """(CPUs this process can actually keep busy, how that was found): the affinity mask, cut down to the container's CPU-time
    quota (cgroup v2 cpu.max / v1 cpu.cfs_quota_us) — a box that shows 256 CPUs under a 16-CPU quota runs 256 workers SLOWER
    than 16 (profiles/r04_cpu_sweep.jsonl)."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = f"{n} CPUs in the affinity mask"
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except Exception:
            pass
    if quota is not None and quota < n:
        n = max(1, int(math.floor(quota + 1e-9)))
        how = f"cgroup CPU quota of {quota:g} CPUs ({how})"
    return n, how


def sample_rows(height: int, n_rows: int):
    step = max(height // n_rows, 1)
    return list(range(step // 2, height, step))[:n_rows]


def oracle_row_sample(sc, seeds, rows, threads, count: bool, pin: bool = False):
    """Run the CPU oracle on whole rows of the view. Returns (samples, seconds, counters)."""
    import ctypes as C
    from oracle import binding
    port = binding.port()
    h = binding.SceneHandle(sc)
    res = np.zeros(3 * sc.width * sc.height, np.float32)
    gids = (np.asarray(rows, np.int64)[:, None] * sc.width + np.arange(sc.width)[None, :]).reshape(-1).astype(np.int32)
    if count:
        port.counters(enable=True, reset=True)
        port.counters(reset=True)
    port.lib.port_set_pinning.argtypes = [C.c_int]
    port.lib.port_set_pinning.restype = None
    port.lib.port_set_pinning(1 if pin else 0)
    t0 = time.perf_counter()
    port.render_gids(h, seeds, gids, res=res, threads=threads)
    dt = time.perf_counter() - t0
    port.lib.port_set_pinning(0)
    c = port.counters(enable=False, reset=True) if count else None
    return gids.size * len(seeds), dt, c


def golden_rows(name):
    """(seeds, row indices, rows [n, W, 3]) of a timed view in tests/golden/timed_rows.npz, or None."""
    path = os.path.join(ROOT, "tests", "golden", "timed_rows.npz")
    if not name or not os.path.exists(path):
        return None
    g = np.load(path)
    if name + "_res" not in g.files:
        return None
    return g["seeds"], g[name + "_rows"], g[name + "_res"]


def compare_golden(image, gold, width, against="tests/golden/timed_rows.npz (rows rendered by the reference build, oracle/_ref)"):
    """image: float32 [3*W*H] (the read-back); gold from golden_rows.  -> the image_check object."""
    _, rows, want = gold
    img = image.reshape(-1, width, 3)
    got = img[np.asarray(rows, np.int64)]
    same = (np.ascontiguousarray(got).view(np.uint32) == np.ascontiguousarray(want).view(np.uint32)).all(axis=2)
    with np.errstate(all="ignore"):
        rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-6)
    return {"rows": [int(y) for y in rows], "pixels": int(same.size), "pixels_differing": int((~same).sum()),
            "bit_identical": bool(same.all()), "max_rel_err": float(np.nanmax(rel)) if rel.size else 0.0,
            "passes": int(len(gold[0])), "against": against}


def live_golden(sc, n_rows=12, n_passes=4):
    """For a view without reference-rendered rows in tests/golden: whole rows rendered HERE by the CPU oracle (oracle/port.c, the C
    restatement — bit-identical to the reference build on every scene tried), in golden_rows' shape."""
    from oracle import binding
    from chunkyclplugin_amd import native
    port = binding.port()
    seeds = native.java_random_ints(n_passes)
    rows = sample_rows(sc.height, n_rows)
    gids = (np.asarray(rows, np.int64)[:, None] * sc.width + np.arange(sc.width)[None, :]).reshape(-1).astype(np.int32)
    res = port.render_gids(binding.SceneHandle(sc), seeds, gids, threads=usable_cpus()[0])
    return seeds, np.asarray(rows), res.reshape(-1, sc.width, 3)[np.asarray(rows)].copy()


def pmc_entry(config, info, passes_per_launch, samples_per_launch, kernel_variant):
    """The committed PMC summary collected for exactly this launch shape, or None."""
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tp) or kernel_variant != 0:
        return None
    try:
        pm = json.load(open(tp))
        for e in pm.get("entries", [pm]):
            if (e.get("config", 2) == config and e.get("kernel_info") == [info["tree"], info["group"], int(info["bvh"]), info["pool"]]
                    and e.get("passes_per_launch") == passes_per_launch and e.get("samples_per_launch") == samples_per_launch):
                return e
    except Exception:
        pass
    return None


def end_to_end_leg(loader, sc, spp, make_renderer):
    """The reference's whole loop through chunky_render_run_ex, merge interval 1024: cold (fresh target) and warm."""
    r = make_renderer(loader)
    n = 3 * sc.width * sc.height
    out = {"spp": spp, "merge_interval": MERGE_INTERVAL}
    for label in ("cold", "warm"):
        buf = np.zeros(n, np.float64)
        merges, marks = [], []
        r.kernel_time()
        t0 = time.perf_counter()
        got = r.render_ex(buf, 0, spp, merge_interval=MERGE_INTERVAL, progress=lambda s: marks.append((s, time.perf_counter())),
                          merged=lambda s: (merges.append(s), marks.append((-s, time.perf_counter()))))
        dt = time.perf_counter() - t0
        ms, launches = r.kernel_time()
        assert got == spp
        out[("cold_" if label == "cold" else "") + "value"] = round(sc.width * sc.height * spp / dt / 1e6, 3)
        # where the wall time went: [spp after the launch, ms since the start] per launch; a negative spp marks the end of a merge
        line = [[int(s), round((t - t0) * 1e3, 2)] for s, t in marks]
        if len(line) > 24:
            line = line[:12] + [["..."]] + line[-11:]
        out[label] = {"seconds": round(dt, 5), "launches": launches, "kernel_ms": round(ms, 3), "readbacks": len(merges),
                      "host_ms": round(dt * 1e3 - ms, 3), "timeline_ms": line}
        out["readbacks"] = len(merges)
    out["unit"] = "Msamples/s"
    out["includes"] = ("launch-size climb (cold), every read-back into host memory, the double-precision merge into the sample "
                       "buffer (OpenClPathTracingRenderer.java:162-178)")
    out["finite"] = bool(np.isfinite(buf).all())
    r.close()
    return out


def kernel_label(info):
    bvh_tag = ",bvh" if info["bvh"] else (",sorted" if info.get("sorted") else "")  # sorted: full cubes and model blocks in phases of their own
    return ("render_pool<%d,%d%s>+fold_kernel" % (info["tree"], info["pool"], bvh_tag) if info["pool"] >= 0 else
            "render_waves<%d,%d,%s>" % (info["tree"], info["group"], "bvh" if info["bvh"] else "no-bvh"))


def nearest_ceiling(pm, physical_frac):
    """Which hardware resource the kernel sits closest to, from the committed PMC entry of this launch shape: the largest
    used / available fraction among VALU issue slots, L1 tag look-ups, the scalar pipe, L2 requests and physical HBM bytes."""
    if not pm:
        return None
    lim = pm.get("limits", {})
    l1 = lim.get("l1_tag_lookup_frac")
    if l1 is None and lim.get("l1_tag_lookups_per_cycle") is not None:  # (an entry collected before round 5: per cycle, against the measured 1.56)
        l1 = lim["l1_tag_lookups_per_cycle"] * 0.64
    cand = {"valu_issue": lim.get("valu_issue_frac"), "l1_tag_lookups": l1,
            "scalar_pipe": lim.get("scalar_pipe_frac"), "l2_requests": lim.get("l2_request_frac_of_34.5TBps"), "hbm_physical": physical_frac}
    cand = {k: float(v) for k, v in cand.items() if v is not None}
    if not cand:
        return None
    best = max(cand, key=cand.get)
    return {"resource": best, "frac": round(cand[best], 4), "all": {k: round(v, 4) for k, v in sorted(cand.items(), key=lambda kv: -kv[1])},
            "wave_wait_frac": lim.get("wave_wait_frac"), "valu_lane_util": lim.get("valu_lane_util"),
            "note": "used / available per resource over the launch (profiles/pmc_traffic.json): VALU issue slots (not useful lanes: x valu_lane_util), "
                    "L1 look-ups against the measured gather rate (1.56 per cycle and CU), scalar pipe = (SALU + branch) per CU-cycle, L2 requests, "
                    "physical HBM bytes.  No unit saturated + a large wave_wait_frac = latency-bound (paths in flight x latency).  The contract "
                    "`frac` above is a work-rate convention; this is the hardware view"}


def physical_bound(pm, physical_frac):
    """What binds the kernel on the HARDWARE, beside the contract's "bound": "hbm" (a work-rate convention): the fullest resource
    when one is above 80 % of its rate, else "latency" (no unit saturated, waves waiting: paths in flight x latency), or
    "issue+latency" when the VALU issue slots are the fullest unit and more than half used."""
    nc = nearest_ceiling(pm, physical_frac)
    if not nc:
        return None
    if nc["frac"] >= 0.8:
        return nc["resource"]
    return "issue+latency" if nc["resource"] in ("valu_issue", "scalar_pipe") and nc["frac"] >= 0.5 else "latency"


def roofline_object(sc, config, info, launch_ms, launches, samples_per_launch, passes_per_launch, seeds, threads, n_rows, kernel_variant,
                    attach_pmc=True, build_oracle=True):
    """The contract roofline of one measured leg + what the committed PMC entry of this launch shape says."""
    from oracle import binding
    binding.port(build=build_oracle)
    rows = sample_rows(sc.height, n_rows)
    n_s, cal_dt, ctr = oracle_row_sample(sc, seeds[:1], rows, threads, count=True, pin=True)
    bytes_per_sample = binding.algorithmic_bytes(ctr)
    achieved = bytes_per_sample * samples_per_launch / (launch_ms * 1e-3) / 1e9 if (launch_ms > 0 and bytes_per_sample) else 0.0
    pm = pmc_entry(config, info, passes_per_launch, samples_per_launch, kernel_variant) if attach_pmc else None
    traffic = pm.get("hbm_bytes_per_launch") if pm else None
    physical = traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic and launch_ms > 0 else None
    frac = achieved / HBM_PEAK_GBS
    roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(frac, 5), "traffic": traffic,
            "achieved_is": "algorithmic bytes of the reference access stream / launch time (SURVEY 8d), "
                           "not physical HBM traffic: the scene is cache-resident",
            "frac_above_one": bool(frac > 1.0),
            "physical_frac": round(physical, 5) if physical is not None else None,
            "nearest_ceiling": nearest_ceiling(pm, physical),
            "physical_bound": physical_bound(pm, physical),
            "valu": pm.get("valu") if pm else None, "limits": pm.get("limits") if pm else None,
            "pmc_source": pm.get("source") if pm else None, "pmc_collected": pm.get("collected") if pm else None,
            "algorithmic_bytes_per_sample": round(bytes_per_sample, 1) if bytes_per_sample else None,
            "kernel": kernel_label(info), "launches": launches, "launch_ms": round(launch_ms, 4),
            "samples_per_launch": samples_per_launch,
            "counted_on": f"{n_s} samples ({len(rows)} rows of this view, seed 0)"}
    if frac > 1.0:
        roof["frac_note"] = ("above 1: the counted bytes are the REFERENCE's access stream (a root restart per march step); this kernel makes two "
                             "tree reads per step from cache, so it does not perform the counted accesses — see nearest_ceiling for what binds")
    return roof, n_s / max(cal_dt, 1e-6), rows


def other_config_leg(inst, config, args, threads, steps=3):
    """One of BASELINE configs[1], [3], [4] on this GPU, warm: `steps` steps of its own passes per step (one launch each), the
    launch time from HIP events, the image check against the reference's rows and the contract roofline."""
    from chunkyclplugin_amd import native
    from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader
    t_leg = time.perf_counter()
    sc, passes, golden_name, what, _ = workload(config, args)
    n_pix = sc.width * sc.height
    loader = HipSceneLoader(inst)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    try:
        r.set_camera(sc.projector_type, sc.camera)
        r.set_option(native.OPT_KERNEL, args.kernel)
        seeds = native.java_random_ints((steps + 1) * passes)
        r.render_passes(seeds[:passes])          # warm-up at the timed launch shape (staging array, scene derived data)
        r.reset()
        r.kernel_time()
        t0 = time.perf_counter()
        for k in range(steps):
            r.render_passes(seeds[(k + 1) * passes:(k + 2) * passes], first_buffer_spp=k * passes, sync=False)
        r.sync()
        dt = time.perf_counter() - t0
        ms, launches = r.kernel_time()
        info = r.kernel_info()
        launch_ms = ms / max(launches, 1)
        out = {"baseline_config": config, "workload": f"{what}, {sc.width}x{sc.height}, draw-depth 256, {passes} spp per step",
               "value": round(n_pix * steps * passes / dt / 1e6, 3), "unit": "Msamples/s", "steps": steps, "passes_per_step": passes,
               "ms_per_step": round(dt / steps * 1e3, 4), "launch_ms": round(launch_ms, 4), "launches": launches}
        gold = golden_rows(golden_name)
        if gold is not None:
            r.reset()
            r.render_passes(gold[0])
            out["image_check"] = compare_golden(r.read(), gold, sc.width)
        if not args.no_roofline:
            out["roofline"], _, _ = roofline_object(sc, config, info, launch_ms, launches, n_pix * steps * passes // max(launches, 1),
                                                    steps * passes // max(launches, 1), seeds, threads, 12, args.kernel, build_oracle=not args.no_cpu)
        out["leg_seconds"] = round(time.perf_counter() - t_leg, 2)
        return out
    finally:
        r.close()
        loader.close()


def group_leg(devices, sc, seeds, passes, gold, kernel_variant):
    """One process, len(devices) GPUs behind one context (chunky_group_create): one step + the gather, timed on their own."""
    from chunkyclplugin_amd import native
    from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance
    inst = RendererInstance.group(devices)
    try:
        loader = HipSceneLoader(inst)
        loader.load_packed(sc)
        r = HipPathTracingRenderer(loader, sc.width, sc.height)
        r.set_camera(sc.projector_type, sc.camera)
        r.set_option(native.OPT_KERNEL, kernel_variant)
        r.render_passes(seeds[:passes])  # warm-up at the timed launch shape (staging arrays are allocated here)
        r.gather()
        r.reset()
        r.kernel_time()
        t0 = time.perf_counter()
        r.render_passes(seeds[:passes], sync=False)
        r.sync()
        t1 = time.perf_counter()
        r.gather()
        t2 = time.perf_counter()
        ms, launches = r.kernel_time()
        # the same gather once more through every transport the group can use (the first one above is the default's, warm)
        transports, checked = {}, {}
        default_t = inst.transport()
        for t_id in (native.TRANSPORT_RCCL_SENDRECV, native.TRANSPORT_RCCL_REDUCE, native.TRANSPORT_PEER_COPY):
            try:
                inst.set_transport(t_id)
            except native.ChunkyHipError as e:
                transports[RendererInstance.TRANSPORT_NAMES[t_id]] = {"unavailable": str(e)[:200]}
                continue
            r.gather()  # (communicator channels / buffers of this transport come up here)
            ta = time.perf_counter()
            for _ in range(3):
                r.gather()
            transports[RendererInstance.TRANSPORT_NAMES[t_id]] = {"gather_ms": round((time.perf_counter() - ta) / 3 * 1e3, 3)}
            if gold is not None:  # every transport must deliver the reference's image, not only the default one
                r.reset()
                r.render_passes(gold[0])
                chk = compare_golden(r.read(), gold, sc.width)
                checked[RendererInstance.TRANSPORT_NAMES[t_id]] = {k: chk[k] for k in ("pixels", "pixels_differing", "bit_identical")}
                checked[RendererInstance.TRANSPORT_NAMES[t_id]]["transport_after"] = inst.transport()["name"]  # (a fallback mid-way would show here)
        try:
            inst.set_transport(default_t["transport"])
        except native.ChunkyHipError:
            pass
        out = {"members": len(devices), "devices": [int(d) for d in devices], "peer_status": inst.peer_status(),
               "transport": default_t, "transports_timed": transports, "transports_checked": checked, "transport_after": inst.transport(),
               "peer_status_legend": "0 local (member 0's device), 1 direct (peer access enabled: xGMI), 2 staged (no peer access), < 0 = -hipError",
               "passes": passes, "render_ms": round((t1 - t0) * 1e3, 3), "gather_ms": round((t2 - t1) * 1e3, 3),
               "kernel_ms_slowest_member": round(ms, 3),
               "value": round(sc.width * sc.height * passes / (t2 - t0) / 1e6, 3), "unit": "Msamples/s"}
        if gold is not None:
            r.reset()
            r.render_passes(gold[0])
            out["image_check"] = compare_golden(r.read(), gold, sc.width)
        r.close()
        loader.close()
        return out
    finally:
        inst.close()


def group_leg_in_child(devices, args, timeout_s=240.0):
    """group_leg in a process of its own, under a watchdog: the first contact of chunky_group_create (peer access, an RCCL
    communicator over every GPU of the node) with real multi-GPU hardware must not be able to take the bench line with it.  The
    child is started, never exec'ed in place; if it does not answer in time it is ended by its own PID and the line says so."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--group-leg-child", ",".join(str(d) for d in devices), "--config", str(args.config),
           "--kernel", str(args.kernel), "--width", str(args.width), "--height", str(args.height), "--chunks", str(args.chunks)]
    if args.passes > 0:
        cmd += ["--passes", str(args.passes)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK")}
    t0 = time.perf_counter()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    try:
        so, se = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        proc.kill()
        try:
            so, se = proc.communicate(timeout=20)
        except Exception:
            so, se = "", ""
        return {"error": f"no answer within {timeout_s:.0f} s (the child was ended)", "stderr_tail": (se or "")[-600:]}
    lines = [ln for ln in (so or "").splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        return {"error": f"child exited with {proc.returncode}", "stderr_tail": (se or "")[-600:]}
    out = json.loads(lines[-1])
    out["ran_in"] = f"a child process of rank 0 ({time.perf_counter() - t0:.1f} s incl. start-up and scene upload)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=(1, 2, 3, 4, 5), help="BASELINE.json configs[n] (default 2: the headline); 5 = the beyond-cache "
                                                                                   "world (not a BASELINE configuration: how the design degrades when the tree leaves the caches)")
    ap.add_argument("--passes", type=int, default=0, help="passes (spp) per step; 0 = the configuration's (256 for the headline: the most a launch carries)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--chunks", type=int, default=32)
    ap.add_argument("--kernel", type=int, default=0, help="kernel variant (CHUNKY_OPT_KERNEL)")
    ap.add_argument("--tile", type=int, default=0, help="shard tiles: 0 = 16x16-pixel blocks (default), n > 0 = runs of n pixel indices")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the oracle row-sample too: nothing under oracle/ is loaded, built or spawned (profiler runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip the end_to_end / image_check / group_check legs (profiler runs)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short legs of BASELINE configs[1], [3], [4] in the default line")
    ap.add_argument("--group", type=int, default=0, help="N > 0: ONE process, N GPUs behind one context (chunky_group_create); "
                                                         "members share GPU 0 when the box has fewer than N")
    ap.add_argument("--detail", default="", help="where the FULL result object goes (default: bench_detail.json beside this script); stdout "
                                                 "carries only the compact line (< 4000 bytes)")
    ap.add_argument("--dump", default="", help="rank 0 writes the final (reduced) framebuffer to this .npy file")
    ap.add_argument("--emulate-world", type=int, default=0, help="rig: render only rank 0's share of an N-GPU split on one GPU")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for rigs")
    ap.add_argument("--one-device", action="store_true", help="rig: every rank uses GPU 0 (1-GPU box, with --backend gloo)")
    ap.add_argument("--spawn", action="store_true", help="run even N = 1 as a child rank through the self-spawn path (rig)")
    ap.add_argument("--group-leg-child", default="", help="internal: run the group_check leg on these devices (d0,d1,...) and print its JSON object")
    args = ap.parse_args()

    if args.group_leg_child:
        from chunkyclplugin_amd import native
        devices = [int(d) for d in args.group_leg_child.split(",")]
        sc, passes, golden_name, _what, _spp = workload(args.config, args)
        if args.passes > 0:
            passes = args.passes
        seeds = native.java_random_ints(max(passes, MERGE_INTERVAL))
        print(json.dumps(group_leg(devices, sc, seeds, passes, golden_rows(golden_name), args.kernel)), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (args.gpus > 1 or args.spawn) and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process has not touched the GPU (numpy only so far) and never will — it
        # starts one CHILD per GPU with the torch.distributed environment, relays rank 0's JSON line and exits with the
        # children's code.  (Under torch.distributed.run RANK is set and this branch is not taken.)
        from chunkyclplugin_amd import parallel
        sys.exit(parallel.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if world != args.gpus:
        args.gpus = world

    import torch
    import torch.distributed as dist
    from chunkyclplugin_amd import native, parallel
    from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance

    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    on_gpu = args.backend == "nccl"
    rccl_error, ctl = None, None
    if world > 1:
        # CUDA tensors travel over RCCL, CPU tensors over gloo: if the first RCCL collective fails (below), the read-back is staged
        # through the host instead of losing the whole run
        dist.init_process_group(args.backend)
        # the control plane (barriers, timings, rank reports) runs on a gloo group of its own: host-side, no spinning kernel on the
        # GPUs, and it keeps working when the first RCCL collective fails (below) — the read-back is then staged through the host
        ctl = dist.new_group(backend="gloo") if on_gpu else None

    sc, passes, golden_name, what, e2e_spp = workload(args.config, args)
    if args.passes > 0:
        passes = args.passes
    n_pix = sc.width * sc.height
    gold = golden_rows(golden_name)
    if gold is None and args.config == 5 and not args.no_extras and not args.no_roofline:
        gold = live_golden(sc)   # (before the GPU is busy: a few seconds of the host's cores)
    group_devices = None
    if args.group > 0:
        n_dev = RendererInstance.device_count()
        group_devices = list(range(args.group)) if n_dev >= args.group else [0] * args.group
        inst = RendererInstance.group(group_devices)
    else:
        inst = RendererInstance.get(local_rank)
    loader = HipSceneLoader(inst)
    loader.load_packed(sc)

    def make_renderer(ld):
        q = HipPathTracingRenderer(ld, sc.width, sc.height)
        q.set_camera(sc.projector_type, sc.camera)
        q.set_option(native.OPT_KERNEL, args.kernel)
        return q

    r = make_renderer(loader)
    r.set_shard(rank, args.emulate_world or world, args.tile)
    fb = torch.zeros(3 * n_pix, dtype=torch.float32, device="cuda")
    # the read-back lands in a buffer of its own: rank 0's framebuffer must keep holding only rank 0's tiles
    image = torch.zeros_like(fb) if world > 1 else fb
    # ... and from there in pinned host memory on rank 0 (the reference's passBuffer, OpenClPathTracingRenderer.java:164-166)
    host_image = torch.empty(3 * n_pix, dtype=torch.float32, pin_memory=True) if rank == 0 else None
    torch.cuda.synchronize()  # the fill runs on torch's stream, the passes on the library's: order them (chunky_hip.h)
    r.set_device_buffer(fb.data_ptr())

    total_passes = (args.warmup + args.steps) * passes
    seeds = native.java_random_ints(max(total_passes, MERGE_INTERVAL))

    def barrier():
        r.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group=ctl)
            torch.cuda.synchronize()

    reduce_ms, d2h_ms = [], []

    def read_back():
        """What the reference does every merge interval (:162-178) up to the merge: the exchange that assembles the image on
        rank 0 / member 0 — N > 1: the one RCCL reduce of the per-rank framebuffers (into `image`); a group: the library's
        one exchange (chunky_render_gather); N = 1: none — and then, at EVERY N, rank 0's copy of the image into pinned host
        memory (clEnqueueReadBuffer, :164-166).  Both are inside the timed region."""
        r.sync()
        t = time.perf_counter()
        if world > 1:
            image.copy_(fb)
            parallel.reduce_framebuffer(image, dst=0, group=None if on_gpu else ctl)  # (without a usable RCCL: staged through the host)
            torch.cuda.synchronize()
        elif group_devices:
            r.read(out=host_image.numpy())  # ONE call: the library's exchange, then member 0's copy to the host (timed together as the exchange)
        t1 = time.perf_counter()
        if rank == 0 and not group_devices:
            host_image.copy_(image)
            torch.cuda.synchronize()
        reduce_ms.append((t1 - t) * 1e3)
        d2h_ms.append((time.perf_counter() - t1) * 1e3)

    if world > 1:  # the read-back collective once, untimed, on a scratch buffer: communicator and channel set-up are not the path
        if on_gpu:
            # the ranks first agree over gloo (host side) on who sits where: two ranks on one device can never share an RCCL
            # communicator, and if only ONE of them failed inside the first collective the others would wait in it for ever —
            # so that case never reaches RCCL at all
            mine = [None] * world
            dist.all_gather_object(mine, (local_rank, os.uname().nodename), group=ctl)
            collide = len(set(mine)) < world
            ok = 0 if collide else 1
            if collide:
                rccl_error = f"ranks share a device ({sorted(mine)}): no RCCL communicator is possible"
            else:
                try:
                    parallel.reduce_framebuffer(torch.zeros_like(fb), dst=0)
                    torch.cuda.synchronize()
                except Exception as e:  # e.g. IPC refused: say so in the line and stage through the host
                    ok, rccl_error = 0, f"{type(e).__name__}: {str(e)[:300]}"
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctl)   # every rank takes the same path
            if int(flag.item()) == 0:
                on_gpu = False
                parallel.HOST_STAGING = True
        if not on_gpu:
            parallel.reduce_framebuffer(torch.zeros(16), dst=0, group=ctl)

    # several GPUs: the steps between two read-backs are handed over in ONE call, so that a share of the image — whose staged
    # samples fit — runs them as one launch of up to 1024 passes and pays the end-of-launch tail once (the library cuts the call
    # into launches itself; on one GPU a step stays a call of its own: 256 passes, one launch, as profiled)
    batch_steps = world > 1 or bool(group_devices) or bool(args.emulate_world)

    def run_steps(n_steps, first_seed):
        """n_steps steps of `passes` passes, read back + restart of the running mean every MERGE_INTERVAL spp and at the end."""
        spp_in_buffer, used, backs, k = 0, first_seed, 0, 0
        while k < n_steps:
            m = 1
            if batch_steps:  # as many steps as still fit before the next read-back
                while k + m < n_steps and spp_in_buffer + m * passes < MERGE_INTERVAL:
                    m += 1
            r.render_passes(seeds[used:used + m * passes], first_buffer_spp=spp_in_buffer, sync=False)
            used += m * passes
            spp_in_buffer += m * passes
            k += m
            last = k == n_steps
            if spp_in_buffer >= MERGE_INTERVAL or last:
                read_back()
                backs += 1
                if not last:
                    r.reset()
                    spp_in_buffer = 0
        return backs

    if args.warmup:
        run_steps(args.warmup, 0)
        r.reset()
    barrier()
    r.kernel_time()  # discard warmup launches
    reduce_ms.clear()
    d2h_ms.clear()
    t0 = time.perf_counter()
    readbacks = run_steps(args.steps, args.warmup * passes)
    barrier()
    dt = time.perf_counter() - t0
    my_d2h_ms = list(d2h_ms)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
        dt = float(t.item())
    kernel_ms, launches = r.kernel_time()
    info = r.kernel_info()
    my_reduce_ms = list(reduce_ms)
    per_rank = None
    devices_seen = [local_rank]
    if world > 1:
        objs = [None] * world
        dist.all_gather_object(objs, (local_rank, torch.cuda.get_device_name(local_rank), kernel_ms, my_reduce_ms), group=ctl)
        devices_seen = [o[0] for o in objs]
        per_rank = {"kernel_ms": [round(float(o[2]), 3) for o in objs], "reduce_ms": [[round(x, 3) for x in o[3]] for o in objs],
                    "device_names": sorted(set(o[1] for o in objs)),
                    "note": "kernel_ms = sum of this rank's launches in the timed region (HIP events); reduce_ms = host time of each "
                            "read-back (copy + RCCL reduce) incl. waiting for the slowest rank"}
    if rank == 0 and args.dump:
        np.save(args.dump, (r.read() if group_devices else image.cpu().numpy()))

    # ---- image check: 4 passes of the same view through the same shards and the same collective, against the reference's rows ----
    image_check = None
    if gold is not None and not args.no_extras and not args.emulate_world:
        r.reset()
        r.render_passes(gold[0], sync=False)
        read_back()
        if rank == 0:
            image_check = compare_golden(r.read() if group_devices else image.cpu().numpy(), gold, sc.width,
                                         **({"against": "rows rendered in this run by oracle/port.c (the C restatement of the reference kernel)"} if args.config == 5 else {}))
    if world > 1:
        dist.barrier(group=ctl)

    # ---- scenes with entity BVHs: the same steps once more with CHUNKY_OPT_BVH_CULL_BEHIND (an EXTENSION, default off: children
    #      entirely behind the ray origin count as missed — the reference walks them), reported beside `value`, never as it ----
    behind_cull = None
    if info["bvh"] and not args.no_extras and not args.emulate_world:
        r.set_option(native.OPT_BVH_CULL_BEHIND, 1)
        r.reset()
        barrier()
        r.kernel_time()
        reduce_ms.clear()
        t1 = time.perf_counter()
        run_steps(args.steps, args.warmup * passes)
        barrier()
        dt2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt2], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
            dt2 = float(t.item())
        ms2, n2 = r.kernel_time()
        check2 = None
        if gold is not None:
            r.reset()
            r.render_passes(gold[0], sync=False)
            read_back()
            if rank == 0:
                check2 = compare_golden(r.read() if group_devices else image.cpu().numpy(), gold, sc.width)
        r.set_option(native.OPT_BVH_CULL_BEHIND, 0)
        if rank == 0:
            behind_cull = {"option": "CHUNKY_OPT_BVH_CULL_BEHIND = 1 (extension, default 0; include/chunky_hip.h)",
                           "value": round(n_pix * args.steps * passes / dt2 / 1e6, 3), "unit": "Msamples/s",
                           "launch_ms": round(ms2 / max(n2, 1), 4), "image_check": check2,
                           "note": "the reference's walk visits every box the ray's LINE pierces, half of them behind the origin; with the "
                                   "option those count as missed.  Identical to the reference build's rows here, but NOT the reference's "
                                   "result in general (40 of 10^8 grazing traces on grid-aligned boxes differ, EXPERIMENTS.md 5.3): "
                                   "`value` above is the reference's walk"}
        if world > 1:
            dist.barrier(group=ctl)

    # ---- configs[3] is stated with "NEE on": the reference kernel has no emitter sampling, so the same steps run once more with
    #      CHUNKY_OPT_EMITTER_NEE (an EXTENSION specified by oracle/port.c) and are reported beside `value` ----
    emitter_nee = None
    if args.config == 3 and not args.no_extras and not args.emulate_world and args.kernel == 0:
        r.set_option(native.OPT_EMITTER_NEE, 1)
        r.reset()
        run_steps(1, 0)          # the extended instantiation's first launch (its staging array, its emitter list) is not timed
        r.reset()
        barrier()
        r.kernel_time()
        reduce_ms.clear()
        t1 = time.perf_counter()
        run_steps(args.steps, args.warmup * passes)
        barrier()
        dt3 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt3], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
            dt3 = float(t.item())
        ms3, n3 = r.kernel_time()
        info3 = r.kernel_info()
        r.set_option(native.OPT_EMITTER_NEE, 0)
        if rank == 0:
            emitter_nee = {"option": "CHUNKY_OPT_EMITTER_NEE = 1 (extension, default 0; include/chunky_hip.h)",
                           "value": round(n_pix * args.steps * passes / dt3 / 1e6, 3), "unit": "Msamples/s",
                           "launch_ms": round(ms3 / max(n3, 1), 4), "kernel": "render_pool<%d,%d,ext>+fold_kernel" % (info3["tree"], info3["pool"]),
                           "note": "one emitter block, face and point sampled per diffuse vertex plus its shadow ray (two more traces per vertex); "
                                   "no reference implementation exists, parity is against its own specification (tests/test_gpu_extensions.py)"}
        if world > 1:
            dist.barrier(group=ctl)

    if rank == 0:
        local_slots = int(parallel.owned_gids(n_pix, 0, args.emulate_world or world, args.tile, sc.width).size)  # pixels rank 0 renders
        # an emulated share renders only rank 0's tiles: count what was rendered, and say so
        samples = (min(local_slots, n_pix) if args.emulate_world else n_pix) * args.steps * passes
        value = samples / dt / 1e6
        # ---- roofline: algorithmic bytes of the reference access stream on this view ---------------
        threads, threads_how = usable_cpus()
        # what a launch really carried: chunky_render_passes cuts a step into launches of at most info["passes_per_launch"]
        # passes (256 unless the staged samples would not fit); the timed region's samples over its launches is exact either way
        launch_ms = kernel_ms / max(launches, 1)
        share = n_pix if group_devices else min(local_slots, n_pix)  # (a group reports its slowest member's kernels for the whole image)
        samples_per_launch = share * args.steps * passes // max(launches, 1)
        passes_per_launch = args.steps * passes // max(launches, 1)
        roof, cal, rows = None, None, []
        if not args.no_roofline:
            roof, cal, rows = roofline_object(sc, args.config, info, launch_ms, launches, samples_per_launch, passes_per_launch, seeds, threads, 36,
                                              args.kernel, attach_pmc=not group_devices, build_oracle=not args.no_cpu)
        else:
            roof = {"bound": "hbm", "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None,
                    "kernel": kernel_label(info), "launches": launches, "launch_ms": round(launch_ms, 4), "samples_per_launch": samples_per_launch}
        group_transport = inst.transport() if group_devices else None
        how = (f"{len(group_devices)} GPU(s) behind one context in one process (chunky_group_create), 16x16-pixel blocks round-robin, "
               f"one exchange per read-back inside the library ({group_transport['name']})"
               if group_devices else
               f"image tiles ({'16x16-pixel blocks' if args.tile == 0 else f'runs of {args.tile} px'}) round-robin over {world} GPU(s), scene replicated, "
               f"one RCCL reduce per read-back")
        box = {"hostname": os.uname().nodename, "gpu": inst.device_name() if not group_devices else torch.cuda.get_device_name(local_rank),
               "gpus_visible": torch.cuda.device_count()}
        out = {
            "metric": "Msamples/s, 32x32-chunk scene @1920x1080" if args.config == 2 else
                      (f"Msamples/s, beyond-cache world (not a BASELINE configuration) @{sc.width}x{sc.height}" if args.config == 5 else
                       f"Msamples/s, BASELINE configs[{args.config}] @{sc.width}x{sc.height}"),
            "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": len(group_devices) if group_devices else world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{what}, {sc.width}x{sc.height}, draw-depth 256, {passes} spp per step",
                       "baseline_config": args.config, "passes_per_step": passes, "passes_per_launch": passes_per_launch, "spp_timed": args.steps * passes,
                       "readbacks_timed": readbacks, "merge_interval": MERGE_INTERVAL,
                       "octree_ints": int(sc.octree.size), "octree_depth": int(sc.octree_depth),
                       "entity_bvh_ints": int(len(sc.world_bvh) + len(sc.actor_bvh)),
                       "parallelism": how, "kernel_variant": args.kernel},
            "roofline": roof, "box": box,
        }
        # the same samples without the read-backs (exchange + copy to the host): what the kernels alone sustain with everything in HBM
        rb_s = (sum(my_reduce_ms) + sum(my_d2h_ms)) * 1e-3
        out["value_hbm_resident"] = round(samples / max(dt - rb_s, 1e-9) / 1e6, 3)
        out["readback"] = {"count": readbacks, "exchange_ms": [round(x, 3) for x in my_reduce_ms], "to_host_ms": [round(x, 3) for x in my_d2h_ms],
                           "bytes_to_host": 12 * n_pix, "in_value": True,
                           "note": "every merge interval and at the end, at every N: the exchange that assembles the image on rank 0 (none at N = 1), "
                                   "then rank 0's copy into pinned host memory — the reference's clEnqueueReadBuffer (OpenClPathTracingRenderer.java:164-166)"}
        if args.emulate_world:
            out["emulated_world"] = args.emulate_world
            out["metric"] += f" — EMULATED rank-0 share of a {args.emulate_world}-GPU split on one GPU (not a multi-GPU result)"
        # what the collective actually ran on: the backend and world size torch.distributed reports ("nccl" IS RCCL on ROCm);
        # a single process runs no communicator at all
        backend_used = None if world == 1 else ("nccl" if on_gpu else "gloo")
        if group_devices:  # one process: the exchange runs inside libchunky_hip — RCCL bound from C++, or its peer-copy fallback
            backend_used = group_transport["backend"]
        out["rccl_ranks"] = (dist.get_world_size() if backend_used == "nccl" else
                             (len(group_devices) if group_devices and backend_used == "rccl" else 0))
        out["collective"] = {"backend": backend_used, "called_from": "libchunky_hip (C++, chunky_render_gather)" if group_devices else
                                                                      ("torch.distributed" if world > 1 else None),
                             "ranks": dist.get_world_size() if world > 1 else (len(group_devices) if group_devices else 1),
                             "devices": sorted(set(devices_seen)) if not group_devices else group_devices,
                             "readback_ms": [round(x, 3) for x in my_reduce_ms],
                             "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else
                                         ("bench.py self-spawn" if "RANK" in os.environ else "none")}
        if rccl_error or (world > 1 and args.backend == "nccl" and not on_gpu):
            out["collective"]["rccl_failed"] = rccl_error or "on another rank"
            out["collective"]["note"] = "the first RCCL collective raised: read-backs were staged through the host and reduced over gloo"
        if group_devices:
            out["collective"]["transport"] = group_transport
            out["group"] = {"members": len(group_devices), "devices": group_devices, "peer_status": inst.peer_status(), "transport": group_transport,
                            "peer_status_legend": "0 local (member 0's device), 1 direct (peer access enabled: xGMI), 2 staged (no peer access), < 0 = -hipError",
                            "gather_ms": [round(x, 3) for x in my_reduce_ms]}
        if per_rank:
            out["per_rank"] = per_rank
        if image_check is not None:
            out["image_check"] = image_check
        if behind_cull is not None:
            out["extension_behind_cull"] = behind_cull
        if emitter_nee is not None:
            out["extension_emitter_nee"] = emitter_nee

    # ---- N > 1: rank 0 alone opens all GPUs behind one context — the in-process path a JVM binds (the others wait) ----------
    if world > 1 and not args.no_extras:
        # (the one-device rig runs it too, with members that share GPU 0)  the other ranks wait on the rendezvous STORE (host side), not in a collective: an RCCL barrier would keep a spinning
        # kernel on every GPU the group is about to render on
        import datetime
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            store = None
        if rank == 0:
            try:
                out["group_check"] = group_leg_in_child(sorted(devices_seen), args)
            except Exception as e:  # first contact with real multi-GPU hardware: report, do not fail the bench
                out["group_check"] = {"error": f"{type(e).__name__}: {e}"}
            if store is not None:
                store.set("chunky_group_check", "done")
        elif store is not None:
            torch.cuda.synchronize()
            try:
                store.wait(["chunky_group_check"], datetime.timedelta(minutes=10))
            except Exception:
                pass
        dist.barrier(group=ctl)

    if rank == 0:
        if world == 1 and not args.no_extras and not args.emulate_world:
            try:
                out["end_to_end"] = end_to_end_leg(loader, sc, e2e_spp, make_renderer)
            except Exception as e:
                out["end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not group_devices and args.config == 2 and not args.no_extras and not args.emulate_world and not args.no_other_configs:
            # the other BASELINE configurations, short and warm, in the same run on the same box (the headline's target is gone:
            # its 6.4 GB staging array is not needed beside the 4K one of configs[4])
            r.close()
            legs = []
            for cfg in (1, 3, 4):
                try:
                    legs.append(other_config_leg(inst, cfg, args, threads))
                except Exception as e:
                    legs.append({"baseline_config": cfg, "error": f"{type(e).__name__}: {e}"})
            out["other_configs"] = legs
        if world == 1 and not args.no_cpu and not args.no_roofline and cal:
            # bounded CPU leg on the SAME view: whole image x P passes when the budget allows, else evenly spread whole rows x 1
            # pass — grown from the rate just measured until a run costs about --cpu-seconds on the usable host CPUs
            rate, done, spent, p_cpu, leg_rows = cal, 0, 0.0, 1, rows
            for _ in range(4):
                target = rate * args.cpu_seconds
                if target >= n_pix:
                    p_cpu, leg_rows = int(min(64, max(1, target // n_pix))), sample_rows(sc.height, sc.height)
                else:
                    p_cpu, leg_rows = 1, sample_rows(sc.height, int(max(36, min(sc.height, target // sc.width))))
                done, spent, _c = oracle_row_sample(sc, seeds[:p_cpu], leg_rows, threads, count=False, pin=True)
                rate = done / spent
                if spent >= 0.6 * args.cpu_seconds or (p_cpu >= 64 and len(leg_rows) >= sc.height):
                    break
            cpu_v = done / spent / 1e6
            out["cpu_baseline"] = {"value": round(cpu_v, 4), "unit": "Msamples/s", "cores": threads, "per_thread": round(cpu_v / threads, 5),
                                   "kind": "port", "pinned": True, "cores_are": threads_how,
                                   "sample": f"{done} samples = {len(leg_rows)} whole rows of the same {sc.width}x{sc.height} view, {p_cpu} pass(es), "
                                             f"{spent:.1f} s of oracle/port.c (C restatement of the reference kernel) "
                                             f"with OpenMP on {threads} threads, one worker pinned per CPU",
                                   "scaling": "profiles/r04_cpu_sweep.jsonl (1 ... 256 threads on this box type: 0.86 of linear at 8 threads, "
                                              "flat from the quota on, slower beyond it)"}
            # where the reference build itself travelled with the tree (oracle/_ref: the reference's rayTracer.cl compiled for the
            # host by clang in the build container, never rebuilt here): the SAME sample through it — the baseline is then the
            # reference kernel (kind "reference"), and the restatement's rate on that sample stays beside it
            try:
                from oracle import binding
                refk = binding.ref(build=False)
            except Exception:
                refk = None
            if refk is not None:
                h = binding.SceneHandle(sc)
                res = np.zeros(3 * n_pix, np.float32)
                band = len(leg_rows)            # whole rows: all of them when the budget bought whole passes, else a band around the centre
                lo = ((sc.height - band) // 2) * sc.width
                t_r = time.perf_counter()
                refk.render_passes(h, seeds[:p_cpu], res=res, gid_range=(lo, lo + band * sc.width), threads=threads)
                dt_r = time.perf_counter() - t_r
                n_r = band * sc.width * p_cpu
                port_leg = out["cpu_baseline"]
                out["cpu_baseline"] = {
                    "value": round(n_r / dt_r / 1e6, 4), "unit": "Msamples/s", "cores": threads, "per_thread": round(n_r / dt_r / 1e6 / threads, 5),
                    "kind": "reference", "pinned": False, "cores_are": threads_how,
                    "sample": f"{n_r} samples = {band} whole rows of the same {sc.width}x{sc.height} view, {p_cpu} pass(es), {dt_r:.1f} s of oracle/_ref — the "
                              f"reference's own rayTracer.cl compiled for x86-64 by clang (oracle/Makefile; its OpenCL builtins from rt_math.h) — on "
                              f"{threads} host threads",
                    "restatement": {"value": port_leg["value"], "kind": "port", "pinned": True, "sample": port_leg["sample"]},
                    "scaling": port_leg["scaling"]}
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            # Chunky's own Java PathTracingRenderer (se.llbit:chunky-core) is timed only where a JDK and a
            # Chunky jar exist on the box; say which it is instead of leaving the question open
            import glob
            import shutil
            jars = [p for pat in ("/usr/share/java/chunky*.jar", os.path.expanduser("~/.chunky/lib/chunky-core*.jar"), os.path.join(ROOT, "chunky-core*.jar"))
                    for p in glob.glob(pat)]
            out["cpu_baseline"]["chunky_java_renderer"] = (
                "unavailable: " + ("no JDK (`java` not on PATH)" if not shutil.which("java") else "a JDK but no chunky-core jar") + " on this box"
                if not (shutil.which("java") and jars) else f"found {jars[0]} (not driven by this bench)")
        emit(out, args.detail)
        # the line is the LAST thing on stdout: whatever a native library prints while it shuts down (RCCL announces itself on
        # stdout) goes to stderr from here on
        try:
            sys.stdout.flush()
            os.dup2(2, 1)
        except Exception:
            pass

    try:  # the line is out: a hiccup while tearing down must not turn a measured run into a failed one
        r.close()
        loader.close()
        if group_devices:
            inst.close()
    except Exception as e:
        print(f"bench.py: teardown: {e}", file=sys.stderr)
    if world > 1:
        try:
            dist.destroy_process_group()
        except Exception:
            pass  # (a communicator that failed at its first collective may refuse to shut down cleanly)


if __name__ == "__main__":
    main()
