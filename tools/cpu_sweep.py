#!/usr/bin/env python3
"""tools/cpu_sweep.py [threads...] — how the CPU baseline leg (oracle/port.c, OpenMP) scales on THIS box: the bench view
(BASELINE configs[2], 1920x1080), a fixed row sample, `threads` workers, unpinned and pinned (port_set_pinning), three
repetitions each.  One JSON line per (threads, pinned): Msamples/s (best and spread of the repetitions), per thread, and the
efficiency against one thread.  The GPU is not used.

    python tools/cpu_sweep.py 1 8 64 128 256 > profiles/rNN_cpu_sweep.jsonl"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chunkyclplugin_amd import scenes  # noqa: E402
from oracle import binding  # noqa: E402

sc = scenes.cached_outdoor_world(chunks=32, height=256, width=1920, img_height=1080)
port = binding.port()
port.lib.port_set_pinning.argtypes = [C.c_int]
port.lib.port_set_pinning.restype = None
h = binding.SceneHandle(sc)
seeds = scenes.java_random_ints(64)
ncpu = os.cpu_count() or 1
try:  # the container's CPU-time quota, if any (cgroup v2)
    _q, _p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    quota = None if _q == "max" else int(_q) / int(_p)
except Exception:
    quota = None
threads = [int(a) for a in sys.argv[1:]] or sorted({1, 8, int(quota) if quota else 8, min(64, ncpu), min(128, ncpu), ncpu})
base = None
for i, t in enumerate(threads):
    for pinned in ((0, 1) if i % 2 == 0 else (1, 0)):  # alternate which goes first: a cold first run must not look like an effect of pinning
        port.lib.port_set_pinning(pinned)
        # about 4 s of work per repetition: rows spread over the image, passes scaled with the thread count
        rows = list(range(4, sc.height, max(sc.height // min(4 * t, sc.height), 1)))[:min(4 * t, sc.height)]
        gids = (np.asarray(rows, np.int64)[:, None] * sc.width + np.arange(sc.width)[None, :]).reshape(-1).astype(np.int32)
        passes = 1
        rates = []
        for rep in range(4):
            res = np.zeros(3 * sc.width * sc.height, np.float32)
            t0 = time.perf_counter()
            port.render_gids(h, seeds[:passes], gids, res=res, threads=t)
            dt = time.perf_counter() - t0
            if rep == 0:  # calibration (and warm-up of the worker pool): size the repetitions for about 4 s
                passes = int(max(1, min(64, round(passes * 4.0 / max(dt, 1e-3)))))
                continue
            rates.append(gids.size * passes / dt / 1e6)
        best = max(rates)
        if base is None:
            base = best / t
        print(json.dumps({"threads": t, "pinned": bool(pinned), "Msamples/s": round(best, 4), "runs": [round(r, 4) for r in rates],
                          "spread": round((max(rates) - min(rates)) / best, 4), "per_thread": round(best / t, 5),
                          "efficiency_vs_first_row": round(best / t / base, 3), "samples_per_run": int(gids.size * passes),
                          "host_cpus": ncpu, "cgroup_cpu_quota": quota}), flush=True)
port.lib.port_set_pinning(0)
