#!/usr/bin/env python3
"""BASELINE.json configs[0]: the reference's benchmark/OpenCL_test scene on the CPU path, 256x256, 16 spp — the plumbing
configuration (no GPU).  Chunky's own Java PathTracingRenderer cannot run here (no JDK / chunky-core jar), so the CPU path
is the C restatement of the reference kernel (oracle/port.c, OpenMP) and, where it was built, the reference kernel itself
compiled for x86-64 (oracle/_ref); both render the same image bit for bit.  Writes profiles/r05_config0_cpu.json."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chunkyclplugin_amd import octree2, scenes  # noqa: E402
from oracle import binding  # noqa: E402

sc = octree2.cached_benchmark_scene(256, 256)
seeds = scenes.java_random_ints(16)
threads = binding.usable_threads()
out = {"config": "BASELINE configs[0]: benchmark/OpenCL_test, 256x256, 16 spp, sun+sky, procedural asset pack (octree2.py: block models by name and properties, hashed 16x16 textures)",
       "host_threads": threads, "samples": 256 * 256 * 16}
h = binding.SceneHandle(sc)
t0 = time.perf_counter()
res = binding.port().render_passes(h, seeds, threads=threads)
dt = time.perf_counter() - t0
out["port"] = {"seconds": round(dt, 3), "Msamples/s": round(256 * 256 * 16 / dt / 1e6, 4), "mean_radiance": float(res.mean()),
               "sha256_of_res": hashlib.sha256(res.tobytes()).hexdigest()}
ref = binding.ref()
if ref is not None:
    t0 = time.perf_counter()
    res2 = ref.render_passes(h, seeds, threads=threads)
    dt = time.perf_counter() - t0
    out["reference_kernel_x86"] = {"seconds": round(dt, 3), "Msamples/s": round(256 * 256 * 16 / dt / 1e6, 4),
                                   "bit_identical_to_port": bool(np.array_equal(res.view(np.uint32), res2.view(np.uint32)))}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, "profiles", "r05_config0_cpu.json"), "w"), indent=1)
