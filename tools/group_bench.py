#!/usr/bin/env python3
"""tools/group_bench.py [members...] — the in-process multi-GPU group (chunky_group_create) on the bench workload: for each
member count n, n members (on the devices given by CHUNKY_GROUP_DEVICES, default: all on device 0 — a 1-GPU box) render
BASELINE configs[2] at 1920x1080, 256 passes, and the read-back exchange (pack, copy to member 0, scatter) is timed on its own.
With members sharing one device the render time says nothing about scaling (they compete for the same CUs); the exchange
time and the bit-identity of the image are what this measures.  One JSON line per member count."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chunkyclplugin_amd import native, scenes  # noqa: E402
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance  # noqa: E402

W, H, PASSES = 1920, 1080, 256
sc = scenes.cached_outdoor_world(chunks=32, height=256, width=W, img_height=H)
seeds = native.java_random_ints(PASSES)
devs = [int(x) for x in os.environ.get("CHUNKY_GROUP_DEVICES", "").split(",") if x.strip()]
ref = None
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    devices = (devs[:n] if len(devs) >= n else [0] * n)
    inst = RendererInstance.group(devices) if n > 1 else RendererInstance(devices[0])
    loader = HipSceneLoader(inst)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, W, H)
    r.set_camera(sc.projector_type, sc.camera)
    r.render_passes(seeds)              # warm-up at the timed launch shape: the staging arrays are allocated here
    r.gather()
    r.reset()
    t0 = time.perf_counter()
    r.render_passes(seeds, sync=False)
    r.sync()
    t1 = time.perf_counter()
    r.gather()
    t2 = time.perf_counter()
    img = r.read()
    if ref is None:
        ref = img
    print(json.dumps({"members": n, "devices": devices, "render_ms": round((t1 - t0) * 1e3, 3), "exchange_ms": round((t2 - t1) * 1e3, 3),
                      "Msamples/s": round(W * H * PASSES / (t2 - t0) / 1e6, 1), "kernel_ms_slowest_member": round(r.kernel_time()[0], 3),
                      "image_equals_first": bool(np.array_equal(img.view(np.uint32), ref.view(np.uint32)))}), flush=True)
    r.close()
    loader.close()
    inst.close()
