#!/usr/bin/env python3
"""Msamples/s of the other BASELINE.json configurations (GPU box): indoor emitter room (configs[3]) and
entity-heavy world (configs[4], 1 GPU share), each with a parity spot-check against the oracle on a crop."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chunkyclplugin_amd import native, scenes  # noqa: E402
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance  # noqa: E402
from oracle import binding  # noqa: E402


def run(sc, passes=32, launches=3, check_rows=(60, 250, 440, 630, 820, 1010)):
    loader = HipSceneLoader(RendererInstance.get(0))
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    seeds = native.java_random_ints(passes * (launches + 1))
    r.render_passes(seeds[:passes])
    r.kernel_time()
    t0 = time.perf_counter()
    for k in range(launches):
        r.render_passes(seeds[(k + 1) * passes:(k + 2) * passes], first_buffer_spp=(k + 1) * passes, sync=False)
    r.sync()
    dt = time.perf_counter() - t0
    ms, n = r.kernel_time()
    # parity spot check: 2 passes, three rows
    r.reset()
    r.render_passes(seeds[:2])
    got = r.read().reshape(sc.height, sc.width, 3)
    rows = [min(y, sc.height - 1) for y in check_rows]
    gids = np.concatenate([np.arange(y * sc.width, (y + 1) * sc.width) for y in rows]).astype(np.int32)
    port = binding.port()
    port.counters(enable=True, reset=True)
    port.counters(reset=True)
    want = port.render_gids(sc, seeds[:2], gids, threads=os.cpu_count()).reshape(sc.height, sc.width, 3)
    bps = binding.algorithmic_bytes(port.counters(enable=False, reset=True))
    same = all(np.array_equal(got[y].view(np.uint32), want[y].view(np.uint32)) for y in rows)
    out = {"scene": sc.name, "size": [sc.width, sc.height], "Msamples/s": sc.width * sc.height * passes * launches / dt / 1e6,
           "launch_ms": ms / n, "rows_bit_identical_to_oracle": bool(same),
           "algorithmic_bytes_per_sample": bps}
    out["algorithmic_GBps"] = bps * sc.width * sc.height * passes / (ms / n * 1e-3) / 1e9
    out["frac_of_8TBps"] = out["algorithmic_GBps"] / 8000.0
    r.close()
    loader.close()
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["benchmark", "indoor", "entities"]
    res = []
    if "benchmark" in which:  # BASELINE configs[1]: the reference's own benchmark scene, 1920x1080
        from chunkyclplugin_amd import octree2
        res.append(run(octree2.cached_benchmark_scene(1920, 1080), passes=64, launches=3))
    if "indoor" in which:
        res.append(run(scenes.indoor_room(size=64, width=1920, img_height=1080)))
    if "entities" in which:
        base = scenes.cached_outdoor_world(chunks=32, height=256)
        t0 = time.time()
        sc = scenes.add_entities(base, 100000, seed=11, actor_tris=5000,
                                 region=((40, 90, 40), (470, 170, 470)))
        print("entities built in %.1fs" % (time.time() - t0), file=sys.stderr)
        res.append(run(sc, passes=16, launches=2))
    for x in res:
        print(json.dumps(x))
