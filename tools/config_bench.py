#!/usr/bin/env python3
"""Msamples/s of the other BASELINE.json configurations on one MI355X (GPU box), each with a parity spot-check of whole image
rows against the oracle AT THE TIMED PASS COUNT (the instantiation that is timed is the one that is checked):

  benchmark   configs[1]  the reference's benchmark/OpenCL_test scene (tests/golden/benchmark_OpenCL_test.npz), 1920x1080
  benchmark_entities      the same with the scene's 4 188 entities + 389 actors as box proxies in the two BVHs
  indoor      configs[3]  emitter-lit room, sun flag 0, 1920x1080 — as the reference renders it (implicit emitter hits)
  indoor_nee  configs[3]  the same with CHUNKY_OPT_EMITTER_NEE (the configuration's "NEE on": an extension)
  entities    configs[4]  32x32-chunk world + 100 000 world / 5 000 actor triangles, 1920x1080 on one GPU
  entities4k  configs[4]  the same at 3840x2160, rank 0's share of an 8-GPU tile split (what one GPU of the stated config renders)
  entities1m  configs[4]  1 000 000 world triangles (the upper end of the configuration's 10^5 - 10^6 range), 1920x1080 (on request)
  entities_cull / entities1m_cull   the two entity worlds with CHUNKY_OPT_BVH_CULL_BEHIND (extension; checked against its own oracle mode)

    python tools/config_bench.py [names...] > profiles/rNN_config_bench.jsonl"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chunkyclplugin_amd import native, parallel, scenes  # noqa: E402
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance  # noqa: E402
from oracle import binding  # noqa: E402
from oracle.binding import PortExt  # noqa: E402


def run(sc, name, passes=32, launches=3, world=1, ext=None, check_rows=(60, 250, 440, 630, 820, 1010)):
    loader = HipSceneLoader(RendererInstance.get(0))
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    r.set_shard(0, world, 0)   # 16 x 16-pixel blocks, as bench.py --gpus N
    if os.environ.get("CHUNKY_BENCH_KERNEL"):   # tuning runs: an experimental kernel variant (CHUNKY_OPT_KERNEL)
        r.set_option(native.OPT_KERNEL, int(os.environ["CHUNKY_BENCH_KERNEL"]))
    for k, v in (ext or {}).items():
        r.set_option({"nee": native.OPT_EMITTER_NEE, "bsdf": native.OPT_BSDF, "cull": native.OPT_BVH_CULL_BEHIND}[k], v)
    seeds = native.java_random_ints(passes * (launches + 1))
    r.render_passes(seeds[:passes])
    r.kernel_time()
    t0 = time.perf_counter()
    for k in range(launches):
        r.render_passes(seeds[(k + 1) * passes:(k + 2) * passes], first_buffer_spp=(k + 1) * passes, sync=False)
    r.sync()
    dt = time.perf_counter() - t0
    ms, n = r.kernel_time()
    info = r.kernel_info()
    # parity spot check: the timed pass count, whole rows (this rank's pixels of them)
    r.reset()
    r.render_passes(seeds[:passes])
    got = r.read().reshape(-1, 3)
    rows = [min(y * sc.height // 1080, sc.height - 1) for y in check_rows]
    gids = np.concatenate([np.arange(y * sc.width, (y + 1) * sc.width) for y in rows]).astype(np.int32)
    gids = np.intersect1d(gids, parallel.owned_gids(sc.width * sc.height, 0, world, 0, sc.width)).astype(np.int32)
    port = binding.port()
    port.counters(enable=True, reset=True)
    port.counters(reset=True)
    if ext and "cull" in ext:
        with binding.PortCull(port):
            want = port.render_gids(sc, seeds[:passes], gids, threads=binding.usable_threads()).reshape(-1, 3)
    elif ext:
        with PortExt(port, sc, **ext):
            want = port.render_gids(sc, seeds[:passes], gids, threads=binding.usable_threads()).reshape(-1, 3)
    else:
        want = port.render_gids(sc, seeds[:passes], gids, threads=binding.usable_threads()).reshape(-1, 3)
    bps = binding.algorithmic_bytes(port.counters(enable=False, reset=True))
    same = bool(np.array_equal(got[gids].view(np.uint32), want[gids].view(np.uint32)))
    n_local = int(parallel.owned_gids(sc.width * sc.height, 0, world, 0, sc.width).size)
    out = {"config": name, "scene": sc.name, "size": [sc.width, sc.height], "share": f"1/{world}", "passes_per_launch": passes,
           "kernel": info, "Msamples/s": n_local * passes * launches / dt / 1e6, "launch_ms": ms / n,
           "rows_bit_identical_to_oracle": same, "pixels_checked": int(gids.size), "algorithmic_bytes_per_sample": bps}
    out["algorithmic_GBps"] = bps * n_local * passes / (ms / n * 1e-3) / 1e9
    out["frac_of_8TBps"] = out["algorithmic_GBps"] / 8000.0
    r.close()
    loader.close()
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["benchmark", "benchmark_entities", "indoor", "indoor_nee", "entities", "entities4k"]
    res = []
    if "benchmark" in which:
        from chunkyclplugin_amd import octree2
        res.append(run(octree2.cached_benchmark_scene(1920, 1080), "configs[1]", passes=64, launches=3))
    if "benchmark_entities" in which:
        from chunkyclplugin_amd import octree2
        res.append(run(octree2.cached_benchmark_scene(1920, 1080, entities=True), "configs[1] + the scene's entities (box proxies)",
                       passes=32, launches=2))
    if "benchmark_entities_cull" in which:
        from chunkyclplugin_amd import octree2
        res.append(run(octree2.cached_benchmark_scene(1920, 1080, entities=True), "configs[1] + the scene's entities, CHUNKY_OPT_BVH_CULL_BEHIND (extension)",
                       passes=32, launches=2, ext={"cull": 1}))
    if "indoor" in which:
        res.append(run(scenes.indoor_room(size=64, width=1920, img_height=1080), "configs[3] (reference light transport)", passes=32))
    if "indoor_nee" in which:
        res.append(run(scenes.indoor_room(size=64, width=1920, img_height=1080), "configs[3] with emitter NEE (extension)", passes=32,
                       ext={"nee": 1}))
    if "entities" in which or "entities4k" in which or "entities_cull" in which:
        base = scenes.cached_outdoor_world(chunks=32, height=256)
        sc = scenes.add_entities(base, 100000, seed=11, actor_tris=5000, region=((40, 90, 40), (470, 170, 470)))
        if "entities" in which:
            res.append(run(sc, "configs[4] at 1920x1080 on one GPU", passes=16, launches=2))
        if "entities_cull" in which:
            res.append(run(sc, "configs[4] at 1920x1080 with CHUNKY_OPT_BVH_CULL_BEHIND (extension)", passes=16, launches=2, ext={"cull": 1}))
        if "entities4k" in which:
            res.append(run(sc.with_view(3840, 2160), "configs[4] at 3840x2160, rank 0 of 8", passes=16, launches=2, world=8))
    if "entities1m" in which:   # the upper end of configs[4]'s range: 10^6 world triangles
        res.append(run(scenes.cached_entity_world(1000000), "configs[4] with 1 000 000 world triangles, 1920x1080 on one GPU", passes=8, launches=2,
                       check_rows=(250, 630, 1010)))
    if "entities1m_cull" in which:
        res.append(run(scenes.cached_entity_world(1000000), "configs[4] with 1 000 000 world triangles and CHUNKY_OPT_BVH_CULL_BEHIND (extension)",
                       passes=8, launches=2, check_rows=(250, 630, 1010), ext={"cull": 1}))
    for x in res:
        print(json.dumps(x), flush=True)
