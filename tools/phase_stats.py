#!/usr/bin/env python3
"""Per-phase profile of the wave-scheduled kernel on the bench workload (GPU box)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chunkyclplugin_amd import native, scenes  # noqa: E402
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 4
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kw = {}
if os.environ.get("CHUNKY_STATS_MODELS") == "0":  # the same world with no slab / plant blocks: BLOCK then runs the cube path only
    kw = dict(aabb_frac=0.0, quad_frac=0.0)
sc = scenes.cached_outdoor_world(chunks=32, height=256, **kw)
if os.environ.get("CHUNKY_STATS_SCENE") == "entities":  # BASELINE configs[4]; BLOCK then includes the BVH walk
    sc = scenes.add_entities(sc, 100000, seed=11, actor_tris=5000, region=((40, 90, 40), (470, 170, 470)))
if os.environ.get("CHUNKY_STATS_SCENE") == "indoor":    # BASELINE configs[3]
    sc = scenes.indoor_room(size=64, width=1920, img_height=1080)
if os.environ.get("CHUNKY_STATS_SCENE") == "city":      # BASELINE configs[1]
    from chunkyclplugin_amd import octree2
    sc = octree2.cached_benchmark_scene(1920, 1080)
loader = HipSceneLoader(RendererInstance.get(0))
loader.load_packed(sc)
r = HipPathTracingRenderer(loader, sc.width, sc.height)
r.set_camera(sc.projector_type, sc.camera)
r.set_option(native.OPT_KERNEL, variant)
r.set_shard(0, world, 256)
seeds = native.java_random_ints(2 * passes)
r.render_passes(seeds[:passes])
r.phase_stats(reset=True)
r.kernel_time()
r.render_passes(seeds[passes:], first_buffer_spp=passes)
ms, n = r.kernel_time()
st = r.phase_stats()
samples = sc.width * sc.height * passes // world
w = st.pop("waves")
ho = st.pop("handover")
parts = st.pop("parts")
model = st.pop("model")
tot = sum(v["cycles"] for v in st.values()) + model["cycles"]   # (march, block, shade + the model blocks' phase of the sorted kernel)
out = {"variant": variant, "launch_ms": ms / n, "Msamples/s": samples / (ms / n) / 1e3}
for k, v in st.items():
    out[k] = {"execs_per_sample": v["execs"] * 64 / samples, "lanes_per_exec": v["lanes"] / max(v["execs"], 1),
              "cycles_per_exec": v["cycles"] / max(v["execs"], 1), "time_share": v["cycles"] / max(tot, 1)}
info = r.kernel_info()
out["kernel"] = info
if info["pool"] >= 0:  # render_pool: the two counters are swap rounds and paths swapped
    out["swaps"] = {"rounds_per_sample": ho["execs"] * 64 / samples, "paths_per_round": ho["cycles"] / max(ho["execs"], 1),
                    "paths_swapped_per_sample": ho["cycles"] / samples}
else:
    out["handover"] = {"execs_per_sample": ho["execs"] * 64 / samples, "cycles_per_exec": ho["cycles"] / max(ho["execs"], 1), "share_of_total": ho["cycles"] / max(tot, 1)}
out["parts_share_of_total"] = {k: round(v / max(tot, 1), 4) for k, v in parts.items()}
if info["pool"] >= 0:  # render_pool reuses three slots: swap cycles, loop iterations, march-loop entries
    out["parts_share_of_total"].pop("open_pixel"); out["parts_share_of_total"].pop("hand_out")
    out["parts_share_of_total"]["swap"] = out["parts_share_of_total"].pop("fold")
    out["model"] = {"execs_per_sample": model["execs"] * 64 / samples, "lanes_per_exec": model["lanes"] / max(model["execs"], 1),
                    "cycles_per_exec": model["cycles"] / max(model["execs"], 1), "time_share": model["cycles"] / max(tot, 1)}
    out["loop"] = {"iterations_per_64_samples": parts["open_pixel"] * 64 / samples, "march_entries_per_64_samples": parts["hand_out"] * 64 / samples}
out["waves"] = {"n": w["n"], "mean_life_over_max": w["life_sum"] / max(w["n"], 1) / max(w["life_max"], 1)}
print(json.dumps(out, indent=1))
