#!/bin/bash
# GPU box: the LDS-staged tree-top experiment the north star names.  Writes gpurun_out/r02_lds_top.json.
one() {  # label, env..., -- bench args
  python3 - "$@" <<'PY'
import json, os, subprocess, sys
label = sys.argv[1]; rest = sys.argv[2:]; k = rest.index("--"); env = dict(os.environ)
for kv in rest[:k]:
    a, b = kv.split("=", 1); env[a] = b
p = subprocess.run([sys.executable, "bench.py", "--no-cpu", "--no-roofline", "--steps", "4"] + rest[k + 1:], env=env, capture_output=True, text=True)
d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
print(json.dumps({"case": label, "Msamples/s": d["value"], "launch_ms": d["roofline"]["launch_ms"], "kernel": d["roofline"]["kernel"]}))
PY
}
V=$PWD/.variants/libchunky_hip_ldstop.so
{
one "A: dense 64^3 top node in L2 + one level of 8^3 nodes, 48 parked paths (the default)" --
one "B: the same tree, 32 parked paths (the LDS budget of C)" -- --kernel 128
one "C: 16^3 top node in L2 + 8^3 + 4^3 nodes (generic three-level walk), 32 parked paths" CHUNKY_DEBUG_WIDE_BITS=4,3,2 -- --kernel 128
one "D: C with the 16^3 top node staged in LDS (16 KB per workgroup)" CHUNKY_DEBUG_WIDE_BITS=4,3,2 CHUNKY_HIP_LIB=$V -- --kernel 128
} | tee gpurun_out/r02_lds_top.jsonl
CHUNKY_DEBUG_WIDE_BITS=4,3,2 CHUNKY_HIP_LIB=$V timeout 300 python3 - <<'PY' | tee -a gpurun_out/r02_lds_top.jsonl
import json, os, numpy as np
from chunkyclplugin_amd import native, scenes
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance
from oracle import binding
sc = scenes.cached_outdoor_world(chunks=32, height=256)
loader = HipSceneLoader(RendererInstance.get(0)); loader.load_packed(sc)
r = HipPathTracingRenderer(loader, sc.width, sc.height); r.set_camera(sc.projector_type, sc.camera); r.set_option(native.OPT_KERNEL, 128)
seeds = native.java_random_ints(8); r.render_passes(seeds)
gids = np.concatenate([np.arange(y * sc.width, (y + 1) * sc.width) for y in (101, 540, 931)]).astype(np.int32)
want = binding.port().render_gids(sc, seeds, gids, threads=os.cpu_count()).reshape(-1, 3)[gids]
got = r.read().reshape(-1, 3)[gids]
print(json.dumps({"case": "D parity: three rows of the 1080p view, 8 passes, against the oracle", "kernel": r.kernel_info(),
                  "bit_identical": bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))}))
PY
