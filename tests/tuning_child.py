"""Child process of the tests that need a tuning / rig environment variable: those exist in the -DCHUNKY_TUNING build of the
library only (native.build_tuning; the parent points CHUNKY_HIP_LIB at it), and a library is bound once per process.  Renders a
golden scene with whatever the environment says and compares image and per-trace records with the reference build's.  One JSON line.

    tuning_child.py <golden scene>"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import golden_scenes as gs  # noqa: E402
from chunkyclplugin_amd import native  # noqa: E402
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance  # noqa: E402


def main():
    name = sys.argv[1]
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    sc = gs.make(name)
    inst = RendererInstance.get(0)
    loader = HipSceneLoader(inst)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    r.render_passes(g["seeds"])
    out = {"library": os.path.basename(native.LIB_PATH), "kernel": r.kernel_info()}
    out["identical"] = bool(np.array_equal(r.read().view(np.uint32), g["res"].view(np.uint32)))
    rec, cnt, _ = r.trace_records(int(g["seeds"][0]), gs.RECORD_GIDS)
    same = bool(np.array_equal(cnt, g["counts"]))
    for i in range(len(gs.RECORD_GIDS)):
        n = int(cnt[i])
        same = same and rec[i, :n]["material"].tolist() == g["records"][i, :n]["material"].tolist()
        same = same and bool(np.array_equal(rec[i, :n]["distance"].view(np.uint32), g["records"][i, :n]["distance"].view(np.uint32)))
    out["records_identical"] = same
    r.close()
    loader.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
