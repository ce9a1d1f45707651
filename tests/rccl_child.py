"""Child process of tests/test_gpu_rccl_transport.py: one group render with the environment the parent chose (which RCCL file
libchunky_hip binds, which transport, whether member 0's own blocks travel too), compared bit for bit with the reference's golden
image.  A process of its own because the RCCL binding is made once per process (csrc/rccl_dyn.hpp).  Prints one JSON line.

    rccl_child.py <devices, e.g. 0,0,0> <golden scene> [transport to switch to after the first read-back]"""
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
    devices = [int(d) for d in sys.argv[1].split(",")]
    name = sys.argv[2]
    then = int(sys.argv[3]) if len(sys.argv) > 3 else None
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    sc = gs.make(name)
    inst = RendererInstance.group(devices)
    out = {"members": inst.group_size(), "before": inst.transport()}
    loader = HipSceneLoader(inst)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    seeds = g["seeds"]
    half = len(seeds) // 2

    def same(a):
        return bool(np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(g["res"]).view(np.uint32)))

    # two launches with a read-back in between (the exchange runs twice on a growing mean), then the whole again after a reset
    r.render_passes(seeds[:half])
    first = r.read()
    out["after_first"] = inst.transport()
    if then is not None:
        try:
            inst.set_transport(then)
            out["switched"] = inst.transport()
        except native.ChunkyHipError as e:
            out["switch_error"] = {"code": e.code, "message": str(e)}
    r.render_passes(seeds[half:], first_buffer_spp=half)
    out["identical"] = same(r.read())
    out["first_nonzero"] = bool(first.any())
    r.reset()
    r.render_passes(seeds)
    out["identical_again"] = same(r.read())
    out["after"] = inst.transport()
    r.close()
    loader.close()
    inst.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
