#!/usr/bin/env python3
"""tools/variant_check.py <CHUNKY_OPT_KERNEL value> — an experimental kernel variant against the reference's golden images
(tests/golden/*.npz, all scenes), bit for bit, three renders each.  Prints one line per scene."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_scenes as gs  # noqa: E402
from chunkyclplugin_amd import native  # noqa: E402
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance  # noqa: E402

variant = int(sys.argv[1])
inst = RendererInstance.get(0)
bad = 0
for name in ("outdoor", "outdoor_nosun", "indoor", "indoor_sun", "inside", "water", "dof", "pregen", "atlas_layers", "entities"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    sc = gs.make(name)
    loader = HipSceneLoader(inst)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    r.set_option(native.OPT_KERNEL, variant)
    same = True
    for rep in range(3):
        r.reset()
        r.render_passes(g["seeds"])
        same = same and np.array_equal(r.read().view(np.uint32), g["res"].view(np.uint32))
    bad += not same
    info = r.kernel_info()
    print(name, "identical" if same else "DIFFERS", info["tree"], info["pool"], info["blocks"])
    r.close()
    loader.close()
print("variant", variant, "scenes differing:", bad)
sys.exit(1 if bad else 0)
