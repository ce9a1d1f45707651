"""Child process of tests/test_gpu_hostile.py: renders scenes whose palettes / BVHs / octree leaves were damaged at random (pointers
outside their arrays, absurd counts) through every kernel family.  A GPU memory fault would abort this process; it has to end with 0.
Prints one JSON line: how many renders ran, how many were refused with CHUNKY_E_INVALID."""
import dataclasses
import json
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from chunkyclplugin_amd import native, scenes  # noqa: E402
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance  # noqa: E402

NASTY = np.array([-1, -7, 0x7FFFFFFF, -0x80000000, 1 << 30, 255, 256, 100000, -100000], np.int64)


def damage(rng, a, hits):
    a = np.array(a, np.int32).copy()
    for _ in range(hits):
        if a.size:
            a[rng.integers(a.size)] = np.int32(rng.choice(NASTY)) if rng.random() < 0.7 else np.int32(rng.integers(-2 ** 31, 2 ** 31))
    return a


def main(seed0, count):
    inst = RendererInstance.get(0)
    ran = refused = 0
    differ = []
    for it in range(count):
        rng = np.random.default_rng(seed0 + it)
        sc = scenes.tiny_scene(seed=int(rng.integers(1, 10 ** 6)), size=int(rng.choice([16, 32])), width=40, height=24,
                               entities=int(rng.choice([0, 24, 120])), sun_flag=True)
        what = rng.integers(7)
        repl = {}
        if what == 0:
            repl["block_palette"] = damage(rng, sc.block_palette, int(rng.integers(1, 4)))
        elif what == 1:
            repl["aabb_models"] = damage(rng, sc.aabb_models, int(rng.integers(1, 4)))
            repl["quad_models"] = damage(rng, sc.quad_models, int(rng.integers(1, 4)))
        elif what == 2:
            repl["material_palette"] = np.array(sc.material_palette, np.int32)[: max(6, sc.material_palette.size - int(rng.integers(1, 13)))]
        elif what == 3:
            repl["bvh_trigs"] = damage(rng, sc.bvh_trigs, int(rng.integers(1, 4)))
        elif what == 4:   # octree leaves pointing beyond the block palette (branch values stay valid: set_octree checks those)
            t = np.array(sc.octree, np.int32).copy()
            leaves = np.flatnonzero(t <= 0)
            for k in rng.choice(leaves, size=min(6, leaves.size), replace=False):
                t[k] = -np.int32(sc.block_palette.size + int(rng.integers(0, 1000)) * 2)
            repl["octree"] = t
        elif what == 5:
            repl["block_palette"] = np.array(sc.block_palette, np.int32)[:-1]   # an odd number of ints
        else:
            repl["material_palette"] = np.zeros(0, np.int32)                    # no materials at all: no block can hit
        bad = dataclasses.replace(sc, **repl)
        images = {}
        for variant in (0, 1, 8, 9, 3):
            loader = HipSceneLoader(inst)
            try:
                loader.load_packed(bad)
                r = HipPathTracingRenderer(loader, bad.width, bad.height)
                r.set_camera(bad.projector_type, bad.camera)
                r.set_option(native.OPT_KERNEL, variant)
                if variant == 0 and it % 3 == 0:   # the extended integrator reads the emitter list and material word 5 as well
                    r.set_option(native.OPT_EMITTER_NEE, 1)
                    r.set_option(native.OPT_BSDF, 1)
                try:
                    r.render_passes(native.java_random_ints(2))
                    img = r.read()
                    r.preview()
                    ran += 1
                    if not (variant == 0 and it % 3 == 0):
                        images[variant] = np.array(img, np.float32).copy()
                except native.ChunkyHipError as e:
                    if e.code != native.E_INVALID:
                        raise
                    refused += 1
                r.close()
            except native.ChunkyHipError as e:   # an upload the library refuses outright
                if e.code != native.E_INVALID:
                    raise
                refused += 1
            loader.close()
        # whatever a damaged scene renders as, every kernel family renders it the same (the sorted block tests of render_pool,
        # variant 0, against the kernels that test a block as it comes)
        ref = images.get(1)
        for variant, img in images.items():
            if ref is not None and not np.array_equal(img.view(np.uint32), ref.view(np.uint32)):
                differ.append([seed0 + it, int(what), variant])
    print(json.dumps({"rendered": ran, "refused": refused, "kernels_differ": differ}))


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))
