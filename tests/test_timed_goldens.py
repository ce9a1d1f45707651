"""The oracle closure at the sizes that are timed.  tests/golden/timed_rows.npz holds whole image rows of the BASELINE views
as bench.py / tools/config_bench.py render them — configs[1] (the reference's benchmark city, without and with its
entities), configs[2] (32x32-chunk world), configs[3] (indoor room), configs[4] (100 000 + 5 000 triangles) at 1920x1080 and at
its stated 3840x2160 — rendered by the REFERENCE build (oracle/_ref, tests/golden/generate.py timed).  The C restatement
must reproduce them (CPU), and so must the HIP kernels (GPU), so parity on depth-9/10 octrees and height-17 BVHs does not
rest on the restatement being right."""
import os

import numpy as np
import pytest

import golden_scenes as gs
from oracle import binding

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "timed_rows.npz"))
THREADS = binding.usable_threads()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def views():
    cache = {}

    def get(name):
        if name not in cache:
            sc = gs.timed_view(name)
            assert gs.input_digest(sc) == str(GOLD[name + "_digest"]), "regenerated scene differs from the one the golden rows were made from"
            cache[name] = sc
        return cache[name]
    return get


def row_gids(sc, rows):
    return np.concatenate([np.arange(y * sc.width, (y + 1) * sc.width) for y in rows]).astype(np.int32)


@pytest.mark.parametrize("name", gs.TIMED_VIEWS)
def test_restatement_matches_the_reference_at_timed_sizes(port, views, name):
    sc = views(name)
    rows = GOLD[name + "_rows"]
    assert rows.tolist() == gs.timed_rows(sc)
    gids = row_gids(sc, rows)
    got = port.render_gids(binding.SceneHandle(sc), GOLD["seeds"], gids, threads=THREADS).reshape(-1, 3)[gids]
    np.testing.assert_array_equal(bits(got), bits(GOLD[name + "_res"].reshape(-1, 3)))


def test_reference_still_gives_the_committed_rows(ref, views):
    """Where the reference build exists: one row of the heaviest view is what it returns today."""
    sc = views("entities")
    y = int(GOLD["entities_rows"][1])
    full = ref.render_passes(binding.SceneHandle(sc), GOLD["seeds"], gid_range=(y * sc.width, (y + 1) * sc.width), threads=THREADS)
    np.testing.assert_array_equal(bits(full.reshape(-1, 3)[y * sc.width:(y + 1) * sc.width]), bits(GOLD["entities_res"][1]))


# (tree form, entity-BVH phases) of the instantiation bench.py / tools/config_bench.py time on each view
TIMED_KERNEL = {"city": (17, False), "city_entities": (17, True), "outdoor": (17, False), "indoor": (17, False),
                "entities": (17, True), "entities4k": (17, True)}


@pytest.mark.gpu
@pytest.mark.parametrize("name", gs.TIMED_VIEWS)
def test_hip_matches_the_reference_at_timed_sizes(gpu_instance, views, name):
    from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader
    sc = views(name)
    loader = HipSceneLoader(gpu_instance)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    r.render_passes(GOLD["seeds"])
    info = r.kernel_info()
    tree, bvh = TIMED_KERNEL[name]
    assert (info["tree"], info["bvh"]) == (tree, bvh) and info["pool"] == (64 if not bvh else info["pool"]) and info["pool"] > 0, info
    # full cubes and model blocks in phases of their own: where model blocks are common (the city, 11 % of its leaves), never with entity BVHs
    assert info["sorted"] == (name == "city"), info
    gids = row_gids(sc, GOLD[name + "_rows"])
    got = r.read().reshape(-1, 3)[gids]
    want = GOLD[name + "_res"].reshape(-1, 3)
    same = (bits(got) == bits(want)).all(axis=1)
    assert same.all(), f"{name}: {int((~same).sum())} of {len(gids)} pixels differ from the reference build's rows (first gid {int(gids[np.argmin(same)])})"
    r.close()
    loader.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,variant,sorted_", [("outdoor", 256, True), ("indoor", 256, True), ("city", 512, False)])
def test_the_other_block_test_order_gives_the_same_rows(gpu_instance, views, name, variant, sorted_):
    """render_pool tests full cubes and model blocks in phases of their own where model blocks are common (the city) and as they come
    elsewhere: the instantiation a timed view does NOT run by default (CHUNKY_OPT_KERNEL bit 8 / bit 9) renders the reference
    build's rows as well."""
    from chunkyclplugin_amd import native
    from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader
    sc = views(name)
    loader = HipSceneLoader(gpu_instance)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    r.set_option(native.OPT_KERNEL, variant)
    r.render_passes(GOLD["seeds"])
    info = r.kernel_info()
    assert info["tree"] == 17 and info["pool"] == 64 and info["sorted"] == sorted_, info
    gids = row_gids(sc, GOLD[name + "_rows"])
    got = r.read().reshape(-1, 3)[gids]
    want = GOLD[name + "_res"].reshape(-1, 3)
    same = (bits(got) == bits(want)).all(axis=1)
    assert same.all(), f"{name}, variant {variant}: {int((~same).sum())} of {len(gids)} pixels differ from the reference build's rows"
    r.close()
    loader.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,passes", [("outdoor", 8), ("city", 4), ("city_entities", 2), ("indoor", 4), ("entities", 2), ("entities4k", 1)])
def test_hip_matches_the_live_reference_build_on_the_whole_image(gpu_instance, ref, views, name, passes):
    """Where the reference build travelled to the GPU box (oracle/_ref; skipped elsewhere): the WHOLE image of a timed view (1920x1080; 3840x2160 for entities4k) — every pixel, `passes` passes of a java.util.Random stream the committed fixture does not hold — rendered by the reference
    kernel on the host's CPUs and by the HIP kernels, bit for bit (2 073 600 pixels, 16.6 M samples for the headline view)."""
    from chunkyclplugin_amd import native
    from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader
    sc = views(name)
    seeds = native.java_random_ints(passes, seed=987654321)
    want = ref.render_passes(binding.SceneHandle(sc), seeds, threads=THREADS)
    loader = HipSceneLoader(gpu_instance)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    r.render_passes(seeds)
    info = r.kernel_info()
    assert info["tree"] == 17 and info["pool"] > 0, info
    same = (bits(r.read()) == bits(want)).reshape(-1, 3).all(axis=1)
    assert same.all(), f"{name}: {int((~same).sum())} of {same.size} pixels differ from the live reference build (first gid {int(np.argmin(same))})"
    r.close()
    loader.close()
