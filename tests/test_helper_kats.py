"""Helper-level known answers (SURVEY.md section 4 tier 1, section 8c item 2): tests/golden/helpers.npz holds what the
REFERENCE object's own exported helpers return (AABB_*, BlockPalette_intersectBlock, Triangle_intersect, Sun_*,
Sky_intersect, nextPath, Atlas_read_uv, Material_sample, Octree_octreeIntersect, Bvh_intersect — driven by
oracle/ref_shim.cpp ref_helpers from tests/golden/generate.py) on seeded input rows.  The C restatement's counterparts
(CPU) and the device functions the kernels are built from (GPU, chunky_selftest_helpers) must give the same bits, so a
parity failure names a function instead of a pixel."""
import os

import numpy as np
import pytest

import golden_scenes as gs

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "helpers.npz"))
NAMES = {0: "AABB_quick_intersect", 1: "AABB_exit", 2: "AABB_full_intersect", 3: "AABB_full_intersect_map_2",
         4: "BlockPalette_intersectBlock", 6: "Triangle_intersect", 7: "Sun_sampleDirection", 8: "Sun_intersect",
         9: "Sky_intersect", 10: "nextPath", 11: "Atlas_read_uv", 12: "Material_sample", 14: "Octree_octreeIntersect",
         15: "Bvh_intersect"}
# columns the device side does not carry: the sky / sun-disc alpha (sampled by the reference, never read again,
# K/kernel.h:30) and Sun_intersect's return value (the kernels add the texel or do not)
SKIP_COLUMNS = {8: (3,), 9: (3,)}


@pytest.fixture(scope="module")
def scene():
    sc = gs.make(gs.HELPER_SCENE)
    assert gs.input_digest(sc) == str(GOLD["digest"])
    return sc


def rows_for(sc, which):
    rows = gs.helper_rows(sc, which)
    assert gs.rows_digest(rows) == str(GOLD[f"in{which}_sha256"]), "regenerated input rows differ from the ones the answers were made from"
    return rows


def assert_same(got, want, which, skip=()):
    keep = [c for c in range(want.shape[1]) if c not in skip]
    got, want = got[:, keep], want[:, keep]
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        r, c = np.argwhere(~same)[0]
        pytest.fail(f"{NAMES[which]}: {int((~same).any(axis=1).sum())} of {len(want)} rows differ; first: row {r} column {keep[c]} "
                    f"got {got[r, c]!r} want {want[r, c]!r}")


@pytest.mark.parametrize("which", gs.HELPER_KINDS)
def test_restatement_helpers_match_the_reference(port, scene, which):
    want = GOLD[f"out{which}"]
    assert np.isfinite(want[:, 0]).mean() > 0.4 and (want[:, 0] != 0).mean() > 0.15, "the rows do not exercise the helper"
    assert_same(port.helpers(scene, which, rows_for(scene, which)), want, which)


@pytest.mark.parametrize("which", gs.HELPER_KINDS)
def test_reference_helpers_still_give_the_committed_answers(ref, scene, which):
    """Where the reference build exists: the committed answers are what it returns today."""
    assert_same(ref.helpers(scene, which, rows_for(scene, which)), GOLD[f"out{which}"], which)


@pytest.fixture(scope="module")
def device_scene(gpu_instance, scene):
    from chunkyclplugin_amd.renderer import HipSceneLoader
    loader = HipSceneLoader(gpu_instance)
    loader.load_packed(scene)
    yield loader
    loader.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", gs.HELPER_KINDS)
def test_device_helpers_match_the_reference(device_scene, scene, which):
    got, _tree = device_scene.selftest_helpers(which, rows_for(scene, which))
    want = GOLD[f"out{which}"].copy()
    if which == 8:
        got[:, 4] = want[:, 4] = 0   # (the return value: see SKIP_COLUMNS)
    assert_same(got, want, which, SKIP_COLUMNS.get(which, ()))


@pytest.mark.gpu
def test_device_octree_march_in_both_layouts_and_pool_bvh_walk(device_scene, scene):
    """Row kind 14 on the reference octree layout (K/octree.h:81-89 as written) and on the wide tree in the form the render
    kernels pick; row kind 18 = the world-BVH walk as render_pool performs it (aligned records, LDS stacks) on kind 15's rows."""
    rows = rows_for(scene, 14)
    want = GOLD["out14"]
    a, ta = device_scene.selftest_helpers(14, rows, tree=0)
    b, tb = device_scene.selftest_helpers(14, rows, tree=1)
    assert ta == 0 and tb >= 16, (ta, tb)
    assert_same(a, want, 14)
    assert_same(b, want, 14)
    walk, _ = device_scene.selftest_helpers(18, rows_for(scene, 15))
    assert_same(walk, GOLD["out15"], 15)
