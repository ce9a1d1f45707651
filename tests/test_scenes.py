"""Host-side packers/generators (wire formats of SURVEY.md Appendix A)."""
import numpy as np

from chunkyclplugin_amd import scenes


def lookup(tree, depth, x, y, z):
    """Octree_get (K/octree.h:23-39) in Python."""
    level, data = depth, int(tree[0])
    while data > 0:
        level -= 1
        data = int(tree[data + ((((x >> level) & 1) << 2) | (((y >> level) & 1) << 1) | ((z >> level) & 1))])
    return -data, level


def test_octree_roundtrip_and_merge():
    rng = np.random.default_rng(0)
    depth, S = 4, 16
    t = np.zeros((S, S, S), np.int32)
    t[:8, :8, :8] = 3                      # mergeable 8^3 region
    t[8:, 8:, 8:] = rng.integers(0, 5, size=(8, 8, 8))
    t[12, 3, 3] = scenes.ANY_TYPE
    for order in ("bfs", "dfs"):
        tree = scenes.build_octree(t, depth, order)
        assert (tree.size - 1) % 8 == 0
        for _ in range(300):
            x, y, z = rng.integers(0, S, size=3)
            v, level = lookup(tree, depth, int(x), int(y), int(z))
            want = int(t[x, y, z])
            assert v == (scenes.ANY_TYPE if want == scenes.ANY_TYPE else 2 * want)
        v, level = lookup(tree, depth, 1, 2, 3)
        assert (v, level) == (6, 3)        # merged into one level-3 leaf
    a, b = scenes.build_octree(t, depth, "bfs"), scenes.build_octree(t, depth, "dfs")
    assert a.size == b.size


def test_atlas_first_fit_and_location_bits():
    ab = scenes.AtlasBuilder(4, 4)
    rng = np.random.default_rng(1)
    ids = [ab.add(scenes.noise_texture(rng, (10 * i, 20, 30))) for i in range(5)]
    big = ab.add(scenes.noise_texture(rng, (200, 200, 200), size=32))
    atlas, recs = ab.build()
    assert atlas.shape == (1, 64, 64, 4)
    size, loc = recs[big]
    assert size == (32 << 16) | 32 and loc == 0          # largest first, at tile (0,0), layer 0
    # x outer, y inner: after the 2x2 tile, the next free tiles are (0,2), (0,3), (1,2)...
    assert [(recs[i][1] >> 22) & 0x1FF for i in ids[:3]] == [0, 0, 1]
    assert [(recs[i][1] >> 13) & 0x1FF for i in ids[:3]] == [2, 3, 2]
    for i in ids:
        size, loc = recs[i]
        x, y = ((loc >> 22) & 0x1FF) * 16, ((loc >> 13) & 0x1FF) * 16
        assert (atlas[0, y:y + 16, x:x + 16] == ab._tex[i]).all()


def test_bvh_layout():
    sc = scenes.add_entities(scenes.tiny_scene(entities=0), 64, actor_tris=16)
    for nodes in (sc.world_bvh, sc.actor_bvh):
        n = nodes.reshape(-1, 7)
        inner = n[:, 0] > 0
        assert (n[inner, 0] % 7 == 0).all() and (n[inner, 0] < nodes.size).all()
        leaves = -n[~inner, 0]
        assert (leaves >= 0).all() and (leaves < sc.bvh_trigs.size).all()
        cnt = sc.bvh_trigs[leaves]
        assert (cnt >= 1).all() and (cnt <= 4).all()
    e = scenes.empty_bvh()
    assert e[0] == 0 and np.isnan(e[1:].view(np.float32)).all()


def test_sun_and_camera_packing():
    sun = scenes.pack_sun(0.6, 1.2, 1.25, True, (0x200020, 5))
    assert sun[0] == 1 and sun[1] == 0x200020 and sun[2] == 5
    assert np.allclose(sun[3:].view(np.float32), [1.25, 0.6, 1.2])
    cam = scenes.look_at_camera((1, 2, 3), (4, 2, 7))
    assert cam.shape == (15,)
    M = cam[3:12].reshape(3, 3)
    assert np.allclose(M @ M.T, np.eye(3), atol=1e-6)
    assert np.allclose(M[:, 2], np.array([3, 0, 4]) / 5, atol=1e-6)   # third column = forward


def test_wide_tree_relayout_preserves_every_lookup():
    """The upload-time re-layout (widetree.hpp) must return the reference's (leaf value, leaf level)
    for every cell — K/octree.h:81-89 is a pure function of the cell."""
    import pytest
    from chunkyclplugin_amd import native
    rng = np.random.default_rng(5)
    sc = scenes.outdoor_world(chunks=2, height=48, seed=101)          # depth 6, with ANY_TYPE interior
    S = 1 << sc.octree_depth
    xyz = rng.integers(0, S, size=(4000, 3)).astype(np.int32)
    want = np.array([lookup(sc.octree, sc.octree_depth, *map(int, c)) for c in xyz])
    for bits in (None, [3, 3], [2, 2, 2], [1, 1, 1, 1, 1, 1], [6], [4, 2], [1, 2, 3]):
        data, level, n = native.widetree_lookup(sc.octree, sc.octree_depth, xyz, bits)
        np.testing.assert_array_equal(data, want[:, 0], err_msg=str(bits))
        np.testing.assert_array_equal(level, want[:, 1], err_msg=str(bits))
        assert n >= 8
    assert (want[:, 0] == scenes.ANY_TYPE).any() and (want[:, 1] > 1).any()
    # single-leaf world, and rejects what it cannot express
    d, l, n = native.widetree_lookup(np.array([-6], np.int32), 4, [[1, 2, 3]])
    assert (d[0], l[0]) == (6, 4)
    with pytest.raises(native.ChunkyHipError):
        native.widetree_lookup(sc.octree, sc.octree_depth, xyz, [3, 2])     # bits do not sum to depth
    with pytest.raises(native.ChunkyHipError):
        native.widetree_lookup(np.array([-(1 << 28)], np.int32), 3, [[0, 0, 0]])  # pointer too large


def test_octree2_reader_on_the_reference_benchmark_scene():
    """`.octree2` reader: 3091 palette entries, depth 10, 345 557 branches,
    2 764 457 packed ints; every leaf is a valid block pointer or ANY_TYPE."""
    import os
    import pytest
    from chunkyclplugin_amd import octree2
    src = "/root/reference/benchmark/OpenCL_test/OpenCL_test.octree2"
    if not os.path.exists(src):
        pytest.skip("reference benchmark fixture not present on this machine")
    palette, depth, stream = octree2.read_octree2(src)
    assert (len(palette), depth) == (3091, 10)
    assert int((stream == -1).sum()) == 345557 and stream.size == 345557 + 2418900
    assert palette[0]["Name"] == "minecraft:air" and palette[1]["Name"] == "minecraft:stone"
    tree = octree2.pack_preorder(stream, len(palette))
    assert tree.size == 2764457
    leaves = -tree[tree <= 0].astype(np.int64)
    ok = (leaves == scenes.ANY_TYPE) | ((leaves % 2 == 0) & (leaves < 2 * 3091))
    assert ok.all() and int((leaves == scenes.ANY_TYPE).sum()) == 312369
    # first leaf cell of the stream round-trips through the packed layout
    sc = octree2.cached_benchmark_scene(64, 36)
    assert sc.octree.size == tree.size and sc.octree_depth == 10


def test_asset_pack_gives_palette_entries_their_model_class():
    """octree2.asset_pack (row f1): a palette entry's NAME and PROPERTIES decide its model — the geometry Minecraft gives the
    class — and every block name gets one texture; entries of one class and material share a model."""
    from chunkyclplugin_amd import octree2
    cls, boxes = octree2._model_class, octree2._model_boxes
    assert boxes(cls("stone_brick_slab", {"type": "top"})) == [(0, 1, 0.5, 1, 0, 1)]
    assert boxes(cls("stone_brick_slab", {"type": "double"})) is None                       # a full cube
    assert boxes(cls("spruce_stairs", {"half": "bottom", "facing": "east"})) == [(0, 1, 0, 0.5, 0, 1), (0.5, 1, 0.5, 1, 0, 1)]
    assert boxes(cls("spruce_stairs", {"half": "top", "facing": "north"})) == [(0, 1, 0.5, 1, 0, 1), (0, 1, 0, 0.5, 0, 0.5)]
    fence = boxes(cls("oak_fence", {"north": "true", "east": "false", "south": "false", "west": "true"}))
    assert fence[0] == (0.375, 0.625, 0, 1, 0.375, 0.625) and len(fence) == 3               # post + two arms
    assert boxes(cls("iron_bars", {}))[0] == (0.4375, 0.5625, 0, 1, 0.4375, 0.5625)
    assert boxes(cls("rail", {})) == [(0, 1, 0, 0.0625, 0, 1)] and boxes(cls("oak_trapdoor", {"half": "top"})) == [(0, 1, 0.8125, 1, 0, 1)]
    assert cls("poppy", {}) == ("plant",) and cls("stone", {}) == ("cube",) and boxes(cls("stone", {})) is None
    palette = [{"Name": "minecraft:air"}, {"Name": "minecraft:stone"}, {"Name": "minecraft:oak_slab", "Properties": {"type": "bottom"}},
               {"Name": "minecraft:oak_slab", "Properties": {"type": "bottom", "waterlogged": "true"}}, {"Name": "minecraft:poppy"},
               {"Name": "minecraft:oak_slab", "Properties": {"type": "double"}}]
    pal, atlas, recs, tsun = octree2.asset_pack(palette)
    blocks, mats, aabbs, quads = pal.arrays()
    b = blocks.reshape(-1, 2)
    assert b[:, 0].tolist() == [0, 1, 2, 2, 3, 1]                                            # invisible, cube, AABB, AABB, quads, cube
    assert b[2, 1] == b[3, 1] and aabbs[0] == 1 and quads[0] == 4                            # the two bottom slabs share one model
    assert mats.size == 6 * 3 and (mats.reshape(-1, 6)[:, 0] == 4).all()                     # three names, all textured
    assert atlas.shape[1] % 16 == 0 and recs[tsun][0] == (32 << 16) | 32


def test_benchmark_fixture_is_what_the_loader_makes_of_the_reference_files():
    """Where the reference's data files exist: the committed fixture equals load_scene's output array for array."""
    import os
    import pytest
    from chunkyclplugin_amd import octree2
    if not os.path.exists(octree2.REFERENCE_SCENE + ".octree2"):
        pytest.skip("reference benchmark files not present on this machine")
    made = octree2.load_scene(octree2.REFERENCE_SCENE + ".octree2", octree2.REFERENCE_SCENE + ".json")
    kept = octree2.cached_benchmark_scene(made.width, made.height)
    for f in ("octree", "block_palette", "material_palette", "aabb_models", "quad_models", "atlas", "sky", "sun", "camera"):
        np.testing.assert_array_equal(np.asarray(getattr(made, f)), np.asarray(getattr(kept, f)), err_msg=f)
    kinds = kept.block_palette.reshape(-1, 2)[:, 0]
    assert (kinds == 2).sum() > 1000 and (kinds == 3).sum() >= 20 and (kinds == 1).sum() > 1500   # model blocks are part of the city now


def test_slab_octree_builder_equals_the_dense_builder():
    """build_octree_slab (the beyond-cache world of bench.py --config 5 is 2048 x 256 x 2048: its dense cube would be 34 GB) packs
    a slab exactly as build_octree packs the same world padded with air: same merging, same breadth-first numbering."""
    rng = np.random.default_rng(3)
    n, h, depth = 64, 16, 6
    t = np.zeros((n, h, n), np.int16)
    hm = (rng.random((n, n)) * 12).astype(int)
    for x in range(n):
        for z in range(n):
            t[x, :hm[x, z] + 1, z] = 1 + (x * 7 + z) % 3
    t[5:9, 2:4, 5:9] = 255
    t[16:32, 0:16, 32:48] = 2          # a whole 16^3 cell: merges one level above the slab's thickness
    dense = np.zeros((n, n, n), np.int32)
    dense[:, :h, :] = t
    dense[dense == 255] = scenes.ANY_TYPE
    np.testing.assert_array_equal(scenes.build_octree_slab(t, depth, 255), scenes.build_octree(dense, depth, "bfs"))
    # a slab as thick as the world is wide, and an all-air world
    cube = (rng.random((8, 8, 8)) < 0.3).astype(np.int16)
    np.testing.assert_array_equal(scenes.build_octree_slab(cube, 3, 255), scenes.build_octree(cube.astype(np.int32), 3, "bfs"))
    assert list(scenes.build_octree_slab(np.zeros((8, 2, 8), np.int16), 3, 255)) == [0]


def test_big_world_generator_small():
    """big_outdoor_world at 16 x 16 chunks: the outdoor world's palettes, a depth-8 octree whose wide re-layout answers every sampled
    cell as the reference walk does."""
    from chunkyclplugin_amd import native
    sc = scenes.big_outdoor_world(chunks=16, width=64, img_height=48)
    small = scenes.outdoor_world(chunks=1, height=16, width=64, img_height=48)
    assert sc.octree_depth == 8 and np.array_equal(sc.block_palette, small.block_palette) and np.array_equal(sc.atlas, small.atlas)
    rng = np.random.default_rng(5)
    xyz = rng.integers(0, 256, size=(4000, 3)).astype(np.int32)
    data, level, _n = native.widetree_lookup(sc.octree, sc.octree_depth, xyz)
    for (x, y, z), d, l in zip(xyz[:600], data[:600], level[:600]):
        lv, v = sc.octree_depth, int(sc.octree[0])
        while v > 0:
            lv -= 1
            v = int(sc.octree[v + ((((x >> lv) & 1) << 2) | (((y >> lv) & 1) << 1) | ((z >> lv) & 1))])
        assert (-v, lv) == (int(d), int(l))
