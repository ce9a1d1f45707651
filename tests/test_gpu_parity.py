"""Parity of the HIP path (through the C ABI) with the oracle, on a real MI355X.

Bars (BASELINE.json north_star): integer block-hit / material indices bit-exact; per-pixel radiance
within 1e-5 relative.  Because both sides compute with the same IEEE contract (rt_math.h, no
contraction) the radiance tests below demand bit equality and report the relative error if that
ever fails."""
import os

import numpy as np
import pytest

import golden_scenes as gs
from chunkyclplugin_amd import native, scenes
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader
from oracle import binding

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
REL_TOL = 1e-5  # north_star radiance tolerance


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_radiance(got, want, what):
    got, want = np.asarray(got, np.float32).reshape(-1), np.asarray(want, np.float32).reshape(-1)
    same = bits(got) == bits(want)
    if same.all():
        return
    rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-6)
    bad = int((~same).sum())
    assert np.nanmax(rel) <= REL_TOL, f"{what}: {bad} values differ, max rel err {np.nanmax(rel):.3e}"
    pytest.fail(f"{what}: within {REL_TOL} but not bit-exact ({bad} values differ) — the arithmetic contract is broken")


# CHUNKY_OPT_KERNEL variants that must all be bit-identical: 0 = default (render_pool, 64 parked paths per wave, wide-tree
# lookup), bit 0 = the reference-layout octree walk of K/octree.h:81-89, bit 1 = one lane per path (render_lanes),
# bit 3 = the fallback kernel render_waves (bits 4-5 = its lanes per pixel forced to 1 / 8 / 16: test_pixel_groups covers them
# all), bits 6-7 = render_pool with no / 32 parked paths, bit 8 / bit 9 = render_pool testing full cubes and model blocks in phases
# of their own always / never (by default: where model blocks are common)
VARIANTS = [0, 1, 2, 3, 64, 128, 8, 9, 8 | 16, 256, 512]


def make_renderer(gpu_instance, sc, variant=0):
    loader = HipSceneLoader(gpu_instance)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    r.set_option(native.OPT_KERNEL, variant)
    return loader, r


# ---- the arithmetic contract itself, device vs host --------------------------------------------
@pytest.mark.parametrize("which", range(16))
def test_device_math_bit_equals_host(gpu_instance, port, which):
    rng = np.random.default_rng(which)
    n = 1 << 16
    if which in (2, 3):
        a = rng.uniform(-1.05, 1.05, n)
    elif which in (0, 1):
        a = rng.uniform(-8, 8, n)
    elif which == 13:
        a = rng.uniform(-1e6, 1e6, n)
    else:
        a = rng.normal(size=n) * 10.0 ** rng.integers(-3, 4, n)
    b = rng.normal(size=n) * 10.0 ** rng.integers(-3, 4, n)
    a, b = a.astype(np.float32), b.astype(np.float32)
    specials = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 0.5, -0.5, 1e-38, 3e38], np.float32)
    if which != 13:
        a[:121] = np.repeat(specials, 11)
        b[:121] = np.tile(specials, 11)
    if which in (0, 1):
        # rt_sincos is specified for |x| <= 1e4 (rt_math.h): beyond that the quadrant index overflows
        a = np.where(np.abs(a) > 1e4, np.float32(1e4), a).astype(np.float32)
    dev = gpu_instance.selftest_math(which, a, b)
    if which <= 9:
        host = port.math(which, a, b)
    elif which == 10:
        host = (np.float32(1) / np.sqrt(_fma3(a, b)))
    elif which == 11:
        host = _dot_fma(a, b)
    elif which == 12:
        host = np.floor(a)
    elif which == 13:
        host = np.trunc(a).astype(np.float32)
    elif which == 14:
        with np.errstate(invalid="ignore"):
            ai = np.where(np.isfinite(a) & (np.abs(a) < 2e9), a, 0).astype(np.int32)
        a = ai.astype(np.float32)
        dev = gpu_instance.selftest_math(which, a, b)
        host = ((ai.astype(np.uint32) & 0xFF).astype(np.float64) / 255.0).astype(np.float32)
    else:
        host = (-0.5 + (a * b).astype(np.float64)).astype(np.float32)
    ok = (bits(dev) == bits(host)) | (np.isnan(dev) & np.isnan(host))
    assert ok.all(), (which, a[~ok][:4], b[~ok][:4], dev[~ok][:4], host[~ok][:4])


def test_device_floor_to_int(gpu_instance):
    """v_cvt_flr_i32_f32 (self test 18) == (int)floor(x), saturating outside the int range; NaN gives INT_MAX
    (a cell outside any world, like the INT_MIN the x86 build of the reference produces there)."""
    rng = np.random.default_rng(18)
    a = np.concatenate([rng.uniform(-600, 600, 40000), rng.normal(size=20000) * 10.0 ** rng.integers(-8, 12, 20000),
                        np.arange(-70, 70) * 0.5, [0.0, -0.0, np.inf, -np.inf, np.nan, 2147483520.0, -2147483648.0, 3e9, -3e9,
                                                  0.99999994, -1e-45, 1e-45, 8388607.5, -8388607.5]]).astype(np.float32)
    dev = gpu_instance.selftest_math(18, a, a)
    with np.errstate(invalid="ignore"):
        want = np.clip(np.floor(a.astype(np.float64)), -2147483648.0, 2147483647.0)
    want = np.where(np.isnan(a), 2147483647.0, want).astype(np.int64).astype(np.float32)
    np.testing.assert_array_equal(dev, want)


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)  # exact product in f64


def _dot_fma(x, y):
    # rt_dot3(x, y, x, y, x, y) = fma(x, y, fma(y, x, x*y))
    with np.errstate(all="ignore"):
        return _fma(x, y, _fma(y, x, (x * y).astype(np.float32)))


def _fma3(x, y):
    with np.errstate(all="ignore"):
        return _fma(x, x, _fma(y, y, (x * x).astype(np.float32)))


# ---- golden images: outputs of the reference kernel itself -------------------------------------
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", gs.NAMES)
def test_render_matches_reference_goldens(gpu_instance, name, variant):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sc = gs.make(name)
    assert gs.input_digest(sc) == str(g["digest"])
    loader, r = make_renderer(gpu_instance, sc, variant)
    r.render_passes(g["seeds"])
    assert_radiance(r.read(), g["res"], f"{name} res")
    np.testing.assert_array_equal(r.preview(), g["preview"])          # integer image: exact
    rec, cnt, rad = r.trace_records(int(g["seeds"][0]), gs.RECORD_GIDS)
    np.testing.assert_array_equal(cnt, g["counts"])
    for i in range(len(gs.RECORD_GIDS)):
        n = int(cnt[i])
        got, want = rec[i, :n], g["records"][i, :n]
        assert got["hit"].tolist() == want["hit"].tolist(), (name, i)
        assert got["material"].tolist() == want["material"].tolist(), (name, i)   # block indices: exact
        for f in ("distance", "normal", "color", "emittance"):
            assert_radiance(got[f], want[f], f"{name} gid {gs.RECORD_GIDS[i]} {f}")
        hit = want["hit"] == 1
        assert_radiance(got["point"][hit], want["point"][hit], f"{name} point")
    assert_radiance(rad, g["radiance"], f"{name} radiance")
    r.close()
    loader.close()


# ---- seeded inputs vs the CPU oracle -----------------------------------------------------------
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name,w,h,passes,first", [("outdoor", 160, 96, 6, 0), ("entities", 128, 80, 3, 4),
                                                    ("indoor_sun", 128, 80, 5, 0), ("inside", 96, 64, 4, 1)])
def test_render_matches_oracle(gpu_instance, port, name, w, h, passes, first, variant):
    sc = gs.make(name).with_view(w, h)
    seeds = scenes.java_random_ints(passes + 2)[2:]
    want = port.render_passes(sc, seeds, first_spp=first)
    loader, r = make_renderer(gpu_instance, sc, variant)
    r.render_passes(seeds, first_buffer_spp=first)
    assert_radiance(r.read(), want, name)
    np.testing.assert_array_equal(r.preview(), port.preview(sc))
    r.close()
    loader.close()


def test_launch_chunking_is_invisible(gpu_instance, port):
    """n passes in one call == n calls of one pass (the reference's one launch per spp), including
    more passes than one launch carries (256)."""
    sc = gs.make("outdoor").with_view(48, 30)
    seeds = scenes.java_random_ints(262)
    loader, r = make_renderer(gpu_instance, sc)
    r.render_passes(seeds)
    a = r.read()
    r.reset()
    for k, s in enumerate(seeds):
        r.render_passes([s], first_buffer_spp=k, sync=False)
    r.sync()
    b = r.read()
    np.testing.assert_array_equal(bits(a), bits(b))
    assert_radiance(a, port.render_passes(sc, seeds), "262 passes")
    r.close()
    loader.close()


@pytest.mark.parametrize("group,variant", [(1, 8 | 16), (8, 8 | 32), (16, 8 | 48), (32, 8)])
def test_every_group_size_gives_the_same_image(gpu_instance, port, group, variant):
    """render_waves' lanes per pixel (variant bits 4-5 force 1 / 8 / 16; a small image with 64 passes gets 32 by itself) only
    changes who computes which pass: bit-identical."""
    sc = gs.make("outdoor").with_view(96, 54)
    seeds = scenes.java_random_ints(64)
    loader, r = make_renderer(gpu_instance, sc, variant=variant)
    r.render_passes(seeds)
    assert r.kernel_info()["group"] == group and r.kernel_info()["pool"] < 0, r.kernel_info()
    assert_radiance(r.read(), port.render_passes(sc, seeds), f"group {group}")
    r.close()
    loader.close()


@pytest.mark.parametrize("name,draw,depth,scale", [("outdoor", 24, 3, 13.0), ("indoor", 256, 2, 2.5), ("indoor_sun", 256, 5, 0.0),
                                                  ("entities", 40, 4, 7.0)])
def test_render_loop_options_match_oracle(gpu_instance, port, name, draw, depth, scale):
    """CHUNKY_OPT_DRAW_DEPTH / _MAX_DEPTH / _EMITTER_SCALE (the constants 256 / 5 / 13 of K/rayTracer.cl:94-107; a scale
    of 0 is Chunky's "emitters off") against the C restatement run with the same values."""
    from oracle.binding import PortOptions
    sc = gs.make(name).with_view(96, 60)
    seeds = scenes.java_random_ints(3)
    loader, r = make_renderer(gpu_instance, sc)
    r.set_option(native.OPT_DRAW_DEPTH, draw)
    r.set_option(native.OPT_MAX_DEPTH, depth)
    r.set_option(native.OPT_EMITTER_SCALE, scale)
    r.render_passes(seeds)
    got = r.read()
    with PortOptions(port, draw, depth, scale):
        want = port.render_passes(sc, seeds)
    assert_radiance(got, want, f"{name} draw {draw} depth {depth} scale {scale}")
    assert not np.array_equal(bits(want), bits(port.render_passes(sc, seeds))), "the options changed nothing"
    r.close()
    loader.close()


def test_sorted_block_tests_run_where_asked(gpu_instance, port):
    """render_pool tests full cubes and model blocks in phases of their own where model blocks are common (from 30 per thousand of a
    world's leaves on: capi.hip model_leaf_permille — the timed city, tests/test_timed_goldens.py; the golden worlds hold 29 and 0);
    CHUNKY_OPT_KERNEL bit 8 / bit 9 force it.  Every combination renders the oracle's image."""
    seeds = scenes.java_random_ints(3)
    for name in ("outdoor", "outdoor_nosun", "indoor_sun"):
        sc = gs.make(name).with_view(64, 40)
        want = port.render_passes(sc, seeds)
        for variant, sorted_ in ((0, False), (256, True), (512, False), (256 | 512, False)):
            loader, r = make_renderer(gpu_instance, sc, variant)
            r.render_passes(seeds)
            info = r.kernel_info()
            assert info["pool"] == 64 and info["sorted"] == sorted_, (name, variant, info)
            assert_radiance(r.read(), want, f"{name}, variant {variant}")
            r.close()
            loader.close()


def test_draw_depth_above_16_bits(gpu_instance, port):
    """render_pool's parked record counts march steps in 16 bits: a draw depth above 65535 must run another kernel
    (render_waves) and still give the reference's image (the depth never binds in a world this small)."""
    sc = gs.make("outdoor").with_view(64, 40)
    seeds = scenes.java_random_ints(2)
    loader, r = make_renderer(gpu_instance, sc)
    r.set_option(native.OPT_DRAW_DEPTH, 65535)
    r.render_passes(seeds)
    assert r.kernel_info()["pool"] >= 0
    r.reset()
    r.set_option(native.OPT_DRAW_DEPTH, 1 << 20)
    r.render_passes(seeds)
    assert r.kernel_info()["pool"] < 0, r.kernel_info()
    assert_radiance(r.read(), port.render_passes(sc, seeds), "draw depth 2^20")
    r.close()
    loader.close()


def test_path_depth_255_marks_a_fresh_path_so_deeper_renders_take_another_kernel(gpu_instance, port):
    """render_pool encodes "this lane holds no path" as path depth 255 (round 6: the state of a path is then its class): a maximum
    depth of 254 still runs it, 255 — the most the option takes — runs render_waves, and both give the C restatement's image at that
    depth (an indoor room: every path bounces until the limit)."""
    from oracle.binding import PortOptions
    sc = gs.make("indoor_sun").with_view(40, 24)
    seeds = scenes.java_random_ints(2)
    loader, r = make_renderer(gpu_instance, sc)
    for depth, pool in ((254, True), (255, False)):
        r.reset()
        r.set_option(native.OPT_MAX_DEPTH, depth)
        r.render_passes(seeds)
        assert (r.kernel_info()["pool"] >= 0) == pool, (depth, r.kernel_info())
        with PortOptions(port, max_depth=depth):
            want = port.render_passes(sc, seeds)
        assert_radiance(r.read(), want, f"max depth {depth}")
    r.close()
    loader.close()


@pytest.mark.parametrize("seed,size,entities,sun", [(11, 16, 0, True), (12, 16, 60, True), (13, 32, 30, False),
                                                       (14, 48, 0, True), (15, 16, 24, True), (16, 32, 0, False)])
def test_seeded_small_worlds_match_oracle(gpu_instance, port, seed, size, entities, sun):
    """Small worlds from other seeds than the golden ones (dense in slab and plant models, emitters, a few entities;
    octree depths 5-7, so the one- and two-level forms of the wide tree): bit-identical to the C restatement."""
    sc = scenes.tiny_scene(seed=seed, size=size, width=72, height=48, entities=entities, sun_flag=sun)
    seeds = scenes.java_random_ints(5, seed=seed)
    loader, r = make_renderer(gpu_instance, sc)
    r.render_passes(seeds)
    assert_radiance(r.read(), port.render_passes(sc, seeds), f"seed {seed} size {size}")
    np.testing.assert_array_equal(r.preview(), port.preview(sc))
    r.close()
    loader.close()


def test_reference_benchmark_scene_matches_oracle(gpu_instance, port):
    """The reference's own benchmark octree (depth 10: a 16^3 top node over two 8^3 levels), its saved camera,
    a small view: bit-identical to the C restatement."""
    from chunkyclplugin_amd import octree2
    sc = octree2.cached_benchmark_scene(160, 90)  # committed fixture tests/golden/benchmark_OpenCL_test.npz: never skipped
    assert sc.octree_depth == 10
    seeds = scenes.java_random_ints(4)
    loader, r = make_renderer(gpu_instance, sc)
    r.render_passes(seeds)
    assert_radiance(r.read(), port.render_passes(sc, seeds), "benchmark/OpenCL_test")
    r.close()
    loader.close()


def test_edge_cases(gpu_instance, port):
    sc = gs.make("outdoor").with_view(33, 17)          # ragged: not a multiple of the block or tile size
    loader, r = make_renderer(gpu_instance, sc)
    r.render_passes([])                                 # empty pass list is legal and does nothing
    assert not r.read().any()
    r.render_passes([7])
    assert_radiance(r.read(), port.render_passes(sc, [7]), "ragged")
    # draw depth 0: every trace misses -> pure sky (K/octree.h:66)
    r.reset()
    r.set_option(native.OPT_DRAW_DEPTH, 0)
    r.render_passes([7])
    sky_only = r.read()
    assert np.isfinite(sky_only).all() and sky_only.mean() > 0
    with pytest.raises(native.ChunkyHipError):
        r.set_camera(3, np.zeros(15, np.float32))       # unsupported projector
    with pytest.raises(native.ChunkyHipError):
        r.set_camera(0, np.zeros(14, np.float32))
    r.close()
    # zero-length palettes are legal (ClIntBuffer.java:15-18)
    L = native.lib()
    native.check(L.chunky_scene_set_palette(loader._h, native.PALETTE_QUAD, None, 0))
    with pytest.raises(native.ChunkyHipError):
        bad = np.array([5, 0, 0], np.int32)             # branch pointer outside the array
        native.check(L.chunky_scene_set_octree(loader._h, bad.ctypes.data, 3, 2))
    loader.close()


def test_native_octree_remap(gpu_instance, port):
    """chunky_scene_load_octree = the leaf remap of ClSceneLoader.java:52-63."""
    sc = gs.make("outdoor")
    raw = sc.octree.copy()
    leaf = (raw <= 0) & (raw != -scenes.ANY_TYPE)
    raw[leaf] = raw[leaf] // 2                          # Chunky's treeData holds -paletteIndex
    mapping = (2 * np.arange(len(sc.block_palette) // 2)).astype(np.int32)
    loader, r = make_renderer(gpu_instance, sc)
    loader.load_octree(raw, sc.octree_depth, mapping)
    r.render_passes([11])
    assert_radiance(r.read(), port.render_passes(sc, [11]), "remap")
    r.close()
    loader.close()


# ---- tiles: N shards sum to the 1-GPU image bit for bit ----------------------------------------
@pytest.mark.parametrize("world,tile", [(2, 256), (3, 64), (8, 256)])
def test_shards_sum_to_full_image(gpu_instance, world, tile):
    sc = gs.make("outdoor").with_view(100, 60)
    seeds = scenes.java_random_ints(3)
    loader, r = make_renderer(gpu_instance, sc)
    r.render_passes(seeds)
    full = r.read()
    total = np.zeros_like(full)
    owned = np.zeros(full.size // 3, np.int32)
    for rank in range(world):
        r.set_shard(rank, world, tile)
        r.reset()
        r.render_passes(seeds)
        part = r.read()
        owned += (part.reshape(-1, 3) != 0).any(axis=1)
        total += part                                    # what the RCCL SUM reduce does
    assert owned.max() <= 1
    np.testing.assert_array_equal(bits(total), bits(full))
    r.close()
    loader.close()


# ---- host pass loop ----------------------------------------------------------------------------
def test_render_run_matches_reference_host_loop(gpu_instance, port):
    """chunky_render_run vs the loop of OpenClPathTracingRenderer.java:95-184 emulated with the
    oracle: seeds from Random(0), bufferSpp restarting after each merge, double merge."""
    sc = gs.make("indoor").with_view(48, 32)
    target, interval = 10, 4
    loader, r = make_renderer(gpu_instance, sc)
    sample = np.zeros(sc.width * sc.height * 3, np.float64)
    spp = r.render(sample, 0, target, merge_interval=interval)
    assert spp == target
    seeds = scenes.java_random_ints(target)
    want = np.zeros_like(sample)
    done = 0
    while done < target:
        m = min(interval, target - done)
        pass_buf = port.render_passes(sc, seeds[done:done + m]).astype(np.float64)
        want = (want * done + pass_buf * m) * (1.0 / (done + m))
        done += m
    np.testing.assert_array_equal(sample, want)
    # postRender returning true stops the loop
    calls = []
    r.set_post_render(lambda: calls.append(1) or True)
    spp2 = r.render(sample, spp, spp + 100, merge_interval=interval)
    assert calls and spp2 < spp + 100
    r.close()
    loader.close()


def test_render_run_ex_hooks(gpu_instance, port):
    """chunky_render_run_ex: a save event (isSaveEvent, OpenClPathTracingRenderer.java:150) cuts the launch at its spp and
    forces a merge there; `merged` sees every merge, `progress` every launch, `regenerate_camera` runs between launches
    (here it installs the same pinhole camera again, :146-148).  Image = the oracle driven through the same merges."""
    sc = gs.make("indoor").with_view(48, 32)
    target, interval, save_at = 12, 5, 7
    loader, r = make_renderer(gpu_instance, sc)
    sample = np.zeros(sc.width * sc.height * 3, np.float64)
    progress, merges, regen = [], [], []
    spp = r.render_ex(sample, 0, target, merge_interval=interval, progress=progress.append, merged=merges.append,
                      save_event=lambda s: s == save_at,
                      regenerate_camera=lambda: (regen.append(1), r.set_camera(sc.projector_type, sc.camera)))
    assert spp == target and merges == [5, 7, 12], (spp, merges)
    assert progress == sorted(set(progress)) and progress[-1] == target and save_at in progress
    assert len(regen) == len(progress)
    seeds = scenes.java_random_ints(target)
    want = np.zeros_like(sample)
    done = 0
    for upto in merges:
        m = upto - done
        pass_buf = port.render_passes(sc, seeds[done:upto]).astype(np.float64)
        want = (want * done + pass_buf * m) * (1.0 / (done + m))
        done = upto
    np.testing.assert_array_equal(sample, want)
    r.close()
    loader.close()


def test_max_depth_above_the_reference_constant(gpu_instance, port):
    """CHUNKY_OPT_MAX_DEPTH is a real option: 9 bounces against the C restatement run with the same constant."""
    from oracle.binding import PortOptions
    sc = gs.make("indoor_sun").with_view(80, 50)
    seeds = scenes.java_random_ints(4)
    loader, r = make_renderer(gpu_instance, sc)
    r.set_option(native.OPT_MAX_DEPTH, 9)
    r.render_passes(seeds)
    with PortOptions(port, 256, 9, 13.0):
        want = port.render_passes(sc, seeds)
    assert_radiance(r.read(), want, "max depth 9")
    with pytest.raises(native.ChunkyHipError):
        r.trace_records(1, [0])                       # the record array holds 10 traces per sample
    r.close()
    loader.close()


# ---- full-size, size-independent properties ----------------------------------------------------
def test_full_size_properties(gpu_instance):
    """BASELINE config 3 at full resolution: the oracle is too slow here, so check properties:
    determinism, pass-order independence of per-pass images, tiles == full, running-mean identity."""
    sc = scenes.cached_outdoor_world(chunks=32, height=256)
    loader, r = make_renderer(gpu_instance, sc)
    seeds = scenes.java_random_ints(2)
    r.render_passes(seeds)
    a = r.read()
    assert np.isfinite(a).all() and a.min() >= 0
    r.reset()
    r.render_passes(seeds)
    np.testing.assert_array_equal(bits(a), bits(r.read()))            # deterministic
    # mean identity: res after passes (s0, s1) == (img(s0)*1 + img(s1)) / 2 computed in float
    r.reset(); r.render_passes(seeds[:1]); i0 = r.read()
    r.reset(); r.render_passes(seeds[1:]); i1 = r.read()
    np.testing.assert_array_equal(bits((i0 * np.float32(1) + i1) / np.float32(2)), bits(a))
    # a checksum of tile checksums equals the checksum of the full image
    total = np.zeros_like(a)
    for rank in range(4):
        r.set_shard(rank, 4, 256); r.reset(); r.render_passes(seeds); total += r.read()
    np.testing.assert_array_equal(bits(total), bits(a))
    r.close()
    loader.close()


@pytest.mark.parametrize("variant", [0, 1, 2, 8])
def test_aabb_plus_z_face_on_the_device(gpu_instance, port, variant):
    """+z faces of AABB models read an unset material in the reference (K/primitives.h:209-234); the reference build takes the
    EAST material (tests/test_oracle_pinning.py::test_aabb_plus_z_face pins that against the reference object).  The HIP
    kernels must do the same — through the aligned model records (default) and through the packed palettes (variant bit 0) —
    in the hit record and in the rendered image."""
    sc = gs.plus_z_scene()
    loader, r = make_renderer(gpu_instance, sc, variant)
    rec, cnt, _rad = r.trace_records(1, [gs.PLUS_Z_GID])
    assert cnt[0] >= 1 and rec[0, 0]["hit"] == 1 and rec[0, 0]["normal"].tolist() == [0, 0, 1]
    np.testing.assert_array_equal(rec[0, 0]["color"], gs.PLUS_Z_EAST)
    seeds = scenes.java_random_ints(6)
    r.render_passes(seeds)
    assert_radiance(r.read(), port.render_passes(sc, seeds), f"+z face image, variant {variant}")
    r.close()
    loader.close()


def test_run_callbacks_of_an_older_host(gpu_instance, port):
    """chunky_run_callbacks carries the caller's sizeof: a host compiled before `poll_gate` existed passes a shorter struct and
    the library must not read past it; a size of 0, or one that cuts a member in half, is refused."""
    import ctypes as C
    sc = gs.make("outdoor").with_view(40, 24)
    loader, r = make_renderer(gpu_instance, sc)
    L = native.lib()
    sample = np.zeros(sc.width * sc.height * 3, np.float64)
    merges = []
    keep = native.PROGRESS_FN(lambda _u, s: merges.append(s))

    class Guarded(C.Structure):  # the real struct followed by a word the library would call if it read `poll_gate` regardless
        _fields_ = [("cb", native.RunCallbacks), ("canary", C.c_void_p)]
    g = Guarded()
    g.cb.struct_size = native.RunCallbacks.poll_gate.offset   # the six-member struct of the previous header
    g.cb.merged = keep
    g.cb.poll_gate = C.cast(C.c_void_p(0xdeadbeef), native.POST_RENDER_FN)  # beyond struct_size: must never be called
    spp = C.c_int32(0)
    rc = L.chunky_render_run_ex(r._h, native.ptr(sample), C.byref(spp), 6, 4, C.cast(C.byref(g), C.POINTER(native.RunCallbacks)))
    assert rc == 0 and spp.value == 6 and merges == [4, 6]
    seeds = scenes.java_random_ints(6)
    want = port.render_passes(sc, seeds[:4]).astype(np.float64)
    want = (want * 4 + port.render_passes(sc, seeds[4:]).astype(np.float64) * 2) * (1.0 / 6)
    np.testing.assert_array_equal(sample, want)
    for bad in (0, 4, native.RunCallbacks.poll_gate.offset + 3):
        g.cb.struct_size = bad
        assert L.chunky_render_run_ex(r._h, native.ptr(sample), C.byref(spp), 8, 4, C.cast(C.byref(g), C.POINTER(native.RunCallbacks))) == native.E_INVALID
    r.close()
    loader.close()


@pytest.mark.parametrize("layout", ["5,3", "0,2", "4096,32"])
def test_entity_bvh_record_placement_is_invisible(layout):
    """Where the entity-BVH records sit in memory (CHUNKY_BVH_LAYOUT: a breadth-first top over depth-first treelets,
    capi.hip relayout_bvh_records) changes addresses only: image and per-trace records stay the reference's.  The variable is
    read by the -DCHUNKY_TUNING build only (the shipping library reads no tuning variable): a child process loads that build."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, CHUNKY_HIP_LIB=native.build_tuning(), CHUNKY_BVH_LAYOUT=layout)
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning_child.py"), "entities"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["library"] == "libchunky_hip_tuning.so" and out["kernel"]["bvh"] and out["kernel"]["pool"] > 0, out
    assert out["identical"] and out["records_identical"], out


@pytest.mark.timeout(120)
@pytest.mark.parametrize("variant", [24, 8, 40, 2])
def test_fallback_kernels_at_256_passes_per_launch(gpu_instance, port, variant):
    """The most passes a launch of the fallback kernels carries.  render_waves with one lane per pixel (variant 8 | 16) keeps the
    pass index in an 8-bit field and used to count before comparing: at exactly 256 passes the index wrapped and the launch
    never ended (found by the fuzz of round 4).  256 and 300 passes (two launches) against the oracle."""
    sc = scenes.tiny_scene(seed=5, size=32, width=64, height=30, entities=24)
    loader, r = make_renderer(gpu_instance, sc, variant)
    for passes in (256, 300):
        seeds = scenes.java_random_ints(passes)
        r.reset()
        r.render_passes(seeds)
        assert r.kernel_info()["pool"] < 0
        assert_radiance(r.read(), port.render_passes(sc, seeds), f"variant {variant}, {passes} passes")
    r.close()
    loader.close()
