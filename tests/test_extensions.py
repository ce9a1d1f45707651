"""SURVEY.md section 8 row f2 — the light-transport options the north star names and the reference lacks (specular /
metal / roughness from material word 5, emitter next-event estimation, sunEnabled / emittersEnabled).  There is no
reference to pin them to: oracle/port.c `trace_sample_ext` IS the specification, and these tests
check that specification analytically.  The GPU half (tests/test_gpu_extensions.py) then demands bit equality with it.

* with every material's word 5 at zero and the options at the values that mean "as the reference", the extended
  integrator reproduces the reference integrator bit for bit (it is a different function: this pins its common part);
* white furnace: no sample ever exceeds the radiance of the uniform sky, whatever spec / metal / rough are;
* rough = 0, spec = 255 is a perfect mirror: the image equals the sky seen along the reflected rays;
* emitter NEE on and off converge to the same image mean, and NEE has the lower variance;
* emittersEnabled = false equals an emitter scale of 0; sun sampling forced on equals the reference on a scene whose sun
  texture is drawn."""
import dataclasses

import numpy as np
import pytest

import golden_scenes as gs
from chunkyclplugin_amd import scenes
from oracle.binding import PortExt, PortOptions


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("name", ["outdoor", "indoor_sun", "entities", "water"])
def test_extended_integrator_reduces_to_the_reference(port, name):
    sc = gs.make(name)                                   # every material of the generators has word 5 = 0
    assert not np.asarray(sc.material_palette)[5::6].any()
    seeds = scenes.java_random_ints(3)
    want = port.render_passes(sc, seeds)
    with PortExt(port, sc, bsdf=1):                      # a different code path, no extra draws when max(spec, metal) = 0
        got = port.render_passes(sc, seeds)
    np.testing.assert_array_equal(bits(got), bits(want))
    if int(sc.sun[0]) & 1:
        with PortExt(port, sc, sun_sampling=1):          # forced on == the reference's own gate when the flag is set
            got = port.render_passes(sc, seeds)
        np.testing.assert_array_equal(bits(got), bits(want))


def furnace(spec_words, albedo_byte=255, size=16, view=(64, 48)):
    """A few cubes with the given material words 5 floating in a uniform white sky (no sun)."""
    rng = np.random.default_rng(5)
    ab = scenes.AtlasBuilder(4, 4)
    tex = ab.add(np.full((16, 16, 4), albedo_byte, np.uint8))
    tsun = ab.add(np.full((16, 16, 4), 255, np.uint8))
    atlas, recs = ab.build()
    pal = scenes.Palettes()
    blocks = [pal.block_invisible()]
    for w in spec_words:
        blocks.append(pal.block_cube(pal.material(texture=recs[tex], spec=int(w))))
    depth = int(np.log2(size))
    types = np.zeros((size, size, size), np.int32)
    cells = rng.integers(2, size - 2, (60, 3))
    for i, (x, y, z) in enumerate(cells):
        types[x, y, z] = blocks[1 + i % len(spec_words)]
    types[:, 1, :] = blocks[1]                            # a floor
    octree = scenes.build_octree(types, depth)
    b, m, a, q = pal.arrays()
    sky = np.full((8, 8, 4), 255, np.uint8)
    cam = scenes.look_at_camera((size * 0.5, size * 0.8, -size * 0.4), (size * 0.5, size * 0.3, size * 0.5), 70.0)
    return scenes.PackedScene(octree=octree, octree_depth=depth, block_palette=b, material_palette=m, aabb_models=a, quad_models=q,
                              world_bvh=scenes.empty_bvh(), actor_bvh=scenes.empty_bvh(), bvh_trigs=np.zeros(1, np.int32),
                              atlas=atlas, sky=sky, sky_intensity=1.0, sun=scenes.pack_sun(0.6, 1.2, 1.0, False, recs[tsun]), camera=cam,
                              width=view[0], height=view[1], name="furnace")


def word5(spec, metal, rough):
    return spec | (metal << 8) | (rough << 16)


def test_white_furnace_energy_conservation(port):
    """A uniform sky of radiance 1 and white surfaces: the radiance arriving anywhere is 1, so a path that escapes carries
    exactly its throughput (<= 1) and one cut off at the depth limit carries 0: no sample may exceed 1, for any
    spec / metal / rough; with albedo < 1 the image can only get darker."""
    words = [0, word5(255, 0, 0), word5(128, 0, 64), word5(0, 255, 0), word5(40, 200, 255), word5(255, 255, 128)]
    sc = furnace(words)
    seeds = scenes.java_random_ints(24)
    with PortExt(port, sc, sun_sampling=0, bsdf=1):
        per_pass = np.stack([port.render_passes(sc, [s]) for s in seeds])      # one sample per pixel each
        grey = furnace(words, albedo_byte=180)
        with PortOptions(port, 256, 12, 13.0):
            deep = port.render_passes(sc, seeds)
        dark = port.render_passes(grey, seeds)
    assert per_pass.max() <= 1.0 + 1e-5
    assert per_pass.min() >= 0.0
    escaped = per_pass[per_pass > 0]
    assert np.abs(escaped - 1.0).max() < 1e-5           # white surfaces: a sample is 0 or the full sky
    assert deep.mean() > np.mean(per_pass) and deep.max() <= 1.0 + 1e-5   # deeper paths: fewer zeros, never more than 1
    assert dark.max() <= 1.0 + 1e-5 and dark.mean() < np.mean(per_pass)


def test_mirror_limit(port):
    """spec = 255, rough = 0: the hit reflects d about n exactly and keeps its throughput, so a mirror floor under a
    patterned sky shows the sky along the reflected rays — rendered here as a second, empty scene whose pre-generated rays
    ARE those reflected rays."""
    size, W, H = 16, 40, 30
    rng = np.random.default_rng(9)
    ab = scenes.AtlasBuilder(2, 2)
    tex = ab.add(np.full((16, 16, 4), 255, np.uint8))
    atlas, recs = ab.build()
    pal = scenes.Palettes()
    air, mirror = pal.block_invisible(), pal.block_cube(pal.material(texture=recs[tex], spec=word5(255, 0, 0)))
    types = np.zeros((size, size, size), np.int32)
    types[:, 2, :] = mirror
    b, m, a, q = pal.arrays()
    sky = rng.integers(0, 256, (32, 32, 4)).astype(np.uint8)
    sky[..., 3] = 255
    o = np.array([8.0, 9.0, 8.0], np.float32)
    d = np.stack([rng.uniform(-0.6, 0.6, W * H), rng.uniform(-1.0, -0.5, W * H), rng.uniform(-0.6, 0.6, W * H)], 1).astype(np.float32)
    rays = np.concatenate([np.tile(o, (W * H, 1)), d], 1).astype(np.float32)
    common = dict(octree_depth=4, block_palette=b, material_palette=m, aabb_models=a, quad_models=q, world_bvh=scenes.empty_bvh(),
                  actor_bvh=scenes.empty_bvh(), bvh_trigs=np.zeros(1, np.int32), atlas=atlas, sky=sky, sky_intensity=1.0,
                  sun=scenes.pack_sun(0.6, 1.2, 1.0, False), width=W, height=H, projector_type=-1)
    floor = scenes.PackedScene(octree=scenes.build_octree(types, 4), camera=rays.reshape(-1), name="mirror", **common)
    with PortExt(port, floor, sun_sampling=0, bsdf=1):
        got = port.render_passes(floor, [123]).reshape(-1, 3)
    # the reflected rays: the floor's top face is y = 3, normal (0, 1, 0)
    t = (3.0 - o[1]) / d[:, 1]
    hitp = o[None, :] + d * t[:, None]
    refl = d.copy()
    refl[:, 1] = -d[:, 1]
    rays2 = np.concatenate([hitp + refl * 1e-4, refl], 1).astype(np.float32)
    empty = scenes.PackedScene(octree=scenes.build_octree(np.zeros((size, size, size), np.int32), 4), camera=rays2.reshape(-1),
                               name="empty", **common)
    want = port.render_passes(empty, [123]).reshape(-1, 3)
    inside = (np.abs(hitp[:, 0] - 8) < 7.9) & (np.abs(hitp[:, 2] - 8) < 7.9)
    assert inside.mean() > 0.9
    np.testing.assert_allclose(got[inside], want[inside], rtol=2e-5, atol=1e-6)


@pytest.fixture(scope="module")
def lit_room():
    return scenes.indoor_room(size=16, seed=3, width=40, img_height=30, emitter_frac=0.04)


def test_emitter_nee_converges_to_the_same_mean_with_less_noise(port, lit_room):
    """Same expectation, different estimator: in a dim room with a dozen emitter blocks the implicit path finds a light
    rarely, next-event estimation at every diffuse vertex finds one often — equal means, about half the noise.  (In a
    room crowded with emitters the implicit hits are frequent already and the two are equally noisy: only the means are
    compared there.)"""
    dim = scenes.indoor_room(size=24, seed=3, width=40, img_height=30, emitter_frac=0.003)
    for sc, n_min, quieter in ((dim, 8, True), (lit_room, 30, False)):
        seeds = scenes.java_random_ints(192)
        half = len(seeds) // 2
        with PortExt(port, sc) as e:
            assert e.n_emitters >= n_min
        off = [port.render_passes(sc, seeds[i * half:(i + 1) * half]) for i in range(2)]   # the reference path: implicit hits only
        with PortExt(port, sc, nee=1):
            on = [port.render_passes(sc, seeds[i * half:(i + 1) * half]) for i in range(2)]
        m_off, m_on = float(np.mean(off)), float(np.mean(on))
        assert m_off > 0.02
        assert abs(m_on - m_off) / m_off < 0.05, (m_on, m_off)
        n_off = np.abs(off[0] - off[1]).mean()      # noise: two independent halves of the sample set against each other
        n_on = np.abs(on[0] - on[1]).mean()
        if quieter:
            assert n_on < 0.7 * n_off, (n_on, n_off)


def test_emitters_disabled_and_sun_switches(port, lit_room):
    sc = lit_room
    seeds = scenes.java_random_ints(4)
    with PortOptions(port, 256, 5, 0.0):
        scale0 = port.render_passes(sc, seeds)
    with PortExt(port, sc, emitters=0):
        off = port.render_passes(sc, seeds)
    np.testing.assert_array_equal(bits(off), bits(scale0))
    assert port.render_passes(sc, seeds).mean() > 4 * off.mean() + 1e-6
    # sunEnabled = false on an outdoor scene: darker than the reference, identical to a scene whose sun flag is clear
    # except for the sun DISC, which drawTexture keeps painting on the sky
    out = gs.make("outdoor")
    ref = port.render_passes(out, seeds)
    with PortExt(port, out, sun_sampling=0):
        nosun = port.render_passes(out, seeds)
    assert nosun.mean() < ref.mean()
    flagless = dataclasses.replace(out, sun=scenes.pack_sun(0.6, 1.2, 1.25, False, (int(out.sun[1]), int(out.sun[2]))))
    with PortExt(port, flagless, sun_sampling=1):
        forced = port.render_passes(flagless, seeds)
    assert forced.mean() > port.render_passes(flagless, seeds).mean()
