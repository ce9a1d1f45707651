"""The EXPERIMENTAL light-transport options on the GPU: the HIP path (render_pool's extended
instantiations, through the C ABI) against their specification, oracle/port.c trace_sample_ext — bit for bit, like
everything else.  tests/test_extensions.py checks the specification itself analytically (CPU)."""
import dataclasses

import numpy as np
import pytest

import golden_scenes as gs
from chunkyclplugin_amd import native, scenes
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader
from oracle.binding import PortExt, PortOptions
from test_extensions import furnace, word5

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def make(gpu_instance, sc, **opts):
    loader = HipSceneLoader(gpu_instance)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    for k, v in opts.items():
        r.set_option({"sun_sampling": native.OPT_SUN_SAMPLING, "emitters": native.OPT_EMITTERS, "bsdf": native.OPT_BSDF,
                      "nee": native.OPT_EMITTER_NEE}[k], v)
    return loader, r


def with_spec_words(sc, seed=1):
    """The same scene with a random word 5 (spec | metal << 8 | rough << 16) on every material."""
    rng = np.random.default_rng(seed)
    m = np.asarray(sc.material_palette).copy().reshape(-1, 6)
    spec, metal, rough = rng.integers(0, 256, len(m)), rng.integers(0, 256, len(m)), rng.integers(0, 256, len(m))
    third = np.arange(len(m)) % 3
    spec[third == 0] = 0
    metal[third != 2] = 0                                  # a third diffuse-only, a third dielectric, a third metallic
    rough[rng.random(len(m)) < 0.3] = 0                    # some perfect mirrors
    m[:, 5] = spec | (metal << 8) | (rough << 16)
    return dataclasses.replace(sc, material_palette=m.reshape(-1).astype(np.int32))


def test_emitter_list_matches_the_specification(gpu_instance, port):
    for sc in (scenes.indoor_room(size=24, seed=7, width=32, img_height=24, emitter_frac=0.03), gs.make("outdoor")):
        loader, r = make(gpu_instance, sc)
        with PortExt(port, sc) as e:
            want = e.emitters[:e.n_emitters]
        np.testing.assert_array_equal(loader.emitters(), want)
        assert len(want) > 0
        r.close()
        loader.close()


CASES = [("outdoor", dict(bsdf=1)), ("outdoor", dict(sun_sampling=0)), ("outdoor_nosun", dict(sun_sampling=1, bsdf=1)),
         ("indoor", dict(nee=1)), ("indoor", dict(emitters=0)), ("indoor_sun", dict(nee=1, bsdf=1)),
         ("entities", dict(bsdf=1, nee=1)), ("water", dict(bsdf=1, sun_sampling=1, nee=1)), ("inside", dict(bsdf=1, nee=1))]


@pytest.mark.parametrize("name,opts", CASES)
def test_extended_kernels_match_the_specification(gpu_instance, port, name, opts):
    sc = with_spec_words(gs.make(name).with_view(96, 64))
    seeds = scenes.java_random_ints(6)
    loader, r = make(gpu_instance, sc, **opts)
    r.render_passes(seeds)
    info = r.kernel_info()
    assert info["ext"] and info["pool"] > 0, info
    with PortExt(port, sc, **opts):
        want = port.render_passes(sc, seeds)
    np.testing.assert_array_equal(bits(r.read()), bits(want))
    base = port.render_passes(sc, seeds)
    assert not np.array_equal(bits(base), bits(want)), "the options changed nothing on this scene"
    # back to the defaults: the reference kernels again, the reference image again
    for o in (native.OPT_BSDF, native.OPT_EMITTER_NEE):
        r.set_option(o, 0)
    r.set_option(native.OPT_SUN_SAMPLING, -1)
    r.set_option(native.OPT_EMITTERS, 1)
    r.reset()
    r.render_passes(seeds)
    assert not r.kernel_info()["ext"]
    np.testing.assert_array_equal(bits(r.read()), bits(base))
    r.close()
    loader.close()


def test_extended_kernels_deep_paths_and_furnace(gpu_instance, port):
    """Twelve bounces through mirrors and rough metals in the white furnace: still the specification's image, still <= 1."""
    sc = furnace([0, word5(255, 0, 0), word5(128, 0, 64), word5(0, 255, 0), word5(40, 200, 255), word5(255, 255, 128)], view=(96, 72))
    seeds = scenes.java_random_ints(8)
    loader, r = make(gpu_instance, sc, bsdf=1, sun_sampling=0)
    r.set_option(native.OPT_MAX_DEPTH, 12)
    r.render_passes(seeds)
    got = r.read()
    with PortExt(port, sc, bsdf=1, sun_sampling=0), PortOptions(port, 256, 12, 13.0):
        want = port.render_passes(sc, seeds)
    np.testing.assert_array_equal(bits(got), bits(want))
    assert got.max() <= 1.0 + 1e-5
    r.close()
    loader.close()


def test_extended_options_need_the_default_kernel(gpu_instance):
    sc = gs.make("indoor")
    loader, r = make(gpu_instance, sc, nee=1)
    r.set_option(native.OPT_KERNEL, 8)
    with pytest.raises(native.ChunkyHipError):
        r.render_passes([1])
    with pytest.raises(native.ChunkyHipError):
        r.set_option(native.OPT_SUN_SAMPLING, 2)
    r.close()
    loader.close()
