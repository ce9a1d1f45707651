"""Several GPUs behind ONE context in one process (chunky_group_create; SURVEY.md section 8b "init(device_ids[], n)" /
8e): scenes replicated on every member, the image's 16 x 16-pixel blocks dealt round-robin, one gather of the owned
blocks per read-back.  A 1-GPU box runs it with members that share device 0 — every piece of the path (fan-out of the
uploads, per-member shards, concurrent launches on the members' streams, pack / copy / scatter, the host loop on top) is
the code n distinct GPUs run; only the copy is device-to-device instead of peer-to-peer.  The image must be bit for bit
the one-context image, the reference's golden image and the oracle's."""
import os

import numpy as np
import pytest

from oracle import binding

import golden_scenes as gs
from chunkyclplugin_amd import native, scenes
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def renderer_on(instance, sc):
    loader = HipSceneLoader(instance)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    return loader, r


@pytest.fixture(scope="module")
def group3():
    g = RendererInstance.group([0, 0, 0])
    yield g
    g.close()


@pytest.mark.parametrize("name", ["outdoor", "entities", "pregen", "indoor"])
def test_group_image_is_the_reference_golden(group3, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sc = gs.make(name)
    assert group3.group_size() == 3
    loader, r = renderer_on(group3, sc)
    r.render_passes(g["seeds"])
    np.testing.assert_array_equal(bits(r.read()), bits(g["res"]))
    np.testing.assert_array_equal(r.preview(), g["preview"])
    info = r.kernel_info()
    assert info["pool"] >= 0  # the members run the pool kernel on their block shares
    r.close()
    loader.close()


def test_group_continues_the_running_mean_and_resets(group3, gpu_instance, port):
    """Two launches (bufferSpp continues), a read in between, a reset, and again: the same as one context."""
    sc = scenes.outdoor_world(chunks=4, height=64, seed=11, width=200, img_height=120, aabb_frac=0.05, quad_frac=0.03)
    seeds = native.java_random_ints(9)
    lg, rg = renderer_on(group3, sc)
    l1, r1 = renderer_on(gpu_instance, sc)
    for r in (rg, r1):
        r.render_passes(seeds[:4])
    a, b = rg.read(), r1.read()
    np.testing.assert_array_equal(bits(a), bits(b))
    for r in (rg, r1):
        r.render_passes(seeds[4:], first_buffer_spp=4)
    a, b = rg.read(), r1.read()
    np.testing.assert_array_equal(bits(a), bits(b))
    np.testing.assert_array_equal(bits(a), bits(port.render_passes(sc, seeds)))
    rg.reset()
    rg.render_passes(seeds[:2])
    np.testing.assert_array_equal(bits(rg.read()), bits(port.render_passes(sc, seeds[:2])))
    for x in (rg, r1, lg, l1):
        x.close()


def test_group_host_loop(group3, port):
    """chunky_render_run_ex on a group: the reference's pass loop (seeds, read-back cadence, double merge) over the members."""
    sc = gs.make("outdoor")
    lg, rg = renderer_on(group3, sc)
    buf = np.zeros(3 * sc.width * sc.height, np.float64)
    merges = []
    spp = rg.render_ex(buf, 0, 10, merge_interval=4, merged=merges.append)
    assert spp == 10 and merges == [4, 8, 10]
    want = np.zeros_like(buf)
    seeds = native.java_random_ints(10)
    done = 0
    for m in (4, 4, 2):  # OpenClPathTracingRenderer.java:167-173
        part = port.render_passes(sc, seeds[done:done + m]).astype(np.float64)
        want = (want * done + part * m) * (1.0 / (done + m))
        done += m
    np.testing.assert_array_equal(buf, want)
    rg.close()
    lg.close()


def test_group_inside_an_outer_shard(gpu_instance, port):
    """A group that is itself rank 1 of 2 (chunky_render_set_shard on the group): its members render as ranks 1 and 3 of 4;
    with a plain context as rank 0 of 2 the two halves add up to the whole image."""
    sc = scenes.outdoor_world(chunks=4, height=64, seed=12, width=160, img_height=96, aabb_frac=0.05, quad_frac=0.03)
    seeds = native.java_random_ints(3)
    g2 = RendererInstance.group([0, 0])
    lg, rg = renderer_on(g2, sc)
    rg.set_shard(1, 2, 0)
    l0, r0 = renderer_on(gpu_instance, sc)
    r0.set_shard(0, 2, 0)
    for r in (rg, r0):
        r.render_passes(seeds)
    half1, half0 = rg.read(), r0.read()
    assert (half1.reshape(-1, 3).any(axis=1) & half0.reshape(-1, 3).any(axis=1)).sum() == 0  # disjoint blocks
    np.testing.assert_array_equal(bits(half0 + half1), bits(port.render_passes(sc, seeds)))
    for x in (rg, r0, lg, l0):
        x.close()
    g2.close()


def test_group_on_the_timed_workload(gpu_instance, port):
    """BASELINE configs[2] at 1920x1080 cut over four members: the timed instantiation on every member, whole rows against
    the oracle, and the gather leaves the image in member 0's device buffer."""
    W, H, P = 1920, 1080, 16
    sc = scenes.cached_outdoor_world(chunks=32, height=256, width=W, img_height=H)
    seeds = native.java_random_ints(P)
    g4 = RendererInstance.group([0, 0, 0, 0])
    lg, rg = renderer_on(g4, sc)
    rg.render_passes(seeds, sync=False)
    rg.gather()
    info = rg.kernel_info()
    assert (info["tree"], info["pool"], info["bvh"], info["passes_per_launch"]) == (17, 64, False, 1024)  # a quarter share: long launches fit
    got = rg.read().reshape(-1, 3)
    rows = (3, 271, 540, 811, 1077)
    gids = np.concatenate([np.arange(y * W, (y + 1) * W) for y in rows]).astype(np.int32)
    ref = port.render_gids(sc, seeds, gids, threads=binding.usable_threads()).reshape(-1, 3)[gids]
    np.testing.assert_array_equal(bits(got[gids]), bits(ref))
    assert np.isfinite(got).all() and got.any(axis=1).mean() > 0.99  # every block arrived
    rg.close()
    lg.close()
    g4.close()


def test_group_errors():
    L = native.lib()
    import ctypes as C
    h = C.c_void_p()
    arr = (C.c_int * 2)(0, 99)
    assert L.chunky_group_create(arr, 2, C.byref(h)) == native.E_NO_DEVICE and not h.value
    assert L.chunky_group_create(arr, 0, C.byref(h)) == native.E_INVALID


def test_members_without_a_block_render_nothing(port):
    """An image of two 16 x 16 blocks on a group of three: the third member owns no block (nothing to render, nothing to
    gather), the image is still the one-context image; likewise a plain rank without blocks."""
    sc = gs.make("outdoor").with_view(30, 12)
    seeds = native.java_random_ints(3)
    g3 = RendererInstance.group([0, 0, 0])
    lg, rg = renderer_on(g3, sc)
    rg.render_passes(seeds)
    np.testing.assert_array_equal(bits(rg.read()), bits(port.render_passes(sc, seeds)))
    rg.set_shard(5, 8, 0)   # the group as rank 5 of 8: none of its members owns anything
    rg.reset()
    rg.render_passes(seeds)
    assert not rg.read().any()
    rg.close()
    lg.close()
    g3.close()


@pytest.mark.parametrize("case", ["variant8", "variant1_entities", "draw_depth", "lanes"])
def test_group_when_the_pool_kernel_does_not_apply(group3, gpu_instance, port, case):
    """Every member of a group holds a share of 16 x 16 blocks, which render_pool maps itself.  Scenes / options it does not take
    (variant bit 3 = round 1's kernel, bit 0 with entities = the packed BVH walk, a draw depth beyond its 16-bit step counter,
    bit 1 = one lane per path) run the fallback kernels on the same pixels from a list (capi.hip block_pixel_list): the image
    is still the one-context image and the oracle's."""
    from oracle.binding import PortOptions
    if case == "variant1_entities":
        sc = gs.make("entities")
    else:
        sc = scenes.outdoor_world(chunks=4, height=64, seed=21, width=200, img_height=120, aabb_frac=0.05, quad_frac=0.03)
    seeds = native.java_random_ints(5)
    lg, rg = renderer_on(group3, sc)
    l1, r1 = renderer_on(gpu_instance, sc)
    depth = 256
    for r in (rg, r1):
        if case == "variant8":
            r.set_option(native.OPT_KERNEL, 8)
        elif case == "variant1_entities":
            r.set_option(native.OPT_KERNEL, 1)
        elif case == "lanes":
            r.set_option(native.OPT_KERNEL, 2)
        else:
            depth = 70000
            r.set_option(native.OPT_DRAW_DEPTH, depth)
        r.render_passes(seeds[:3])
        r.render_passes(seeds[3:], first_buffer_spp=3)
    assert rg.kernel_info()["pool"] < 0, rg.kernel_info()  # the members did run a fallback kernel
    got = rg.read()
    np.testing.assert_array_equal(bits(got), bits(r1.read()))
    with PortOptions(port, draw_depth=depth):
        np.testing.assert_array_equal(bits(got), bits(port.render_passes(sc, seeds)))
    # back to the default kernel on the same target: the share is mapped by render_pool again
    rg.set_option(native.OPT_KERNEL, 0)
    rg.set_option(native.OPT_DRAW_DEPTH, 256)
    rg.reset()
    rg.render_passes(seeds)
    assert rg.kernel_info()["pool"] >= 0
    np.testing.assert_array_equal(bits(rg.read()), bits(port.render_passes(sc, seeds)))
    for x in (rg, r1, lg, l1):
        x.close()


def test_group_peer_status(group3, gpu_instance):
    """chunky_group_peer_status: members that share member 0's device need no peer copy; a plain context reports one LOCAL entry."""
    assert group3.peer_status() == [native.PEER_LOCAL] * 3
    assert gpu_instance.peer_status() == [native.PEER_LOCAL]
    import ctypes as C
    out = (C.c_int * 2)()
    assert native.lib().chunky_group_peer_status(group3._h, out, 2) == native.E_INVALID  # room for fewer members than the group has


def test_group_long_launches(group3, port):
    """More passes in one call than the kernel-argument segment holds seeds for: every member runs them as one launch with its
    seeds in device memory (its own buffer on its own device), and the gathered image is the oracle's."""
    sc = scenes.outdoor_world(chunks=4, height=64, seed=31, width=200, img_height=120, aabb_frac=0.05, quad_frac=0.03)
    seeds = native.java_random_ints(700)
    lg, rg = renderer_on(group3, sc)
    rg.kernel_time()
    rg.render_passes(seeds)
    assert rg.kernel_info()["passes_per_launch"] == 1024 and rg.kernel_time()[1] == 1
    np.testing.assert_array_equal(bits(rg.read()), bits(port.render_passes(sc, seeds, threads=binding.usable_threads())))
    rg.close()
    lg.close()
