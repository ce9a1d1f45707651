"""The context is shared between threads (Chunky's render-manager thread runs render(), a ForkJoin task regenerates
camera rays meanwhile, sceneReset arrives between renders, and handles may be freed from a GC cleaner thread): every entry point takes
the context's mutex.  ctypes drops the GIL around each call, so these threads really are inside the library together."""
import threading

import numpy as np
import pytest

import golden_scenes as gs
from chunkyclplugin_amd import scenes
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader


def run_threads(fns):
    errors = []

    def wrap(f):
        def go():
            try:
                f()
            except BaseException as e:  # noqa: BLE001 — reported to the test below
                errors.append(e)
        return go

    ts = [threading.Thread(target=wrap(f)) for f in fns]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a thread is stuck inside the library"
    if errors:
        raise errors[0]


@pytest.mark.gpu
def test_many_threads_one_context(gpu_instance, port):
    """eight threads, each with its own scene and target on the SAME context: upload, render, preview, read, destroy — interleaved"""
    names = ["indoor", "outdoor", "entities", "dof", "indoor", "outdoor_nosun", "entities", "outdoor"]
    seeds = scenes.java_random_ints(3)
    scs = [gs.make(n).with_view(40 + 8 * i, 24 + 4 * i) for i, n in enumerate(names)]
    want = [port.render_passes(sc, seeds) for sc in scs]
    want_prev = [port.preview(sc) for sc in scs]
    got, got_prev = [None] * len(scs), [None] * len(scs)

    def worker(i):
        def go():
            for _ in range(3):   # three rounds of create / render / destroy per thread
                loader = HipSceneLoader(gpu_instance)
                loader.load_packed(scs[i])
                r = HipPathTracingRenderer(loader, scs[i].width, scs[i].height)
                r.set_camera(scs[i].projector_type, scs[i].camera)
                r.render_passes(seeds[:1])
                r.render_passes(seeds[1:], first_buffer_spp=1, sync=False)
                got_prev[i] = r.preview()
                got[i] = r.read()
                r.close()
                loader.close()
        return go

    run_threads([worker(i) for i in range(len(scs))])
    for i in range(len(scs)):
        np.testing.assert_array_equal(got[i].view(np.uint32), want[i].view(np.uint32))
        np.testing.assert_array_equal(got_prev[i], want_prev[i])


@pytest.mark.gpu
def test_many_threads_one_group(port):
    """the same on a group context (three members behind one handle): every call fans out under the group's lock"""
    from chunkyclplugin_amd.renderer import RendererInstance
    group = RendererInstance.group([0, 0, 0])
    seeds = scenes.java_random_ints(2)
    scs = [gs.make(n).with_view(48 + 16 * i, 32 + 8 * i) for i, n in enumerate(["indoor", "entities", "outdoor", "indoor"])]
    want = [port.render_passes(sc, seeds) for sc in scs]
    got = [None] * len(scs)

    def worker(i):
        def go():
            for _ in range(3):
                loader = HipSceneLoader(group)
                loader.load_packed(scs[i])
                r = HipPathTracingRenderer(loader, scs[i].width, scs[i].height)
                r.set_camera(scs[i].projector_type, scs[i].camera)
                r.render_passes(seeds, sync=False)
                got[i] = r.read()
                r.close()
                loader.close()
        return go

    run_threads([worker(i) for i in range(len(scs))])
    for i in range(len(scs)):
        np.testing.assert_array_equal(got[i].view(np.uint32), want[i].view(np.uint32))
    group.close()


@pytest.mark.gpu
def test_camera_thread_beside_the_pass_loop(gpu_instance, port):
    """render() on one thread while another keeps installing the camera (the regenerating ForkJoin task, ClCamera.java:99-104) and
    asking for previews; the camera it installs is the same one, so the image must be the oracle's"""
    sc = gs.make("indoor").with_view(64, 40)
    loader = HipSceneLoader(gpu_instance)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    target, interval = 24, 8
    sample = np.zeros(sc.width * sc.height * 3, np.float64)
    done = threading.Event()
    spp = []
    pokes = [0]

    def loop():
        try:
            spp.append(r.render(sample, 0, target, merge_interval=interval))
        finally:
            done.set()

    def camera():
        while not done.is_set():
            r.set_camera(sc.projector_type, sc.camera)
            r.preview()
            pokes[0] += 1

    run_threads([loop, camera])
    assert spp == [target] and pokes[0] > 0
    seeds = scenes.java_random_ints(target)
    want = np.zeros_like(sample)
    for lo in range(0, target, interval):
        pass_buf = port.render_passes(sc, seeds[lo:lo + interval]).astype(np.float64)
        want = (want * lo + pass_buf * interval) * (1.0 / (lo + interval))
    np.testing.assert_array_equal(sample.view(np.uint64), want.view(np.uint64))
    r.close()
    loader.close()


@pytest.mark.gpu
def test_handles_freed_on_a_foreign_thread(gpu_instance, port):
    """NativeCleaner frees from a cleaner thread (NativeCleaner.java:45-53): objects made here are destroyed over there, while this
    thread keeps rendering with others"""
    sc = gs.make("outdoor").with_view(48, 32)
    seeds = scenes.java_random_ints(2)
    want = port.render_passes(sc, seeds)
    made = []
    for _ in range(6):
        loader = HipSceneLoader(gpu_instance)
        loader.load_packed(sc)
        r = HipPathTracingRenderer(loader, sc.width, sc.height)
        r.set_camera(sc.projector_type, sc.camera)
        r.render_passes(seeds, sync=False)   # still queued when the other thread frees it
        made.append((loader, r))
    keep_loader, keep = made.pop()
    images = []

    def cleaner():
        for loader, r in made:
            r.close()
            loader.close()

    def renderer():
        for _ in range(4):
            keep.reset()
            keep.render_passes(seeds)
            images.append(keep.read())

    run_threads([cleaner, renderer])
    for img in images:
        np.testing.assert_array_equal(img.view(np.uint32), want.view(np.uint32))
    keep.close()
    keep_loader.close()
