"""A plugin session lasts hours: scenes are reloaded, targets re-created, groups come and go.  Device memory and host memory have to come
back — every hipMalloc of a scene, a target (frame buffer, staging array, events, gather buffers) and a context is released with it."""
import gc
import os

import numpy as np
import pytest

import golden_scenes as gs
from chunkyclplugin_amd import scenes
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance


def device_free_bytes():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")   # the runtime the library itself is linked against
    assert hip.hipDeviceSynchronize() == 0
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def host_rss_bytes():
    import psutil
    return psutil.Process(os.getpid()).memory_info().rss


def one_session(inst, sc, seeds):
    loader = HipSceneLoader(inst)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    r.render_passes(seeds)
    img = r.read()
    r.preview()
    sample = np.zeros(sc.width * sc.height * 3, np.float64)
    r.render(sample, 0, 4, merge_interval=2)
    r.close()
    loader.close()
    return img


@pytest.mark.gpu
def test_memory_comes_back(gpu_instance):
    seeds = scenes.java_random_ints(2)
    scs = [gs.make(n).with_view(w, h) for n, w, h in (("entities", 160, 96), ("indoor", 320, 200), ("outdoor", 96, 64))]
    first = [one_session(gpu_instance, sc, seeds) for sc in scs]   # warm: the allocator's pools, the kernels' code objects
    warm = RendererInstance.group([0, 0, 0])   # (the runtime keeps ~54 MiB for the first extra streams it is asked for)
    one_session(warm, scs[1], seeds)
    warm.close()
    gc.collect()
    free0, rss0 = device_free_bytes(), host_rss_bytes()
    for k in range(60):
        img = one_session(gpu_instance, scs[k % 3], seeds)
        if k < 3:
            np.testing.assert_array_equal(img.view(np.uint32), first[k].view(np.uint32))
    for _ in range(12):   # groups: member contexts, replicated scenes, gather buffers
        g = RendererInstance.group([0, 0, 0])
        one_session(g, scs[1], seeds)
        g.close()
    gc.collect()
    free1, rss1 = device_free_bytes(), host_rss_bytes()
    # 72 sessions would leak hundreds of MB if a frame buffer (0.8 MB), a staging array or a scene (several MB) stayed behind each time
    assert free0 - free1 < 16 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 72 sessions"
    assert rss1 - rss0 < 96 << 20, f"host memory grew by {(rss1 - rss0) >> 20} MiB over 72 sessions"
