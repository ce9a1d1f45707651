"""The C-ABI library loads and exports every symbol include/chunky_hip.h declares; its host-only
entry points behave; without a GPU the device path fails loudly instead of falling back."""
import ctypes as C

import numpy as np
import pytest

from chunkyclplugin_amd import native, scenes


def test_exports_every_declared_symbol():
    L = native.lib()
    names = native.declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/chunky_hip.h but not exported"
    assert b"gfx950" in L.chunky_version()


def test_java_seed_stream_native():
    want = [-1155484576, -723955400, 1033096058, -1690734402, -1557280266, 1327362106, -1930858313, 502539523]
    assert native.java_random_ints(8).tolist() == want
    assert native.java_random_ints(100, seed=12345).tolist() == scenes.java_random_ints(100, seed=12345).tolist()


def test_null_handles_are_errors_not_crashes():
    L = native.lib()
    assert L.chunky_shutdown(None) == native.E_INVALID
    assert L.chunky_scene_destroy(None) == native.E_INVALID
    assert L.chunky_render_sync(None) == native.E_INVALID
    assert L.chunky_init(0, None) == native.E_INVALID
    assert b"NULL" in L.chunky_last_error()


def test_no_device_means_error_not_fallback():
    L = native.lib()
    if L.chunky_device_count() > 0:
        pytest.skip("a HIP device is present")
    h = C.c_void_p()
    assert L.chunky_init(0, C.byref(h)) == native.E_NO_DEVICE
    assert not h.value
    assert b"no CPU fallback" in L.chunky_last_error()
    from chunkyclplugin_amd.renderer import RendererInstance
    with pytest.raises(native.ChunkyHipError):
        RendererInstance(0)


def test_group_without_devices_fails_loudly():
    """chunky_group_create has no fallback either; bad arguments are errors, not crashes."""
    L = native.lib()
    h = C.c_void_p()
    two = (C.c_int * 2)(0, 0)
    assert L.chunky_group_create(None, 2, C.byref(h)) == native.E_INVALID
    assert L.chunky_group_create(two, 0, C.byref(h)) == native.E_INVALID
    assert L.chunky_group_create(two, 2, None) == native.E_INVALID
    assert L.chunky_group_size(None) == native.E_INVALID
    assert L.chunky_render_gather(None) == native.E_INVALID
    if L.chunky_device_count() == 0:
        assert L.chunky_group_create(two, 2, C.byref(h)) == native.E_NO_DEVICE and not h.value
        assert b"member 0" in L.chunky_last_error()


RIG_VARIABLES = (b"CHUNKY_WIDE_LEVELS", b"CHUNKY_WIDE_TOP_BITS", b"CHUNKY_DEBUG_WIDE_BITS", b"CHUNKY_BVH_LAYOUT", b"CHUNKY_GROUP_TRANSPORT",
                 b"CHUNKY_GROUP_SELF_EXCHANGE", b"CHUNKY_GROUP_NO_PROBE", b"CHUNKY_GROUP_TIMEOUT_MS", b"CHUNKY_RCCL_TRY_SHARED")


def test_the_shipping_library_reads_no_tuning_variable():
    """What a JVM loads must not be steerable through the environment: the names of the tuning / test-rig variables do not even
    occur in libchunky_hip.so (the one variable it reads is CHUNKY_RCCL_LIB, a deployment's own librccl file); they exist in the
    -DCHUNKY_TUNING build that rig tests load in child processes."""
    native.build()
    shipping = open(native.LIB_PATH, "rb").read()
    for name in RIG_VARIABLES:
        assert name not in shipping, name
    assert b"CHUNKY_RCCL_LIB" in shipping
    tuning = open(native.build_tuning(), "rb").read()
    for name in RIG_VARIABLES:
        assert name in tuning, name


def test_rccl_is_not_needed_to_build_the_library_and_the_local_declarations_match_it(tmp_path):
    """csrc/rccl_dyn.hpp declares the few RCCL types and enumerators itself (the functions come from dlsym): no RCCL header is
    included by a normal build.  Where <rccl/rccl.h> is installed the compiler compares the two."""
    import os
    import subprocess
    text = open(os.path.join(native.CSRC, "rccl_dyn.hpp")).read()
    assert "#if defined(CHUNKY_CHECK_RCCL_HEADER)\n#include <rccl/rccl.h>" in text and text.count("#include <rccl/rccl.h>") == 1
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no RCCL header on this box")
    src = tmp_path / "check.cpp"
    src.write_text('#include "rccl_dyn.hpp"\nint main() { return chunky::rccl_api().usable() ? 0 : 0; }\n')
    p = subprocess.run(["hipcc", "-x", "hip", "--offload-host-only", "-std=c++17", "-DCHUNKY_CHECK_RCCL_HEADER", "-I" + native.CSRC, "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
