"""ctypes binding of the C ABI declared in include/chunky_hip.h (libchunky_hip.so).

This is the only way Python reaches the device path; there is no fallback.  If the shared library
is missing it is built with hipcc (`build()`); if that is impossible, or no HIP device is present
when a context is created, the error propagates — nothing here routes to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
from typing import List, Optional

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("CHUNKY_HIP_LIB") or os.path.join(PKG_DIR, "libchunky_hip.so")  # override: tuning builds (tools/variants.sh)
HEADER = os.path.join(os.path.dirname(PKG_DIR), "include", "chunky_hip.h")
SOURCES = ["render_pool.hip", "render_fallback.hip", "aux_kernels.hip", "filter.hip", "capi.hip", "widetree.cpp"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"]

MAX_TRACES = 10
HIT_DTYPE = np.dtype([("hit", "<i4"), ("material", "<i4"), ("distance", "<f4"), ("normal", "<f4", 3),
                      ("color", "<f4", 4), ("emittance", "<f4"), ("point", "<f4", 3)])

PALETTE_BLOCK, PALETTE_MATERIAL, PALETTE_AABB, PALETTE_QUAD, PALETTE_TRIG = range(5)
BVH_WORLD, BVH_ACTOR = 0, 1
OPT_DRAW_DEPTH, OPT_MAX_DEPTH, OPT_EMITTER_SCALE, OPT_KERNEL, OPT_SUN_SAMPLING, OPT_EMITTERS, OPT_BSDF, OPT_EMITTER_NEE, OPT_BVH_CULL_BEHIND = range(9)
PEER_LOCAL, PEER_DIRECT, PEER_STAGED = 0, 1, 2
TRANSPORT_PEER_COPY, TRANSPORT_RCCL_SENDRECV, TRANSPORT_RCCL_REDUCE = 0, 1, 2
E_INVALID, E_NO_DEVICE, E_HIP, E_STATE, E_ABORTED = -1, -2, -3, -4, -5


class ChunkyHipError(RuntimeError):
    """Raised for any non-zero status (the JNI layer would throw RuntimeException the same way;
    the reference gets CLException from JOCL, RendererInstance.java:36)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"chunky-hip error {code}: {message}")
        self.code = code


def _needs_build() -> bool:
    if os.environ.get("CHUNKY_HIP_LIB"):
        return False
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    # the sources (files only: csrc/build/ holds objects and tools that are written after the link)
    deps = [p for p in (os.path.join(CSRC, f) for f in os.listdir(CSRC)) if os.path.isfile(p)] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, extra_flags=(), out: Optional[str] = None, objdir: Optional[str] = None) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or _needs_build():
        # one hipcc per translation unit, side by side (objects under csrc/build/, git-ignored), then one link
        import fcntl
        from concurrent.futures import ThreadPoolExecutor
        objdir = objdir or os.path.join(CSRC, "build")
        os.makedirs(objdir, exist_ok=True)
        # several ranks of one job may arrive here at once (bench.py --gpus N on a box without the library): one builds,
        # the others wait for it and find the library there
        lock = open(os.path.join(objdir, ".lock"), "w")
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _needs_build():
            lock.close()
            return out or LIB_PATH
        flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags)

        def compile_one(src):
            obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
            cmd = ["hipcc", *flags, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
            proc = subprocess.run(cmd, capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n" + proc.stderr[-4000:])
            return obj

        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        proc = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out or LIB_PATH], capture_output=True, text=True)
        lock.close()  # (also released when an exception unwinds past here: the file object goes away)
        if proc.returncode != 0:
            raise RuntimeError("hipcc link failed:\n" + proc.stderr[-4000:])
    return out or LIB_PATH


TUNING_LIB_PATH = os.path.join(PKG_DIR, "libchunky_hip_tuning.so")


def build_tuning(force: bool = False) -> str:
    """The same library compiled with -DCHUNKY_TUNING: the build that reads the tuning / test-rig environment variables
    (CHUNKY_WIDE_LEVELS, CHUNKY_WIDE_TOP_BITS, CHUNKY_DEBUG_WIDE_BITS, CHUNKY_BVH_LAYOUT, CHUNKY_GROUP_TRANSPORT,
    CHUNKY_GROUP_SELF_EXCHANGE, CHUNKY_GROUP_NO_PROBE, CHUNKY_GROUP_TIMEOUT_MS, CHUNKY_RCCL_TRY_SHARED).  The shipping
    library reads none of them; tests and tools that need one run a child process with CHUNKY_HIP_LIB pointing here."""
    stale = not os.path.exists(TUNING_LIB_PATH)
    if not stale:
        t = os.path.getmtime(TUNING_LIB_PATH)
        deps = [p for p in (os.path.join(CSRC, f) for f in os.listdir(CSRC)) if os.path.isfile(p)] + [HEADER]
        stale = any(os.path.getmtime(d) > t for d in deps)
    if force or stale:
        build(force=True, extra_flags=["-DCHUNKY_TUNING"], out=TUNING_LIB_PATH, objdir=os.path.join(CSRC, "build", "tuning"))
    return TUNING_LIB_PATH


def declared_symbols() -> List[str]:
    """Every function include/chunky_hip.h declares."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chunky_[a-z_0-9]+)\s*\(", text)) - {"chunky_post_render_fn"})


_lib: Optional[C.CDLL] = None
POST_RENDER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)
SAVE_EVENT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32)
REGEN_FN = C.CFUNCTYPE(None, C.c_void_p)


class RunCallbacks(C.Structure):
    """chunky_run_callbacks (include/chunky_hip.h)."""
    _fields_ = [("struct_size", C.c_size_t), ("post_render", POST_RENDER_FN), ("progress", PROGRESS_FN), ("merged", PROGRESS_FN),
                ("save_event", SAVE_EVENT_FN), ("regenerate_camera", REGEN_FN), ("user", C.c_void_p),
                ("poll_gate", POST_RENDER_FN)]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.chunky_last_error.restype = C.c_char_p
        L.chunky_version.restype = C.c_char_p
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        sig = {
            "chunky_device_count": [],
            "chunky_device_name": [C.c_int, C.c_char_p, C.c_int],
            "chunky_init": [C.c_int, C.POINTER(vp)],
            "chunky_shutdown": [vp],
            "chunky_group_create": [vp, C.c_int, C.POINTER(vp)],
            "chunky_group_size": [vp],
            "chunky_group_device": [vp, C.c_int],
            "chunky_group_peer_status": [vp, vp, C.c_int],
            "chunky_group_transport": [vp, C.POINTER(C.c_int), C.c_char_p, C.c_int],
            "chunky_group_set_transport": [vp, C.c_int],
            "chunky_scene_create": [vp, C.POINTER(vp)],
            "chunky_scene_destroy": [vp],
            "chunky_scene_set_octree": [vp, vp, i64, C.c_int],
            "chunky_scene_load_octree": [vp, vp, i64, C.c_int, vp, i64],
            "chunky_scene_set_palette": [vp, C.c_int, vp, i64],
            "chunky_scene_set_bvh": [vp, C.c_int, vp, i64],
            "chunky_scene_set_atlas": [vp, vp, C.c_int, C.c_int, C.c_int],
            "chunky_scene_write_atlas_tile": [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp],
            "chunky_scene_set_sky": [vp, vp, C.c_int, C.c_int, f32],
            "chunky_scene_set_sun": [vp, vp],
            "chunky_scene_emitters": [vp, vp, i32, C.POINTER(i32)],
            "chunky_render_create": [vp, vp, C.c_int, C.c_int, C.POINTER(vp)],
            "chunky_render_destroy": [vp],
            "chunky_render_set_camera": [vp, C.c_int, vp, i64],
            "chunky_render_set_option": [vp, C.c_int, i32],
            "chunky_render_set_shard": [vp, C.c_int, C.c_int, C.c_int],
            "chunky_render_set_device_buffer": [vp, vp],
            "chunky_render_device_buffer": [vp, C.POINTER(vp)],
            "chunky_render_reset": [vp],
            "chunky_render_passes": [vp, vp, C.c_int, C.c_int],
            "chunky_render_sync": [vp],
            "chunky_render_read": [vp, vp, i64],
            "chunky_render_gather": [vp],
            "chunky_render_kernel_time": [vp, C.POINTER(f32), C.POINTER(C.c_int)],
            "chunky_render_preview": [vp, vp],
            "chunky_render_phase_stats": [vp, vp, C.c_int],
            "chunky_render_kernel_info": [vp, vp],
            "chunky_render_trace_records": [vp, i32, vp, C.c_int, vp, vp, vp],
            "chunky_render_run": [vp, vp, C.POINTER(i32), i32, i32, POST_RENDER_FN, vp],
            "chunky_render_run_ex": [vp, vp, C.POINTER(i32), i32, i32, C.POINTER(RunCallbacks)],
            "chunky_java_random_ints": [i64, vp, C.c_int],
            "chunky_selftest_math": [vp, C.c_int, C.c_int, vp, vp, vp],
            "chunky_selftest_helpers": [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.POINTER(i32)],
            "chunky_selftest_gamma_scan": [vp, C.c_int, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(f32)],
            "chunky_filter_frame": [vp, C.c_int, C.c_int, C.c_double, vp, vp, C.c_int],
            "chunky_filter_gamma_thresholds": [vp],
            "chunky_filter_frame_device": [vp, i64, f32, vp, vp, C.c_int, C.c_int, C.POINTER(f32)],
            "chunky_widetree_lookup": [vp, i64, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, C.POINTER(i64)],
        }
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise ChunkyHipError(rc, (lib().chunky_last_error() or b"").decode("utf-8", "replace"))


def ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def java_random_ints(n: int, seed: int = 0) -> np.ndarray:
    out = np.zeros(n, np.int32)
    check(lib().chunky_java_random_ints(seed, ptr(out), n))
    return out


def widetree_lookup(tree: np.ndarray, depth: int, xyz: np.ndarray, level_bits=None):
    """Host-side check of the upload-time octree re-layout: (data, level, n_entries) for each cell."""
    tree = np.ascontiguousarray(tree, np.int32)
    xyz = np.ascontiguousarray(xyz, np.int32).reshape(-1, 3)
    data = np.zeros(len(xyz), np.int32)
    level = np.zeros(len(xyz), np.int32)
    n_entries = C.c_int64()
    lb = None if level_bits is None else np.ascontiguousarray(level_bits, np.int32)
    check(lib().chunky_widetree_lookup(ptr(tree), tree.size, depth, None if lb is None else ptr(lb),
                                       0 if lb is None else lb.size, ptr(xyz), len(xyz), ptr(data), ptr(level),
                                       C.byref(n_entries)))
    return data, level, n_entries.value
