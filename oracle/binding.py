"""ctypes bindings for the two CPU checkers.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this
module; nothing under `chunkyclplugin_amd/` does.

* `ref()`  — oracle/_ref/libchunky_ref.so: the reference OpenCL kernel itself, compiled in place
             from /root/reference for x86-64 (exists only where it was built: the build container;
             it does not travel to the GPU box).
* `port()` — oracle/libchunky_port.so: oracle/port.c, the plain-C restatement (buildable
             anywhere with gcc).
* `port_libm()` / `ref_libm()` — the same two on a second platform layer (glibc libm, unfused
             vector builtins): they must agree bit for bit, which pins the restatement's logic
             under builtins that do not come from rt_math.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_TRACES = 10


class OracleScene(C.Structure):
    _fields_ = [
        ("projector_type", C.c_int32), ("camera_settings", C.c_void_p),
        ("octree_depth", C.c_int32), ("octree", C.c_void_p),
        ("block_palette", C.c_void_p), ("quad_models", C.c_void_p), ("aabb_models", C.c_void_p),
        ("world_bvh", C.c_void_p), ("actor_bvh", C.c_void_p), ("bvh_trigs", C.c_void_p),
        ("atlas", C.c_void_p), ("atlas_w", C.c_int32), ("atlas_h", C.c_int32), ("atlas_layers", C.c_int32),
        ("material_palette", C.c_void_p),
        ("sky", C.c_void_p), ("sky_w", C.c_int32), ("sky_h", C.c_int32), ("sky_intensity", C.c_float),
        ("sun", C.c_void_p),
        ("width", C.c_int32), ("height", C.c_int32),
    ]


HIT_DTYPE = np.dtype([("hit", "<i4"), ("material", "<i4"), ("distance", "<f4"), ("normal", "<f4", 3),
                      ("color", "<f4", 4), ("emittance", "<f4"), ("point", "<f4", 3)])
assert HIT_DTYPE.itemsize == 56

COUNTER_NAMES = ("samples", "traces", "steps", "node", "block", "model_hdr", "aabb", "quad", "mat",
                 "texel", "bvh_inner", "leaf_hdr", "tri", "sky", "hits")


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


class SceneHandle:
    """Keeps the numpy arrays alive and exposes the OracleScene struct."""

    def __init__(self, sc):
        self.keep = dict(
            cam=np.ascontiguousarray(sc.camera, np.float32),
            octree=np.ascontiguousarray(sc.octree, np.int32),
            blocks=np.ascontiguousarray(sc.block_palette, np.int32),
            quads=np.ascontiguousarray(sc.quad_models, np.int32),
            aabbs=np.ascontiguousarray(sc.aabb_models, np.int32),
            wbvh=np.ascontiguousarray(sc.world_bvh, np.int32),
            abvh=np.ascontiguousarray(sc.actor_bvh, np.int32),
            trigs=np.ascontiguousarray(sc.bvh_trigs, np.int32),
            atlas=np.ascontiguousarray(sc.atlas, np.uint8),
            mats=np.ascontiguousarray(sc.material_palette, np.int32),
            sky=np.ascontiguousarray(sc.sky, np.uint8),
            sun=np.ascontiguousarray(sc.sun, np.int32),
        )
        k = self.keep
        L, H, W, _ = k["atlas"].shape
        self.struct = OracleScene(
            int(sc.projector_type), _ptr(k["cam"]), int(sc.octree_depth), _ptr(k["octree"]),
            _ptr(k["blocks"]), _ptr(k["quads"]), _ptr(k["aabbs"]), _ptr(k["wbvh"]), _ptr(k["abvh"]),
            _ptr(k["trigs"]), _ptr(k["atlas"]), W, H, L, _ptr(k["mats"]), _ptr(k["sky"]),
            k["sky"].shape[1], k["sky"].shape[0], float(sc.sky_intensity), _ptr(k["sun"]),
            int(sc.width), int(sc.height))
        self.width, self.height = int(sc.width), int(sc.height)


class _Lib:
    prefix = ""

    def __init__(self, path: str):
        self.path = path
        self.lib = C.CDLL(path)
        p = self.prefix
        f = getattr(self.lib, p + "_render_passes")
        f.argtypes = [C.POINTER(OracleScene), C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64,
                      C.c_void_p, C.c_int]
        f.restype = C.c_int
        f = getattr(self.lib, p + "_preview")
        f.argtypes = [C.POINTER(OracleScene), C.c_void_p, C.c_int]
        f.restype = C.c_int
        f = getattr(self.lib, p + "_trace_records")
        f.argtypes = [C.POINTER(OracleScene), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        f.restype = C.c_int
        f = getattr(self.lib, p + "_math")
        f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        f.restype = None
        f = getattr(self.lib, p + "_filter")
        f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int]
        f.restype = None
        f = getattr(self.lib, p + "_pow")
        f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        f.restype = None

    def render_passes(self, sc, seeds, first_spp: int = 0, res: Optional[np.ndarray] = None,
                      gid_range=None, threads: int = 8) -> np.ndarray:
        """res[3*W*H] running mean after the given passes (rayTracer.cl:109-112)."""
        h = sc if isinstance(sc, SceneHandle) else SceneHandle(sc)
        seeds = np.ascontiguousarray(seeds, np.int32)
        n = h.width * h.height
        if res is None:
            res = np.zeros(3 * n, np.float32)
        b, e = gid_range if gid_range is not None else (0, n)
        getattr(self.lib, self.prefix + "_render_passes")(
            C.byref(h.struct), _ptr(seeds), len(seeds), first_spp, b, e, _ptr(res), threads)
        return res

    def preview(self, sc, threads: int = 8) -> np.ndarray:
        h = sc if isinstance(sc, SceneHandle) else SceneHandle(sc)
        out = np.zeros(h.width * h.height, np.int32)
        getattr(self.lib, self.prefix + "_preview")(C.byref(h.struct), _ptr(out), threads)
        return out

    def trace_records(self, sc, seed: int, gid: int):
        h = sc if isinstance(sc, SceneHandle) else SceneHandle(sc)
        hits = np.zeros(MAX_TRACES, HIT_DTYPE)
        rad = np.zeros(3, np.float32)
        n = getattr(self.lib, self.prefix + "_trace_records")(
            C.byref(h.struct), int(seed), int(gid), _ptr(hits), _ptr(rad))
        return hits[:n], rad

    def filter(self, samples, exposure: float, type_: int, width: int = 0, height: int = 0) -> np.ndarray:
        """The tone-map kernel on `samples` (3 doubles per pixel): one ARGB word per pixel."""
        samples = np.ascontiguousarray(samples, np.float64).reshape(-1)
        n = samples.size // 3
        out = np.zeros(n, np.uint32)
        getattr(self.lib, self.prefix + "_filter")(n, width or n, height or 1, float(np.float32(exposure)), _ptr(samples),
                                                   _ptr(out), int(type_))
        return out

    def pow(self, a, b) -> np.ndarray:
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        out = np.empty_like(a)
        getattr(self.lib, self.prefix + "_pow")(a.size, _ptr(a), _ptr(b), _ptr(out))
        return out

    def helpers(self, sc, which: int, rows) -> np.ndarray:
        """Helper-level known answers: helper `which` (numbering and row layouts in oracle/ref_shim.cpp ref_helpers) on rows
        of 32 floats; 12 floats out per row.  The reference build calls the reference's own exported functions, the
        restatement its counterparts."""
        h = sc if isinstance(sc, SceneHandle) else SceneHandle(sc)
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 32)
        out = np.zeros((len(rows), 12), np.float32)
        f = getattr(self.lib, self.prefix + "_helpers")
        f.argtypes = [C.POINTER(OracleScene), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        f.restype = None
        f(C.byref(h.struct), int(which), len(rows), _ptr(rows), _ptr(out))
        return out

    def math(self, which: int, a, b=None) -> np.ndarray:
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(a if b is None else b, np.float32)
        out = np.empty_like(a)
        getattr(self.lib, self.prefix + "_math")(which, a.size, _ptr(a), _ptr(b), _ptr(out))
        return out


class PortOptions:
    """with PortOptions(port, draw_depth=.., max_depth=.., emitter_scale=..): the C restatement renders with the
    render-loop constants the HIP library exposes as options; restored to the reference's (256, 5, 13) on exit."""

    def __init__(self, port, draw_depth=256, max_depth=5, emitter_scale=13.0):
        self.port, self.args = port, (int(draw_depth), int(max_depth), float(emitter_scale))

    def __enter__(self):
        self.port.lib.port_set_options.argtypes = [C.c_int, C.c_int, C.c_float]
        self.port.lib.port_set_options.restype = None
        self.port.lib.port_set_options(*self.args)
        return self

    def __exit__(self, *exc):
        self.port.lib.port_set_options(256, 5, 13.0)
        return False


class PortCull:
    """with PortCull(port): the C restatement walks the entity BVHs with CHUNKY_OPT_BVH_CULL_BEHIND (children entirely behind the
    ray origin count as missed) — the specification of that option; the reference's walk again on exit."""

    def __init__(self, port, on: bool = True):
        self.port, self.on = port, on
        port.lib.port_set_bvh_cull.argtypes = [C.c_int]
        port.lib.port_set_bvh_cull.restype = None

    def __enter__(self):
        self.port.lib.port_set_bvh_cull(1 if self.on else 0)
        return self

    def __exit__(self, *exc):
        self.port.lib.port_set_bvh_cull(0)
        return False


class PortExt:
    """with PortExt(port, scene, sun_sampling=.., emitters=.., bsdf=.., nee=..): the C restatement renders with the
    EXPERIMENTAL light-transport options of DESIGN.md section 9 (oracle/port.c trace_sample_ext is their specification);
    the emitter list the NEE option samples is built here by port_list_emitters.  Defaults restored on exit."""

    def __init__(self, port, sc, sun_sampling=-1, emitters=1, bsdf=0, nee=0):
        self.port, self.args = port, (int(sun_sampling), int(emitters), int(bsdf), int(nee))
        self.handle = sc if isinstance(sc, SceneHandle) else SceneHandle(sc)
        L = port.lib
        L.port_set_ext.argtypes = [C.c_int] * 4
        L.port_set_ext.restype = None
        L.port_list_emitters.argtypes = [C.POINTER(OracleScene), C.c_void_p, C.c_int]
        L.port_list_emitters.restype = C.c_int
        L.port_use_emitters.argtypes = [C.c_void_p, C.c_int]
        L.port_use_emitters.restype = None
        n = L.port_list_emitters(C.byref(self.handle.struct), None, 0)
        self.emitters = np.zeros((max(n, 1), 4), np.int32)
        L.port_list_emitters(C.byref(self.handle.struct), _ptr(self.emitters), n)
        self.n_emitters = n

    def __enter__(self):
        self.port.lib.port_use_emitters(_ptr(self.emitters), self.n_emitters)
        self.port.lib.port_set_ext(*self.args)
        return self

    def __exit__(self, *exc):
        self.port.lib.port_set_ext(-1, 1, 0, 0)
        self.port.lib.port_use_emitters(None, 0)
        return False


class RefLib(_Lib):
    prefix = "ref"

    def __init__(self, path):
        super().__init__(path)
        self.lib.ref_pcg_stream.argtypes = [C.c_uint, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.ref_sun_basis.argtypes = [C.c_void_p, C.c_void_p]

    def pcg_stream(self, state: int, n: int):
        s = np.zeros(n, np.uint32)
        f = np.zeros(n, np.float32)
        self.lib.ref_pcg_stream(state, n, _ptr(s), _ptr(f))
        return s, f

    def sun_basis(self, sun: np.ndarray) -> np.ndarray:
        sun = np.ascontiguousarray(sun, np.int32)
        out = np.zeros(9, np.float32)
        self.lib.ref_sun_basis(_ptr(sun), _ptr(out))
        return out


class PortLib(_Lib):
    prefix = "port"

    def __init__(self, path):
        super().__init__(path)
        self.lib.port_counters_reset.restype = None
        self.lib.port_counters_read.argtypes = [C.c_void_p]
        self.lib.port_counters_enable.argtypes = [C.c_int]
        self.lib.port_render_gids.argtypes = [C.POINTER(OracleScene), C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                              C.c_int64, C.c_void_p, C.c_int]

    def render_gids(self, sc, seeds, gids, first_spp: int = 0, res: Optional[np.ndarray] = None,
                    threads: int = 8) -> np.ndarray:
        h = sc if isinstance(sc, SceneHandle) else SceneHandle(sc)
        seeds = np.ascontiguousarray(seeds, np.int32)
        gids = np.ascontiguousarray(gids, np.int32)
        if res is None:
            res = np.zeros(3 * h.width * h.height, np.float32)
        self.lib.port_render_gids(C.byref(h.struct), _ptr(seeds), len(seeds), first_spp, _ptr(gids), gids.size,
                                  _ptr(res), threads)
        return res

    def geom(self, which: int, a, b=None) -> np.ndarray:
        """dot (0) / cross (1) / normalize (2) / general fmod (3) on rows of 3 floats, as rt_math.h defines them."""
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(a if b is None else b, np.float32).reshape(-1, 3)
        out = np.zeros_like(a)
        self.lib.port_geom.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.port_geom.restype = None
        self.lib.port_geom(which, len(a), _ptr(a), _ptr(b), _ptr(out))
        return out

    def mirror_linear(self, s, w: int):
        s = np.ascontiguousarray(s, np.float32)
        i0, i1, a = np.zeros(s.size, np.int32), np.zeros(s.size, np.int32), np.zeros(s.size, np.float32)
        self.lib.port_mirror_linear.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.port_mirror_linear.restype = None
        self.lib.port_mirror_linear(s.size, _ptr(s), int(w), _ptr(i0), _ptr(i1), _ptr(a))
        return i0, i1, a

    def sample_linear(self, st, rgba) -> np.ndarray:
        st = np.ascontiguousarray(st, np.float32).reshape(-1, 2)
        rgba = np.ascontiguousarray(rgba, np.uint8)
        out = np.zeros((len(st), 4), np.float32)
        self.lib.port_sample_linear.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.lib.port_sample_linear.restype = None
        self.lib.port_sample_linear(len(st), _ptr(st), _ptr(rgba), rgba.shape[1], rgba.shape[0], _ptr(out))
        return out

    def gamma_scan(self, lo_bits: int, hi_bits: int, thresholds, threads: int = 8) -> int:
        """Bit patterns in [lo, hi] whose GAMMA byte breaks monotonicity or disagrees with the threshold table."""
        t = np.ascontiguousarray(thresholds, np.float32)
        assert t.size == 256
        self.lib.port_gamma_scan.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
        self.lib.port_gamma_scan.restype = C.c_int64
        return int(self.lib.port_gamma_scan(lo_bits, hi_bits, _ptr(t), threads))

    def counters(self, enable: Optional[bool] = None, reset: bool = False) -> dict:
        if enable is not None:
            self.lib.port_counters_enable(1 if enable else 0)
        out = np.zeros(len(COUNTER_NAMES), np.int64)
        self.lib.port_counters_read(_ptr(out))
        if reset:
            self.lib.port_counters_reset()
        return dict(zip(COUNTER_NAMES, (int(v) for v in out)))


def algorithmic_bytes(c: dict) -> float:
    """algorithmic bytes of the reference access stream, per sample."""
    total = (4 * c["node"] + 8 * c["block"] + 4 * c["model_hdr"] + 52 * c["aabb"] + 60 * c["quad"]
             + 24 * c["mat"] + 4 * c["texel"] + 56 * c["bvh_inner"] + 4 * c["leaf_hdr"] + 80 * c["tri"]
             + 16 * c["sky"] + 24 * c["samples"])
    return total / max(c["samples"], 1)


def usable_threads() -> int:
    """Worker threads worth starting for the checkers: the affinity mask cut down to the container's CPU-time quota (cgroup
    cpu.max).  Under a 16-CPU quota on a 256-thread host, 256 workers are slower than 16 (profiles/r04_cpu_sweep.jsonl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(period))))
    except Exception:
        pass
    return n


_ref: Optional[RefLib] = None
_port: Optional[PortLib] = None
_port_libm: Optional[PortLib] = None
_ref_libm: Optional[RefLib] = None


def _make(target: str) -> None:
    # CHUNKY_ORACLE_NO_BUILD=1 (set by tools/pmc.sh / kt_trace.sh after they have built everything): never spawn a child —
    # under `rocprofv3 --pmc` the profiler's preloaded library initialises the GPU in every child process, and a child
    # that then execs a compiler is exactly the exec-after-GPU-init this pool forbids
    if os.environ.get("CHUNKY_ORACLE_NO_BUILD"):
        return
    subprocess.run(["make", "-C", HERE, target], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.PIPE)


def ref(build: bool = True) -> Optional[RefLib]:
    """The reference-kernel oracle, or None where it cannot exist (no /root/reference and no
    prebuilt .so)."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libchunky_ref.so")
        if build and os.path.isdir("/root/reference"):
            _make("ref")
        if not os.path.exists(path):
            return None
        _ref = RefLib(path)
    return _ref


def port_libm(build: bool = True) -> PortLib:
    """oracle/port.c built with -DPORT_LIBM: the restatement on the second platform layer (glibc libm, unfused dot / cross /
    normalize — what ref_shim.cpp gives the compiled reference under REF_SHIM_LIBM)."""
    global _port_libm
    if _port_libm is None:
        path = os.path.join(HERE, "libchunky_port_libm.so")
        if build:
            _make("port_libm")
        _port_libm = PortLib(path)
    return _port_libm


def ref_libm(build: bool = True) -> Optional[RefLib]:
    """The compiled reference on the second platform layer, or None where it cannot exist."""
    global _ref_libm
    if _ref_libm is None:
        path = os.path.join(HERE, "_ref", "libchunky_ref_libm.so")
        if build and os.path.isdir("/root/reference"):
            _make("ref_libm")
        if not os.path.exists(path):
            return None
        _ref_libm = RefLib(path)
    return _ref_libm


def port(build: bool = True) -> PortLib:
    global _port
    if _port is None:
        path = os.path.join(HERE, "libchunky_port.so")
        if build:
            _make("port")
        _port = PortLib(path)
    return _port
