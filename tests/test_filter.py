"""The tone-map path (`filter`, tonemap/include/post_processing_filter.cl:5-51).

CPU: oracle/port.c against the words the reference kernel produced (tests/golden/filter.npz, written by
oracle/_ref), against oracle/_ref itself where it is built, and against an independent float64 evaluation
of the same curves; rt_pow against mpmath.  GPU: the HIP kernel through the C ABI, bit for bit.
"""
import os

import mpmath as mp
import numpy as np
import pytest

from oracle import binding

GOLD = os.path.join(os.path.dirname(__file__), "golden", "filter.npz")


def curves64(x, exposure, type_):
    """The four curves in float64 (what the reference computes in float32)."""
    c = x.astype(np.float32).astype(np.float64) * float(np.float32(exposure))
    if type_ == 0:
        c = c ** (1 / 2.2)
    elif type_ == 1:
        c = np.maximum(0, c - 0.004)
        c = (c * (6.2 * c + 0.5)) / (c * (6.2 * c + 1.7) + 0.06)
    elif type_ == 2:
        c = (c * (2.51 * c + 0.03)) / (c * (2.43 * c + 0.59) + 0.14)
        c = np.clip(c, 0, 1) ** (1 / 2.2)
    elif type_ == 3:
        def h(v):
            return ((v * (0.15 * v + 0.10 * 0.50) + 0.20 * 0.02) / (v * (0.15 * v + 0.50) + 0.20 * 0.30)) - 0.02 / 0.30
        c = h(16 * c) / h(11.2)
    return np.clip(np.floor(c * 255 + 0.5), 0, 255).astype(np.int64)


def channels(argb):
    argb = argb.astype(np.uint32)
    return np.stack([(argb >> 16) & 255, (argb >> 8) & 255, argb & 255], axis=-1).reshape(-1).astype(np.int64)


def test_port_matches_reference_goldens(port):
    g = np.load(GOLD)
    for ti, t in enumerate(g["types"]):
        for ei, e in enumerate(g["exposures"]):
            got = port.filter(g["samples"], float(e), int(t))
            np.testing.assert_array_equal(got, g["argb"][ti, ei], err_msg=f"type {t} exposure {e}")
    np.testing.assert_array_equal(port.pow(g["pow_a"], g["pow_b"]).view(np.uint32), g["pow"].view(np.uint32))


def test_port_matches_reference_build(port, ref):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0, 3, 30000), 10.0 ** rng.uniform(-300, 6, 3000)])
    for t in (0, 1, 2, 3, -1):
        for e in (1.0, 0.123, 1.9):
            np.testing.assert_array_equal(port.filter(x, e, t), ref.filter(x, e, t))


def test_goldens_are_the_curves(port):
    """The reference words agree with a float64 evaluation of the same formulae to one code value
    (float32 rounding), alpha is always 0xFF, and an unknown type applies the exposure only."""
    g = np.load(GOLD)
    x = g["samples"]
    for ti, t in enumerate(g["types"]):
        for ei, e in enumerate(g["exposures"]):
            argb = g["argb"][ti, ei]
            assert ((argb >> 24) == 0xFF).all()
            want = curves64(x, e, int(t))
            assert np.abs(channels(argb) - want).max() <= 1, (t, e)


def test_filter_edge_values(port):
    # empty frame, a ragged pixel count, zero / subnormal / huge samples; out-of-domain samples saturate
    assert port.filter(np.zeros(0), 1.0, 0).size == 0
    x = np.array([0.0, 5e-324, 1e-310, 1e30, 1e300, np.inf, -1.0, np.nan, 0.5], np.float64)
    for t in range(4):
        w = port.filter(x, 1.0, t)
        assert w.size == 3 and ((w >> 24) == 0xFF).all()
        assert (w[0] & 0xFFFFFF) in (0x000000,) or t == 3  # zero samples stay black (HABLE: a rounding residue of 0)
    w = port.filter(x, 1.0, 0)
    assert w[1] == 0xFFFFFFFF and w[2] == 0xFF0000BA  # 1e30, 1e300->inf, inf clamp to 255; -1 -> NaN -> 0; NaN -> 0; 0.5^(1/2.2)


def test_pow_accuracy_and_specials(port):
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.uniform(0, 4, 1500), 10.0 ** rng.uniform(-44, 8, 500)]).astype(np.float32)
    b = np.concatenate([rng.uniform(-3, 3, 1000), np.full(1000, 1.0 / 2.2)]).astype(np.float32)
    got = port.pow(a, b).astype(np.float64)
    exact = np.array([float(mp.power(mp.mpf(float(x)), mp.mpf(float(y)))) for x, y in zip(a, b)])
    fin = np.isfinite(exact) & (np.abs(exact) > 1e-37) & (np.abs(exact) < 3e38)
    ulp = np.spacing(np.abs(exact[fin]).astype(np.float32)).astype(np.float64)
    assert (np.abs(got[fin] - exact[fin]) / ulp).max() <= 0.501  # OpenCL allows 16
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    cases = [(0.0, 0.5, 0.0), (-0.0, 3.0, -0.0), (0.0, -2.0, inf), (-0.0, -3.0, -inf), (inf, 0.5, inf), (inf, -1.0, 0.0),
             (-inf, 3.0, -inf), (-8.0, 3.0, -512.0), (-8.0, 2.0, 64.0), (2.0, inf, inf), (0.5, inf, 0.0), (2.0, -inf, 0.0),
             (-1.0, inf, 1.0), (1.0, nan, 1.0), (nan, 0.0, 1.0), (2.0, -150.0, 0.0), (2.0, 128.0, inf), (2.0, -149.0, 1.4e-45)]
    x = np.array([c[0] for c in cases], np.float32)
    y = np.array([c[1] for c in cases], np.float32)
    want = np.array([c[2] for c in cases], np.float32)
    np.testing.assert_array_equal(port.pow(x, y).view(np.uint32), want.view(np.uint32))
    assert np.isnan(port.pow(np.array([-1.0, nan, 2.0], np.float32), np.array([0.5, 1.0, nan], np.float32))).all()


def test_gamma_thresholds_are_the_gamma_curve(port):
    """The HIP kernel evaluates the GAMMA and ACES curves' last steps — pow(c, 1/2.2) * 255 + 0.5 -> (uint) -> min 255 — by
    comparing c with 255 thresholds instead of evaluating pow.  That is the same function iff the byte never decreases with
    c and steps exactly at the thresholds: checked here for EVERY float from 0 to 1.125 (1.07e9 bit patterns, the C
    restatement's rt_pow), and above on a sample up to +inf; NaN and negative values give byte 0 either way."""
    import os
    from chunkyclplugin_amd import native
    T = np.zeros(256, np.float32)
    native.check(native.lib().chunky_filter_gamma_thresholds(T.ctypes.data))
    assert T[0] == 0 and (np.diff(T[1:]) > 0).all() and 0.99 < T[255] < 1.0
    assert port.gamma_scan(0, 0x3F900000, T, threads=binding.usable_threads()) == 0
    for lo in range(0x3F900000, 0x7F800000, 0x01000000):            # the rest of the positive floats, 65 536 at a time
        assert port.gamma_scan(lo, lo + 0xFFFF, T, threads=2) == 0
    assert port.gamma_scan(0x7F7F0000, 0x7F800000, T, threads=2) == 0   # up to +inf
    # NaN, negative, -0: pow gives NaN / NaN / +0 -> byte 0, and every comparison with a threshold fails for them too;
    # -inf is the exception on both sides: pow(-inf, 1/2.2) = +inf (C99) -> byte 255
    x = np.array([np.nan, -1.0, -0.0, -1e-30, -np.inf], np.float64)
    words = port.filter(np.repeat(x, 3), 1.0, 0)
    assert (words[:4] == 0xFF000000).all() and words[4] == 0xFFFFFFFF


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_gamma_thresholds_edge_values(gpu_instance, port):
    """GAMMA and ACES on the device at, just below and just above every threshold, and on special values: the words of the
    C restatement (which evaluates pow)."""
    from chunkyclplugin_amd import native
    T = np.zeros(256, np.float32)
    native.check(native.lib().chunky_filter_gamma_thresholds(T.ctypes.data))
    bits = T[1:].view(np.uint32).astype(np.int64)
    near = np.concatenate([bits + d for d in (-2, -1, 0, 1, 2)]).astype(np.uint32).view(np.float32)
    rng = np.random.default_rng(12)
    x = np.concatenate([near, [0.0, -0.0, np.inf, -np.inf, np.nan, -1.0, 1.0, 0.5, 1e-45, 1e-38, 3e38, 2.0],
                        rng.uniform(0, 1.2, 30000), 10.0 ** rng.uniform(-45, 5, 5000)]).astype(np.float64)
    x = np.concatenate([x, np.zeros((-x.size) % 3)])
    n = x.size // 3
    for t in (0, 2):
        for e in (1.0, 0.73, 2.5):
            out = np.zeros(n, np.uint32)
            native.check(native.lib().chunky_filter_frame(gpu_instance._h, n, 1, e, native.ptr(x), native.ptr(out), t))
            np.testing.assert_array_equal(out, port.filter(x, e, t), err_msg=f"type {t} exposure {e}")


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [0, 2])
def test_gpu_gamma_fast_path_every_float(gpu_instance, curve):
    """The device evaluates the GAMMA / ACES byte from a hardware log2 / exp2 (ACES: and reciprocal) estimate and consults the
    threshold table (ACES: after the correctly rounded division) only when the estimate lies within 1/8192 of a step.  That
    is the byte of the reference's arithmetic for EVERY float: all 2^32 bit patterns (NaNs, negative values, denormals,
    infinities included) through both, and the estimate never strays further beyond its byte's interval than a quarter of
    the guard band."""
    bad, worst = 0, 0.0
    for part in range(4):
        b, w = gpu_instance.selftest_gamma_scan(curve, part << 30, 1 << 30)
        bad += b
        worst = max(worst, w)
    assert bad == 0
    assert worst < 0.25 / 8192, worst


@pytest.mark.gpu
def test_gpu_filter_matches_reference_goldens(gpu_instance):
    from chunkyclplugin_amd.renderer import HipPostProcessingFilter
    g = np.load(GOLD)
    x = np.ascontiguousarray(g["samples"])
    n = x.size // 3
    ids = {0: "GAMMA", 1: "TONEMAP1", 2: "TONEMAP2", 3: "TONEMAP3"}
    for ti, t in enumerate(g["types"]):
        if int(t) not in ids:
            continue
        f = HipPostProcessingFilter(ids[int(t)], gpu_instance)
        assert f.get_id() == ids[int(t)]
        for ei, e in enumerate(g["exposures"]):
            out = np.zeros(n, np.uint32)
            f.process_frame(n, 1, x, out, float(e))
            np.testing.assert_array_equal(out, g["argb"][ti, ei], err_msg=f"type {t} exposure {e}")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(0, 0), (1, 1), (17, 3), (511, 1), (512, 1), (513, 1), (640, 360), (1921, 1081)])
def test_gpu_filter_matches_oracle(gpu_instance, port, shape):
    from chunkyclplugin_amd import native
    w, h = shape
    rng = np.random.default_rng(w * 7 + h)
    x = np.concatenate([rng.uniform(0, 2.5, 3 * w * h - 3 * (w * h // 5)), 10.0 ** rng.uniform(-320, 9, 3 * (w * h // 5))])
    rng.shuffle(x)
    if x.size >= 9:
        x[:9] = [0.0, 5e-324, 1e-310, 1e30, 1e300, np.inf, -1.0, np.nan, 0.5]
    for t in (0, 1, 2, 3, 9):
        for e in (1.0, 0.61):
            out = np.full(w * h, 0xDEADBEEF, np.uint32)
            native.check(native.lib().chunky_filter_frame(gpu_instance._h, w, h, e, native.ptr(x) if x.size else None,
                                                          native.ptr(out) if out.size else None, t))
            np.testing.assert_array_equal(out, port.filter(x, e, t), err_msg=f"{shape} type {t} exposure {e}")


class _Hip:
    """hipMalloc / hipMemcpy straight from the HIP runtime the C-ABI library already loaded (no torch: a second
    copy of the runtime, loaded after the first has initialised the device, finds no GPU)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        loaded = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln]
        assert loaded, "the C-ABI library has not loaded a HIP runtime"
        self.rt = C.CDLL(loaded[0])
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]

    def alloc(self, nbytes):
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), nbytes) == 0
        return p.value

    def upload(self, dst, arr):
        assert self.rt.hipMemcpy(dst, arr.ctypes.data, arr.nbytes, 1) == 0

    def download(self, arr, src):
        assert self.rt.hipMemcpy(arr.ctypes.data, src, arr.nbytes, 2) == 0

    def free(self, p):
        self.rt.hipFree(p)


@pytest.mark.gpu
def test_gpu_filter_device_buffers_and_pow(gpu_instance, port):
    from chunkyclplugin_amd.renderer import HipPostProcessingFilter
    hip = _Hip()
    n = 1920 * 1080 + 37  # ragged last tile
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 2, 3 * n)
    want = port.filter(x, 1.25, 2)
    d_in, d_out = hip.alloc(x.nbytes + 8), hip.alloc(4 * n)
    out = np.zeros(n, np.uint32)
    f = HipPostProcessingFilter("TONEMAP2", gpu_instance)
    hip.upload(d_in, x)
    ms = f.process_device(n, 1.25, d_in, d_out, repeat=3)
    assert ms > 0
    hip.download(out, d_out)
    np.testing.assert_array_equal(out, want)
    # a source that is only 8-byte aligned takes the scalar-load path; same words
    hip.upload(d_in + 8, x)
    hip.upload(d_out, np.zeros(n, np.uint32))
    f.process_device(n, 1.25, d_in + 8, d_out)
    hip.download(out, d_out)
    np.testing.assert_array_equal(out, want)
    hip.free(d_in)
    hip.free(d_out)
    # rt_pow on the device == on the host
    a = np.concatenate([rng.uniform(0, 4, 60000), 10.0 ** rng.uniform(-44, 8, 5000), [0, -0.0, np.inf, -np.inf, np.nan, 1, -8, -8]])
    b = np.concatenate([rng.uniform(-3, 3, 35000), np.full(30000, 1 / 2.2), [0.5, 3, -1, 3, 1, np.nan, 3, 2.5]])
    a, b = a.astype(np.float32), b.astype(np.float32)
    dev, host = gpu_instance.selftest_math(16, a, b), port.pow(a, b)
    ok = (dev.view(np.uint32) == host.view(np.uint32)) | (np.isnan(dev) & np.isnan(host))
    assert ok.all(), (a[~ok][:4], b[~ok][:4], dev[~ok][:4], host[~ok][:4])
