"""The float contract of rt_math.h: accuracy against mpmath within the OpenCL 1.2 ULP bounds, the
special-value rules the tracer relies on, and agreement with the values the reference build saw
(tests/golden/kats.npz, produced through oracle/_ref's builtin shim)."""
import os

import mpmath as mp
import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SIN, COS, ASIN, ACOS, ATAN2, FMOD1, FMIN, FMAX, SQRT, DIV = range(10)


def ulp_err(got: np.ndarray, exact) -> np.ndarray:
    got = got.astype(np.float64)
    exact = np.array([float(e) for e in exact])
    ulp = np.spacing(np.abs(exact).astype(np.float32)).astype(np.float64)
    return np.abs(got - exact) / ulp


@pytest.mark.parametrize("which,fn,lo,hi,bound", [
    (SIN, mp.sin, -7.0, 7.0, 4), (COS, mp.cos, -7.0, 7.0, 4),
    (ASIN, mp.asin, -1.0, 1.0, 4), (ACOS, mp.acos, -1.0, 1.0, 4)])
def test_unary_within_opencl_ulp(port, which, fn, lo, hi, bound):
    rng = np.random.default_rng(which)
    x = np.concatenate([np.linspace(lo, hi, 3001), rng.uniform(lo, hi, 3000)]).astype(np.float32)
    got = port.math(which, x)
    exact = [fn(mp.mpf(float(v))) for v in x]
    err = ulp_err(got, exact)
    # absolute floor near zeros of sin/cos (OpenCL bounds are relative; reduced-argument error
    # of a float input is what any single-precision implementation has there)
    ok = (err <= bound) | (np.abs(got.astype(np.float64) - np.array([float(e) for e in exact])) < 2e-7)
    assert ok.all(), (x[~ok][:5], err[~ok][:5])


def test_atan2_within_opencl_ulp(port):
    rng = np.random.default_rng(4)
    y = rng.normal(size=5000).astype(np.float32)
    x = rng.normal(size=5000).astype(np.float32)
    got = port.math(ATAN2, y, x)
    exact = [mp.atan2(mp.mpf(float(a)), mp.mpf(float(b))) for a, b in zip(y, x)]
    assert ulp_err(got, exact).max() <= 6


def test_domain_and_specials(port):
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    assert np.isnan(port.math(ASIN, [1.0000001, -2.0, nan])).all()
    assert np.isnan(port.math(ACOS, [1.0000001, -2.0, nan])).all()
    assert port.math(ACOS, [1.0])[0] == 0.0
    np.testing.assert_array_equal(port.math(ATAN2, [0.0, 1.0, -1.0], [1.0, 0.0, 0.0]).view(np.uint32),
                                  np.array([0.0, np.float32(np.pi / 2), -np.float32(np.pi / 2)], np.float32).view(np.uint32))
    # fmin/fmax return the non-NaN operand (K/primitives.h:37-41 rely on it for 0*inf)
    np.testing.assert_array_equal(port.math(FMIN, [nan, 2.0, 3.0], [1.0, nan, -inf]), [1.0, 2.0, -inf])
    np.testing.assert_array_equal(port.math(FMAX, [nan, 2.0, 3.0], [1.0, nan, inf]), [1.0, 2.0, inf])
    # signed zeros ordered -0 < +0
    z = port.math(FMIN, [0.0, -0.0], [-0.0, 0.0])
    assert np.signbit(z).all()
    z = port.math(FMAX, [0.0, -0.0], [-0.0, 0.0])
    assert not np.signbit(z).any()
    # fmod(x, 1) is exact and keeps the sign of x
    x = np.array([2.75, -2.75, 0.5, -0.5, 1e9, -3.0], np.float32)
    np.testing.assert_array_equal(port.math(FMOD1, x), np.fmod(x, np.float32(1)))


def test_matches_reference_build_kats(port):
    k = np.load(os.path.join(GOLD, "kats.npz"))
    for which, xs, key in ((SIN, "xs", "sin"), (COS, "xs", "cos"), (ASIN, "us", "asin"), (ACOS, "us", "acos"),
                           (FMOD1, "xs", "fmod1")):
        np.testing.assert_array_equal(port.math(which, k[xs]).view(np.uint32), k[key].view(np.uint32), err_msg=key)
    np.testing.assert_array_equal(port.math(ATAN2, k["ya"], k["xa"]).view(np.uint32), k["atan2"].view(np.uint32))


# ---- the rest of the platform layer: geometric builtins, general fmod, the linear / mirrored-repeat sampler ------------
EPS = float(np.finfo(np.float32).eps)


def test_dot_and_cross_within_opencl_tolerance(port):
    """OpenCL 1.2 section 7.4: dot has an absolute error of at most max|a_i| * max|b_i| * (2n - 1) * FLT_EPSILON,
    cross of max|a_i| * max|b_i| * 3 * FLT_EPSILON per component — against exact (mpmath) values."""
    rng = np.random.default_rng(31)
    a = (rng.normal(size=(4000, 3)) * 10.0 ** rng.integers(-3, 4, (4000, 1))).astype(np.float32)
    b = (rng.normal(size=(4000, 3)) * 10.0 ** rng.integers(-3, 4, (4000, 1))).astype(np.float32)
    b[:500] = a[:500] * np.float32(-1.0000001)            # heavy cancellation in the dot product
    a[500:1000] = b[500:1000]                              # cross of (nearly) parallel vectors
    scale = np.abs(a).max(axis=1).astype(np.float64) * np.abs(b).max(axis=1)
    dot = port.geom(0, a, b)[:, 0].astype(np.float64)
    exact = np.array([float(mp.fdot([mp.mpf(float(x)) for x in u], [mp.mpf(float(x)) for x in v])) for u, v in zip(a, b)])
    assert (np.abs(dot - exact) <= 5 * EPS * scale).all()
    cr = port.geom(1, a, b).astype(np.float64)
    A, B = a.astype(np.float64), b.astype(np.float64)        # products of two floats are exact in binary64; one rounding in the sum
    ex = np.stack([A[:, 1] * B[:, 2] - A[:, 2] * B[:, 1], A[:, 2] * B[:, 0] - A[:, 0] * B[:, 2], A[:, 0] * B[:, 1] - A[:, 1] * B[:, 0]], 1)
    assert (np.abs(cr - ex) <= 3 * EPS * scale[:, None] * (1 + 1e-9)).all()
    # and the definitions themselves: one fused chain, fixed association (rt_math.h) — evaluated here step by step
    f32 = lambda x: np.float32(x)
    for u, v, d in zip(a[:300], b[:300], port.geom(0, a[:300], b[:300])[:, 0]):
        t = f32(np.float64(u[0]) * np.float64(v[0]))
        t = f32(np.float64(u[1]) * np.float64(v[1]) + np.float64(t))
        t = f32(np.float64(u[2]) * np.float64(v[2]) + np.float64(t))
        assert t.view(np.uint32) == d.view(np.uint32)


def test_normalize_within_opencl_ulp(port):
    """normalize: at most 2 + n = 5 ULP per component (OpenCL 1.2 section 7.4), for lengths that do not overflow."""
    rng = np.random.default_rng(32)
    a = (rng.normal(size=(5000, 3)) * 10.0 ** rng.integers(-6, 7, (5000, 1))).astype(np.float32)
    got = port.geom(2, a)
    for u, g in zip(a[:1500], got[:1500]):
        n = mp.sqrt(sum(mp.mpf(float(x)) ** 2 for x in u))
        exact = [mp.mpf(float(x)) / n for x in u]
        assert ulp_err(g, exact).max() <= 5, (u, g)
    ln = np.sqrt((got.astype(np.float64) ** 2).sum(axis=1))
    assert np.abs(ln - 1).max() < 4 * EPS


def test_general_fmod_is_exact(port):
    """fmod(x, y) for any y (the kernel only calls fmod(., 1), K/sky.h:102; the general branch is C's fmod): exact by
    definition, sign of x."""
    rng = np.random.default_rng(33)
    x = (rng.normal(size=4000) * 10.0 ** rng.integers(-3, 6, 4000)).astype(np.float32)
    y = (rng.normal(size=4000) * 10.0 ** rng.integers(-3, 3, 4000)).astype(np.float32)
    y[:400] = 1.0
    a = np.zeros((4000, 3), np.float32); a[:, 0] = x
    b = np.zeros((4000, 3), np.float32); b[:, 0] = y
    got = port.geom(3, a, b)[:, 0]
    want = np.array([float(mp.fmod(mp.mpf(float(p)), mp.mpf(float(q)))) if q != 0 else np.nan for p, q in zip(x, y)])
    want = np.where(np.signbit(x) & (want > 0), want - np.abs(y), np.where(~np.signbit(x) & (want < 0), want + np.abs(y), want))
    ok = (got.astype(np.float64) == want) | (np.isnan(got) & np.isnan(want))
    assert ok.all(), (x[~ok][:4], y[~ok][:4], got[~ok][:4], want[~ok][:4])


def test_mirror_linear_follows_the_opencl_sampler_formulae(port):
    """OpenCL 1.2 section 8.2, CLK_ADDRESS_MIRRORED_REPEAT + CLK_FILTER_LINEAR with normalised coordinates:
    s' = |s - 2 rint(s / 2)|, u = s' w, i0 = floor(u - 0.5), i1 = i0 + 1, both clamped to [0, w - 1], weight
    a = frac(u - 0.5).  Evaluated here in binary64 from the formulae; the float32 contract must pick the same texels
    away from texel boundaries, the same weights to float precision, and the same filtered values everywhere."""
    rng = np.random.default_rng(34)
    for w in (16, 100, 128):
        s = np.concatenate([rng.uniform(-3, 3, 6000), np.linspace(-2, 2, 2001), (np.arange(4 * w + 1) - 2 * w) / w,
                            (np.arange(2 * w) + 0.5) / w]).astype(np.float32)
        i0, i1, a = port.mirror_linear(s, w)
        S = s.astype(np.float64)
        sp = np.abs(S - 2 * np.rint(S / 2))
        um = sp * w - 0.5
        f0 = np.floor(um)
        e0, e1, ea = np.clip(f0, 0, w - 1).astype(int), np.clip(f0 + 1, 0, w - 1).astype(int), um - f0
        assert ((0 <= i0) & (i0 <= i1) & (i1 <= w - 1) & (a >= 0) & (a < 1)).all()
        away = np.minimum(ea, 1 - ea) > 1e-4             # u - 0.5 not within 1e-4 of an integer
        np.testing.assert_array_equal(i0[away], e0[away])
        np.testing.assert_array_equal(i1[away], e1[away])
        assert np.abs(a[away] - ea[away]).max() < 1e-4
        # the filtered value is continuous across texel boundaries: compare on a random 1-D texture everywhere
        tex = rng.uniform(0, 1, w)
        got = (1 - a.astype(np.float64)) * tex[i0] + a * tex[i1]
        want = (1 - ea) * tex[e0] + ea * tex[e1]
        assert np.abs(got - want).max() < 2e-4
    # two axes + UNORM8 conversion: the sampler on an RGBA8 image against the same formulae in binary64
    img = rng.integers(0, 256, (32, 64, 4)).astype(np.uint8)
    st = rng.uniform(-2, 2, (3000, 2)).astype(np.float32)
    got = port.sample_linear(st, img).astype(np.float64)
    def axis(c, n):
        sp = np.abs(c - 2 * np.rint(c / 2)); um = sp * n - 0.5; f0 = np.floor(um)
        return np.clip(f0, 0, n - 1).astype(int), np.clip(f0 + 1, 0, n - 1).astype(int), um - f0
    x0, x1, ax = axis(st[:, 0].astype(np.float64), 64)
    y0, y1, ay = axis(st[:, 1].astype(np.float64), 32)
    T = img.astype(np.float64) / 255.0
    want = ((1 - ax) * (1 - ay))[:, None] * T[y0, x0] + (ax * (1 - ay))[:, None] * T[y0, x1] + \
           ((1 - ax) * ay)[:, None] * T[y1, x0] + (ax * ay)[:, None] * T[y1, x1]
    assert np.abs(got - want).max() < 3e-4
