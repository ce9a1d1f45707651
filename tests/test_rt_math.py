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
