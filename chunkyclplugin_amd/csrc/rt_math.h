/* rt_math.h — the float arithmetic contract of the chunky-hip path tracer.
 *
 * The reference kernel (ThatRedox/ChunkyClPlugin, src/main/opencl/kernel/include/ headers) leans on
 * OpenCL C builtins whose results the OpenCL 1.2 spec only bounds in ULPs: sin, cos, acos, asin,
 * atan2, fmod, dot, cross, normalize, and the image samplers.  Two conforming OpenCL drivers
 * therefore produce different bits.  To make "same seeds -> same hits" a testable statement, this
 * header DEFINES those builtins once, from exactly-rounded IEEE-754 binary32 primitives
 * (+ - * / sqrt fma floor rint trunc) that x86-64 and gfx950 both implement identically.  It is
 * included by
 *   - the HIP kernels (device code, gfx950),
 *   - oracle/ref_shim.cpp, which supplies these functions as the OpenCL builtins when the
 *     reference rayTracer.cl is compiled for x86-64 in place,
 *   - oracle/port.c, the plain-C restatement of the reference algorithm.
 * Accuracy is checked against mpmath/numpy in tests/test_rt_math.py (all within the OpenCL 1.2
 * ULP bounds: sin/cos/asin/acos <= 4, atan2 <= 6).
 *
 * Rules for everything in this file: no reliance on compiler contraction (build with
 * -ffp-contract=off everywhere), fused multiply-adds only where written as rt_fma, no libm
 * transcendental calls, no fast-math.
 */
#ifndef CHUNKY_RT_MATH_H
#define CHUNKY_RT_MATH_H

#if defined(__HIPCC__)
#define RT_FN __host__ __device__ static __forceinline__
#else
#define RT_FN static inline __attribute__((always_inline))
#endif

#define RT_PI_F 3.14159274101257f      /* OpenCL M_PI_F   */
#define RT_PI_2_F 1.57079637050629f    /* OpenCL M_PI_2_F */
#define RT_1_PI_F 0.31830987334251f    /* OpenCL M_1_PI_F */

RT_FN unsigned rt_f2u(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return u; }
RT_FN float rt_u2f(unsigned u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

RT_FN float rt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RT_FN float rt_fabs(float x) { return __builtin_fabsf(x); }
RT_FN float rt_sqrt(float x) { return __builtin_sqrtf(x); }
RT_FN float rt_floor(float x) { return __builtin_floorf(x); }
RT_FN float rt_trunc(float x) { return __builtin_truncf(x); }
RT_FN float rt_rint(float x) { return __builtin_rintf(x); }
RT_FN int rt_isnan(float x) { return x != x; }
RT_FN float rt_nan(void) { return rt_u2f(0x7fc00000u); }
RT_FN float rt_inf(void) { return rt_u2f(0x7f800000u); }

/* OpenCL fmin/fmax: if one operand is NaN return the other (IEEE minNum/maxNum). The slab tests of
 * the reference (primitives.h:37-41,59-60) depend on this for 0*inf.  Signed zeros are ordered
 * -0 < +0, which is what gfx950's v_min_f32 / v_max_f32 do; the host form spells that out so both
 * sides agree bit for bit (checked on hardware by chunky_selftest_math). */
#if defined(__HIP_DEVICE_COMPILE__)
RT_FN float rt_fmin(float a, float b) { return __builtin_fminf(a, b); }
RT_FN float rt_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
#else
RT_FN float rt_fmin(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a < b) return a;
    if (b < a) return b;
    return (rt_f2u(a) >> 31) ? a : b;
}
RT_FN float rt_fmax(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a < b) return b;
    if (b < a) return a;
    return (rt_f2u(a) >> 31) ? b : a;
}
#endif
/* OpenCL clamp(x, lo, hi) = fmin(fmax(x, lo), hi) */
RT_FN float rt_clamp(float x, float lo, float hi) { return rt_fmin(rt_fmax(x, lo), hi); }
RT_FN int rt_clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* dot / cross / normalize: fused chains, fixed association. */
RT_FN float rt_dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return rt_fma(az, bz, rt_fma(ay, by, ax * bx));
}
/* one component of a cross product: a*b - c*d */
RT_FN float rt_cross_c(float a, float b, float c, float d) { return rt_fma(a, b, -(c * d)); }
/* 1/|v| used by normalize: v * rt_rlen3(v) */
RT_FN float rt_rlen3(float x, float y, float z) { return 1.0f / rt_sqrt(rt_dot3(x, y, z, x, y, z)); }

/* fmod(x, 1.0f): exact (x - trunc(x) is representable), sign of x kept like C fmod. */
RT_FN float rt_fmod1(float x) {
    float r = x - rt_trunc(x);
    return __builtin_copysignf(r, x);
}

/* ---- sin / cos ---------------------------------------------------------------------------
 * Cody-Waite reduction by pi/2 in three float parts (fma keeps k*part exact enough for |x|<=1e4),
 * then the classic minimax polynomials on [-pi/4, pi/4] (coefficients: Cephes sinf/cosf).
 * Domain used by the path tracer: [0, 2pi) bounce angles, sun altitude/azimuth, 0.03. */
RT_FN void rt_sincos(float x, float* s_out, float* c_out) {
    float kf = rt_rint(x * 0.6366197466850281f);
    int k = (int)kf;
    float r = rt_fma(kf, -1.5707963705062866f, x);
    r = rt_fma(kf, 4.371138828673793e-08f, r);
    r = rt_fma(kf, 1.7151245100058819e-15f, r);
    float z = r * r;
    float sp = rt_fma(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = rt_fma(sp, z, -1.6666654611e-1f);
    float s = rt_fma(sp * z, r, r);
    float cp = rt_fma(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = rt_fma(cp, z, 4.166664568298827e-2f);
    float c = rt_fma(cp * z, z, rt_fma(z, -0.5f, 1.0f));
    float ss = (k & 1) ? c : s;
    float cc = (k & 1) ? s : c;
    if (k & 2) ss = -ss;
    if ((k + 1) & 2) cc = -cc;
    *s_out = ss;
    *c_out = cc;
}
RT_FN float rt_sin(float x) { float s, c; rt_sincos(x, &s, &c); return s; }
RT_FN float rt_cos(float x) { float s, c; rt_sincos(x, &s, &c); return c; }

/* ---- asin / acos -------------------------------------------------------------------------
 * |x| <= 0.5: x + x*z*P(z), z = x*x (Cephes asinf polynomial); otherwise the half-angle identity
 * asin(x) = pi/2 - 2*asin(sqrt((1-|x|)/2)).  |x| > 1 -> NaN. */
RT_FN float rt_asin_poly(float x, float z) {
    float p = rt_fma(z, 4.2163199048e-2f, 2.4181311049e-2f);
    p = rt_fma(p, z, 4.5470025998e-2f);
    p = rt_fma(p, z, 7.4953002686e-2f);
    p = rt_fma(p, z, 1.6666752422e-1f);
    return rt_fma(p * z, x, x);
}
/* (Both ranges through ONE evaluation of the polynomial, its operands selected first: a wave whose lanes fall into both ranges —
 * the usual case — otherwise runs it twice.  Per value the operations are the same.) */
RT_FN float rt_asin(float x) {
    float a = rt_fabs(x);
    if (!(a <= 1.0f)) return rt_nan();
    const int small = a <= 0.5f;
    const float z = small ? a * a : (1.0f - a) * 0.5f;
    const float s = small ? a : rt_sqrt(z);
    const float t = rt_asin_poly(s, z);
    /* pi/2 - 2t, with pi/2 split hi+lo so the subtraction keeps the low bits */
    const float r = small ? t : rt_fma(-2.0f, t, 1.5707963705062866f) + -4.371138828673793e-08f;
    return __builtin_copysignf(r, x);
}
RT_FN float rt_acos(float x) {
    float a = rt_fabs(x);
    if (!(a <= 1.0f)) return rt_nan();
    const int small = a <= 0.5f;
    const float z = small ? x * x : (1.0f - a) * 0.5f;
    const float s = small ? x : rt_sqrt(z);
    const float t = rt_asin_poly(s, z);
    if (small) return (1.5707963705062866f - t) + -4.371138828673793e-08f;
    if (x > 0.0f) return 2.0f * t;
    return rt_fma(-2.0f, t, 3.1415927410125732f) + -8.742277657347586e-08f;
}

/* ---- atan2 -------------------------------------------------------------------------------
 * atan on [0, inf) by the Cephes atanf reduction (tan(3pi/8), tan(pi/8)), quadrant fix-up as C
 * atan2f for finite inputs; atan2(0,0) = 0 with the sign rules of C for the cases the tracer can
 * reach (+0/-0 y with x>=0 -> +-0, x<0 -> +-pi). */
RT_FN float rt_atan_pos(float t) { /* t >= 0 */
    float y0, x;
    if (t > 2.414213562373095f) {
        y0 = 1.5707963705062866f;
        x = -(1.0f / t);
    } else if (t > 0.4142135623730950f) {
        y0 = 0.7853981852531433f;
        x = (t - 1.0f) / (t + 1.0f);
    } else {
        y0 = 0.0f;
        x = t;
    }
    float z = x * x;
    float p = rt_fma(z, 8.05374449538e-2f, -1.38776856032e-1f);
    p = rt_fma(p, z, 1.99777106478e-1f);
    p = rt_fma(p, z, -3.33329491539e-1f);
    float r = rt_fma(p * z, x, x);
    return y0 + r;
}
RT_FN float rt_atan2(float y, float x) {
    if ((x != x) || (y != y)) return rt_nan();
    float ay = rt_fabs(y), ax = rt_fabs(x);
    float r;
    if (ax == 0.0f && ay == 0.0f) {
        r = 0.0f;
    } else if (ay == rt_inf() && ax == rt_inf()) {
        r = 0.7853981852531433f;
    } else {
        /* atan(ay/ax) for ax >= ay, pi/2 - atan(ax/ay) otherwise: ONE quotient and ONE evaluation of rt_atan_pos, the operands
         * selected first (a wave with lanes on both sides otherwise divides twice and runs the polynomial twice) */
        const int wide = ax >= ay;
        const float q = rt_atan_pos((wide ? ay : ax) / (wide ? ax : ay));
        r = wide ? q : 1.5707963705062866f - q;
    }
    if (rt_f2u(x) >> 31) r = 3.1415927410125732f - r; /* x negative (incl. -0) */
    return __builtin_copysignf(r, y);
}

/* ---- image sampling contract (gfx950 has no image instructions; images are flat RGBA8) -----
 * UNORM8 -> float: b * fl(1/255); exact at 0 and 255 as the OpenCL spec requires. */
RT_FN float rt_unorm8(unsigned b) { return (float)b * 0.003921568859368563f; }

/* CLK_NORMALIZED_COORDS_TRUE | CLK_ADDRESS_MIRRORED_REPEAT | CLK_FILTER_LINEAR along one axis
 * (OpenCL 1.2 spec 8.2): s' = |s - 2*rint(s/2)|, u = s'*w, i0 = floor(u-0.5), i1 = i0+1, both
 * clamped to [0, w-1], weight a = frac(u-0.5). */
RT_FN void rt_mirror_linear(float s, int w, int* i0, int* i1, float* a) {
    float sp = 2.0f * rt_rint(0.5f * s);
    sp = rt_fabs(s - sp);
    float u = sp * (float)w;
    float um = u - 0.5f;
    float fl = rt_floor(um);
    int j0 = (int)fl;
    int j1 = j0 + 1;
    *a = um - fl;
    *i0 = j0 < 0 ? 0 : j0;
    *i1 = j1 > w - 1 ? w - 1 : j1;
}

/* ---- pow (the tone-map kernel, reference tonemap/include/post_processing_filter.cl:24-44) --------
 * OpenCL bounds pow at 16 ULP; this definition is x^y = 2^(y*log2|x|) evaluated in binary64 from
 * + - * fma and three 32-entry tables, rounded once to binary32 (<= 0.51 ULP), with C99's special
 * cases.  These operations are IEEE-exact on x86-64 and gfx950 alike, so host and device agree bit for bit.
 *   log2 x = e + logc[i] + log2(1 + r),  r = m*invc[i] - 1,  i = top five mantissa bits, |r| <= 1/64 (degree 6)
 *   2^t    = 2^q * exp2tab[j] * 2^f,     t = q + j/32 + f,   |f| <= 1/64 (degree 4)
 * Tables: csrc/rt_pow_tables.inc (tools/gen_pow_tables.py; hexadecimal literals). */
#include "rt_pow_tables.inc"
#if defined(__HIPCC__)
__device__ __constant__ static const double rt_pow_invc_dev[32] = {RT_POW_INVC};
__device__ __constant__ static const double rt_pow_logc_dev[32] = {RT_POW_LOGC};
__device__ __constant__ static const double rt_pow_exp2_dev[32] = {RT_POW_EXP2};
#endif
static const double rt_pow_invc_host[32] = {RT_POW_INVC};
static const double rt_pow_logc_host[32] = {RT_POW_LOGC};
static const double rt_pow_exp2_host[32] = {RT_POW_EXP2};
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_POW_TAB(name, i) rt_pow_##name##_dev[i]
#else
#define RT_POW_TAB(name, i) rt_pow_##name##_host[i]
#endif
/* A binary64 constant of a polynomial: on the device it is pinned to a scalar register pair, so that each Horner
 * step is one v_fma_f64 with a scalar addend instead of a 64-bit register copy plus v_fmac_f64 (the constants
 * cannot be literals of a VOP3 instruction).  Same value either way. */
#if defined(__HIP_DEVICE_COMPILE__)
RT_FN double rt_kd(double c) { asm("" : "+s"(c)); return c; }
#else
RT_FN double rt_kd(double c) { return c; }
#endif
RT_FN unsigned long long rt_d2u(double d) { unsigned long long u; __builtin_memcpy(&u, &d, 8); return u; }
RT_FN double rt_u2d(unsigned long long u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
/* log2 of a positive, finite, normal double, to about 2^-44 absolute */
RT_FN double rt_log2_d(double x) {
    const unsigned long long u = rt_d2u(x);
    const int e = (int)((u >> 52) & 0x7ffu) - 1023;
    const int i = (int)((u >> 47) & 31u);
    const double m = rt_u2d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1, 2) */
    const double r = __builtin_fma(m, RT_POW_TAB(invc, i), -1.0);
    /* log2(1 + r) = r (1/ln2) (1 - r/2 + r^2/3 - r^3/4 + r^4/5 - r^5/6) */
    double p = rt_kd(-1.4426950408889634 / 6.0);
    p = __builtin_fma(p, r, rt_kd(1.4426950408889634 / 5.0));
    p = __builtin_fma(p, r, rt_kd(-1.4426950408889634 / 4.0));
    p = __builtin_fma(p, r, rt_kd(1.4426950408889634 / 3.0));
    p = __builtin_fma(p, r, rt_kd(-1.4426950408889634 / 2.0));
    p = __builtin_fma(p, r, rt_kd(1.4426950408889634));
    return __builtin_fma(p, r, (double)e + RT_POW_TAB(logc, i));
}
/* 2^t for |t| <= 200, to about 2^-39 relative */
RT_FN double rt_exp2_d(double t) {
    const double k = __builtin_rint(t * 32.0);
    const double f = __builtin_fma(k, -0.03125, t); /* exact; |f| <= 1/64 */
    const int ki = (int)k;
    /* 2^f = 1 + f ln2 (1 + f ln2/2 (1 + f ln2/3 (1 + f ln2/4))) expanded */
    double p = rt_kd(0.009618129107628477);   /* ln2^4 / 24 */
    p = __builtin_fma(p, f, rt_kd(0.05550410866482158));  /* ln2^3 / 6 */
    p = __builtin_fma(p, f, rt_kd(0.2402265069591007));   /* ln2^2 / 2 */
    p = __builtin_fma(p, f, rt_kd(0.6931471805599453));   /* ln2 */
    p = __builtin_fma(p, f, 1.0);
    const double scale = rt_u2d((unsigned long long)((ki >> 5) + 1023) << 52);
    return (p * RT_POW_TAB(exp2, ki & 31)) * scale;
}
/* Written without branches (every rule is a select, the later ones override the earlier ones): the three
 * channels of a pixel then run as independent instruction streams whose table reads overlap. */
RT_FN float rt_pow(float x, float y) {
    const float ax = rt_fabs(x), ay = rt_fabs(y);
    const int neg = (int)(rt_f2u(x) >> 31);
    const int y_int = rt_trunc(y) == y;
    const int y_odd = y_int && ay < 16777216.0f && (((int)(ay < 16777216.0f ? ay : 0.0f)) & 1);
    const int ax_plain = ax > 0.0f && ax < rt_inf(); /* not 0, inf or NaN */
    double t = (double)y * rt_log2_d((double)(ax_plain ? ax : 1.0f));
    t = t > 200.0 ? 200.0 : (t < -200.0 ? -200.0 : t);
    t = t == t ? t : 0.0; /* y NaN */
    float r = (float)rt_exp2_d(t);
    r = (neg && y_odd) ? -r : r;
    /* a negative base with a non-integer exponent */
    r = (neg && !y_int) ? rt_nan() : r;
    /* x = +-0 or +-inf */
    {
        const float z = ((ax == 0.0f) == (y < 0.0f)) ? rt_inf() : 0.0f;
        r = ax_plain ? r : ((neg && y_odd) ? -z : z);
    }
    /* y = +-inf */
    {
        const float z = ax == 1.0f ? 1.0f : (((ax < 1.0f) == (y < 0.0f)) ? rt_inf() : 0.0f);
        r = ay == rt_inf() ? z : r;
    }
    r = (x != x || y != y) ? rt_nan() : r;
    r = (y == 0.0f || x == 1.0f) ? 1.0f : r;
    return r;
}

/* PCG hash (reference randomness.h:6-11) — 32-bit wraparound arithmetic. */
RT_FN unsigned rt_pcg_next(unsigned* state) {
    unsigned s = *state * 47796405u + 2891336453u;
    s = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    s = (s >> 22u) ^ s;
    *state = s;
    return s;
}
/* reference randomness.h:15-17: top 24 bits / 2^24 (exact in float) */
RT_FN float rt_pcg_float(unsigned* state) { return (float)(rt_pcg_next(state) >> 8) / 16777216.0f; }

#endif /* CHUNKY_RT_MATH_H */
