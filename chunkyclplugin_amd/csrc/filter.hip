// filter.hip — the tone-map kernel of tonemap/include/post_processing_filter.cl and its self test.
//
// Compiled with -ffp-contract=off (see rt_device.hpp).
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "rt_device.hpp"

namespace chunky {

// ---------------------------------------------------------------------------------------------
// `filter` — tonemap/include/post_processing_filter.cl:5-51: 3 doubles per pixel in, one ARGB word
// out; 28 bytes of HBM traffic per pixel and nothing to reuse, so the kernel is a streaming copy
// with arithmetic in the shadow of the loads.  A workgroup takes 512 pixels = 1536 doubles at a
// time: every lane reads three consecutive 16-byte pairs of the tile (fully coalesced), the pairs are
// narrowed to float (double.h:19-21, fp64 present) into LDS, and each lane then picks up the three
// channels of its two pixels (stride-3 LDS reads, conflict-free).
constexpr int kFilterTile = 512;

DEV unsigned filter_to_uint(float f) {  // (uint) of color_to_argb (rgba.h:9-14); saturating outside the uint range
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (unsigned)f;
}

DEV float filter_tonemap1(float c) {
    c = rt_fmax(0.0f, c - 0.004f);
    return (c * (6.2f * c + 0.5f)) / (c * (6.2f * c + 1.7f) + 0.06f);
}
DEV float filter_aces_num(float c) { return c * (2.51f * c + 0.03f); }
DEV float filter_aces_den(float c) { return c * (2.43f * c + 0.59f) + 0.14f; }
DEV float filter_aces(float c) { return rt_clamp(filter_aces_num(c) / filter_aces_den(c), 0.0f, 1.0f); }
DEV float filter_hable(float c) {
    c *= 16.0f;
    c = ((c * (0.15f * c + 0.10f * 0.50f) + 0.20f * 0.02f) / (c * (0.15f * c + 0.50f) + 0.20f * 0.30f)) - 0.02f / 0.30f;
    const float white = ((11.2f * (0.15f * 11.2f + 0.10f * 0.50f) + 0.20f * 0.02f) / (11.2f * (0.15f * 11.2f + 0.50f) + 0.20f * 0.30f)) - 0.02f / 0.30f;
    return c / white;
}
DEV unsigned filter_pack(float r, float g, float b) {
    unsigned ur = filter_to_uint(r * 255.0f + 0.5f), ug = filter_to_uint(g * 255.0f + 0.5f), ub = filter_to_uint(b * 255.0f + 0.5f);
    ur = ur > 255u ? 255u : ur;
    ug = ug > 255u ? 255u : ug;
    ub = ub > 255u ? 255u : ub;
    return 0xFF000000u | (ur << 16) | (ug << 8) | ub;  // alpha: (uint)(1 * 255 + 0.5) = 255
}

// GAMMA and ACES end in pow(c, 1/2.2) -> c * 255 + 0.5 -> (uint) -> clamp to 255 (post_processing_filter.cl:24-27,33-38,
// rgba.h:9-14): a monotone step function of the float c with at most 255 steps.  Its thresholds (kT[k] = the smallest float
// whose byte is >= k, found on the host by bisection with the same rt_pow: capi.hip gamma_thresholds; monotonicity is checked
// exhaustively by tests/test_filter.py) replace the six binary64 rt_pow per lane that made these two curves issue-bound.
// The byte is estimated with the hardware log2 / exp2 (v = 2^(log2(c) / 2.2) * 255 + 0.5, measured: off by at most 2.3e-5, and
// by less below), and only an estimate that falls within kGammaGuard of a step can be wrong, by one: those — one value in
// four thousand — are settled against the neighbouring thresholds, and a wave none of whose lanes is that close skips the
// table altogether (gamma_bytes3).  Same byte for every float: chunky_selftest_gamma_scan compares the two over all 2^32 bit
// patterns on the device.  NaN and negative values fail every comparison (byte 0, as rt_pow's NaN does), except -inf, whose
// power is +inf (C99 pow(-inf, y > 0)).
constexpr float kGammaGuard = 1.0f / 8192.0f;  // eight times the worst stray of an estimate over all 2^32 floats (1.5e-5: one ulp of 255)
DEV float gamma_estimate(float c) { return __builtin_amdgcn_exp2f((float)(1.0 / 2.2) * __builtin_amdgcn_logf(c)) * 255.0f + 0.5f; }
DEV int gamma_estimate_byte(float est) { return est >= 255.0f ? 255 : (est > 0.0f ? (int)est : 0); }  // NaN -> 0
DEV bool gamma_near_step(float est) {
    const float f = est - __builtin_floorf(est);
    return est > 0.5f && est < 255.5f && (f < kGammaGuard || f > 1.0f - kGammaGuard);
}
DEV int gamma_settle(float c, int k, const float* __restrict__ kT) {  // k is off by at most one
    const bool up = k < 255 && c >= kT[k < 255 ? k + 1 : 255];
    const bool down = k > 0 && !(c >= kT[k]);
    return k + (int)up - (int)down;
}
// the bytes of three channel values, packed as the ARGB word's low 24 bits
DEV unsigned gamma_bytes3(float r, float g, float b, const float* __restrict__ kT) {
    const float er = gamma_estimate(r), eg = gamma_estimate(g), eb = gamma_estimate(b);
    int kr = gamma_estimate_byte(er), kg = gamma_estimate_byte(eg), kb = gamma_estimate_byte(eb);
    if (__ballot(gamma_near_step(er) || gamma_near_step(eg) || gamma_near_step(eb))) {  // wave-uniform
        kr = gamma_settle(r, kr, kT);
        kg = gamma_settle(g, kg, kT);
        kb = gamma_settle(b, kb, kT);
    }
    kr = r == -rt_inf() ? 255 : kr;
    kg = g == -rt_inf() ? 255 : kg;
    kb = b == -rt_inf() ? 255 : kb;
    return ((unsigned)kr << 16) | ((unsigned)kg << 8) | (unsigned)kb;
}
// ACES (post_processing_filter.cl:33-38): the curve's quotient, clamped to [0, 1], then the same power and byte.  The estimate
// takes the quotient from the hardware reciprocal (1 ulp: 1e-5 of a byte); the correctly rounded division the reference
// performs is evaluated only when a lane of the wave is within the guard band of a step (or its denominator is out of the
// reciprocal's range), and it is that exact value which is estimated again and settled against the thresholds.  NaN clamps
// to 0 either way (fmin / fmax drop it); the denominator is never below 0.1.
DEV unsigned aces_bytes3(float r, float g, float b, const float* __restrict__ kT) {
    const float nr = filter_aces_num(r), ng = filter_aces_num(g), nb = filter_aces_num(b);
    const float dr = filter_aces_den(r), dg = filter_aces_den(g), db = filter_aces_den(b);
    const float er = gamma_estimate(rt_clamp(nr * __builtin_amdgcn_rcpf(dr), 0.0f, 1.0f)),
                eg = gamma_estimate(rt_clamp(ng * __builtin_amdgcn_rcpf(dg), 0.0f, 1.0f)),
                eb = gamma_estimate(rt_clamp(nb * __builtin_amdgcn_rcpf(db), 0.0f, 1.0f));
    int kr = gamma_estimate_byte(er), kg = gamma_estimate_byte(eg), kb = gamma_estimate_byte(eb);
    // (a denominator beyond 2^100 — its reciprocal would be flushed to zero — or NaN takes the exact path as well)
    const bool big = !(dr < 0x1p100f) || !(dg < 0x1p100f) || !(db < 0x1p100f);
    if (__ballot(big || gamma_near_step(er) || gamma_near_step(eg) || gamma_near_step(eb))) {  // wave-uniform
        kr = gamma_estimate_byte(gamma_estimate(rt_clamp(nr / dr, 0.0f, 1.0f)));  // the estimate of the exact quotient: off by one at most
        kg = gamma_estimate_byte(gamma_estimate(rt_clamp(ng / dg, 0.0f, 1.0f)));
        kb = gamma_estimate_byte(gamma_estimate(rt_clamp(nb / db, 0.0f, 1.0f)));
        kr = gamma_settle(rt_clamp(nr / dr, 0.0f, 1.0f), kr, kT);
        kg = gamma_settle(rt_clamp(ng / dg, 0.0f, 1.0f), kg, kT);
        kb = gamma_settle(rt_clamp(nb / db, 0.0f, 1.0f), kb, kT);
    }
    return ((unsigned)kr << 16) | ((unsigned)kg << 8) | (unsigned)kb;  // the clamped value is never -inf
}
// Every float of [first, first + count) (as bit patterns) through gamma_bytes3 (curve 0) or aces_bytes3 (curve 2) and through
// the reference's arithmetic followed by a plain search of the threshold table: counts the values whose bytes differ, and
// returns the largest distance of an estimate from the table's byte interval.
__global__ void __launch_bounds__(256) gamma_scan_kernel(unsigned first, unsigned long long count, int curve, const float* __restrict__ thresholds,
                                                         unsigned long long* __restrict__ mismatches, float* __restrict__ worst) {
    __shared__ float kT[256];
    kT[threadIdx.x] = thresholds[threadIdx.x];
    __syncthreads();
    unsigned long long bad = 0;
    float far = 0.0f;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256) {
        const float x = __uint_as_float(first + (unsigned)i);
        const float c = curve == 2 ? filter_aces(x) : x;  // what the reference hands to pow
        int want = 0;  // the largest k with c >= kT[k] (kT[0] = 0; NaN and negative values: 0)
        for (int step = 128; step > 0; step >>= 1)
            if (want + step < 256 && c >= kT[want + step]) want += step;
        if (c == -rt_inf()) want = 255;
        const unsigned got = curve == 2 ? aces_bytes3(x, x, x, kT) : gamma_bytes3(x, x, x, kT);
        if (got != (((unsigned)want << 16) | ((unsigned)want << 8) | (unsigned)want)) bad += 1;
        const float est = curve == 2 ? gamma_estimate(rt_clamp(filter_aces_num(x) * __builtin_amdgcn_rcpf(filter_aces_den(x)), 0.0f, 1.0f))
                                     : gamma_estimate(x);
        if (c > 0.0f && want > 0 && want < 255 && !(curve == 2 && !(filter_aces_den(x) < 0x1p100f))) {  // how far the estimate strays outside [want, want + 1)
            const float d = est < (float)want ? (float)want - est : (est >= (float)(want + 1) ? est - (float)(want + 1) : 0.0f);
            far = d > far ? d : far;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
    if (far > 0.0f) atomicMax((unsigned*)worst, __float_as_uint(far));  // non-negative floats order like their bit patterns
}

// The curve of K's switch (post_processing_filter.cl:23-45) applied to the N channel values of a lane at once: the
// switch is taken once, and inside a case the N evaluations are independent instruction streams in one basic
// block, so the long dependent chain of each rt_pow overlaps with the others'.
template <int N>
DEV void filter_curve(float (&c)[N], float exposure, int type) {
#pragma unroll
    for (int i = 0; i < N; i++) c[i] *= exposure;
    const float gamma = (float)(1.0 / 2.2);
    switch (type) {
        case 0:  // GAMMA
#pragma unroll
            for (int i = 0; i < N; i++) c[i] = rt_pow(c[i], gamma);
            break;
        case 1:  // TONEMAP1
#pragma unroll
            for (int i = 0; i < N; i++) c[i] = filter_tonemap1(c[i]);
            break;
        case 2:  // ACES
#pragma unroll
            for (int i = 0; i < N; i++) c[i] = rt_pow(filter_aces(c[i]), gamma);
            break;
        case 3:  // HABLE
#pragma unroll
            for (int i = 0; i < N; i++) c[i] = filter_hable(c[i]);
            break;
        default: break;  // the reference's switch has no default: exposure only
    }
}

__global__ __launch_bounds__(256) void filter_kernel(long long n, float exposure, const double* __restrict__ in,
                                                     unsigned* __restrict__ out, int type, int vec_ok, const float* __restrict__ thresholds) {
    __shared__ float stage[3 * kFilterTile];
    __shared__ float kT[256];
    const int t = threadIdx.x;
    const bool bytes = thresholds != nullptr && (type == 0 || type == 2);  // GAMMA, ACES through the threshold table
    if (bytes) kT[t] = thresholds[t];
    const long long total = 3 * n;
    for (long long base = (long long)blockIdx.x * kFilterTile; base < n; base += (long long)gridDim.x * kFilterTile) {
        const long long first = 3 * base;
        if (vec_ok && base + kFilterTile <= n) {
            const double2* __restrict__ src = reinterpret_cast<const double2*>(in + first);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double2 v = src[t + 256 * k];
                stage[2 * (t + 256 * k)] = (float)v.x;
                stage[2 * (t + 256 * k) + 1] = (float)v.y;
            }
        } else {
            for (int k = t; k < 3 * kFilterTile; k += 256) stage[k] = first + k < total ? (float)in[first + k] : 0.0f;
        }
        __syncthreads();
        float c[6];  // the three channels of this lane's two pixels (a pixel beyond n computes on zeros, stores nothing)
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int px = t + 256 * k;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) c[3 * k + ch] = stage[3 * px + ch];
        }
        if (bytes) {
#pragma unroll
            for (int i = 0; i < 6; i++) c[i] *= exposure;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int px = t + 256 * k;
                const unsigned word = 0xFF000000u | (type == 2 ? aces_bytes3(c[3 * k], c[3 * k + 1], c[3 * k + 2], kT)
                                                               : gamma_bytes3(c[3 * k], c[3 * k + 1], c[3 * k + 2], kT));
                if (base + px < n) out[base + px] = word;
            }
        } else {
            filter_curve<6>(c, exposure, type);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int px = t + 256 * k;
                if (base + px < n) out[base + px] = filter_pack(c[3 * k], c[3 * k + 1], c[3 * k + 2]);
            }
        }
        __syncthreads();
    }
}
hipError_t launch_gamma_scan(unsigned first, unsigned long long count, int curve, const float* thresholds, unsigned long long* mismatches,
                             float* worst, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(gamma_scan_kernel, dim3(8192), dim3(256), 0, stream, first, count, curve, thresholds, mismatches, worst);
    return hipGetLastError();
}

hipError_t launch_filter(long long n_pixels, float exposure, const double* in, unsigned* out, int type, hipStream_t stream,
                         const float* thresholds) {
    if (n_pixels <= 0) return hipSuccess;
    long long tiles = (n_pixels + kFilterTile - 1) / kFilterTile;
    int blocks = (int)(tiles < 4096 ? tiles : 4096);
    int vec_ok = (reinterpret_cast<uintptr_t>(in) & 15u) == 0;
    hipLaunchKernelGGL(filter_kernel, dim3(blocks), dim3(256), 0, stream, n_pixels, exposure, in, out, type, vec_ok, thresholds);
    return hipGetLastError();
}

}  // namespace chunky
