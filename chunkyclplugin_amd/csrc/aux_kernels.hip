// aux_kernels.hip — kernels beside the hot path: trace_records (per-trace hit records for the parity tests), preview
// (K/rayTracer.cl:115-217) and the self tests (rt_math.h contract; helper-level known answers).
//
// Compiled with -ffp-contract=off (see rt_device.hpp).
#include <hip/hip_runtime.h>

#include "path_state.hpp"
#include "pool_walk.hpp"

namespace chunky {

template <int TREE>
__global__ void __launch_bounds__(256) trace_records_kernel(SceneView S, CameraView C, RenderOpts O, int seed,
                                                             const int* __restrict__ gids, int n,
                                                             HitRecord* __restrict__ out, int* __restrict__ counts,
                                                             float* __restrict__ radiance) {
    extern __shared__ int lds[];
    LdsStack stack{lds + threadIdx.x, (int)blockDim.x};
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    HitRecord local[kMaxTraces];
    int cnt = 0;
    f3 c = sample_path<true, TREE>(S, C, O, seed, gids[i], stack, local, &cnt);
    for (int k = 0; k < cnt; k++) out[(size_t)i * kMaxTraces + k] = local[k];
    counts[i] = cnt;
    radiance[3 * i] = c.x;
    radiance[3 * i + 1] = c.y;
    radiance[3 * i + 2] = c.z;
}

// preview — K/rayTracer.cl:115-217
template <int TREE>
__global__ void __launch_bounds__(256) preview_lanes(SceneView S, CameraView C, RenderOpts O, int* __restrict__ argb) {
    extern __shared__ int lds[];
    LdsStack stack{lds + threadIdx.x, (int)blockDim.x};
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int W = C.width, H = C.height;
    if (gid >= W * H) return;
    int px = gid % W, py = gid / W;
    if ((px == W / 2 && (py >= H / 2 - 5 && py <= H / 2 + 5)) || (py == H / 2 && (px >= W / 2 - 5 && px <= W / 2 + 5))) {
        argb[gid] = (int)0xFFFFFFFFu;
        return;
    }
    unsigned rng = 0;
    rt_pcg_next(&rng);
    const RayOD pr = primary_ray(C, gid, rng, true);
    const f3 o = pr.o, d = pr.d;
    Hit h;
    h.distance = rt_inf();
    h.material = 0;
    h.normal = mk3(0, 0, 0);
    h.color = f4{0, 0, 0, 0};
    h.emittance = 0;
    f3 point;
    f4 c;
    if (closest_hit<TREE>(S, o, d, O.draw_depth, h, point, stack)) {
        float shading = dot(h.normal, mk3(0.25f, 0.866f, 0.433f));
        shading = rt_fmax(0.3f, shading);
        c = f4{h.color.x * shading, h.color.y * shading, h.color.z * shading, 0};
    } else {
        c = sky_color(S, d);
        sun_disc(S, d, c);
    }
    int r = (int)rt_floor(rt_clamp(rt_sqrt(c.x) * 255.0f, 0.0f, 255.0f));
    int g = (int)rt_floor(rt_clamp(rt_sqrt(c.y) * 255.0f, 0.0f, 255.0f));
    int b = (int)rt_floor(rt_clamp(rt_sqrt(c.z) * 255.0f, 0.0f, 255.0f));
    argb[gid] = (int)(0xFF000000u | ((unsigned)r << 16) | ((unsigned)g << 8) | (unsigned)b);
}

// Device evaluation of the rt_math.h contract, compared bit-for-bit with the host by the tests.
__global__ void math_selftest_kernel(int which, int n, const float* __restrict__ a, const float* __restrict__ b,
                                     float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i], r = 0;
    switch (which) {
        case 0: r = rt_sin(x); break;
        case 1: r = rt_cos(x); break;
        case 2: r = rt_asin(x); break;
        case 3: r = rt_acos(x); break;
        case 4: r = rt_atan2(x, y); break;
        case 5: r = rt_fmod1(x); break;
        case 6: r = rt_fmin(x, y); break;
        case 7: r = rt_fmax(x, y); break;
        case 8: r = rt_sqrt(x); break;
        case 9: r = x / y; break;
        case 10: r = rt_rlen3(x, y, x); break;
        case 11: r = rt_dot3(x, y, x, y, x, y); break;
        case 12: r = rt_floor(x); break;
        case 13: r = (float)(int)x; break;
        case 14: r = (float)((double)((unsigned)(int)x & 0xFF) / 255.0); break;
        case 15: r = (float)(-0.5 + (double)(x * y)); break;
        case 16: r = rt_pow(x, y); break;
        case 18: r = (float)floor_to_int(x); break;
        case 17: r = (float)((double)x * (double)y); break;
        default: break;
    }
    out[i] = r;
}

// Helper-level known answers (chunky_selftest_helpers): the device functions the kernels are made of, one call per input
// row, against tests/golden/helpers.npz — the answers of the reference object's own exported helpers (oracle/ref_shim.cpp
// ref_helpers has the row layouts and the `which` numbering, K/primitives.h:30-409, K/sky.h:42-106, K/kernel.h:46-98,
// K/block.h:30-118, K/octree.h:41-109, K/bvh.h:22-113).  18 = the world-BVH walk as render_pool performs it (rwalk_step on
// the aligned records, to-visit stacks in LDS), same rows and answers as 15.
constexpr int kHelperIn = 32, kHelperOut = 12;
struct ScratchStack {
    int a[kBvhStackEntries];
    DEV void push(int slot, int v) { a[slot] = v; }
    DEV int pop(int slot) { return a[slot]; }
};
template <int TREE>
__global__ void __launch_bounds__(64) helpers_selftest_kernel(SceneView S, int which, int n, const float* __restrict__ in_rows,
                                                              float* __restrict__ out_rows) {
    extern __shared__ int lds[];
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n) return;
    const float* in = in_rows + (size_t)r * kHelperIn;
    float out[kHelperOut];
    for (int k = 0; k < kHelperOut; k++) out[k] = 0;
    Hit h;
    h.distance = rt_inf();
    h.material = 0;
    h.normal = mk3(0, 0, 0);
    h.color = f4{0, 0, 0, 0};
    h.emittance = 0;
    h.spec = 0;
    const f3 a6 = mk3(in[6], in[7], in[8]), a9 = mk3(in[9], in[10], in[11]);
    switch (which) {
        case 0: out[0] = box_quick(in[0], in[1], in[2], in[3], in[4], in[5], a6, rcp3(a9)); break;
        case 1: out[0] = box_exit(in[0], in[1], in[2], in[3], in[4], in[5], a6, rcp3(a9)); break;
        case 2: case 3: {
            const f3 inv = which == 2 ? rcp3(mk3(in[12], in[13], in[14])) : rcp3(a9);
            const Slabs sl = which == 2 ? slabs(0, 1, 0, 1, 0, 1, a6, inv) : slabs(in[0], in[1], in[2], in[3], in[4], in[5], a6, inv);
            const float tn = slab_near(sl), tf = slab_far(sl);
            if (tf < tn) {
                out[0] = rt_nan();
            } else {
                const Face f = which == 2 ? face_unit(sl, tn, a6 + a9 * tn) : face_map2(sl, tn, a6 + a9 * tn);
                out[0] = tn; out[1] = f.n.x; out[2] = f.n.y; out[3] = f.n.z; out[4] = f.u; out[5] = f.v;
            }
            break;
        }
        case 4: {
            const f3 pos = mk3(in[4], in[5], in[6]), dir = mk3(in[7], in[8], in[9]);
            out[0] = block_hit(S, __float_as_int(in[0]), (int)in[1], (int)in[2], (int)in[3], pos, dir, rcp3(dir), h);
            out[1] = h.normal.x; out[2] = h.normal.y; out[3] = h.normal.z;
            out[4] = h.color.x; out[5] = h.color.y; out[6] = h.color.z; out[7] = h.color.w; out[8] = h.emittance;
            break;
        }
        case 6: {
            f3 nn = mk3(0, 0, 0);
            float u = 0, v = 0;
            int mat = 0;
            out[0] = triangle_hit((const int*)in, in[26], mk3(in[20], in[21], in[22]), mk3(in[23], in[24], in[25]), nn, u, v, mat);
            out[1] = nn.x; out[2] = nn.y; out[3] = nn.z; out[4] = u; out[5] = v; out[6] = __int_as_float(mat);
            break;
        }
        case 7: {
            unsigned rng = (unsigned)__float_as_int(in[0]);
            const f3 d = sun_sample(S, rng);
            out[0] = d.x; out[1] = d.y; out[2] = d.z;
            out[3] = rt_fabs(dot(d, mk3(in[1], in[2], in[3])));  // record.emittance, K/sky.h:90 (sample_path / shade_phase write it the same way)
            out[4] = __int_as_float((int)rng);
            break;
        }
        case 8: {
            f4 c = f4{in[3], in[4], in[5], in[6]};
            const f4 before = c;
            sun_disc(S, mk3(in[0], in[1], in[2]), c);
            out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;  // (alpha: the kernels never read it; the test compares x, y, z)
            out[4] = (c.x != before.x || c.y != before.y || c.z != before.z) ? 1.0f : 0.0f;
            break;
        }
        case 9: {
            const f4 c = sky_color(S, mk3(in[0], in[1], in[2]));
            out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
            break;
        }
        case 10: {
            unsigned rng = (unsigned)__float_as_int(in[0]);
            const f3 d = diffuse_bounce(mk3(in[1], in[2], in[3]), rng);
            const f3 o = mk3(in[4], in[5], in[6]) + d * kOffset;
            out[0] = d.x; out[1] = d.y; out[2] = d.z; out[3] = o.x; out[4] = o.y; out[5] = o.z; out[6] = __int_as_float((int)rng);
            break;
        }
        case 11: {
            const f4 c = unpack_unorm8(atlas_texel(S, in[0], in[1], __float_as_int(in[2]), __float_as_int(in[3])));
            out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
            break;
        }
        case 12: {
            out[0] = material_sample(S, __float_as_int(in[0]), in[1], in[2], h) ? 1.0f : 0.0f;
            out[1] = h.color.x; out[2] = h.color.y; out[3] = h.color.z; out[4] = h.color.w; out[5] = h.emittance;
            break;
        }
        case 14: case 15: case 18: {
            const f3 o = mk3(in[0], in[1], in[2]), d = mk3(in[3], in[4], in[5]);
            bool hit = false;
            if (which == 14) {
                hit = octree_hit<TREE>(S, o, d, 256, h);
            } else if (which == 15) {
                ScratchStack stack;
                h.distance = in[6];
                hit = bvh_hit(S, S.world_bvh, o, d, h, stack);
            } else {
                LaneState L;
                L.o = o; L.d = d; L.inv = rcp3(d);
                L.h = h;
                L.h.distance = in[6];
                L.shadow = false; L.oct_hit = false; L.trace_hit = false;
                L.bvh_dist = in[6];
                L.pid = (int)threadIdx.x;
                SceneView W = S;
                W.actor_bvh_empty = 1;  // the world BVH alone, like row 15
                PathStacks K{lds, (int)blockDim.x};
                int st = rbvh_enter(W, L, 0);
                while (st == ST_BVH) st = rwalk_step(W, L, K);
                hit = L.trace_hit;
                h = L.h;
            }
            int k = 0;
            out[k++] = hit ? 1.0f : 0.0f;
            out[k++] = h.distance;
            if (which == 14) out[k++] = __int_as_float(h.material);
            out[k++] = h.normal.x; out[k++] = h.normal.y; out[k++] = h.normal.z;
            out[k++] = h.color.x; out[k++] = h.color.y; out[k++] = h.color.z; out[k++] = h.color.w;
            out[k++] = h.emittance;
            break;
        }
        default: break;
    }
    for (int k = 0; k < kHelperOut; k++) out_rows[(size_t)r * kHelperOut + k] = out[k];
}
hipError_t launch_trace_records(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, int seed,
                                const int* gids_dev, int n, HitRecord* out, int* counts, float* radiance,
                                hipStream_t stream) {
    const int block = 256;
    int grid = (n + block - 1) / block;
    if (grid <= 0) return hipSuccess;
    if (use_wide(variant, S))
        hipLaunchKernelGGL(trace_records_kernel<-1>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O,
                           seed, gids_dev, n, out, counts, radiance);
    else
        hipLaunchKernelGGL(trace_records_kernel<0>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O,
                           seed, gids_dev, n, out, counts, radiance);
    return hipGetLastError();
}

hipError_t launch_preview(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, int* argb,
                          hipStream_t stream) {
    const int block = 256;
    int grid = (C.width * C.height + block - 1) / block;
    if (use_wide(variant, S))
        hipLaunchKernelGGL(preview_lanes<-1>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O, argb);
    else
        hipLaunchKernelGGL(preview_lanes<0>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O, argb);
    return hipGetLastError();
}

hipError_t launch_math_selftest(int which, int n, const float* a, const float* b, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(math_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, which, n, a, b, out);
    return hipGetLastError();
}

// `tree`: 0 the reference layout, 1 the wide tree in the form the render kernels would pick for this scene (row 14 only)
hipError_t launch_helpers_selftest(const SceneView& S, int which, int tree, int n, const float* in, float* out, int* tree_used, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (which == 18 && !(S.bvh_rec && S.tri_rec && S.mat8)) return hipErrorNotSupported;
    int t = tree ? tree_form(0, S) : 0;
    typedef void (*Kernel)(SceneView, int, int, const float*, float*);
    Kernel k;
    switch (t) {
        case 0: k = helpers_selftest_kernel<0>; break;
        case 16: k = helpers_selftest_kernel<16>; break;
        case 17: k = helpers_selftest_kernel<17>; break;
        case 18: k = helpers_selftest_kernel<18>; break;
        default: t = -1; k = helpers_selftest_kernel<-1>; break;
    }
    if (tree_used) *tree_used = t;
    const int entries = S.bvh_stack_entries > 0 && S.bvh_stack_entries < kBvhStackEntries ? S.bvh_stack_entries : kBvhStackEntries;
    const size_t lds = which == 18 ? (size_t)entries * 64 * sizeof(int) : 0;
    hipLaunchKernelGGL(k, dim3((unsigned)((n + 63) / 64)), dim3(64), lds, stream, S, which, n, in, out);
    return hipGetLastError();
}


}  // namespace chunky
