// kernels.hip — gfx950 kernels of the Chunky path tracer and their launchers.
//
// render_lanes: one lane owns one pixel for all the passes of a launch (the running mean of
// K/rayTracer.cl:109-112 stays in registers: 1 framebuffer read + 1 write per launch instead of per
// pass, same float recurrence in the same order), walking the path of K/rayTracer.cl:93-107.
// preview_lanes: K/rayTracer.cl:115-217.  trace_records: per-trace hit records for parity tests.
//
// Compiled with -ffp-contract=off (see rt_device.hpp).
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "rt_device.hpp"

namespace chunky {

// Per-lane BVH to-visit stack in LDS: entry e of lane t at lds[e * blockDim.x + t] (conflict-free).
struct LdsStack {
    int* base;
    int stride;
    DEV void push(int slot, int v) { base[slot * stride] = v; }
    DEV int pop(int slot) { return base[slot * stride]; }
};

// Leaf lookup of K/octree.h:81-89: cell (bx,by,bz) -> block pointer `data` and leaf `level`.
// WIDE = false walks the reference layout from the root, one bit per level; WIDE = true walks the
// wide re-layout (widetree.hpp), bits[i] bits per level — same (data, level) for every cell.
template <bool WIDE>
DEV void leaf_lookup(const SceneView& S, int bx, int by, int bz, int& data, int& level) {
    if (!WIDE) {
        const int* __restrict__ tree = S.octree;
        level = S.octree_depth;
        data = tree[0];
        while (data > 0) {
            level--;
            data = tree[data + ((((bx >> level) & 1) << 2) | (((by >> level) & 1) << 1) | ((bz >> level) & 1))];
        }
        data = -data;
    } else {
        const uint32_t* __restrict__ tree = S.wide;
        int e = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (i < S.wide_nlev && e >= 0) {
                const int sh = S.wide_shift[i], b = S.wide_bits[i];
                const unsigned ix = __builtin_amdgcn_ubfe((unsigned)bx, sh, b), iy = __builtin_amdgcn_ubfe((unsigned)by, sh, b),
                               iz = __builtin_amdgcn_ubfe((unsigned)bz, sh, b);
                e = (int)tree[e + (int)((((ix << b) | iy) << b) | iz)];
            }
        }
        level = (e >> 27) & 15;
        unsigned code = (unsigned)e & 0x7FFFFFFu;
        data = code == 0x7FFFFFFu ? kAnyType : (int)code;
    }
}

// Octree_octreeIntersect — K/octree.h:41-109.  Leaf-exit march.
template <bool WIDE>
DEV bool octree_hit(const SceneView& S, f3 o, f3 d, int draw_depth, Hit& h) {
    const int depth = S.octree_depth;
    float dist_march = 0;
    f3 inv = rcp3(d);
    f3 off = d * kOffset;
    int lx = (int)rt_floor(o.x) >> depth, ly = (int)rt_floor(o.y) >> depth, lz = (int)rt_floor(o.z) >> depth;
    if ((lx != 0) | (ly != 0) | (lz != 0)) {
        float size = (float)(1 << depth);
        float dist = box_quick(0, size, 0, size, 0, size, o, inv);
        if (dist != dist || dist < 0) return false;
        dist_march += dist + kOffset;
    }
    for (int i = 0; i < draw_depth; i++) {
        if (dist_march > h.distance) return false;
        f3 pos = o + d * dist_march;
        f3 po = pos + off;
        int bx = (int)rt_floor(po.x), by = (int)rt_floor(po.y), bz = (int)rt_floor(po.z);
        if (((bx | by | bz) >> depth) != 0) return false;  // any coordinate outside [0, 2^depth)
        int level, data;
        leaf_lookup<WIDE>(S, bx, by, bz, data, level);
        if (data != 0) {  // ray->material is always 0 (K/wavefront.h:34, K/octree.h:92)
            float dist = block_hit(S, data, bx, by, bz, pos, d, inv, h);
            if (dist == dist) {
                h.distance = dist_march + dist;
                h.material = data;
                return true;
            }
        }
        lx = bx >> level;
        ly = by >> level;
        lz = bz >> level;
        dist_march += box_exit((float)(lx << level), (float)((lx + 1) << level), (float)(ly << level),
                               (float)((ly + 1) << level), (float)(lz << level), (float)((lz + 1) << level), po,
                               inv) + kOffset;
    }
    return false;
}

// closestIntersect — K/kernel.h:14-24
template <bool WIDE>
DEV bool closest_hit(const SceneView& S, f3 o, f3 d, int draw_depth, Hit& h, f3& point, LdsStack& stack) {
    bool hit = octree_hit<WIDE>(S, o, d, draw_depth, h);
    if (!S.world_bvh_empty) hit |= bvh_hit(S, S.world_bvh, o, d, h, stack);
    if (!S.actor_bvh_empty) hit |= bvh_hit(S, S.actor_bvh, o, d, h, stack);
    if (hit) point = o + d * (h.distance - kOffset);
    return hit;
}

DEV void put_record(HitRecord* out, int& n, bool hit, const Hit& h, f3 point) {
    HitRecord r;
    r.hit = hit;
    r.material = h.material;
    r.distance = h.distance;
    r.normal[0] = h.normal.x; r.normal[1] = h.normal.y; r.normal[2] = h.normal.z;
    r.color[0] = h.color.x; r.color[1] = h.color.y; r.color[2] = h.color.z; r.color[3] = h.color.w;
    r.emittance = h.emittance;
    r.point[0] = point.x; r.point[1] = point.y; r.point[2] = point.z;
    out[n++] = r;
}

// One sample — K/rayTracer.cl:55-107
template <bool RECORD, bool WIDE>
DEV f3 sample_path(const SceneView& S, const CameraView& C, const RenderOpts& O, int seed, int gid, LdsStack& stack,
                   HitRecord* rec_out, int* rec_n) {
    unsigned rng = (unsigned)seed + (unsigned)gid;
    rt_pcg_next(&rng);
    f3 o, d;
    primary_ray(C, gid, rng, false, o, d);
    f3 radiance = mk3(0, 0, 0), throughput = mk3(1, 1, 1);
    Hit h;
    h.distance = rt_inf();
    h.material = 0;
    h.normal = mk3(0, 0, 0);
    h.color = f4{0, 0, 0, 0};
    h.emittance = 0;
    f3 point = mk3(0, 0, 0);
    int depth = 0, nrec = 0;
    for (;;) {
        bool hit = closest_hit<WIDE>(S, o, d, O.draw_depth, h, point, stack);
        if (RECORD) put_record(rec_out, nrec, hit, h, point);
        if (!hit) {
            radiance = radiance + sky_radiance(S, d, throughput, 1.0f);  // record.emittance = 1
            break;
        }
        // applyRayColor — K/kernel.h:33-44
        o = point;
        f3 c = mk3(h.color.x, h.color.y, h.color.z);
        throughput = throughput * c;
        radiance = radiance + (c * (h.emittance * O.emitter_scale)) * throughput;
        if (S.sun_flags & 1) {
            // Sun_sampleDirection + shadow trace — K/rayTracer.cl:101-106: the shadow record is a
            // copy of the main record (distance included), traced along the shared ray from
            // `point` with no offset.
            d = sun_sample(S, rng);
            h.emittance = rt_fabs(dot(d, h.normal));  // written to the main record, then copied (K/sky.h:90)
            Hit sh = h;
            f3 sp = h.normal;  // the copy's dead point = normal (K/wavefront.h:73)
            bool shadowed = closest_hit<WIDE>(S, o, d, O.draw_depth, sh, sp, stack);
            if (RECORD) put_record(rec_out, nrec, shadowed, sh, sp);
            if (!shadowed) radiance = radiance + sky_radiance(S, d, throughput, sh.emittance);
        }
        // nextPath — K/kernel.h:46-98
        o = point;
        d = diffuse_bounce(h.normal, rng);
        o = o + d * kOffset;
        depth += 1;
        h.distance = rt_inf();
        if (!(depth < O.max_depth)) break;
    }
    if (RECORD) *rec_n = nrec;
    return radiance;
}

DEV int shard_gid(const ShardView& T, int local) {
    // local pixel slot -> global pixel index: tiles of T.tile consecutive gids dealt round-robin
    if (T.world == 1) return local;
    int t = local / T.tile, w = local - t * T.tile;
    return (t * T.world + T.rank) * T.tile + w;
}

template <bool WIDE>
__global__ void __launch_bounds__(256) render_lanes(SceneView S, CameraView C, RenderOpts O, ShardView T, PassSeeds P,
                                                     float* __restrict__ res) {
    extern __shared__ int lds[];
    LdsStack stack{lds + threadIdx.x, (int)blockDim.x};
    int local = blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= T.n_local) return;
    int gid = shard_gid(T, local);
    if (gid >= C.width * C.height) return;
    float* px = res + 3 * (size_t)gid;
    f3 mean = mk3(px[0], px[1], px[2]);
    for (int k = 0; k < P.n; k++) {
        f3 c = sample_path<false, WIDE>(S, C, O, P.seed[k], gid, stack, nullptr, nullptr);
        int spp = P.first_spp + k;
        float fs = (float)spp, fs1 = (float)(spp + 1);
        mean = f3{(mean.x * fs + c.x) / fs1, (mean.y * fs + c.y) / fs1, (mean.z * fs + c.z) / fs1};
    }
    px[0] = mean.x;
    px[1] = mean.y;
    px[2] = mean.z;
}

template <bool WIDE>
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
    f3 c = sample_path<true, WIDE>(S, C, O, seed, gids[i], stack, local, &cnt);
    for (int k = 0; k < cnt; k++) out[(size_t)i * kMaxTraces + k] = local[k];
    counts[i] = cnt;
    radiance[3 * i] = c.x;
    radiance[3 * i + 1] = c.y;
    radiance[3 * i + 2] = c.z;
}

// preview — K/rayTracer.cl:115-217
template <bool WIDE>
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
    f3 o, d;
    primary_ray(C, gid, rng, true, o, d);
    Hit h;
    h.distance = rt_inf();
    h.material = 0;
    h.normal = mk3(0, 0, 0);
    h.color = f4{0, 0, 0, 0};
    h.emittance = 0;
    f3 point;
    f4 c;
    if (closest_hit<WIDE>(S, o, d, O.draw_depth, h, point, stack)) {
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
        default: break;
    }
    out[i] = r;
}

// ------------------------------------------------------------------------------------ launchers
// variant bit 0 set = force the reference-layout octree walk (K/octree.h:81-89 as written)
static bool use_wide(int variant, const SceneView& S) { return S.wide != nullptr && !(variant & 1); }

static size_t stack_lds_bytes(const SceneView& S, int block) {
    bool need = !S.world_bvh_empty || !S.actor_bvh_empty;
    return need ? (size_t)kBvhStackEntries * block * sizeof(int) : 0;
}

hipError_t launch_render(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, const ShardView& T,
                         const PassSeeds& P, float* res, hipStream_t stream) {
    const int block = 256;
    int grid = (T.n_local + block - 1) / block;
    if (grid <= 0 || P.n <= 0) return hipSuccess;
    if (use_wide(variant, S))
        hipLaunchKernelGGL(render_lanes<true>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O, T, P, res);
    else
        hipLaunchKernelGGL(render_lanes<false>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O, T, P, res);
    return hipGetLastError();
}

hipError_t launch_trace_records(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, int seed,
                                const int* gids_dev, int n, HitRecord* out, int* counts, float* radiance,
                                hipStream_t stream) {
    const int block = 256;
    int grid = (n + block - 1) / block;
    if (grid <= 0) return hipSuccess;
    if (use_wide(variant, S))
        hipLaunchKernelGGL(trace_records_kernel<true>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O,
                           seed, gids_dev, n, out, counts, radiance);
    else
        hipLaunchKernelGGL(trace_records_kernel<false>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O,
                           seed, gids_dev, n, out, counts, radiance);
    return hipGetLastError();
}

hipError_t launch_preview(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, int* argb,
                          hipStream_t stream) {
    const int block = 256;
    int grid = (C.width * C.height + block - 1) / block;
    if (use_wide(variant, S))
        hipLaunchKernelGGL(preview_lanes<true>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O, argb);
    else
        hipLaunchKernelGGL(preview_lanes<false>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O, argb);
    return hipGetLastError();
}

hipError_t launch_math_selftest(int which, int n, const float* a, const float* b, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(math_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, which, n, a, b, out);
    return hipGetLastError();
}

}  // namespace chunky
