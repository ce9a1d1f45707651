// kernels.hip — gfx950 kernels of the Chunky path tracer and their launchers.
//
// render_pool (the default) + fold_kernel: a persistent grid whose waves each run a pool of 64 + K path state machines phase
// by phase (MARCH / BLOCK / SHADE, plus the entity-BVH walk), voted over the pool, samples claimed per XCD, the running mean
// folded afterwards in pass order; DESIGN.md section 5 describes it.  render_waves: round 1's kernel (paths bound to lanes,
// pixels shared by groups of lanes; CHUNKY_OPT_KERNEL bit 3, and the fallback where render_pool does not apply).
// filter_kernel: the tone-map kernel of tonemap/include/post_processing_filter.cl.
// render_lanes: one lane owns one pixel for all the passes of a launch (the running mean of
// K/rayTracer.cl:109-112 stays in registers: 1 framebuffer read + 1 write per launch instead of per
// pass, same float recurrence in the same order), walking the path of K/rayTracer.cl:93-107.
// preview_lanes: K/rayTracer.cl:115-217.  trace_records: per-trace hit records for parity tests.
//
// Compiled with -ffp-contract=off (see rt_device.hpp).
#include <hip/hip_runtime.h>

#include <cstdlib>

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

#ifndef CHUNKY_LDS_TOP
#define CHUNKY_LDS_TOP 0
#endif
#if CHUNKY_LDS_TOP
// experiment: render_pool<-1, 32> copies the top node (at most 16^3 entries) behind its pool area in dynamic LDS
constexpr int chunky_lds_top_offset = 4 * (32 * 128 + 32 * 8) / 4;
#endif
// Leaf lookup of K/octree.h:81-89: cell (bx,by,bz) -> block pointer `data` and leaf `level`.
// TREE = 0 walks the reference layout from the root, one bit per level; TREE = -1 walks the wide
// re-layout (widetree.hpp) with per-level bit counts from the scene view; TREE = 16 + n walks the
// default split of widetree.cpp — one dense top node of S.wide_bits[0] bits per axis (a wave-uniform
// value) over n levels of 3 bits (compile-time shifts): two dependent reads per cell for a 512^3 world
// where the reference descends nine — same (data, level) for every cell.
// `kind`: 0 full cube, 1 other model, 2 or 3 cannot be hit (air, invisible, ANY_TYPE); the reference
// layout carries no kinds, so every non-air leaf reports 1 there (the general test handles all types).
struct TopCache {
    int* idx;
    int* e;
};
template <int TREE>
DEV void leaf_lookup(const SceneView& S, int bx, int by, int bz, int& data, int& level, int& kind, bool inside = true,
                     TopCache cache = TopCache{nullptr, nullptr}) {
    // `inside` false: the cell is not in the world; the lookup then reads cell (0, 0, 0) (callers discard it)
    if (TREE < 16 && !inside) bx = by = bz = 0;
    if (TREE == 0) {
        const int* __restrict__ tree = S.octree;
        level = S.octree_depth;
        data = tree[0];
        while (data > 0) {
            level--;
            data = tree[data + ((((bx >> level) & 1) << 2) | (((by >> level) & 1) << 1) | ((bz >> level) & 1))];
        }
        data = -data;
        kind = (data == 0 || data == kAnyType) ? 2 : 1;
    } else {
        const uint32_t* __restrict__ tree = S.wide;
        int e = 0;
        if (TREE >= 16) {
            constexpr int N3 = TREE - 16;
            {
                // the cell is inside the world, so the top index needs no masks
                const int tb = S.wide_bits[0];
                unsigned idx = (((((unsigned)bx >> (3 * N3)) << tb) | ((unsigned)by >> (3 * N3))) << tb) | ((unsigned)bz >> (3 * N3));
                idx = inside ? idx : 0u;  // the levels below mask their index bits: any bx, by, bz stay inside the node
                // byte offset in 32 bits (the builder keeps the array under 2^30 entries): SGPR base + VGPR offset
                if (cache.idx) {
                    if ((int)idx != *cache.idx) {
                        *cache.e = *(const int*)((const char*)tree + (idx << 2));
                        *cache.idx = (int)idx;
                    }
                    e = *cache.e;
                } else
                    e = *(const int*)((const char*)tree + (idx << 2));
            }
#pragma unroll
            for (int i = 0; i < N3; i++) {
                if (e >= 0) {
                    const int sh = 3 * (N3 - 1 - i);
                    const unsigned idx = (((unsigned)bx >> sh) & 7u) << 6 | (((unsigned)by >> sh) & 7u) << 3 | (((unsigned)bz >> sh) & 7u);
                    e = *(const int*)((const char*)tree + (((unsigned)e + idx) << 2));
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (i < S.wide_nlev && e >= 0) {
                const int sh = S.wide_shift[i], b = S.wide_bits[i];
                const unsigned ix = __builtin_amdgcn_ubfe((unsigned)bx, sh, b), iy = __builtin_amdgcn_ubfe((unsigned)by, sh, b),
                               iz = __builtin_amdgcn_ubfe((unsigned)bz, sh, b);
#if CHUNKY_LDS_TOP  // experiment (profiles/r02_lds_top_experiment.json): the top node staged in LDS by render_pool
                if (i == 0 && TREE == -1) {
                    extern __shared__ int lds_all[];
                    e = lds_all[chunky_lds_top_offset + ((((ix << b) | iy) << b) | iz)];
                    continue;
                }
#endif
                e = (int)tree[(unsigned)e + ((((ix << b) | iy) << b) | iz)];  // unsigned: 32-bit offset off an SGPR base
            }
        }
        // three field extractions: the builder's annotation pass (widetree.cpp) has set kind 2 on air and on
        // pointers outside the block palette, and ANY_TYPE carries kind 3; `data` means something for kinds 0, 1 only
        level = (e >> 27) & 15;
        kind = (int)(((unsigned)e >> 25) & 3u);
        data = (int)((unsigned)e & 0x1FFFFFFu);
    }
}

// Octree_octreeIntersect — K/octree.h:41-109.  Leaf-exit march.
template <int TREE>
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
        int level, data, kind;
        leaf_lookup<TREE>(S, bx, by, bz, data, level, kind);
        if (kind < 2) {  // not air (ray->material is always 0, K/octree.h:92) and able to intersect
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
template <int TREE>
DEV bool closest_hit(const SceneView& S, f3 o, f3 d, int draw_depth, Hit& h, f3& point, LdsStack& stack) {
    bool hit = octree_hit<TREE>(S, o, d, draw_depth, h);
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
template <bool RECORD, int TREE>
DEV f3 sample_path(const SceneView& S, const CameraView& C, const RenderOpts& O, int seed, int gid, LdsStack& stack,
                   HitRecord* rec_out, int* rec_n) {
    unsigned rng = (unsigned)seed + (unsigned)gid;
    rt_pcg_next(&rng);
    const RayOD pr = primary_ray(C, gid, rng, false);
    f3 o = pr.o, d = pr.d;
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
        bool hit = closest_hit<TREE>(S, o, d, O.draw_depth, h, point, stack);
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
            bool shadowed = closest_hit<TREE>(S, o, d, O.draw_depth, sh, sp, stack);
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

// render_pool's pixel slots: a tile of 256 slots is a 16 x 16 block of pixels (neighbouring paths meet the same part of the
// scene: L1 / L2 hits) — all of them with one rank, every world-th with several (chunky_render_set_shard with tile 0); with a
// run length given instead, a tile is one of the rank's runs of consecutive pixel indices.  Returns width * height for a
// padding slot.
#ifndef CHUNKY_TILE_LOG
#define CHUNKY_TILE_LOG 4  // tiles of 16 x 16 pixels
#endif
constexpr int kTileLog = CHUNKY_TILE_LOG, kTileEdge = 1 << kTileLog, kSampleTile = kTileEdge * kTileEdge;  // pixel slots per tile
// Inside a tile the samples are ordered (sub-block of kSubBlock pixel slots, pass, slot in the sub-block): the 256 samples a
// wave claims at a time are 256 / kSubBlock consecutive passes of one small block of pixels — 64 passes of 2 x 2 pixels — so
// the paths a wave starts together begin as nearly the same ray: their march steps read the same tree entries (L1 hits,
// often the same address), find their candidates together and reach SHADE together.  Measured on the bench, sub-blocks of
// 256 (the tile: one pass per claim) / 128 / 64 / 32 / 16 / 8 / 4 / 2 / 1 slots: 5.96 / 6.01 / 6.03 / 6.06 / 6.11 / 6.12 /
// 6.16 / 6.08 / 5.96 Gsamples/s.
#ifndef CHUNKY_SUBBLOCK
#define CHUNKY_SUBBLOCK 4
#endif
constexpr int kSubBlock = CHUNKY_SUBBLOCK;
// sub-block shapes: 256 = the tile, 128 = 16 x 8, 64 = 8 x 8, 32 = 8 x 4, 16 = 4 x 4, 8 = 4 x 2, 4 = 2 x 2, 2 = 2 x 1 pixels, row-major over the
// tile and inside
constexpr int kSubW = kSubBlock >= 128 ? 16 : (kSubBlock >= 32 ? 8 : (kSubBlock >= 8 ? 4 : (kSubBlock >= 2 ? 2 : 1))), kSubH = kSubBlock / kSubW;
static_assert(kTileLog == 4 && kSubW * kSubH == kSubBlock && kSubH >= 1 && kSubH <= 16, "sub-blocks tile a 16 x 16 tile");
DEV int pool_slot_gid(const ShardView& T, int width, int height, int slot) {
    if (T.world != 1 && T.tile != 0) return slot < T.n_local ? shard_gid(T, slot) : width * height;
    const int bw = (width + kTileEdge - 1) >> kTileLog;
    // (several ranks, T.tile == 0: the image's 16 x 16 blocks dealt round-robin — the rank's t-th tile is block t * world + rank)
    int b = slot >> (2 * kTileLog);
    const int i = slot & (kSampleTile - 1);
    if (T.world != 1) {
        b = b * T.world + T.rank;
        if (b >= bw * ((height + kTileEdge - 1) >> kTileLog)) return width * height;
    }
    const int by = b / bw, bx = b - by * bw;
    // slot i of a tile: sub-block i / kSubBlock (row-major over the tile's sub-blocks), then row-major inside it
    const int sb = i / kSubBlock, px = i % kSubBlock;
    const int x = (bx << kTileLog) + (sb % (kTileEdge / kSubW)) * kSubW + px % kSubW,
              y = (by << kTileLog) + (sb / (kTileEdge / kSubW)) * kSubH + px / kSubW;
    return (x < width && y < height) ? y * width + x : width * height;
}
__host__ __device__ inline long long pool_tiles(const ShardView& T, int width, int height) {
    if (T.world != 1) return ((long long)T.n_local + kSampleTile - 1) / kSampleTile;  // runs of T.tile pixels, or (T.tile == 0) the rank's blocks
    return (long long)((width + kTileEdge - 1) >> kTileLog) * ((height + kTileEdge - 1) >> kTileLog);
}

template <int TREE>
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
        f3 c = sample_path<false, TREE>(S, C, O, P.seed[k], gid, stack, nullptr, nullptr);
        int spp = P.first_spp + k;
        float fs = (float)spp, fs1 = (float)(spp + 1);
        mean = f3{(mean.x * fs + c.x) / fs1, (mean.y * fs + c.y) / fs1, (mean.z * fs + c.z) / fs1};
    }
    px[0] = mean.x;
    px[1] = mean.y;
    px[2] = mean.z;
}

// ---------------------------------------------------------------------------------------------
// render_waves — the wave-scheduled form of the same path.
//
// Each lane is a persistent path-state machine; a lane claims a pixel from a global counter, runs
// all passes of the launch for it (running mean in registers) and claims the next one.  A lane is
// always in one of three states, and every iteration the WAVE votes (ballot + popcount, all
// scalar) for the state most lanes are waiting in and executes only that phase:
//
//   MARCH  one octree march step: limit checks, cell, leaf lookup; air -> leaf-exit, stay;
//          block candidate -> BLOCK; end of trace -> SHADE
//   BLOCK  block-model intersection + material/texture test at the current cell; hit -> SHADE,
//          rejected -> leaf-exit, back to MARCH
//   SHADE  everything between two traces: entity BVHs, sky / sun lookup, throughput update, sun
//          sampling, cosine bounce, accumulation, next pass / next pixel, primary ray, trace setup
//
// Per-path arithmetic is exactly sample_path's (the same helpers in the same order on the same
// values), so the image is bit-identical; only which lanes execute together changes.  On the
// benchmark view the one-lane-per-path form keeps 19 % of the VALU lanes busy (profiles/), because
// a wave waits for its longest march and its deepest path.
// lanes of the wave for which p holds, as a 32-bit scalar (a 64-bit popcount makes the compiler do the
// vote comparisons on the VALU: there is no 64-bit scalar less-than)
DEV int count_lanes(bool p) {
    const unsigned long long m = __ballot(p);
    return __builtin_popcount((unsigned)m) + __builtin_popcount((unsigned)(m >> 32));
}

enum : int {
    ST_MARCH = 0, ST_BLOCK = 1, ST_SHADE = 2,  // voted phases (+ ST_BVH)
    ST_BVH = 9,     // at a node of an entity BVH: inner-node visits are voted as one phase,
    ST_LEAF = 11,   // the triangle tests of a leaf as another
    ST_TRACED = 10, // octree part of the trace finished (transient)
    ST_DONE = 3,   // no pixels left for this lane's group
    ST_NEXT = 4,   // path finished, radiance ready
    ST_SETUP = 5,  // ray ready, trace_setup pending
    ST_IDLE = 7,   // lane is free and waits for a pass of its group's pixel
    ST_START = 8   // begin the sample L.pass
};

// What the leader lane of a pixel group keeps for its group (G > 1) — a property of the lane, not of a path.
struct GroupCtl {
    unsigned cur : 1;              // the open pixel passes are issued from
    unsigned exhausted : 1;        // the pixel queue is empty
    unsigned serial_counter : 23;  // serial of the last pixel opened
};

struct LaneState {
    // Small fields share registers (bit-fields): at the 96-VGPR budget of five waves per SIMD every register of
    // path state that is saved is one spill less; none of these is touched by the march loop.
    // pixel / pass
    int gid;
    int sidx;              // render_pool: index of the sample in the launch (pass * n_local + pixel slot)
    unsigned pass : 8;     // pass index inside the launch (< kMaxPassesPerLaunch)
    unsigned slot : 1;     // G > 1: which open pixel of the group this lane's pass belongs to,
    unsigned serial : 23;  //        and that pixel's serial
    f3 mean;               // G = 1 only (grouped lanes keep the means in LDS)
    unsigned rng;
    // path
    f3 radiance, throughput, o, d;
    unsigned depth : 8;
    unsigned shadow : 1;      // the current trace is the sun-sample trace of K/rayTracer.cl:101-106
    unsigned oct_hit : 1;
    unsigned trace_hit : 1;   // closestIntersect result so far (octree, then the BVHs)
    unsigned cand_level : 4;  // level of the candidate's leaf
    unsigned bvh_which : 1;   // entity BVH walked: 0 world, 1 actor
    // trace
    f3 inv;
    float dist_march;
    int steps;
    int cand_data;
    // entity BVH traversal (K/bvh.h:22-113): current node, stack height,
    // and the shadow ray's own copy of record.distance
    int bvh_cur;
    int bvh_top;             // stack height times the stack's lane stride (= offset of the next free entry)
    const int* bvh_base;     // the BVH being walked (world or actor)
    int bvh_head;  // first word of node bvh_cur (> 0: index of its second child; <= 0: -pointer to a leaf's triangles)
    float bvh_dist;
    f3 far;  // per axis 1.0 where the ray runs towards +axis (inv > 0), else 0.0: selects a leaf's exit plane
    // render_pool: the top-level entry of the wide tree read last and its index (-1: none).  A third of the march steps
    // stay inside the top cell of the step before (8^3 blocks): those skip the first of the two dependent reads
    int top_idx, top_e;
    int pid;  // render_pool with entity BVHs: which of the wave's to-visit stacks (LDS) belongs to this path
    // extended integrator: what the current trace is — 0 the path's ray, 1 the sun shadow ray, 2 the
    // emitter shadow ray — the light that ray brings if it is free, and whether the vertex before sampled the emitters
    unsigned tkind : 2;
    unsigned after_nee : 1;
    f3 pend;
    // main record
    Hit h;
};

// (int)floor(x) in one instruction, saturating like v_cvt_i32_f32 (math self test 18).  NaN converts to
// INT_MAX: a cell outside any world, which ends the march like the INT_MIN of the reference's x86 build does.
DEV int floor_to_int(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// AABB_exit of the leaf that holds cell (bx, by, bz) = floor(po) (K/octree.h:103-106, K/primitives.h:52-61).
// The leaf box is [lv << level, (lv + 1) << level) per axis; as floats: min = float(b & -2^level) and
// max = min + 2^level, both exact (integers below 2^24).  The reference takes, per axis,
// fmax((min - p)*inv, (max - p)*inv).  p lies in [min, max), rounding is monotonic, so the larger product is the
// one with the plane the ray leaves through — max if inv > 0, else min — and only that one is evaluated
// (L.far selects it; the fma is exact).  The one case where that product is NaN while the reference's fmax
// returns the other one (p == min with inv == -inf: NaN against -inf) is restored by the fmax with -inf.
DEV float leaf_exit_distance(const LaneState& L, f3 far, f3 po, int bx, int by, int bz, int level) {
    const int keep = -1 << level;
    const float size = __builtin_ldexpf(1.0f, level);
    const float x0 = (float)(bx & keep), y0 = (float)(by & keep), z0 = (float)(bz & keep);
    const float tx = rt_fmax((rt_fma(far.x, size, x0) - po.x) * L.inv.x, -rt_inf());
    const float ty = rt_fmax((rt_fma(far.y, size, y0) - po.y) * L.inv.y, -rt_inf());
    const float tz = rt_fmax((rt_fma(far.z, size, z0) - po.z) * L.inv.z, -rt_inf());
    return rt_fmin(tx, rt_fmin(ty, tz));
}
// 1.0 per axis where the ray runs towards +axis: render_waves keeps it in LaneState.far, render_pool derives it from
// the sign of inv where it is needed (three registers less to carry and to park)
DEV f3 far_of(const f3& inv) { return mk3(inv.x > 0 ? 1.0f : 0.0f, inv.y > 0 ? 1.0f : 0.0f, inv.z > 0 ? 1.0f : 0.0f); }

template <int TREE, bool FARREG = true>
DEV void leaf_exit(const SceneView& S, LaneState& L, f3 po, int bx, int by, int bz, int level) {
    L.dist_march += leaf_exit_distance(L, FARREG ? L.far : far_of(L.inv), po, bx, by, bz, level) + kOffset;
    L.steps += 1;
}

// Start of Octree_octreeIntersect (K/octree.h:44-64): returns the next state.
template <int END, bool FARREG = true>
DEV int trace_setup(const SceneView& S, LaneState& L) {
    const int depth = S.octree_depth;
    L.inv = rcp3(L.d);
    if (FARREG) L.far = far_of(L.inv);
    L.dist_march = 0;
    L.steps = 0;
    L.oct_hit = false;
    int lx = (int)rt_floor(L.o.x) >> depth, ly = (int)rt_floor(L.o.y) >> depth, lz = (int)rt_floor(L.o.z) >> depth;
    if ((lx != 0) | (ly != 0) | (lz != 0)) {
        float size = (float)(1 << depth);
        float dist = box_quick(0, size, 0, size, 0, size, L.o, L.inv);
        if (dist != dist || dist < 0) return END;
        L.dist_march += dist + kOffset;
    }
    return ST_MARCH;
}

// Written without early returns: the arithmetic runs for every lane of the phase (it is harmless for
// a lane whose trace has ended), only the tree reads are guarded, and the outcome is three selects —
// nested exits cost a copy of every live-out per exit in the compiled code.
// One march step (K/octree.h:66-106) for the lanes of `marching`, written for the whole wave with no branch
// around it and no per-lane state flags: who is marching, who found a candidate and whose trace ended are
// 64-bit lane masks in scalar registers, combined with scalar instructions; the vector unit only sees the
// arithmetic.  A lane outside `marching` computes on stale values and keeps none of it — its tree read is made
// safe by the in-world test alone.  Returns the candidates and the lanes still alive; `data` / `level` are
// the leaf every lane looked at (for a lane that has stopped marching they keep coming out the same: its
// position no longer moves).
typedef unsigned long long LaneMask;
DEV bool in_mask(LaneMask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

template <int TREE>
DEV void march_step(const SceneView& S, const RenderOpts& O, LaneState& L, LaneMask marching, LaneMask& cand_out,
                    LaneMask& live_out, int& data, int& level, LaneMask* model_out = nullptr, bool top_cache = false,
                    const LaneMask* far_masks = nullptr) {
    const int depth = S.octree_depth;
    f3 pos = L.o + L.d * L.dist_march;
    f3 po = pos + L.d * kOffset;
    int bx = floor_to_int(po.x), by = floor_to_int(po.y), bz = floor_to_int(po.z);
    const bool inside = ((bx | by | bz) >> depth) == 0;
    const LaneMask live = marching & __ballot(L.steps < O.draw_depth) & __ballot(!(L.dist_march > L.h.distance)) & __ballot(inside);
    int kind;
    if (top_cache) {
        int ci = L.top_idx, ce = L.top_e;  // locals: LaneState stays promotable to registers
        leaf_lookup<TREE>(S, bx, by, bz, data, level, kind, inside, TopCache{&ci, &ce});
        L.top_idx = ci;
        L.top_e = ce;
    } else
        leaf_lookup<TREE>(S, bx, by, bz, data, level, kind, inside);
    const LaneMask hittable = __ballot(kind < 2);
    const LaneMask cand = live & hittable, go = live & ~hittable;
    if (model_out) *model_out = cand & __ballot(kind == 1);  // candidates that are AABB / quad models (or of unknown kind)
    // render_pool passes the three lane masks "inv > 0" (scalar registers, set when the loop is entered) instead of L.far
    const f3 far = far_masks ? mk3(in_mask(far_masks[0]) ? 1.0f : 0.0f, in_mask(far_masks[1]) ? 1.0f : 0.0f, in_mask(far_masks[2]) ? 1.0f : 0.0f)
                             : L.far;
    const float step = leaf_exit_distance(L, far, po, bx, by, bz, level) + kOffset;  // K/octree.h:103-106
    const bool advance = in_mask(go);
    L.dist_march = advance ? L.dist_march + step : L.dist_march;
    L.steps = advance ? L.steps + 1 : L.steps;
    cand_out = cand;
    live_out = live;
}

// CUBE: every candidate here is a full cube by the tree's leaf kind (render_pool votes cubes and models separately)
template <int TREE, int END, bool CUBE = false, bool FARREG = true>
DEV int block_phase(const SceneView& S, LaneState& L) {
    f3 pos = L.o + L.d * L.dist_march;
    f3 po = pos + L.d * kOffset;
    int bx = (int)rt_floor(po.x), by = (int)rt_floor(po.y), bz = (int)rt_floor(po.z);
    Hit t = L.h;
    float dist = (CUBE && S.cube_info && S.block_info) ? block_hit_cube(S, L.cand_data, bx, by, bz, pos, L.d, L.inv, t)
                                        : block_hit(S, L.cand_data, bx, by, bz, pos, L.d, L.inv, t);
    if (!L.shadow) {  // a rejected cube has already overwritten the normal (K/block.h:59-60)
        L.h.normal = t.normal;
        L.h.color = t.color;
        L.h.emittance = t.emittance;
        L.h.spec = t.spec;
    }
    if (dist == dist) {
        if (!L.shadow) {
            L.h.distance = L.dist_march + dist;
            L.h.material = L.cand_data;
        }
        L.oct_hit = true;
        return END;
    }
    leaf_exit<TREE, FARREG>(S, L, po, bx, by, bz, L.cand_level);
    return ST_MARCH;
}

struct WorkQueue {
    int* next;  // next unclaimed local pixel slot
};

// Pixel slots a wave has claimed but not handed to a lane yet: [next, end), wave-uniform.  One
// returning atomic per kPixelBatch pixels per wave instead of one per shade round (a contended
// device-scope atomic costs ~1-3 us, MI355X_MICROARCH.md "dequeue").
constexpr int kPixelBatch = 32;
struct PixelPool {
    int next, end;
};

template <int BATCH>
DEV int claim_slot(WorkQueue Q, PixelPool& pool, bool need) {
    const unsigned long long mask = __ballot(need);
    if (mask == 0) return 0;
    const int n_need = __popcll(mask);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    const int rem = pool.end - pool.next;
    int slot = pool.next + rank;
    if (rem >= n_need) {
        pool.next += n_need;
    } else {
        // take what is left, then a fresh batch (large enough for every waiting lane)
        const int want = n_need - rem > BATCH ? n_need - rem : BATCH;
        int base = 0;
        if (need && rank == 0) base = atomicAdd(Q.next, want);
        base = __builtin_amdgcn_readfirstlane(__shfl(base, __ffsll((long long)mask) - 1));
        if (rank >= rem) slot = base + (rank - rem);
        pool.next = base + (n_need - rem);
        pool.end = base + want;
    }
    return slot;
}

// The octree part of a trace is over: continue closestIntersect (K/kernel.h:16-18) in the entity BVHs.
// A shadow trace only needs the boolean, so it skips the BVHs once anything was hit.
// The walk keeps the first word of its current node in a register: a visit reads both children whole (their
// first words with their boxes), so stepping down needs no further read — only a pop does.
DEV int bvh_enter(const SceneView& S, LaneState& L, int which) {
    L.bvh_which = which;
    L.bvh_base = which ? S.actor_bvh : S.world_bvh;
    L.bvh_cur = 0;
    L.bvh_top = 0;
    L.bvh_head = L.bvh_base[0];
    return L.bvh_head <= 0 ? ST_LEAF : ST_BVH;
}
DEV int bvh_begin(const SceneView& S, LaneState& L) {
    L.trace_hit = L.oct_hit;
    if (L.shadow && L.trace_hit) return ST_SHADE;
    if (S.world_bvh_empty && S.actor_bvh_empty) return ST_SHADE;
    L.bvh_dist = L.h.distance;
    return bvh_enter(S, L, S.world_bvh_empty ? 1 : 0);
}

// Bvh_intersect (K/bvh.h:47-109), one node per execution, as two voted phases so that a wave does not pay for the
// triangle code at every step of the walk: bvh_phase visits an inner node (two box tests, near-first / push-far
// ordering), leaf_phase tests a leaf's triangles; both leave the lane at its next node.  The to-visit stack
// lives in LDS.
DEV int bvh_finished(const SceneView& S, LaneState& L) {
    if (L.bvh_which == 0 && !S.actor_bvh_empty && !(L.shadow && L.trace_hit)) return bvh_enter(S, L, 1);
    return ST_SHADE;
}
DEV int bvh_pop(const SceneView& S, LaneState& L, LdsStack& stack) {
    if (L.bvh_top == 0) return bvh_finished(S, L);
    L.bvh_top -= stack.stride;
    L.bvh_cur = stack.base[L.bvh_top];
    L.bvh_head = L.bvh_base[L.bvh_cur];
    return L.bvh_head <= 0 ? ST_LEAF : ST_BVH;
}

DEV int bvh_phase(const SceneView& S, LaneState& L, LdsStack& stack) {
    const int* __restrict__ bvh = L.bvh_base;
    const float limit = L.shadow ? L.bvh_dist : L.h.distance;
    const int first = L.bvh_cur + 7, second = L.bvh_head;
    const int* a = bvh + first;
    const int* b = bvh + second;
    const int head_a = a[0], head_b = b[0];
    float t1 = box_quick(as_float(a[1]), as_float(a[2]), as_float(a[3]), as_float(a[4]), as_float(a[5]),
                         as_float(a[6]), L.o, L.inv);
    float t2 = box_quick(as_float(b[1]), as_float(b[2]), as_float(b[3]), as_float(b[4]), as_float(b[5]),
                         as_float(b[6]), L.o, L.inv);
    const bool miss1 = (t1 != t1) || t1 > limit;
    const bool miss2 = (t2 != t2) || t2 > limit;
    if (miss1 & miss2) return bvh_pop(S, L, stack);
    // near child first; the other one is pushed when both are hit (K/bvh.h:86-103: the first child is the near one
    // only when t1 < t2)
    const bool go_first = !miss1 & (miss2 | (t1 < t2));
    if (!miss1 & !miss2) {
        stack.base[L.bvh_top] = go_first ? second : first;
        L.bvh_top += stack.stride;
    }
    L.bvh_cur = go_first ? first : second;
    L.bvh_head = go_first ? head_a : head_b;
    return L.bvh_head <= 0 ? ST_LEAF : ST_BVH;
}

DEV int leaf_phase(const SceneView& S, LaneState& L, LdsStack& stack) {
    const int* __restrict__ trigs = S.trigs;
    float limit = L.shadow ? L.bvh_dist : L.h.distance;
    const int prim = -L.bvh_head;
    const int n = trigs[prim];
    for (int i = 0; i < n; i++) {
        f3 nn;
        float u, v;
        int mat;
        float dist = triangle_hit(trigs + prim + 1 + 20 * i, limit, L.o, L.d, nn, u, v, mat);
        if (dist == dist) {
            Hit t = L.h;
            if (material_sample(S, mat, u, v, t)) {
                if (!L.shadow) {
                    L.h.color = t.color;
                    L.h.emittance = t.emittance;
                    L.h.normal = nn;
                    L.h.distance = dist;
                }
                limit = dist;
                L.trace_hit = true;
            }
        }
    }
    if (L.shadow) L.bvh_dist = limit;
    if (L.shadow && L.trace_hit) return bvh_finished(S, L);
    return bvh_pop(S, L, stack);
}

// All launch parameters travel as ONE by-value struct and are read through the kernel-argument
// segment pointer (constant address space, scalar loads).  The march loop keeps only the handful
// of fields it needs in SGPRs; the BLOCK and SHADE phases re-read theirs through a pointer the
// optimiser cannot see through (`fresh_args`), so those ~100 rarely used scalars are not hoisted
// out of the state-machine loop and spilled into VGPR lanes (v_readlane in the march loop was
// ~30 % of its VALU issue before this).
struct WaveArgs {
    SceneView S;
    CameraView C;
    RenderOpts O;
    ShardView T;
    PassSeeds P;
    WorkQueue Q;
    float* res;
    unsigned long long* stats;
    unsigned stack_bytes;  // size of the BVH-stack area at the start of dynamic LDS
    float* staging;        // render_pool: radiance of every sample of the launch, [tile][pass][slot in tile][3]
    unsigned n_samples;    // render_pool: tiles of kSampleTile pixel slots (the last one padded) x P.n
    unsigned xcd_stripe;   // render_pool: samples per range (xcd_claim): the tiles over kXcdRanges, rounded up, x P.n x kSampleTile
};
static_assert(sizeof(WaveArgs) <= 4096, "launch arguments must fit the 4 KB kernel-argument segment");
typedef const WaveArgs __attribute__((address_space(4))) * WaveArgPtr;

// copy one member struct out of the argument segment (explicit cast: the host pass has no
// address-space-qualified copy constructors; on the device the loads stay scalar)
template <typename T>
DEV T arg_copy(const T __attribute__((address_space(4))) * p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *p;  // constant-address-space struct load: only the fields that are used get (scalar) loads
#else
    (void)p;
    return T{};  // host pass: kernels bodies are parsed but never run
#endif
}

DEV WaveArgPtr fresh_args() {
    WaveArgPtr a = (WaveArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    return a;
}

// Cycle sums of parts of SHADE for the profiling build (chunky_render_phase_stats values 14..23).
struct PartTimers {
    unsigned long long t[10];
    unsigned long long last;
};
enum : int { PT_SKY = 0, PT_SAMPLING, PT_SETUP, PT_DEPOSIT, PT_FOLD, PT_OPEN, PT_HANDOUT, PT_NEWSAMPLE };
template <bool ON>
DEV void part_begin(PartTimers* pt) {
    if (ON) pt->last = __builtin_amdgcn_s_memtime();
}
template <bool ON>
DEV void part_end(PartTimers* pt, int which) {  // charges the time since the last begin/end to `which`
    if (ON) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        pt->t[which] += now - pt->last;
        pt->last = now;
    }
}

// SHADE, part 1: everything from the end of a trace to the start of the next one on the same path.
// Returns ST_SETUP (a ray is ready to be traced) or ST_NEXT (the path is finished).
template <int TREE, bool BVH, bool PROF = false>
DEV int shade_phase(const SceneView& S, const RenderOpts& O, LaneState& L, LdsStack& stack, PartTimers* pt = nullptr) {
    part_begin<PROF>(pt);
    const bool hit = BVH ? L.trace_hit : L.oct_hit;  // closestIntersect (K/kernel.h:14-24) is complete
    // Each block below appears once, so a shade round issues it once however the lanes split.
    const bool main_trace = !L.shadow;
    if (!hit) {  // intersectSky (K/kernel.h:26-31); record.emittance = 1 for the main ray (K/rayTracer.cl:95)
        // a shadow ray's record.emittance is the |dot(sun dir, normal)| stored at its start: nothing writes it during
        // the shadow trace (the block and triangle tests leave the main record alone for shadow rays)
        const float e = main_trace ? 1.0f : L.h.emittance;
        L.radiance = L.radiance + sky_radiance(S, L.d, L.throughput, e);
    }
    part_end<PROF>(pt, PT_SKY);
    // one exit: a main ray that reached the sky is finished; every other lane goes on below
    bool finished = !hit && main_trace;
    bool to_sun = false;  // start a shadow ray towards the sun; otherwise bounce
    if (finished) {
    } else if (main_trace) {
        // the hit point (K/kernel.h:21-23) becomes the origin of the shadow ray and stays there until the bounce
        L.o = L.o + L.d * (L.h.distance - kOffset);
        // applyRayColor (K/kernel.h:33-44)
        f3 c = mk3(L.h.color.x, L.h.color.y, L.h.color.z);
        L.throughput = L.throughput * c;
        L.radiance = L.radiance + (c * (L.h.emittance * O.emitter_scale)) * L.throughput;
        if (S.sun_flags & 1) to_sun = true;
    } else {
        L.shadow = false;
    }
    // Sun_sampleDirection (K/sky.h:68-93) for the lanes that start a shadow ray, nextPath (K/kernel.h:46-98)
    // for the lanes that bounce.  A lane does one or the other, and both have the same skeleton — two draws,
    // sin/cos of 2*pi*x2, a square root, a vector, its reciprocal length — so the expensive steps are issued
    // once for both kinds of lane and only the cheap vector algebra in between is specific.
    if (!finished) {
        const float x1 = rt_pcg_float(&L.rng), x2 = rt_pcg_float(&L.rng);
        float sn, cs;
        rt_sincos(2 * RT_PI_F * x2, &sn, &cs);
        const float cos_a = 1 - x1 + x1 * S.sun_radius_cos;          // sun: cosine of the angle off the sun axis
        const float root = rt_sqrt(to_sun ? 1 - cos_a * cos_a : x1);  // sun: sin_a; bounce: r
        const f3 n = L.h.normal;
        f3 a;        // sun: the unnormalised direction; bounce: the unnormalised tangent u
        float len2;  // its squared length, each written as the reference writes it
        if (to_sun) {
            const f3 u = S.su * (cs * root), v = S.sv * (sn * root), w = S.sw * cos_a;
            a = (u * v) + w;  // component-wise product, as the reference has it
            len2 = dot(a, a);
        } else {
            float xx, xy, xz = 0;
            if ((double)rt_fabs(n.x) > 0.1) {
                xx = 0;
                xy = 1;
            } else {
                xx = 1;
                xy = 0;
            }
            a = mk3(xy * n.z - xz * n.y, xz * n.x - xx * n.z, xx * n.y - xy * n.x);
            len2 = a.x * a.x + a.y * a.y + a.z * a.z;
        }
        const float rl = 1 / rt_sqrt(len2);
        a = a * rl;
        if (to_sun) {
            L.d = a;
            L.h.emittance = rt_fabs(dot(L.d, n));
            L.shadow = true;
        } else {
            const float tx = root * cs, ty = root * sn, tz = rt_sqrt(1 - x1);
            const float vx = a.y * n.z - a.z * n.y, vy = a.z * n.x - a.x * n.z, vz = a.x * n.y - a.y * n.x;
            L.d = f3{a.x * tx + vx * ty + n.x * tz, a.y * tx + vy * ty + n.y * tz, a.z * tx + vz * ty + n.z * tz};
            L.o = L.o + L.d * kOffset;
            L.depth += 1;
            L.h.distance = rt_inf();
            finished = !(L.depth < O.max_depth);
        }
    }
    part_end<PROF>(pt, PT_SAMPLING);
    return finished ? ST_NEXT : ST_SETUP;
}

// cosine-weighted direction about n from two draws — the direction part of nextPath (K/kernel.h:52-90), as diffuse_bounce
DEV f3 cosine_direction(f3 n, float x1, float x2) {
    float r = rt_sqrt(x1);
    float theta = 2 * RT_PI_F * x2;
    float st, ct;
    rt_sincos(theta, &st, &ct);
    float tx = r * ct, ty = r * st, tz = rt_sqrt(1 - x1);
    float xx, xy, xz = 0;
    if ((double)rt_fabs(n.x) > 0.1) {
        xx = 0;
        xy = 1;
    } else {
        xx = 1;
        xy = 0;
    }
    float ux = xy * n.z - xz * n.y;
    float uy = xz * n.x - xx * n.z;
    float uz = xx * n.y - xy * n.x;
    r = 1 / rt_sqrt(ux * ux + uy * uy + uz * uz);
    ux *= r;
    uy *= r;
    uz *= r;
    float vx = uy * n.z - uz * n.y;
    float vy = uz * n.x - ux * n.z;
    float vz = ux * n.y - uy * n.x;
    return f3{ux * tx + vx * ty + n.x * tz, uy * tx + vy * ty + n.y * tz, uz * tx + vz * ty + n.z * tz};
}

// SHADE of the EXTENDED integrator: oracle/port.c trace_sample_ext, operation for operation, as a
// state machine over the kind of trace that just ended (L.tkind: 0 the path's ray, 1 the sun shadow ray, 2 the emitter
// shadow ray).  Returns ST_SETUP (a ray is ready to be traced) or ST_NEXT (the path is finished).
template <int TREE, bool BVH>
DEV int shade_phase_ext(const SceneView& S, const RenderOpts& O, LaneState& L) {
    const bool hit = BVH ? L.trace_hit : L.oct_hit;
    const int kind = L.tkind;
    const f3 n = L.h.normal;
    int next = 0;  // 1 sun sampling, 2 emitter sampling, 3 diffuse bounce, 4 specular bounce (ray already set)
    if (kind == 0) {
        if (!hit) {
            L.radiance = L.radiance + sky_radiance(S, L.d, L.throughput, 1.0f);
            return ST_NEXT;
        }
        const f3 d_in = L.d;
        L.o = L.o + L.d * (L.h.distance - kOffset);  // rec.point
        const f3 base = L.throughput;
        const f3 c = mk3(L.h.color.x, L.h.color.y, L.h.color.z);
        const f3 thr_d = base * c;
        if (O.emitters && !L.after_nee) L.radiance = L.radiance + (c * (L.h.emittance * O.emitter_scale)) * thr_d;
        L.after_nee = false;
        bool specular = false;
        float metal = 0, rough = 0;
        if (O.bsdf) {
            const float spec = rt_unorm8((unsigned)L.h.spec & 0xFFu);
            metal = rt_unorm8(((unsigned)L.h.spec >> 8) & 0xFFu);
            rough = rt_unorm8(((unsigned)L.h.spec >> 16) & 0xFFu);
            const float ps = rt_fmax(spec, metal);
            if (ps > 0) specular = rt_pcg_float(&L.rng) < ps;
        }
        if (specular) {
            L.throughput = mk3(base.x * (c.x * metal + (1 - metal)), base.y * (c.y * metal + (1 - metal)), base.z * (c.z * metal + (1 - metal)));
            f3 refl = d_in - n * (2 * dot(d_in, n));
            if (rough > 0) {
                const float x1 = rt_pcg_float(&L.rng), x2 = rt_pcg_float(&L.rng);
                const f3 dd = cosine_direction(n, x1, x2);
                refl = normalize(dd * rough + refl * (1 - rough));
                const float rn = dot(refl, n);
                if (rn < 0) refl = refl - n * (2 * rn);
            }
            L.d = refl;
            L.o = L.o + refl * kOffset;
            next = 4;
        } else {
            L.throughput = thr_d;
            next = 1;
        }
    } else if (kind == 1) {
        if (!hit) L.radiance = L.radiance + sky_radiance(S, L.d, L.throughput, L.h.emittance);
        next = 2;
    } else {
        if (!hit) L.radiance = L.radiance + L.pend;
        next = 3;
    }
    if (next == 1) {  // Sun_sampleDirection + shadow ray (K/sky.h:68-93, K/rayTracer.cl:101-106): the record keeps its distance
        const bool sun_on = O.sun_sampling < 0 ? (S.sun_flags & 1) != 0 : O.sun_sampling != 0;
        if (sun_on) {
            L.d = sun_sample(S, L.rng);
            L.h.emittance = rt_fabs(dot(L.d, n));
            L.tkind = 1;
            L.shadow = true;
            return ST_SETUP;
        }
        next = 2;
    }
    if (next == 2) {
        next = 3;
        // not at the last vertex: the bounce ray that would find the same light implicitly is never traced there
        if (O.nee && O.emitters && S.n_emitters > 0 && (int)L.depth + 1 < O.max_depth) {
            const float xk = rt_pcg_float(&L.rng), xf = rt_pcg_float(&L.rng), xu = rt_pcg_float(&L.rng), xv = rt_pcg_float(&L.rng);
            int k = (int)(xk * (float)S.n_emitters);
            if (k > S.n_emitters - 1) k = S.n_emitters - 1;
            int face = (int)(xf * 6.0f);
            if (face > 5) face = 5;
            const int4 em = S.emitters[k];
            const int level = (em.w >> 25) & 15, block = em.w & 0x1FFFFFF;
            const float size = (float)(1 << level);
            const float a = xu * size, b = xv * size;
            const float fa = a - rt_floor(a), fb = b - rt_floor(b);
            const float ex = (float)em.x, ey = (float)em.y, ez = (float)em.z;
            f3 pe, nf;
            float tu, tv;
            switch (face) {
                case 0: pe = mk3(ex, ey + a, ez + b); nf = mk3(-1, 0, 0); tu = 1 - fb; tv = fa; break;
                case 1: pe = mk3(ex + size, ey + a, ez + b); nf = mk3(1, 0, 0); tu = fb; tv = fa; break;
                case 2: pe = mk3(ex + a, ey, ez + b); nf = mk3(0, -1, 0); tu = fa; tv = 1 - fb; break;
                case 3: pe = mk3(ex + a, ey + size, ez + b); nf = mk3(0, 1, 0); tu = fa; tv = fb; break;
                case 4: pe = mk3(ex + a, ey + b, ez); nf = mk3(0, 0, -1); tu = fa; tv = fb; break;
                default: pe = mk3(ex + a, ey + b, ez + size); nf = mk3(0, 0, 1); tu = 1 - fa; tv = fb; break;
            }
            const f3 l = pe - L.o;
            const float d2 = dot(l, l);
            const float dist = rt_sqrt(d2);
            const f3 dir = l * (1 / dist);
            const float cs = dot(dir, n), cl = -dot(dir, nf);
            L.after_nee = true;
            if (cs > 0 && cl > 0 && dist > 0.002f) {
                Hit er = L.h;
                if (material_sample(S, S.blocks[block + 1], tu, tv, er) && er.emittance > 0) {
                    const float w = (cs * cl) / (RT_PI_F * d2) * (6.0f * (float)S.n_emitters * (size * size));
                    const f3 ce = mk3(er.color.x, er.color.y, er.color.z);
                    const f3 le = ce * (ce * (er.emittance * O.emitter_scale));
                    L.pend = L.throughput * (le * w);
                    L.d = dir;
                    L.h.distance = dist - 0.001f;  // anything nearer than the emitter's face hides it
                    L.tkind = 2;
                    L.shadow = true;
                    return ST_SETUP;
                }
            }
        }
    }
    if (next == 3) {
        const float x1 = rt_pcg_float(&L.rng), x2 = rt_pcg_float(&L.rng);
        L.d = cosine_direction(n, x1, x2);
        L.o = L.o + L.d * kOffset;
    }
    L.depth += 1;
    L.h.distance = rt_inf();
    L.tkind = 0;
    L.shadow = false;
    return (int)L.depth < O.max_depth ? ST_SETUP : ST_NEXT;
}

// SHADE, part 2 for G = 1 (one lane per pixel), called from wave-uniform control flow (the pixel
// pool must be updated by the whole wave): accumulate finished paths, hand out pixels, start the next
// sample of every lane in ST_NEXT.  The grouped form below does the same for G > 1.
template <int TREE>
DEV int next_sample_single(const SceneView& S, const CameraView& C, const ShardView& T, WaveArgPtr A, PixelPool& pool,
                    LaneState& L, int st, bool fresh) {
    const int first_spp = A->P.first_spp, n_passes = A->P.n;
    float* __restrict__ res = A->res;
    WorkQueue Q = arg_copy(&A->Q);
    const bool nxt = st == ST_NEXT;
    bool need_pixel = fresh;
    if (nxt && !fresh) {
        // ---- accumulate (K/rayTracer.cl:109-112) ----
        int spp = first_spp + L.pass;
        float fs = (float)spp, fs1 = (float)(spp + 1);
        L.mean = f3{(L.mean.x * fs + L.radiance.x) / fs1, (L.mean.y * fs + L.radiance.y) / fs1,
                    (L.mean.z * fs + L.radiance.z) / fs1};
        L.pass += 1;
        if (L.pass >= n_passes) {
            float* px = res + 3 * (size_t)L.gid;
            px[0] = L.mean.x;
            px[1] = L.mean.y;
            px[2] = L.mean.z;
            need_pixel = true;
        }
    }
    const int slot = claim_slot<kPixelBatch>(Q, pool, need_pixel);  // convergent: every lane of the wave is here
    if (!nxt) return st;
    if (need_pixel) {
        int gid = slot < T.n_local ? shard_gid(T, slot) : C.width * C.height;
        if (gid >= C.width * C.height) return ST_DONE;
        L.gid = gid;
        L.pass = 0;
        const float* px = res + 3 * (size_t)gid;
        L.mean = mk3(px[0], px[1], px[2]);
    }
    // ---- new sample (K/rayTracer.cl:55-91) ----
    {
        // locals, not struct members, as out-parameters: keeps LaneState promotable to registers
        unsigned rng = (unsigned)A->P.seed[L.pass] + (unsigned)L.gid;  // per-lane index: a vector load from the argument segment
        rt_pcg_next(&rng);
        const RayOD pr = primary_ray(C, L.gid, rng, false);
        L.rng = rng;
        L.o = pr.o;
        L.d = pr.d;
    }
    L.radiance = mk3(0, 0, 0);
    L.throughput = mk3(1, 1, 1);
    L.depth = 0;
    L.shadow = false;
    L.h.distance = rt_inf();
    return ST_SETUP;
}

// SHADE, part 2 for G > 1, called from wave-uniform control flow (every lane of the wave is here: it
// uses cross-lane operations and updates the wave's pixel pool).
//
// A pixel is shared by a GROUP of G adjacent lanes.  The passes of the launch are handed to the
// group's lanes one at a time, on demand, so a work item is a single path instead of a whole pixel:
// that removes the idle tail of the persistent grid (12 % of wave time with one lane per pixel) and
// keeps every lane busy when a GPU owns few pixels (8-GPU strong scaling leaves one pixel per lane).
// The running mean of K/rayTracer.cl:109-112 must still absorb the passes IN ORDER: a finished path
// parks its radiance in the group's LDS ring at its pass index, tagged {pixel serial, pass}, and the
// group leader folds the parked values strictly by pass index.  Same float recurrence, same order:
// the image is bit-identical for every G.  Two pixels are open per group — passes are issued from
// the newer one while the older one waits for its last paths — so a group never drains between pixels.
#ifndef CHUNKY_HANDOVER_BATCH
#define CHUNKY_HANDOVER_BATCH 8
#endif
constexpr int kHandoverBatch = CHUNKY_HANDOVER_BATCH;
// vote weights: the phase with the largest (waiting lanes x weight) runs next
#ifndef CHUNKY_W_MARCH
#define CHUNKY_W_MARCH 4
#endif
#ifndef CHUNKY_W_BLOCK
#define CHUNKY_W_BLOCK 4
#endif
#ifndef CHUNKY_W_SHADE
#define CHUNKY_W_SHADE 4
#endif
constexpr int kWMarch = CHUNKY_W_MARCH, kWBlock = CHUNKY_W_BLOCK, kWShade = CHUNKY_W_SHADE;
// parked radiances per open pixel (a pass is issued only inside fold + ring): two per lane of the group; the rings
// of a workgroup take 18 KB of LDS either way, which leaves room for five workgroups per CU
constexpr int ring_size(int group) { return 2 * group; }
struct GroupLds {
    float4* rad;  // [2][kRing] {r, g, b, tag}
    int* hdr;     // [2][8]  {gid, fold, issue, serial, mean.x, mean.y, mean.z, -}
};
enum : int { H_GID = 0, H_FOLD = 1, H_ISSUE = 2, H_SERIAL = 3, H_MEAN = 4 };

template <int TREE, int G, bool PROF = false, int RING = ring_size(G)>
DEV int next_sample(const SceneView& S, const CameraView& C, const ShardView& T, WaveArgPtr A, PixelPool& pool,
                    LaneState& L, GroupCtl& ctl, GroupLds lds, int st, PartTimers* pt = nullptr) {
    constexpr int kRing = RING;
    part_begin<PROF>(pt);
    const int first_spp = A->P.first_spp, n_passes = A->P.n;
    float* __restrict__ res = A->res;
    WorkQueue Q = arg_copy(&A->Q);
    const int lane = (int)(threadIdx.x & 63u);
    const int sub = lane & (G - 1);
    const int leader = lane & ~(G - 1);
    const bool is_leader = sub == 0;
    // ---- a finished path parks its radiance at its pass index ----
    if (st == ST_NEXT) {
        lds.rad[L.slot * kRing + (L.pass & (kRing - 1))] =
            make_float4(L.radiance.x, L.radiance.y, L.radiance.z, __int_as_float((L.serial << 8) | L.pass));
        st = ST_IDLE;  // free for another pass
    }
    part_end<PROF>(pt, PT_DEPOSIT);
    // ---- fold parked radiances strictly in pass order.  Six lanes of the group work at once: lanes 0-2 take
    //      the R, G, B channel of open pixel 0, lanes 3-5 those of open pixel 1, so the fold code — one exact
    //      division per pass and channel, K/rayTracer.cl:109-112 — is issued once for all six ----
    int4* const hdr4 = (int4*)lds.hdr;  // per open pixel: {gid, fold, issue, serial}, {mean.x, mean.y, mean.z, -}
    if (sub < 6) {
        const int k = sub >= 3 ? 1 : 0, ch = sub - 3 * k;
        const int4 h = hdr4[2 * k];
        if (h.x >= 0) {
            float mean = __int_as_float(lds.hdr[8 * k + H_MEAN + ch]);
            const float* const ring = (const float*)(lds.rad + k * kRing);
            int fn = h.y;
            while (fn < n_passes) {
                const float* const c = ring + 4 * (fn & (kRing - 1));
                if (__float_as_int(c[3]) != ((h.w << 8) | fn)) break;
                const int spp = first_spp + fn;
                mean = (mean * (float)spp + c[ch]) / (float)(spp + 1);
                fn++;
            }
            if (fn >= n_passes) {  // pixel complete
                res[3 * (size_t)h.x + ch] = mean;
                if (ch == 0) hdr4[2 * k] = make_int4(-1, fn, h.z, h.w);
            } else if (fn != h.y) {
                if (ch == 0) hdr4[2 * k] = make_int4(h.x, fn, h.z, h.w);
                lds.hdr[8 * k + H_MEAN + ch] = __float_as_int(mean);
            }
        }
    }
    part_end<PROF>(pt, PT_FOLD);
    // ---- leader: open a new pixel when the issuing one is used up and a slot is free ----
    bool need_pixel = false;
    int target = 0;
    if (is_leader && !ctl.exhausted) {
        const int4 hc = hdr4[2 * ctl.cur], ho = hdr4[2 * (ctl.cur ^ 1)];
        const bool cur_open = hc.x >= 0;
        if (!cur_open) {
            need_pixel = true;
            target = ctl.cur;
        } else if (hc.z >= n_passes && ho.x < 0) {
            need_pixel = true;
            target = ctl.cur ^ 1;
        }
    }
    const int pos = claim_slot<(64 / G)>(Q, pool, need_pixel);  // small batches: pixels cannot move between waves once claimed
    if (need_pixel) {
        int gid = pos < T.n_local ? shard_gid(T, pos) : -1;
        if (gid >= C.width * C.height) gid = -1;  // padding of the last tile: only padding follows
        if (gid < 0) {
            ctl.exhausted = true;
        } else {
            const float* px = res + 3 * (size_t)gid;
            ctl.serial_counter += 1;
            hdr4[2 * target] = make_int4(gid, 0, 0, ctl.serial_counter);
            hdr4[2 * target + 1] = make_int4(__float_as_int(px[0]), __float_as_int(px[1]), __float_as_int(px[2]), 0);
            ctl.cur = target;
        }
    }
    part_end<PROF>(pt, PT_OPEN);
    // ---- hand passes of the issuing pixel to the lanes that are free ----
    const bool want = st == ST_IDLE;
    const unsigned gmask = (unsigned)(__ballot(want) >> leader) & (G >= 32 ? 0xFFFFFFFFu : (1u << (G & 31)) - 1u);
    const int cur = __shfl(ctl.cur, leader);
    const int exhausted = __shfl((int)ctl.exhausted, leader);
    const int4 hc = hdr4[2 * cur];
    const int gid = hc.x, issue = hc.z, serial = hc.w;
    int limit = hc.y + kRing;  // ring capacity
    limit = limit < n_passes ? limit : n_passes;
    if (want) {
        const int p = issue + __popc(gmask & ((1u << sub) - 1u));
        if (gid >= 0 && p < limit) {
            L.pass = p;
            L.slot = cur;
            L.serial = serial;
            L.gid = gid;
            st = ST_START;
        } else if (exhausted && gid < 0 && hdr4[2 * (cur ^ 1)].x < 0) {
            st = ST_DONE;
        }
    }
    if (is_leader && gid >= 0) {
        int nx = issue + __popc(gmask);
        hdr4[2 * cur] = make_int4(gid, hc.y, nx < limit ? nx : limit, serial);
    }
    part_end<PROF>(pt, PT_HANDOUT);
    if (st != ST_START) return st;
    // ---- new sample (K/rayTracer.cl:55-91) ----
    {
        // locals, not struct members, as out-parameters: keeps LaneState promotable to registers
        unsigned rng = (unsigned)A->P.seed[L.pass] + (unsigned)gid;  // per-lane index: a vector load from the argument segment
        rt_pcg_next(&rng);
        const RayOD pr = primary_ray(C, gid, rng, false);
        L.rng = rng;
        L.o = pr.o;
        L.d = pr.d;
    }
    L.radiance = mk3(0, 0, 0);
    L.throughput = mk3(1, 1, 1);
    L.depth = 0;
    L.shadow = false;
    L.h.distance = rt_inf();
    part_end<PROF>(pt, PT_NEWSAMPLE);
    return ST_SETUP;
}

// STATS = true adds a per-phase profile of the state machine (executions, active lanes, shader
// cycles by s_memtime), summed over waves into stats[phase*3 + {0,1,2}]; used by tools/phase_stats.py.
// BVH = false compiles the entity-BVH state out (scenes whose two BVHs are the empty sentinel).
template <int TREE, bool STATS, int G, bool BVH = false>
// Five workgroups per CU (96 VGPRs; a march step waits on one or two dependent tree reads, and the fifth wave per
// SIMD fills that time: +5 % over four); the entity-BVH kernels need ~125 registers and stay at four, like the
// profiling build (its counters would spill).
#ifndef CHUNKY_WAVES_PER_SIMD
#define CHUNKY_WAVES_PER_SIMD 5
#endif
__global__ void __launch_bounds__(256, ((BVH || STATS) ? 4 : CHUNKY_WAVES_PER_SIMD)) render_waves(WaveArgs unused_by_name) {
    constexpr int END = BVH ? ST_TRACED : ST_SHADE;  // where a lane goes when the octree part of a trace ends
    extern __shared__ int lds[];
    LdsStack stack{lds + threadIdx.x, (int)blockDim.x};
    LaneState L;
    L.h.material = 0;
    L.h.normal = mk3(0, 0, 0);
    L.h.color = f4{0, 0, 0, 0};
    L.h.emittance = 0;
    L.cand_data = 0;
    L.cand_level = 0;
    L.pass = 0;
    L.gid = -1;  // no pixel yet
    L.mean = mk3(0, 0, 0);
    L.slot = 0;
    L.serial = 0;
    GroupCtl ctl;
    ctl.cur = 0;
    ctl.serial_counter = 0;
    ctl.exhausted = false;
    L.steps = 0;
    L.radiance = mk3(0, 0, 0);
    L.oct_hit = false;
    L.trace_hit = false;
    L.bvh_cur = L.bvh_top = L.bvh_which = L.bvh_head = 0;
    L.bvh_base = nullptr;
    L.bvh_dist = 0;
    // per-group radiance buffers behind the BVH stacks in dynamic LDS
    GroupLds glds{nullptr, nullptr};
    if (G > 1) {
        constexpr int kRing = ring_size(G);
        const unsigned stack_bytes = fresh_args()->stack_bytes;
        char* base = (char*)lds + stack_bytes + (threadIdx.x / G) * (2 * kRing * 16 + 64);
        glds.rad = (float4*)base;
        glds.hdr = (int*)(base + 2 * kRing * 16);
        for (int i = threadIdx.x & (G - 1); i < 2 * kRing; i += G) glds.rad[i] = make_float4(0, 0, 0, __int_as_float(-1));
        for (int i = threadIdx.x & (G - 1); i < 16; i += G) glds.hdr[i] = -1;
    }
    unsigned long long prof[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long hand_cycles = 0, hand_execs = 0;  // the pixel/pass hand-over part of SHADE
    PartTimers parts{{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0};
    unsigned long long t_begin = 0;
    if (STATS) t_begin = __builtin_amdgcn_s_memtime();
    PixelPool pool{0, 0};
    // Every lane starts free (G > 1) or finished (G = 1): the first samples are handed out by the loop itself, so the
    // hand-over and the trace set-up exist ONCE in the kernel.  A second copy of them on another path of the loop (the
    // "nobody is tracing" rounds used to have one) makes the compiler keep every path-state register twice and copy
    // between the two sets on each iteration: ~80 v_mov per iteration, 7 % of all instructions issued.
    int st = G == 1 ? ST_NEXT : ST_IDLE;
    bool first_round = true;
    int idle_rounds = 0;
    for (;;) {
        if (BVH && __ballot(st == ST_TRACED)) {  // octree part of some traces just ended: entity BVHs next
            const SceneView S = arg_copy(&fresh_args()->S);
            if (st == ST_TRACED) st = bvh_begin(S, L);
        }
        const int n_march = count_lanes(st == ST_MARCH);
        const int n_block = count_lanes(st == ST_BLOCK);
        const int n_shade = count_lanes(st == ST_SHADE);
        const int n_bvh = BVH ? count_lanes(st == ST_BVH) : 0;
        const int n_leaf = BVH ? count_lanes(st == ST_LEAF) : 0;
        // nobody is tracing: everything is parked, so folding / pixel hand-out can always advance (SHADE branch below)
        const bool idle = (n_march | n_block | n_shade | n_bvh | n_leaf) == 0;
        if (idle && !first_round && (G == 1 || __ballot(st != ST_DONE) == 0 || ++idle_rounds > 64)) break;
        if (!idle) idle_rounds = 0;
        unsigned long long t0 = 0;
        if (STATS) t0 = __builtin_amdgcn_s_memtime();
        int ph, n_ph1 = n_block;  // lanes served by a phase profiled as BLOCK (the BVH phases are)
        const int n_octree = n_march > n_block ? (n_march > n_shade ? n_march : n_shade) : (n_block > n_shade ? n_block : n_shade);
        if (!idle && BVH && n_bvh > 0 && n_bvh >= n_octree && n_bvh >= n_leaf) {
            ph = 1;  // profiled with BLOCK
            n_ph1 = n_bvh;
            const SceneView S = arg_copy(&fresh_args()->S);
            if (STATS) {
                if (st == ST_BVH) st = bvh_phase(S, L, stack);
            } else {
                // keep visiting nodes while the BVH walk holds the majority; a walk only leaves to LEAF or SHADE
                int nv, nl, ns;
                do {
                    if (st == ST_BVH) st = bvh_phase(S, L, stack);
                    nv = count_lanes(st == ST_BVH);
                    nl = count_lanes(st == ST_LEAF);
                    ns = count_lanes(st == ST_SHADE);
                } while (nv > 0 && nv >= nl && nv >= n_march && nv >= n_block && nv >= ns);
            }
        } else if (!idle && BVH && n_leaf > 0 && n_leaf >= n_octree) {
            ph = 1;
            n_ph1 = n_leaf;
            const SceneView S = arg_copy(&fresh_args()->S);
            if (st == ST_LEAF) st = leaf_phase(S, L, stack);
        } else if (!idle && n_march * kWMarch >= n_block * kWBlock && n_march * kWMarch >= n_shade * kWShade) {
            ph = 0;
            // the few scalars MARCH needs are re-read here too (scalar cache hits): kept live across
            // the whole loop they are the first thing the allocator spills to VGPR lanes
            WaveArgPtr A = fresh_args();
            const SceneView Sm = arg_copy(&A->S);
            const RenderOpts Om = arg_copy(&A->O);
            {
                // stay in the march while it keeps the majority: an inner loop whose back-edge carries
                // only what MARCH changes (the outer loop's back-edge re-shuffles ~25 state registers)
                // The lanes' states stay untouched inside the loop; they are written once when it is left.
                // `ne` counts the lanes waiting where a trace that ends here goes next (SHADE, or the entity BVHs)
                int nm = n_march, nb = n_block, ne = BVH ? n_bvh + n_leaf : n_shade;
                const int n_other = BVH ? n_shade : 0;
                const LaneMask entered = __ballot(st == ST_MARCH);
                LaneMask marching = entered, to_block = 0;
                int data, level;
                do {
                    if (STATS) {  // every inner iteration counts as one MARCH execution (cycles are added below)
                        prof[0] += 1;
                        prof[1] += (unsigned long long)nm;
                    }
                    LaneMask cand, live;
                    march_step<TREE>(Sm, Om, L, marching, cand, live, data, level);
                    nb += __popcll(cand);
                    ne += __popcll(marching & ~live);
                    to_block |= cand;
                    marching = live & ~cand;
                    nm = __popcll(marching);
                } while (nm > 0 && nm * kWMarch >= nb * kWBlock && nm * kWMarch >= ne * kWShade && nm >= n_other);
                const bool found = in_mask(to_block);
                L.cand_data = found ? data : L.cand_data;
                L.cand_level = found ? level : L.cand_level;
                st = found ? ST_BLOCK : (in_mask(entered & ~marching & ~to_block) ? END : st);
                if (STATS) {  // undo the one execution the common accounting below adds
                    prof[0] -= 1;
                    prof[1] -= (unsigned long long)n_march;
                }
            }
        } else if (!idle && n_block * kWBlock >= n_shade * kWShade) {
            ph = 1;
            const SceneView S = arg_copy(&fresh_args()->S);
            if (st == ST_BLOCK) st = block_phase<TREE, END>(S, L);
        } else {
            ph = 2;
            WaveArgPtr A = fresh_args();
            const SceneView S = arg_copy(&A->S);
            const RenderOpts O = arg_copy(&A->O);
            if (st == ST_SHADE) st = shade_phase<TREE, BVH, STATS>(S, O, L, stack, &parts);
            // G > 1: hand-over rounds cost ~300 instructions; wait until a few finished paths share one
            if (idle || count_lanes(st == ST_NEXT) >= (G == 1 ? 1 : kHandoverBatch)) {
                unsigned long long th = 0;
                if (STATS) th = __builtin_amdgcn_s_memtime();
                const CameraView C = arg_copy(&A->C);
                const ShardView T = arg_copy(&A->T);
                if (G == 1)
                    st = next_sample_single<TREE>(S, C, T, A, pool, L, st, first_round);
                else
                    st = next_sample<TREE, G, STATS>(S, C, T, A, pool, L, ctl, glds, st, &parts);
                if (STATS) {
                    hand_cycles += __builtin_amdgcn_s_memtime() - th;
                    hand_execs += 1;
                }
            }
            part_begin<STATS>(&parts);
            if (st == ST_SETUP) st = trace_setup<END>(S, L);
            part_end<STATS>(&parts, PT_SETUP);
            first_round = false;
        }
        if (STATS) {
            unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
            int n = ph == 0 ? n_march : (ph == 1 ? n_ph1 : n_shade);
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (ph == k) {
                    prof[3 * k] += 1;
                    prof[3 * k + 1] += (unsigned long long)n;
                    prof[3 * k + 2] += dt;
                }
        }
    }
    if (STATS && (threadIdx.x & 63) == 0) {
        unsigned long long* stats = fresh_args()->stats;
        for (int k = 0; k < 9; k++) atomicAdd(&stats[k], prof[k]);
        // wave lifetimes: [9] sum, [10] max, [11] waves — the gap between mean and max is the launch tail
        const unsigned long long life = __builtin_amdgcn_s_memtime() - t_begin;
        atomicAdd(&stats[9], life);
        atomicMax(&stats[10], life);
        atomicAdd(&stats[11], 1ull);
        atomicAdd(&stats[12], hand_execs);
        atomicAdd(&stats[13], hand_cycles);
        for (int k = 0; k < 10; k++) atomicAdd(&stats[14 + k], parts.t[k]);
    }
}

// ---------------------------------------------------------------------------------------------
// render_pool — the wave-scheduled path state machine with a POOL of paths per wave.
//
// render_waves binds a path to a lane for its whole life, so the phase a wave executes only ever serves the lanes
// that happen to wait in it (39 / 27 / 35 of 64 for MARCH / BLOCK / SHADE on the benchmark view).  Here a wave owns
// 64 + K paths: 64 in its lanes' registers and K parked in LDS (32 dwords each: everything LaneState carries between
// phases).  Before a phase runs, lanes whose path waits for another phase swap it for a parked path that waits for
// this one, so the phase executes for (almost) every lane as long as the pool holds 64 such paths; nothing is shared
// between waves and nothing waits on another wave.  The vote is over the pool, not the lanes.
//
// A work item is one SAMPLE (pass, pixel), claimed from a global counter in pass-major order; its radiance goes to a
// staging array [pass][pixel][3] and fold_kernel applies the running mean of K/rayTracer.cl:109-112 in pass order
// afterwards — the same float recurrence in the same order, so the image is bit-identical, and the pixel groups, LDS
// rings and hand-over rounds of render_waves (15 % of its time) do not exist here.
enum : int {
    ST_FRESH = 12,  // the lane (or parked slot) holds no path and wants a sample; served by the SHADE branch
    ST_MODEL = 13   // at a candidate block that is not a full cube (ST_BLOCK then means: a full cube)
};

struct PoolLds {
    uint4* park;  // [WORDS][K]: 16-byte word g of slot s at park[g * K + s] (consecutive lanes, consecutive 16 bytes)
    int* tags;    // [K] state of the path parked in slot s
    int* list;    // [K] scratch: the slots taking part in a swap, by rank
};

DEV int phase_class(int st) {
    return st == ST_MARCH ? 0 : (st == ST_BLOCK ? 1 : (st == ST_DONE ? 3 : (st == ST_MODEL ? 4 : ((st == ST_BVH || st == ST_LEAF) ? 5 : 2))));
}

// A parked path is WORDS 16-byte words.  7 words without entity BVHs (the march-step count shares word 0 with the flags —
// launch_pool sends draw depths above 65535 to render_waves — and the candidate block takes the place of the BVH cursor's
// word); 8 with them; 9 for the extended integrator.  The flag bits sit where LaneState's bit-fields have them.
template <int WORDS>
DEV void pool_pack(const LaneState& L, uint4 (&v)[WORDS]) {
    constexpr int H = WORDS == 7 ? 5 : 6;  // first of the two words of the main record
    const unsigned misc = (unsigned)L.depth | ((unsigned)L.shadow << 8) | ((unsigned)L.oct_hit << 9) | ((unsigned)L.trace_hit << 10) |
                          ((unsigned)L.cand_level << 11) | ((unsigned)L.bvh_which << 15) |
                          (WORDS == 7 ? (unsigned)L.steps << 16 : ((unsigned)L.pid << 16) | ((unsigned)L.tkind << 24) | ((unsigned)L.after_nee << 26));
    if (WORDS > 8) v[WORDS - 1] = make_uint4(__float_as_uint(L.pend.x), __float_as_uint(L.pend.y), __float_as_uint(L.pend.z), (unsigned)L.h.spec);
    v[0] = make_uint4((unsigned)L.sidx, L.rng, misc, WORDS == 7 ? (unsigned)L.cand_data : (unsigned)L.steps);
    v[1] = make_uint4(__float_as_uint(L.radiance.x), __float_as_uint(L.radiance.y), __float_as_uint(L.radiance.z), __float_as_uint(L.throughput.x));
    v[2] = make_uint4(__float_as_uint(L.throughput.y), __float_as_uint(L.throughput.z), __float_as_uint(L.o.x), __float_as_uint(L.o.y));
    v[3] = make_uint4(__float_as_uint(L.o.z), __float_as_uint(L.d.x), __float_as_uint(L.d.y), __float_as_uint(L.d.z));
    v[4] = make_uint4(__float_as_uint(L.inv.x), __float_as_uint(L.inv.y), __float_as_uint(L.inv.z), __float_as_uint(L.dist_march));
    if (WORDS > 7) v[5] = make_uint4((unsigned)L.bvh_cur, (unsigned)L.bvh_top, __float_as_uint(L.bvh_dist), (unsigned)L.cand_data);
    v[H] = make_uint4(__float_as_uint(L.h.distance), __float_as_uint(L.h.normal.x), __float_as_uint(L.h.normal.y), __float_as_uint(L.h.normal.z));
    v[H + 1] = make_uint4(__float_as_uint(L.h.color.x), __float_as_uint(L.h.color.y), __float_as_uint(L.h.color.z), __float_as_uint(L.h.emittance));
}
template <int WORDS>
DEV void pool_unpack(LaneState& L, const uint4 (&v)[WORDS]) {
    constexpr int H = WORDS == 7 ? 5 : 6;
    if (WORDS > 8) {
        L.pend = mk3(__uint_as_float(v[WORDS - 1].x), __uint_as_float(v[WORDS - 1].y), __uint_as_float(v[WORDS - 1].z));
        L.h.spec = (int)v[WORDS - 1].w;
    }
    L.sidx = (int)v[0].x; L.rng = v[0].y;
    L.depth = v[0].z & 0xFFu; L.shadow = (v[0].z >> 8) & 1u; L.oct_hit = (v[0].z >> 9) & 1u; L.trace_hit = (v[0].z >> 10) & 1u;
    L.cand_level = (v[0].z >> 11) & 15u; L.bvh_which = (v[0].z >> 15) & 1u;
    if (WORDS == 7) {
        L.steps = (int)(v[0].z >> 16);
        L.cand_data = (int)v[0].w;
    } else {
        L.pid = (int)((v[0].z >> 16) & 0xFFu); L.tkind = (v[0].z >> 24) & 3u; L.after_nee = (v[0].z >> 26) & 1u;
        L.steps = (int)v[0].w;
        L.bvh_cur = (int)v[5].x; L.bvh_top = (int)v[5].y; L.bvh_dist = __uint_as_float(v[5].z);
        L.cand_data = (int)v[5].w;
    }
    L.radiance = mk3(__uint_as_float(v[1].x), __uint_as_float(v[1].y), __uint_as_float(v[1].z));
    L.throughput = mk3(__uint_as_float(v[1].w), __uint_as_float(v[2].x), __uint_as_float(v[2].y));
    L.o = mk3(__uint_as_float(v[2].z), __uint_as_float(v[2].w), __uint_as_float(v[3].x));
    L.d = mk3(__uint_as_float(v[3].y), __uint_as_float(v[3].z), __uint_as_float(v[3].w));
    L.inv = mk3(__uint_as_float(v[4].x), __uint_as_float(v[4].y), __uint_as_float(v[4].z));
    L.dist_march = __uint_as_float(v[4].w);
    L.h.distance = __uint_as_float(v[H].x);
    L.h.normal = mk3(__uint_as_float(v[H].y), __uint_as_float(v[H].z), __uint_as_float(v[H].w));
    L.h.color = f4{__uint_as_float(v[H + 1].x), __uint_as_float(v[H + 1].y), __uint_as_float(v[H + 1].z), 0.0f};
    L.h.emittance = __uint_as_float(v[H + 1].w);
}

// LDS traffic between the lanes of ONE wave: the hardware executes a wave's LDS instructions in order; the fence keeps
// the compiler from moving accesses of different lanes to the same word across it.
DEV void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef CHUNKY_SWAP_XCHG
#define CHUNKY_SWAP_XCHG 1  // a swap trades registers and parked record with LDS exchanges, in place (measured: +2 %)
#endif
// Lanes whose path does not wait for phase X trade it for a parked path that does (as many as both sides have).
// Lanes that hold nothing any more (ST_DONE) give their place up first.  Returns the number of swaps.
//
// The state tags of the parked paths live in registers (`ptag` of lane j = tag of slot j).  Partners find each other by
// rank through two small LDS arrays — slot and tag of the r-th parked path that comes in, tag of the r-th lane path that
// goes out — written by everybody first and read after ONE fence; then each swapping lane reads its partner's eight
// 16-byte groups in one burst and writes its own over them.
template <int K, int WORDS = 8>
DEV int pool_swap(PoolLds P, LaneState& L, int& st, int& ptag, int X, int lane) {
    const bool done = st == ST_DONE;
    const bool out = phase_class(st) != X;
    const bool in = lane < K && phase_class(ptag) == X;
    const LaneMask m_done = __ballot(done), m_out = __ballot(out && !done), m_in = __ballot(in);
    const int n_done = __popcll(m_done), n_out = n_done + __popcll(m_out), n_in = __popcll(m_in);
    const int n = n_out < n_in ? n_out : n_in;
    if (n == 0) return 0;  // wave-uniform
    const int r_in = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_in >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_in, 0u));
    const LaneMask m_mine = done ? m_done : m_out;
    const int r_out = (done ? 0 : n_done) + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_mine >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_mine, 0u));
    const bool comes = in && r_in < n, goes = out && r_out < n;
    if (comes) P.list[r_in] = lane | (ptag << 8);  // slot and tag of the r-th path that comes in
    if (goes) P.tags[r_out] = st;                  // tag of the r-th path that goes out
    wave_lds_fence();
    if (comes) ptag = P.tags[r_in];
    if (goes) {
        const int e = P.list[r_out];
        const int s = e & 0xFF;
        uint4 mine[WORDS], theirs[WORDS];
        pool_pack<WORDS>(L, mine);
#if CHUNKY_SWAP_XCHG
        // one LDS exchange per 8 bytes: the parked record and the lane's registers trade places in place
#pragma unroll
        for (int g = 0; g < WORDS; g++) {
            unsigned long long* q = (unsigned long long*)&P.park[g * K + s];
            const unsigned long long lo = __hip_atomic_exchange(q, (unsigned long long)mine[g].x | ((unsigned long long)mine[g].y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            const unsigned long long hi = __hip_atomic_exchange(q + 1, (unsigned long long)mine[g].z | ((unsigned long long)mine[g].w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            theirs[g] = make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
        }
#else
#pragma unroll
        for (int g = 0; g < WORDS; g++) theirs[g] = P.park[g * K + s];
#pragma unroll
        for (int g = 0; g < WORDS; g++) P.park[g * K + s] = mine[g];
#endif
        st = e >> 8;
        pool_unpack<WORDS>(L, theirs);
        L.top_idx = -1;  // the cached top-level entry belonged to the path that left
    }
    wave_lds_fence();  // the next swap's readers (other lanes) come after these writes
    return n;
}

// Entity BVHs in render_pool: Bvh_intersect (K/bvh.h:47-109) on the aligned records of rt_device.hpp, one node per
// execution as in render_waves (inner visits and leaf visits are separate voted phases).  L.bvh_cur is a reference (inner
// record index, or a leaf reference < 0), L.bvh_top the height of the path's to-visit stack; the stacks live in LDS, one
// per PATH of the wave's pool (L.pid), so a path can be parked in the middle of a walk: entry e of stack p at
// base[e * paths + p].
struct PathStacks {
    int* base;
    int paths;  // 64 + K
};
// A walker is at a node: an inner record (bvh_cur >= 0) or a position inside a leaf (bvh_cur < 0:
// -1 - (triangle record << 6 | triangles left)).  One STEP of the walk is one inner-node visit or ONE triangle test;
// both start with the same four 16-byte reads (from one array or the other), so a step of the whole wave is a single
// round trip to memory whatever its lanes are at.  Per path the sequence of box tests, triangle tests, pushes and pops
// is K/bvh.h:47-109's.
DEV int rbvh_enter(const SceneView& S, LaneState& L, int which) {
    L.bvh_which = which;
    L.bvh_cur = which ? S.actor_root : S.world_root;
    L.bvh_top = 0;
    return ST_BVH;
}
DEV int rbvh_begin(const SceneView& S, LaneState& L) {  // closestIntersect after the octree (K/kernel.h:16-18)
    L.trace_hit = L.oct_hit;
    if (L.shadow && L.trace_hit) return ST_SHADE;
    L.bvh_dist = L.h.distance;
    return rbvh_enter(S, L, S.world_bvh_empty ? 1 : 0);
}
DEV int rbvh_finished(const SceneView& S, LaneState& L) {
    if (L.bvh_which == 0 && !S.actor_bvh_empty && !(L.shadow && L.trace_hit)) return rbvh_enter(S, L, 1);
    return ST_SHADE;
}
DEV int rbvh_pop(const SceneView& S, LaneState& L, PathStacks K) {
    if (L.bvh_top == 0) return rbvh_finished(S, L);
    L.bvh_top -= 1;
    L.bvh_cur = K.base[L.bvh_top * K.paths + L.pid];
    return ST_BVH;
}
DEV int rwalk_step(const SceneView& S, LaneState& L, PathStacks K) {
    const int cur = L.bvh_cur;
    const bool inner = cur >= 0;
    const int lref = -1 - cur, tri = lref >> 6, left = lref & 63;
    const int4* __restrict__ p = inner ? S.bvh_rec + (size_t)(unsigned)cur * 4 : S.tri_rec + (size_t)(unsigned)tri * 5;
    const int4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
    const float limit = L.shadow ? L.bvh_dist : L.h.distance;
    if (inner) {
        const int first = r0.x, second = r0.y;
        const float t1 = box_quick(as_float(r1.x), as_float(r1.y), as_float(r1.z), as_float(r1.w), as_float(r2.x), as_float(r2.y), L.o, L.inv);
        const float t2 = box_quick(as_float(r2.z), as_float(r2.w), as_float(r3.x), as_float(r3.y), as_float(r3.z), as_float(r3.w), L.o, L.inv);
        const bool miss1 = (t1 != t1) || t1 > limit;
        const bool miss2 = (t2 != t2) || t2 > limit;
        if (miss1 & miss2) return rbvh_pop(S, L, K);
        const bool go_first = !miss1 & (miss2 | (t1 < t2));  // K/bvh.h:86-103: the first child is the near one only when t1 < t2
        if (!miss1 & !miss2) {
            K.base[L.bvh_top * K.paths + L.pid] = go_first ? second : first;
            L.bvh_top += 1;
        }
        L.bvh_cur = go_first ? first : second;
        return ST_BVH;
    }
    if (left == 0) return rbvh_pop(S, L, K);  // an empty leaf
    // Triangle_intersect (K/primitives.h:368-409) on the record {e1, flags} {e2, material} {o, t1.u} {n, t1.v} {t2.u, t2.v, t3.u, t3.v}
    bool hit = false;
    {
        const int flags = r0.w;
        const f3 e1 = mk3(as_float(r0.x), as_float(r0.y), as_float(r0.z));
        const f3 e2 = mk3(as_float(r1.x), as_float(r1.y), as_float(r1.z));
        const f3 to = mk3(as_float(r2.x), as_float(r2.y), as_float(r2.z));
        const f3 pvec = cross(L.d, e2);
        const float det = dot(e1, pvec);
        const bool facing = ((flags >> 8) & 1) ? !(det > -kEps && det < kEps) : !(det > -kEps);
        if (facing) {
            const float recip = 1 / det;
            const f3 tvec = L.o - to;
            const float uu = dot(tvec, pvec) * recip;
            if (!(uu < 0 || uu > 1)) {
                const f3 qvec = cross(tvec, e1);
                const float vv = dot(L.d, qvec) * recip;
                if (!(vv < 0 || (uu + vv) > 1)) {
                    const float tt = dot(e2, qvec) * recip;
                    if (tt > kEps && tt < limit) {
                        const int4 r4 = p[4];
                        const float w = 1 - uu - vv;
                        const float u = as_float(r2.w) * uu + as_float(r4.x) * vv + as_float(r4.z) * w;
                        const float v = as_float(r3.w) * uu + as_float(r4.y) * vv + as_float(r4.w) * w;
                        Hit t = L.h;
                        if (material_sample8(S, r1.w, u, v, t)) {
                            if (!L.shadow) {
                                L.h.color = t.color;
                                L.h.emittance = t.emittance;
                                L.h.spec = t.spec;
                                L.h.normal = mk3(as_float(r3.x), as_float(r3.y), as_float(r3.z));
                                L.h.distance = tt;
                            } else {
                                L.bvh_dist = tt;
                            }
                            L.trace_hit = true;
                            hit = true;
                        }
                    }
                }
            }
        }
    }
    if (L.shadow && hit) return rbvh_finished(S, L);  // a shadow ray only needs the boolean (K/rayTracer.cl:101-106)
    if (left == 1) return rbvh_pop(S, L, K);
    L.bvh_cur = -1 - (((tri + 1) << 6) | (left - 1));
    return ST_BVH;
}

#ifndef CHUNKY_POOL_WAVES
#define CHUNKY_POOL_WAVES 6  // waves per SIMD: the march is bound by the latency of its two dependent tree reads, every wave counts
#endif
#ifndef CHUNKY_POOL_PARK
#define CHUNKY_POOL_PARK 56  // paths parked per wave (LDS: 7 x 16 + 8 bytes each; 6 x 4 waves x 56 fill 158 of 160 KB)
#endif
constexpr int kPoolPark = CHUNKY_POOL_PARK;
#ifndef CHUNKY_POOL_REFILL
#define CHUNKY_POOL_REFILL 24
#endif
#ifndef CHUNKY_NT
#define CHUNKY_NT 1  // staging array written / read with non-temporal accesses
#endif
#ifndef CHUNKY_TOP_CACHE
#define CHUNKY_TOP_CACHE 0  // 1: keep the last top-level tree entry per lane (measured: no gain — a wave waits for its slowest lane)
#endif
#ifndef CHUNKY_POOL_SPLIT
#define CHUNKY_POOL_SPLIT 0  // 1: full cubes and model blocks are voted as separate phases (measured: the model paths that
                             // wait for company shrink the pool by more than the cheaper cube phase gains)
#endif
#ifndef CHUNKY_MODEL_BATCH
#define CHUNKY_MODEL_BATCH 24
#endif
constexpr int kModelBatch = CHUNKY_MODEL_BATCH;  // model-block candidates that share one execution of their phase
constexpr int kPoolRefill = CHUNKY_POOL_REFILL;  // leave the march loop to refill once this many lanes are free and parked marchers exist
#ifndef CHUNKY_SAMPLE_BATCH
#define CHUNKY_SAMPLE_BATCH 256
#endif
constexpr int kSampleBatch = CHUNKY_SAMPLE_BATCH;  // sample indices a wave claims per atomic (measured: 64 -20 %, 128 -5 %, 512 -0.1 %, 1024 -1.3 %)

// Samples are handed out per XCD.  Each of the eight XCDs has its own L2, and workgroup b of a launch runs on XCD b % 8 (read
// from the hardware: HW_REG_XCC_ID).  The launch's samples — tile-major, so a contiguous range is a stripe of the image with
// all its passes — are cut into kXcdRanges equal ranges with a counter each; a wave starts in a range of the XCD it runs on
// and, when that has run dry (sky stripes finish long before terrain stripes), goes on with the range that follows, where its
// neighbours already are.  The waves that share an L2 — and the paths that share a wave's pool — thus work on neighbouring
// tiles of one stripe, rays of one kind, while the chip as a whole is spread over the image.  Which wave renders a sample
// has no influence on its value.  Measured on the bench (one counter: 5.55 Gsamples/s): eight stripes 5.92; going on with
// the fullest range instead of the next 5.77; tile groups dealt round-robin to the XCDs instead of stripes: groups of 1-8
// tiles -0.6 ... +0.9 %, 16-240 tiles +3 %.
#ifndef CHUNKY_XCD_QUEUES
#define CHUNKY_XCD_QUEUES 1
#endif
#ifndef CHUNKY_XCD_RANGES_PER_XCD
#define CHUNKY_XCD_RANGES_PER_XCD 1
#endif
constexpr int kXcdRanges = 8 * CHUNKY_XCD_RANGES_PER_XCD;  // at most 64
constexpr int kXcdCounters = 64;  // the range counters sit at work_counter[64 ...] (behind the claim counter and the 24 profile words)
DEV int xcd_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return (int)(x & 7u);
}
// the range a wave starts in: the XCD's ranges are consecutive, its workgroups take them in turn
DEV int xcd_first_range() { return xcd_id() * CHUNKY_XCD_RANGES_PER_XCD + (int)((blockIdx.x >> 3) % CHUNKY_XCD_RANGES_PER_XCD); }
struct XcdClaim {
    int q;               // the range this wave draws from; kXcdRanges + the ranges found empty so far, once its first one is
    unsigned next, end;  // claimed and not yet handed out: samples [next, end) of the launch
};
// samples of range x: [x * stripe, min((x + 1) * stripe, n_samples)); stripe is a multiple of kSampleBatch
DEV unsigned xcd_range_size(unsigned x, unsigned stripe, unsigned n_samples) {
    const unsigned lo = x * stripe;
    return lo >= n_samples ? 0u : (n_samples - lo < stripe ? n_samples - lo : stripe);
}
// Every lane with `need` gets a sample index: < n_samples a sample, kClaimNone nothing this time (the tail of a batch: the
// lane asks again), kClaimDone no samples left anywhere.  Convergent.
constexpr unsigned kClaimNone = 0xFFFFFFFEu, kClaimDone = 0xFFFFFFFFu;
DEV unsigned xcd_claim(int* counters, XcdClaim& c, int& tried, bool need, unsigned stripe, unsigned n_samples) {
    const unsigned long long mask = __ballot(need);
    if (mask == 0) return kClaimNone;
    const unsigned n_need = (unsigned)__popcll(mask);
    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    const unsigned rem = c.end - c.next;
    unsigned sidx = kClaimNone;
    if (need && rank < rem) sidx = c.next + rank;
    c.next += n_need < rem ? n_need : rem;
    if (n_need > rem) {
        bool got = false;
        while (tried < kXcdRanges) {
            int b = 0;
            if (need && rank == 0) b = atomicAdd(counters + c.q, kSampleBatch);
            const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane(__shfl(b, __ffsll((long long)mask) - 1));
            const unsigned n_q = xcd_range_size((unsigned)c.q, stripe, n_samples);
            if (base < n_q) {
                c.next = (unsigned)c.q * stripe + base;
                c.end = c.next + (n_q - base < (unsigned)kSampleBatch ? n_q - base : (unsigned)kSampleBatch);
                got = true;
                break;
            }
            c.q = c.q + 1 == kXcdRanges ? 0 : c.q + 1;  // this range is empty for good: on to the next
            tried += 1;
        }
        if (!got) {
            if (need && rank >= rem) sidx = kClaimDone;
            c.next = c.end = 0u;
        } else {
            const unsigned mine = c.next + (rank - rem);
            if (need && rank >= rem && mine < c.end) sidx = mine;
            c.next = c.next + (n_need - rem) < c.end ? c.next + (n_need - rem) : c.end;
        }
    }
    return sidx;
}

// stats (STATS = true), same layout as render_waves: [0..8] executions / lanes / cycles of MARCH, BLOCK (and the entity-BVH
// walk), SHADE; [9..11] wave lifetimes; [12] swap rounds, [13] paths swapped; [14..] parts of SHADE.
#ifndef CHUNKY_W_WALK
#define CHUNKY_W_WALK 1
#endif
#ifndef CHUNKY_WALK_LEAVE
#define CHUNKY_WALK_LEAVE 24
#endif
constexpr int kWalkLeave = CHUNKY_WALK_LEAVE;
#ifndef CHUNKY_W_BVH
#define CHUNKY_W_BVH 1
#endif
#ifndef CHUNKY_W_LEAF
#define CHUNKY_W_LEAF 1
#endif
constexpr int kWWalk = CHUNKY_W_WALK, kWBvh = CHUNKY_W_BVH, kWLeaf = CHUNKY_W_LEAF;
#ifndef CHUNKY_POOL_BVH_WAVES
#define CHUNKY_POOL_BVH_WAVES 5
#endif
template <int TREE, int K, bool STATS, bool BVH = false, bool EXT = false>
__global__ void __launch_bounds__(256, ((STATS || EXT) ? 4 : (BVH ? CHUNKY_POOL_BVH_WAVES : CHUNKY_POOL_WAVES))) render_pool(WaveArgs unused_by_name) {
    constexpr int WORDS = EXT ? 9 : (BVH ? 8 : 7);  // 16-byte words of a parked path (pool_pack)
    constexpr int END = BVH ? ST_TRACED : ST_SHADE;  // where a lane goes when the octree part of a trace ends
    extern __shared__ int lds[];
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    PoolLds P{nullptr, nullptr, nullptr};
    PathStacks stacks{nullptr, 64 + K};
    {
        // per wave: K parked records, their tag / list scratch, then (BVH) one to-visit stack per path of the pool
        const unsigned depth = BVH ? fresh_args()->stack_bytes : 0u;  // entries per stack
        char* base = (char*)lds + wave * (K * 16 * WORDS + K * 8 + (64 + K) * depth * 4);
        P.park = (uint4*)base;
        P.tags = (int*)(base + K * 16 * WORDS);
        P.list = P.tags + K;
        stacks.base = (int*)(base + K * 16 * WORDS + K * 8);
        if (BVH && lane < K) P.park[lane] = make_uint4(0u, 0u, (unsigned)(64 + lane) << 16, 0u);  // the parked slots' stack ids
    }
#if CHUNKY_LDS_TOP
    if (TREE == -1 && K == 32 && !BVH && !EXT) {
        const SceneView S0 = arg_copy(&fresh_args()->S);
        const int n_top = 1 << (3 * S0.wide_bits[0]);
        for (int i = (int)threadIdx.x; i < n_top && i < 4096; i += 256) lds[chunky_lds_top_offset + i] = (int)S0.wide[i];
        __syncthreads();
    }
#endif
    LdsStack stack{lds, 0};  // render_waves' per-lane stacks are not used here
    LaneState L;
    L.h.material = 0;
    L.h.normal = mk3(0, 0, 0);
    L.h.color = f4{0, 0, 0, 0};
    L.h.emittance = 0;
    L.h.distance = 0;
    L.cand_data = 0;
    L.cand_level = 0;
    L.pass = 0;
    L.gid = -1;
    L.sidx = 0;
    L.top_idx = -1;
    L.top_e = 0;
    L.pid = lane;
    L.tkind = 0;
    L.after_nee = false;
    L.pend = mk3(0, 0, 0);
    L.h.spec = 0;
    L.mean = mk3(0, 0, 0);
    L.slot = 0;
    L.serial = 0;
    L.steps = 0;
    L.rng = 0;
    L.depth = 0;
    L.shadow = false;
    L.dist_march = 0;
    L.radiance = mk3(0, 0, 0);
    L.throughput = mk3(0, 0, 0);
    L.o = L.d = L.inv = L.far = mk3(0, 0, 0);
    L.oct_hit = false;
    L.trace_hit = false;
    L.bvh_cur = L.bvh_top = L.bvh_which = L.bvh_head = 0;
    L.bvh_base = nullptr;
    L.bvh_dist = 0;
    unsigned long long prof[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long swap_rounds = 0, swapped = 0;
    PartTimers parts{{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0};
    unsigned long long t_begin = 0;
    if (STATS) t_begin = __builtin_amdgcn_s_memtime();
    PixelPool pool{0, 0};
    XcdClaim claim{xcd_first_range(), 0u, 0u};
    int ranges_tried = 0;  // ranges this wave has found empty
    int st = ST_FRESH;
    int ptag = lane < K ? ST_FRESH : ST_DONE;
    wave_lds_fence();
    for (;;) {
        if (BVH && __ballot(st == ST_TRACED)) {  // octree part of some traces just ended: entity BVHs next
            const SceneView S = arg_copy(&fresh_args()->S);
            if (st == ST_TRACED) st = rbvh_begin(S, L);
        }
        // the pool's census: paths waiting for each phase, in lanes and parked
        const int c_march = count_lanes(st == ST_MARCH) + count_lanes(ptag == ST_MARCH);
        const int c_block = count_lanes(st == ST_BLOCK) + count_lanes(ptag == ST_BLOCK);
        const int c_model = CHUNKY_POOL_SPLIT ? count_lanes(st == ST_MODEL) + count_lanes(ptag == ST_MODEL) : 0;
        const int c_shade = count_lanes(st == ST_SHADE || st == ST_FRESH) + count_lanes(ptag == ST_SHADE || ptag == ST_FRESH);
        const int c_bvh = BVH ? count_lanes(st == ST_BVH) + count_lanes(ptag == ST_BVH) : 0;
        const int c_leaf = BVH ? count_lanes(st == ST_LEAF) + count_lanes(ptag == ST_LEAF) : 0;
        if ((c_march | c_block | c_model | c_shade | c_bvh | c_leaf) == 0) break;  // every lane and every slot is ST_DONE
        // at most 64 paths run at once; among phases that can fill the wave SHADE and BLOCK go first (they feed the march).
        // Model blocks (slabs, plants: loops over boxes / quads, several dependent reads each) are rare and slow: they
        // wait until a fair number of them can share one execution, or nothing else is left.
        const int v_march = (c_march < 64 ? c_march : 64) * kWMarch, v_block = (c_block < 64 ? c_block : 64) * kWBlock,
                  v_shade = (c_shade < 64 ? c_shade : 64) * kWShade;
        int X = (v_shade >= v_block && v_shade >= v_march) ? 2 : (v_block >= v_march ? 1 : 0);
        int v_best = X == 2 ? v_shade : (X == 1 ? v_block : v_march);
        if (c_model >= kModelBatch || (c_model > 0 && v_best == 0)) X = 4;
        if (BVH) {  // the walk through the entity BVHs (inner-node and leaf visits together) is one class of the pool
            const int c_walk = c_bvh + c_leaf;
            const int v_walk = (c_walk < 64 ? c_walk : 64) * kWWalk;
            if (v_walk > v_best) { X = 5; v_best = v_walk; }
        }
        unsigned long long t0 = 0;
        if (STATS) t0 = __builtin_amdgcn_s_memtime();
        if (K > 0) {
            const int n = pool_swap<K, WORDS>(P, L, st, ptag, X, lane);
            if (STATS && n) {
                swap_rounds += 1;
                swapped += (unsigned long long)n;
            }
        }
        if (STATS) {  // parts 4, 5, 6 of the profile: cycles in swaps, loop iterations, entries into the march loop
            parts.t[PT_FOLD] += __builtin_amdgcn_s_memtime() - t0;
            parts.t[PT_OPEN] += 1;
            parts.t[PT_HANDOUT] += X == 0 ? 1 : 0;
        }
        int n_exec = 0;
        if (X == 0) {
            WaveArgPtr A = fresh_args();
            const SceneView Sm = arg_copy(&A->S);
            const RenderOpts Om = arg_copy(&A->O);
            const LaneMask entered = __ballot(st == ST_MARCH);
            int nm = __popcll(entered);
            n_exec = nm;
            const int parked_march = c_march - nm;  // marchers still parked after the swap
            // The wave stays in the march while it runs fuller than anything else could: lanes that leave join the paths
            // waiting for BLOCK or SHADE (`other` of them already), so it leaves once nm would drop below the larger of those
            // crowds — at worst every leaver joins it: nm < other + (n0 - nm) — or once enough lanes are free for a refill
            // from the parked marchers.  One bound, fixed on entry: the loop's bookkeeping is one popcount and one compare.
            const int other_b = c_block < 64 ? c_block : 64, other_s = c_shade < 64 ? c_shade : 64;
            int other = other_b > other_s ? other_b : other_s;
            // (with entity BVHs the walkers are not counted: they are the pool's standing crowd and wait in any case)
            int stay = (other + nm + 1) >> 1;
            if (K > 0 && parked_march >= kPoolRefill && stay < 65 - kPoolRefill) stay = 65 - kPoolRefill;
            if (stay < 1) stay = 1;
            LaneMask marching = entered, to_block = 0, to_model = 0;
            const LaneMask far_masks[3] = {__ballot(L.inv.x > 0), __ballot(L.inv.y > 0), __ballot(L.inv.z > 0)};
            int data, level;
            do {
                if (STATS) {
                    prof[0] += 1;
                    prof[1] += (unsigned long long)nm;
                }
                LaneMask cand, live, model;
                march_step<TREE>(Sm, Om, L, marching, cand, live, data, level, CHUNKY_POOL_SPLIT ? &model : nullptr,
                                 CHUNKY_TOP_CACHE && TREE >= 16, far_masks);
                to_block |= cand;
                if (CHUNKY_POOL_SPLIT) to_model |= model;
                marching = live & ~cand;
                nm = __popcll(marching);
            } while (nm >= stay);
            const bool found = in_mask(to_block);
            L.cand_data = found ? data : L.cand_data;
            L.cand_level = found ? level : L.cand_level;
            st = found ? ((CHUNKY_POOL_SPLIT && in_mask(to_model)) ? ST_MODEL : ST_BLOCK) : (in_mask(entered & ~marching & ~to_block) ? END : st);
            if (STATS) {
                prof[0] -= 1;
                prof[1] -= (unsigned long long)n_exec;
            }
        } else if (X == 1) {
            n_exec = count_lanes(st == ST_BLOCK);
            const SceneView S = arg_copy(&fresh_args()->S);
            if (st == ST_BLOCK) st = block_phase<TREE, END, CHUNKY_POOL_SPLIT != 0, false>(S, L);
        } else if (X == 4) {
            n_exec = count_lanes(st == ST_MODEL);
            const SceneView S = arg_copy(&fresh_args()->S);
            if (st == ST_MODEL) st = block_phase<TREE, END, false, false>(S, L);
        } else if (BVH && X == 5) {
            // The walk: a path alternates between inner-node visits and leaf visits every few nodes, so the two are voted
            // here, in a loop of their own (two counts per round) instead of through the pool's census.  The wave stays while
            // the walkers outnumber what waits elsewhere, or until enough lanes are free for a refill from parked walkers.
            const SceneView S = arg_copy(&fresh_args()->S);
            int nw = count_lanes(st == ST_BVH);
            n_exec = nw;
            const int parked_walk = c_bvh + c_leaf - nw;
            // The walk is nine tenths of the work in a scene with entities and everything else is cheap beside it: the other
            // phases are served as soon as a small crowd waits for them (vote weight kWWalk against 4), so that the pool stays
            // full of walkers; the wave leaves the walk when kWalkLeave lanes have finished theirs (they wait for SHADE now),
            // or when that many are free and parked walkers can take their place.
            int stay = nw - kWalkLeave + 1;
            if (K > 0 && parked_walk > 0) {
                const int refill = parked_walk < kWalkLeave ? parked_walk : kWalkLeave;
                if (stay < 65 - refill) stay = 65 - refill;
            }
            if (stay < 1) stay = 1;
            do {
                if (STATS) {
                    prof[3] += 1;
                    prof[4] += (unsigned long long)nw;
                }
                if (st == ST_BVH) st = rwalk_step(S, L, stacks);
                nw = count_lanes(st == ST_BVH);
            } while (nw >= stay);
            if (STATS) {
                prof[3] -= 1;
                prof[4] -= (unsigned long long)n_exec;
            }
        } else {
            n_exec = count_lanes(st == ST_SHADE || st == ST_FRESH);
            WaveArgPtr A = fresh_args();
            const SceneView S = arg_copy(&A->S);
            const RenderOpts O = arg_copy(&A->O);
            if (st == ST_SHADE) st = EXT ? shade_phase_ext<TREE, BVH>(S, O, L) : shade_phase<TREE, BVH, STATS>(S, O, L, stack, &parts);
            part_begin<STATS>(&parts);
            if (st == ST_NEXT) {  // the path is finished: its radiance waits in the staging array for fold_kernel
                // streamed past the caches (nt): written once, read once by fold_kernel; the L2 stays with the tree
                float* __restrict__ out = A->staging + 3 * (size_t)(unsigned)L.sidx;
#if CHUNKY_NT
                __builtin_nontemporal_store(L.radiance.x, out);
                __builtin_nontemporal_store(L.radiance.y, out + 1);
                __builtin_nontemporal_store(L.radiance.z, out + 2);
#else
                *(f3*)out = L.radiance;
#endif
                st = ST_FRESH;
            }
            part_end<STATS>(&parts, PT_DEPOSIT);
            // ---- new samples (K/rayTracer.cl:55-91).  Sample index = (tile of kSampleTile pixel slots, pass, slot in tile): a
            //      tile gets all its passes before the next tile starts, so the paths in flight on the whole GPU cover a few
            //      thousand neighbouring pixels — a part of the scene that stays in the 4 MB L2s (pass-major order spread
            //      them over a third of the image: L2 hit rate 91 %, 66 GB of fabric reads per launch instead of 4) ----
            const bool need = st == ST_FRESH;
#if CHUNKY_XCD_QUEUES
            const unsigned sidx = xcd_claim(A->Q.next + kXcdCounters, claim, ranges_tried, need, A->xcd_stripe, A->n_samples);  // convergent
#else
            const unsigned sidx = (unsigned)claim_slot<kSampleBatch>(arg_copy(&A->Q), pool, need);  // convergent
#endif
            if (need && sidx != kClaimNone) {
                const unsigned n_samples = A->n_samples;
                if (sidx >= n_samples) {
                    st = ST_DONE;
                } else {
                    const CameraView C = arg_copy(&A->C);
                    const ShardView T = arg_copy(&A->T);
                    // sidx = ((tile * sub-blocks per tile + sub-block) * passes + pass) * kSubBlock + slot in the sub-block
                    const unsigned per_sub = (unsigned)A->P.n * (unsigned)kSubBlock;
                    const unsigned sub = sidx / per_sub, rem = sidx - sub * per_sub;  // sub = tile * (kSampleTile / kSubBlock) + sub-block
                    const unsigned pass = rem / (unsigned)kSubBlock;
                    const int slot = (int)(sub * (unsigned)kSubBlock + (rem & (unsigned)(kSubBlock - 1)));
                    const int gid = pool_slot_gid(T, C.width, C.height, slot);
                    if (gid < C.width * C.height) {  // else: a padding slot, nothing to render (the lane claims again)
                        unsigned rng = (unsigned)A->P.seed[pass] + (unsigned)gid;
                        rt_pcg_next(&rng);
                        const RayOD pr = primary_ray(C, gid, rng, false);
                        L.sidx = (int)sidx;
                        L.rng = rng;
                        L.o = pr.o;
                        L.d = pr.d;
                        L.radiance = mk3(0, 0, 0);
                        L.throughput = mk3(1, 1, 1);
                        L.depth = 0;
                        L.shadow = false;
                        L.tkind = 0;
                        L.after_nee = false;
                        L.h.distance = rt_inf();
                        st = ST_SETUP;
                    }
                }
            }
            part_end<STATS>(&parts, PT_NEWSAMPLE);
            if (st == ST_SETUP) st = trace_setup<END, false>(S, L);
            part_end<STATS>(&parts, PT_SETUP);
        }
        if (STATS) {
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (X == k) {
                    prof[3 * k] += 1;
                    prof[3 * k + 1] += (unsigned long long)n_exec;
                    prof[3 * k + 2] += dt;
                }
            if (X == 5) {  // the entity-BVH walk is profiled with BLOCK
                prof[3] += 1;
                prof[4] += (unsigned long long)n_exec;
                prof[5] += dt;
            }
            if (X == 4) {  // parts 8, 9: lanes served and cycles of the model phase
                parts.t[8] += (unsigned long long)n_exec;
                parts.t[9] += dt;
            }
        }
    }
    if (STATS && lane == 0) {
        unsigned long long* stats = fresh_args()->stats;
        for (int k = 0; k < 9; k++) atomicAdd(&stats[k], prof[k]);
        const unsigned long long life = __builtin_amdgcn_s_memtime() - t_begin;
        atomicAdd(&stats[9], life);
        atomicMax(&stats[10], life);
        atomicAdd(&stats[11], 1ull);
        atomicAdd(&stats[12], swap_rounds);
        atomicAdd(&stats[13], swapped);
        for (int k = 0; k < 10; k++) atomicAdd(&stats[14 + k], parts.t[k]);
    }
}

// The running mean of K/rayTracer.cl:109-112 over the staged samples of a launch, strictly in pass order: one thread per
// pixel and channel, reads coalesced across pixels ([pass][slot][3]).
#ifndef CHUNKY_FOLD_NT
#define CHUNKY_FOLD_NT 0  // with small sub-blocks a thread's consecutive passes share cache lines: plain loads keep them
#endif
__global__ void __launch_bounds__(256) fold_kernel(const float* __restrict__ staging, float* __restrict__ res, ShardView T, int n_pixels,
                                                    int width, long long n_slots, int n_passes, int first_spp) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= 3ll * n_slots) return;
    const int slot = (int)(t / 3), c = (int)(t - 3ll * slot);
    const int gid = pool_slot_gid(T, width, n_pixels / width, slot);
    if (gid >= n_pixels) return;
    float mean = res[3 * (size_t)gid + c];
    // sample (sub-block, pass, i) sits at index (sub-block * n_passes + pass) * kSubBlock + i
    const size_t sub = (size_t)slot / kSubBlock, i = (size_t)slot % kSubBlock;
    const float* p = staging + 3 * (sub * (size_t)n_passes * kSubBlock + i) + c;
#pragma unroll 8
    for (int k = 0; k < n_passes; k++) {
        const int spp = first_spp + k;
#if CHUNKY_NT && CHUNKY_FOLD_NT
        mean = (mean * (float)spp + __builtin_nontemporal_load(p + (size_t)k * (3 * kSubBlock))) / (float)(spp + 1);
#else
        mean = (mean * (float)spp + p[(size_t)k * (3 * kSubBlock)]) / (float)(spp + 1);
#endif
    }
    res[3 * (size_t)gid + c] = mean;
}

// Read-back exchange of a multi-GPU group (capi.hip group_gather): the pixels of the slots of shard T, 3 floats per slot
// in slot order, out of the image (PACK) or back into one.  Slot -> pixel is pool_slot_gid, as in the kernels that rendered
// them; padding slots carry nothing.
template <bool PACK>
__global__ void __launch_bounds__(256) gather_kernel(ShardView T, int width, int height, float* __restrict__ fb, float* __restrict__ packed) {
    const int slot = (int)(blockIdx.x * 256 + threadIdx.x);
    if (slot >= T.n_local) return;
    const int gid = pool_slot_gid(T, width, height, slot);
    if (gid >= width * height) return;
    float* a = fb + 3 * (size_t)gid;
    float* b = packed + 3 * (size_t)slot;
    if (PACK) {
        b[0] = a[0]; b[1] = a[1]; b[2] = a[2];
    } else {
        a[0] = b[0]; a[1] = b[1]; a[2] = b[2];
    }
}

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

// ------------------------------------------------------------------------------------ launchers
// variant bit 0 set = force the reference-layout octree walk (K/octree.h:81-89 as written);
// variant bit 1 set = one lane per path for the whole launch (render_lanes) instead of render_waves
static bool use_wide(int variant, const SceneView& S) { return S.wide != nullptr && !(variant & 1); }

static size_t stack_lds_bytes(const SceneView& S, int block) {
    bool need = !S.world_bvh_empty || !S.actor_bvh_empty;
    int entries = S.bvh_stack_entries > 0 && S.bvh_stack_entries < kBvhStackEntries ? S.bvh_stack_entries : kBvhStackEntries;
    return need ? (size_t)entries * block * sizeof(int) : 0;
}

// render_pool + fold_kernel.  variant bits 6-7 pick the parked paths per wave: 0 = 56 (default), 1 = none, 2 = 32, 3 = 64.
static hipError_t launch_pool(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, const ShardView& T,
                              const PassSeeds& P, float* res, int* work_counter, hipStream_t stream, KernelChoice* chosen,
                              float* staging) {
    const int block = 256;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipGetLastError();
        n_cu = prop.multiProcessorCount;
    }
    if (T.n_local <= 0 || P.n <= 0) return hipSuccess;
    const bool stats = (variant & 4) != 0;
    int tree = 0;
    if (use_wide(variant, S)) {
        tree = S.wide_nlev <= 4 ? 16 + S.wide_nlev - 1 : -1;
        for (int i = 1; i < S.wide_nlev; i++)
            if (S.wide_bits[i] != 3) tree = -1;
    }
    const bool bvh = !S.world_bvh_empty || !S.actor_bvh_empty;
    int depth = 0;  // entries per to-visit stack (the reference reserves 64, K/bvh.h:38; here: the height of the taller BVH + 1)
    if (bvh) depth = S.bvh_stack_entries > 0 && S.bvh_stack_entries < kBvhStackEntries ? S.bvh_stack_entries : kBvhStackEntries;
    int park = kPoolPark;
    switch ((variant >> 6) & 3) {
        case 1: park = 0; break;
        case 2: park = 32; break;
        case 3: park = 64; break;
        default: break;
    }
    if (const char* e = getenv("CHUNKY_DEBUG_POOL")) park = atoi(e);
    typedef void (*Kernel)(WaveArgs);
    Kernel k;
    const bool ext = opts_extended(O);  // EXPERIMENTAL light-transport options: their own instantiations
    int words = bvh ? 8 : 7;
    if (ext) {
        if (tree != 17 && tree != 18) tree = -1;
        words = 9;
        park = bvh ? 16 : 32;
        if (bvh)
            k = tree == 17 ? render_pool<17, 16, false, true, true> : (tree == 18 ? render_pool<18, 16, false, true, true> : render_pool<-1, 16, false, true, true>);
        else
            k = tree == 17 ? render_pool<17, 32, false, false, true> : (tree == 18 ? render_pool<18, 32, false, false, true> : render_pool<-1, 32, false, false, true>);
    } else if (bvh) {
        // every path of the pool owns a to-visit stack in LDS: 32 parked paths when five workgroups per CU still fit, else 16
        if (tree != 17 && tree != 18) tree = -1;
        park = 5 * 4 * (32 * 136 + (64 + 32) * depth * 4) <= 160 * 1024 ? 32 : 16;
        if (const char* e = getenv("CHUNKY_DEBUG_POOL")) park = atoi(e) >= 32 ? 32 : 16;
        if (stats) {
            park = 16;
            k = tree == 17 ? render_pool<17, 16, true, true> : (tree == 18 ? render_pool<18, 16, true, true> : render_pool<-1, 16, true, true>);
        } else if (park == 32) {
            k = tree == 17 ? render_pool<17, 32, false, true> : (tree == 18 ? render_pool<18, 32, false, true> : render_pool<-1, 32, false, true>);
        } else {
            k = tree == 17 ? render_pool<17, 16, false, true> : (tree == 18 ? render_pool<18, 16, false, true> : render_pool<-1, 16, false, true>);
        }
    } else if (stats) {
        if (tree != 17) tree = -1;
        park = kPoolPark;
        k = tree == 17 ? render_pool<17, kPoolPark, true> : render_pool<-1, kPoolPark, true>;
    } else if (park != kPoolPark) {
        if (tree != 17) tree = -1;
        if (park == 0) k = tree == 17 ? render_pool<17, 0, false> : render_pool<-1, 0, false>;
        else if (park == 32) k = tree == 17 ? render_pool<17, 32, false> : render_pool<-1, 32, false>;
        else { park = 64; k = tree == 17 ? render_pool<17, 64, false> : render_pool<-1, 64, false>; }
    } else {
        switch (tree) {
            case 0: k = render_pool<0, kPoolPark, false>; break;
            case 16: k = render_pool<16, kPoolPark, false>; break;
            case 17: k = render_pool<17, kPoolPark, false>; break;
            case 18: k = render_pool<18, kPoolPark, false>; break;
            case 19: k = render_pool<19, kPoolPark, false>; break;
            default: tree = -1; k = render_pool<-1, kPoolPark, false>; break;
        }
    }
    size_t lds = (size_t)(block / 64) * (size_t)(park * 16 * words + park * 8 + (64 + park) * depth * 4);
#if CHUNKY_LDS_TOP
    if (tree == -1 && park == 32 && !bvh && !ext) lds += 4096 * 4;
#endif
    int occ = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, block, lds);
    if (e != hipSuccess) return e;
    const int bpc = occ > 0 ? occ : 1;
    const long long n_tiles = pool_tiles(T, C.width, C.height);
    const long long n_samples = n_tiles * kSampleTile * P.n;  // tiles at the image's edges are padded
    // a wave keeps 64 + park paths in flight: no more workgroups than the samples can feed
    const long long want = (n_samples + (long long)(block / 64) * (64 + park) - 1) / ((long long)(block / 64) * (64 + park));
    int grid = n_cu * bpc;
    if ((long long)grid > want) grid = (int)want;
    if (chosen) *chosen = KernelChoice{tree, 1, bvh ? 1 : 0, grid, park, ext ? 1 : 0};
    e = hipMemsetAsync(work_counter, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(work_counter + kXcdCounters, 0, kXcdRanges * sizeof(int), stream);  // the per-XCD sample ranges (xcd_claim)
    if (e != hipSuccess) return e;
    WaveArgs A{S, C, O, T, P, WorkQueue{work_counter}, res, (unsigned long long*)(work_counter + 2), (unsigned)depth, staging, (unsigned)n_samples,
               (unsigned)((n_tiles + kXcdRanges - 1) / kXcdRanges * kSampleTile * P.n)};
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, stream, A);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long long threads = 3ll * n_tiles * kSampleTile;
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, (const float*)staging, res, T,
                       C.width * C.height, C.width, n_tiles * kSampleTile, P.n, P.first_spp);
    return hipGetLastError();
}

hipError_t launch_render(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, const ShardView& T,
                         const PassSeeds& P, float* res, int* work_counter, hipStream_t stream, KernelChoice* chosen,
                         float* staging) {
    const bool any_bvh = !S.world_bvh_empty || !S.actor_bvh_empty;
    // render_pool: always without entity BVHs; with them when they could be re-laid out (rt_device.hpp bvh_rec / tri_rec)
    // (its 7-word parked record counts march steps in 16 bits: a larger draw depth runs render_waves)
    const bool steps_fit = any_bvh || opts_extended(O) || O.draw_depth <= 65535;
    if (!(variant & 2) && !(variant & 8) && work_counter && staging && steps_fit && (!any_bvh || (S.bvh_rec && S.tri_rec && S.mat8 && !(variant & 1))))
        return launch_pool(variant, S, C, O, T, P, res, work_counter, stream, chosen, staging);
    if (T.world != 1 && T.tile == 0) return hipErrorNotSupported;  // shards of 16 x 16 blocks exist in render_pool only
    if (!(variant & 2) && work_counter) {
        const bool stats = (variant & 4) != 0;  // work_counter[2..] = 9 x u64 phase profile
        // wave-scheduled persistent kernel: one resident grid, lanes pull pixels from a counter
        const int block = 256;
        static int n_cu = 0;
        const bool wide = use_wide(variant, S);
        size_t lds = stack_lds_bytes(S, block);
        if (const char* pad = getenv("CHUNKY_DEBUG_LDS_PAD")) lds += (size_t)atoi(pad);  // occupancy experiments
        if (n_cu == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipGetLastError();
            n_cu = prop.multiProcessorCount;
        }
        // tree kind: 0 reference layout, 16 + n = dense top over n <= 3 levels of 3 bits, -1 any other wide split
        int tree = 0;
        if (wide) {
            tree = S.wide_nlev <= 4 ? 16 + S.wide_nlev - 1 : -1;
            for (int i = 1; i < S.wide_nlev; i++)
                if (S.wide_bits[i] != 3) tree = -1;
        }
        typedef void (*Kernel)(WaveArgs);
        // lanes per pixel (see next_sample): 8; 16 or 32 when this GPU owns few pixels (multi-GPU tile split: the fewer
        // pixels per group, the longer the tail of the launch); CHUNKY_DEBUG_GROUP=1|8|16|32 for experiments
        const size_t stack = lds;
        int group = T.n_local < (3 << 17) ? 32 : (T.n_local < (3 << 18) ? 16 : 8);
        while (group > 8 && P.n < 2 * group) group /= 2;  // a group needs a few passes per lane to stay busy
        if (P.n < 2 * group) group = 1;
        if (const char* g = getenv("CHUNKY_DEBUG_GROUP")) group = atoi(g);
        switch ((variant >> 4) & 3) {  // variant bits 4-5 force the group size (tests cover all three)
            case 1: group = 1; break;
            case 2: group = 8; break;
            case 3: group = 16; break;
            default: break;
        }
        if (group != 1 && group != 8 && group != 16 && group != 32) group = 8;
        const bool has_bvh = !S.world_bvh_empty || !S.actor_bvh_empty;
        if (group == 32 && (has_bvh || stats)) group = 16;
        // the instantiations that exist: every tree form for the plain kernels at G = 1 and 8; the dense-top forms of
        // the two benchmark depths (9 and 10) for the rest; `tree` becomes the form actually used
        Kernel k;
        if (has_bvh && !stats && group > 1) {
            if (tree != 17 && tree != 18) tree = -1;
            if (group == 16)
                k = tree == 17 ? render_waves<17, false, 16, true> : (tree == 18 ? render_waves<18, false, 16, true> : render_waves<-1, false, 16, true>);
            else
                k = tree == 17 ? render_waves<17, false, 8, true> : (tree == 18 ? render_waves<18, false, 8, true> : render_waves<-1, false, 8, true>);
        } else if (has_bvh) {
            tree = -1;
            group = 1;
            k = stats ? render_waves<-1, true, 1, true> : render_waves<-1, false, 1, true>;
        } else if (stats) {
            if (tree != 17) tree = -1;
            if (group != 1) group = 8;
            k = tree == 17 ? (group == 1 ? render_waves<17, true, 1> : render_waves<17, true, 8>)
                           : (group == 1 ? render_waves<-1, true, 1> : render_waves<-1, true, 8>);
        } else if (group == 1) {
            switch (tree) {
                case 0: k = render_waves<0, false, 1>; break;
                case 16: k = render_waves<16, false, 1>; break;
                case 17: k = render_waves<17, false, 1>; break;
                case 18: k = render_waves<18, false, 1>; break;
                case 19: k = render_waves<19, false, 1>; break;
                default: tree = -1; k = render_waves<-1, false, 1>; break;
            }
        } else if (group == 32) {
            if (tree != 17 && tree != 18) tree = -1;
            k = tree == 17 ? render_waves<17, false, 32> : (tree == 18 ? render_waves<18, false, 32> : render_waves<-1, false, 32>);
        } else if (group == 16) {
            if (tree != 17 && tree != 18) tree = -1;
            k = tree == 17 ? render_waves<17, false, 16> : (tree == 18 ? render_waves<18, false, 16> : render_waves<-1, false, 16>);
        } else {
            switch (tree) {
                case 0: k = render_waves<0, false, 8>; break;
                case 16: k = render_waves<16, false, 8>; break;
                case 17: k = render_waves<17, false, 8>; break;
                case 18: k = render_waves<18, false, 8>; break;
                case 19: k = render_waves<19, false, 8>; break;
                default: tree = -1; k = render_waves<-1, false, 8>; break;
            }
        }
        if (group > 1) lds += (size_t)(block / group) * (2 * ring_size(group) * 16 + 64);
        int occ = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, block, lds);
        if (e != hipSuccess) return e;
        int bpc = occ > 0 ? occ : 1;
        // a pixel keeps `group` lanes busy (one pass each), so that many lanes per pixel are worth launching
        const long long want = ((long long)T.n_local * group + block - 1) / block;
        int grid = n_cu * bpc;
        if ((long long)grid > want) grid = (int)want;
        if (grid <= 0 || P.n <= 0) return hipSuccess;
        if (chosen) *chosen = KernelChoice{tree, group, has_bvh ? 1 : 0, grid, -1, 0};
        e = hipMemsetAsync(work_counter, 0, sizeof(int), stream);
        if (e != hipSuccess) return e;
        WaveArgs A{S, C, O, T, P, WorkQueue{work_counter}, res, (unsigned long long*)(work_counter + 2), (unsigned)stack, nullptr, 0u};
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, stream, A);
        return hipGetLastError();
    }
    const int block = 256;
    int grid = (T.n_local + block - 1) / block;
    if (grid <= 0 || P.n <= 0) return hipSuccess;
    if (chosen) *chosen = KernelChoice{use_wide(variant, S) ? -1 : 0, 0, any_bvh ? 1 : 0, grid, -1, 0};
    if (use_wide(variant, S))
        hipLaunchKernelGGL(render_lanes<-1>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O, T, P, res);
    else
        hipLaunchKernelGGL(render_lanes<0>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O, T, P, res);
    return hipGetLastError();
}

hipError_t launch_gather(bool pack, const ShardView& T, int width, int height, float* fb, float* packed, hipStream_t stream) {
    if (T.n_local <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((T.n_local + 255) / 256);
    if (pack)
        hipLaunchKernelGGL(gather_kernel<true>, dim3(grid), dim3(256), 0, stream, T, width, height, fb, packed);
    else
        hipLaunchKernelGGL(gather_kernel<false>, dim3(grid), dim3(256), 0, stream, T, width, height, fb, packed);
    return hipGetLastError();
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
    int t = 0;
    if (tree && S.wide) {
        t = S.wide_nlev <= 4 ? 16 + S.wide_nlev - 1 : -1;
        for (int i = 1; i < S.wide_nlev; i++)
            if (S.wide_bits[i] != 3) t = -1;
    }
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
