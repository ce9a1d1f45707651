// path_state.hpp — device code shared by the render kernels: the leaf lookup and the octree march, one path's state
// (LaneState) and its phases (march step, block test, shade), the launch-argument block.  Included by render_pool.hip,
// render_fallback.hip and aux_kernels.hip; compiled with -ffp-contract=off (see rt_device.hpp).
#pragma once
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
// TREE = 0 walks the reference layout from the root, one bit per level; TREE = -1 walks the wide
// re-layout (widetree.hpp) with per-level bit counts from the scene view; TREE = 16 + n walks the
// default split of widetree.cpp — one dense top node of S.wide_bits[0] bits per axis (a wave-uniform
// value) over n levels of 3 bits (compile-time shifts): two dependent reads per cell for a 512^3 world
// where the reference descends nine — same (data, level) for every cell.
// `kind`: 0 full cube, 1 other model, 2 or 3 cannot be hit (air, invisible, ANY_TYPE); the reference
// layout carries no kinds, so every non-air leaf reports 1 there (the general test handles all types).
template <int TREE>
DEV void leaf_lookup(const SceneView& S, int bx, int by, int bz, int& data, int& level, int& kind, bool inside = true, int* entry = nullptr) {
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
        // (a block pointer beyond the palette — hostile data; the reference's read there is undefined — never intersects)
        kind = (data == 0 || data == kAnyType || (unsigned)data + 1u >= (unsigned)S.n_block_ints) ? 2 : 1;
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
                e = *(const int*)((const char*)tree + (idx << 2));
            }
#pragma unroll
            for (int i = 0; i < N3; i++) {
                if (e >= 0) {
                    // index in the 8^3 node = x:3 y:3 z:3 of the cell's bits [sh, sh + 3): each field shifted into place, then two
                    // v_and_or_b32 (the compiler's form is three v_and and a v_or3)
                    const int sh = 3 * (N3 - 1 - i);
                    const unsigned xs = sh <= 6 ? (unsigned)bx << (6 - sh) : (unsigned)bx >> (sh - 6);
                    const unsigned ys = sh <= 3 ? (unsigned)by << (3 - sh) : (unsigned)by >> (sh - 3);
                    unsigned idx = ((unsigned)bz >> sh) & 7u;
                    asm("v_and_or_b32 %0, %1, 56, %0" : "+v"(idx) : "v"(ys));
                    asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(idx) : "v"(xs), "s"(0x1c0u));  // (no literal operands in VOP3 on gfx9: the mask rides in an SGPR)
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
                e = (int)tree[(unsigned)e + ((((ix << b) | iy) << b) | iz)];  // unsigned: 32-bit offset off an SGPR base
            }
        }
        // two field extractions and ONE compare: the builder's annotation pass (widetree.cpp) has set the no-hit bit (30) on air,
        // on pointers outside the block palette and on ANY_TYPE, so "a leaf that can be hit" is entry < 0xC0000000 as unsigned;
        // `data` means something for such leaves only (callers only ever ask kind < 2)
        level = (e >> 26) & 15;
        kind = (unsigned)e >= 0xC0000000u ? 2 : 0;
        data = (int)((unsigned)e & 0x1FFFFFFu);
        if (entry) *entry = e;  // the whole entry: bit 25 = a model block (types 2, 3), for callers that sort candidates by it
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
    if (T.list) return T.list[local];  // block shards under a kernel without the block mapping: the rank's pixels listed
    int t = local / T.tile, w = local - t * T.tile;
    return (t * T.world + T.rank) * T.tile + w;
}

// render_pool's pixel slots: a tile of 256 slots is a 16 x 16 block of pixels (neighbouring paths meet the same part of the
// scene: L1 / L2 hits) — all of them with one rank, every world-th with several (chunky_render_set_shard with tile 0); with a
// run length given instead, a tile is one of the rank's runs of consecutive pixel indices.  Returns width * height for a
// padding slot.
constexpr int kTileLog = 4, kTileEdge = 1 << kTileLog, kSampleTile = kTileEdge * kTileEdge;  // tiles of 16 x 16 pixels: pixel slots per tile
// Inside a tile the samples are ordered (sub-block of kSubBlock pixel slots, pass, slot in the sub-block): the 256 samples a
// wave claims at a time are 256 / kSubBlock consecutive passes of one small block of pixels — 64 passes of 2 x 2 pixels — so
// the paths a wave starts together begin as nearly the same ray: their march steps read the same tree entries (L1 hits,
// often the same address), find their candidates together and reach SHADE together.  Measured on the bench, sub-blocks of
// 256 (the tile: one pass per claim) / 128 / 64 / 32 / 16 / 8 / 4 / 2 / 1 slots: 5.96 / 6.01 / 6.03 / 6.06 / 6.11 / 6.12 /
// 6.16 / 6.08 / 5.96 Gsamples/s.
constexpr int kSubBlock = 4, kSubW = 2, kSubH = 2;  // sub-blocks of 2 x 2 pixels, row-major over the tile and inside
static_assert(kTileLog == 4 && kSubW * kSubH == kSubBlock, "sub-blocks tile a 16 x 16 tile");
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
// Exact n / d for n < 2^31 by one multiply-high and one shift (d fixed for a launch, the pair made on the host): with 2^s < d <= 2^(s+1)
// and m = ceil(2^(32+s) / d) — a 32-bit number — the error term m * d - 2^(32+s) is below d, and n * d < 2^(32+s) keeps the quotient exact.
struct FastDiv {
    unsigned m;  // 0: d == 1
    int s;
};
inline FastDiv fast_div(unsigned d) {
    if (d <= 1) return FastDiv{0u, 0};
    int s = 0;
    while ((2u << s) < d) s++;  // 2^s < d <= 2^(s+1)
    const unsigned long long m = (((unsigned long long)1 << (32 + s)) + d - 1) / d;
    return FastDiv{(unsigned)m, s};
}
DEV unsigned fast_quotient(unsigned n, FastDiv f) { return f.m ? __umulhi(n, f.m) >> f.s : n; }

// pool_slot_gid with the pixel's column and row, and the division by the tile row length as a multiply (render_pool's new samples)
struct SlotPixel {
    int gid, x, y;
};
DEV SlotPixel pool_slot_pixel(const ShardView& T, int width, int height, int slot, FastDiv by_bw) {
    if (T.world != 1 && T.tile != 0) {
        const int gid = slot < T.n_local ? shard_gid(T, slot) : width * height;
        return SlotPixel{gid, gid % width, gid / width};
    }
    const int bw = (width + kTileEdge - 1) >> kTileLog;
    int b = slot >> (2 * kTileLog);
    const int i = slot & (kSampleTile - 1);
    if (T.world != 1) {
        b = b * T.world + T.rank;
        if (b >= bw * ((height + kTileEdge - 1) >> kTileLog)) return SlotPixel{width * height, 0, 0};
    }
    const int by = (int)fast_quotient((unsigned)b, by_bw), bx = b - by * bw;
    const int sb = i / kSubBlock, px = i % kSubBlock;
    const int x = (bx << kTileLog) + (sb % (kTileEdge / kSubW)) * kSubW + px % kSubW,
              y = (by << kTileLog) + (sb / (kTileEdge / kSubW)) * kSubH + px / kSubW;
    return SlotPixel{(x < width && y < height) ? y * width + x : width * height, x, y};
}
__host__ __device__ inline long long pool_tiles(const ShardView& T, int width, int height) {
    if (T.world != 1) return ((long long)T.n_local + kSampleTile - 1) / kSampleTile;  // runs of T.tile pixels, or (T.tile == 0) the rank's blocks
    return (long long)((width + kTileEdge - 1) >> kTileLog) * ((height + kTileEdge - 1) >> kTileLog);
}
// lanes of the wave for which p holds, as a 32-bit scalar: one s_bcnt1_i32_b64 (the result is cast to int at once — kept as the
// 64-bit value __builtin_popcountll returns, the comparisons that follow would be 64-bit ones, which the scalar unit lacks)
DEV int count_lanes(bool p) { return (int)__builtin_popcountll(__ballot(p)); }
enum : int {
    ST_MARCH = 0, ST_BLOCK = 1, ST_SHADE = 2,  // voted phases (+ ST_BVH)
    ST_MODEL = 6,   // render_pool on the re-laid-out tree: the candidate is a model block (ST_BLOCK then means a full cube)
    ST_BVH = 9,     // at a node of an entity BVH: inner-node visits are voted as one phase,
    ST_LEAF = 11,   // the triangle tests of a leaf as another
    ST_TRACED = 10, // octree part of the trace finished (transient)
    ST_DONE = 3,   // no pixels left for this lane's group
    ST_NEXT = 4,   // path finished, radiance ready
    ST_SETUP = 5,  // ray ready, trace_setup pending
    ST_IDLE = 7,   // lane is free and waits for a pass of its group's pixel
    ST_START = 8   // begin the sample L.pass
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
template <bool GUARD = true>
DEV float leaf_exit_distance(const LaneState& L, f3 far, f3 po, int bx, int by, int bz, int level) {
    const int keep = -1 << level;
    const float size = __builtin_ldexpf(1.0f, level);
    const float x0 = (float)(bx & keep), y0 = (float)(by & keep), z0 = (float)(bz & keep);
    float tx = (rt_fma(far.x, size, x0) - po.x) * L.inv.x, ty = (rt_fma(far.y, size, y0) - po.y) * L.inv.y, tz = (rt_fma(far.z, size, z0) - po.z) * L.inv.z;
    if (GUARD) {  // only a component of the direction that is exactly -0 (inv = -inf) can produce the NaN: callers that know none is may skip this
        tx = rt_fmax(tx, -rt_inf());
        ty = rt_fmax(ty, -rt_inf());
        tz = rt_fmax(tz, -rt_inf());
    }
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

// The world's edge, 2^depth, as a value the optimiser cannot see through (it would turn "x < 1 << depth" back into a shift and
// a compare): made once, outside the march loop.
DEV unsigned world_edge(const SceneView& S) {
    unsigned edge = 1u << S.octree_depth;
    asm("" : "+s"(edge));
    return edge;
}
template <int TREE, bool GUARD = true>
DEV void march_step(const SceneView& S, const RenderOpts& O, LaneState& L, LaneMask marching, LaneMask& cand_out,
                    LaneMask& live_out, int& data, int& level, const unsigned edge, const LaneMask* far_masks = nullptr, int* entry = nullptr) {
    f3 pos = L.o + L.d * L.dist_march;
    f3 po = pos + L.d * kOffset;
    int bx = floor_to_int(po.x), by = floor_to_int(po.y), bz = floor_to_int(po.z);
    const bool inside = (unsigned)(bx | by | bz) < edge;  // every coordinate in [0, 2^depth): one compare (a negative one has the top bit set)
    const LaneMask live = marching & __ballot(L.steps < O.draw_depth) & __ballot(!(L.dist_march > L.h.distance)) & __ballot(inside);
    int kind;
    leaf_lookup<TREE>(S, bx, by, bz, data, level, kind, inside, entry);
    const LaneMask hittable = __ballot(kind < 2);
    const LaneMask cand = live & hittable, go = live & ~hittable;
    // render_pool passes the three lane masks "inv > 0" (scalar registers, set when the loop is entered) instead of L.far
    const f3 far = far_masks ? mk3(in_mask(far_masks[0]) ? 1.0f : 0.0f, in_mask(far_masks[1]) ? 1.0f : 0.0f, in_mask(far_masks[2]) ? 1.0f : 0.0f)
                             : L.far;
    const float step = leaf_exit_distance<GUARD>(L, far, po, bx, by, bz, level) + kOffset;  // K/octree.h:103-106
    const bool advance = in_mask(go);
    L.dist_march = advance ? L.dist_march + step : L.dist_march;
    // steps += 1 for the lanes of `go`: the mask is the carry-in of one add
    asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(L.steps) : "s"(go) : "vcc");
    cand_out = cand;
    live_out = live;
}

template <int TREE, int END, bool FARREG = true, int KINDS = kBlockAny>
DEV int block_phase(const SceneView& S, LaneState& L) {
    f3 pos = L.o + L.d * L.dist_march;
    f3 po = pos + L.d * kOffset;
    int bx = (int)rt_floor(po.x), by = (int)rt_floor(po.y), bz = (int)rt_floor(po.z);
    Hit t = L.h;
    float dist = block_hit<KINDS>(S, L.cand_data, bx, by, bz, pos, L.d, L.inv, t);
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
    FastDiv div_sub;       // render_pool: division of a sample index by P.n x kSubBlock (the samples of one sub-block)
    FastDiv div_bw;        // render_pool: division of a tile index by the tiles per image row
    const int* seeds_dev;  // render_pool: the launch's seeds in device memory when it carries more passes than P.seed holds, else null
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
        // a shadow ray's record.emittance is the |dot(sun dir, normal)| stored at its start (K/sky.h:90) and nothing writes it during
        // the trace — nor L.d or the normal: it is evaluated here, from the same operands, instead of being carried through the trace
        // (render_pool's six-word parked record has no place for it while 1/d and the distance marched are alive)
        const float e = main_trace ? 1.0f : rt_fabs(dot(L.d, L.h.normal));
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
            L.shadow = true;  // (record.emittance = |dot(L.d, n)|, K/sky.h:90: evaluated where it is read, above)
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

// variant bit 0 set = force the reference-layout octree walk (K/octree.h:81-89 as written)
inline bool use_wide(int variant, const SceneView& S) { return S.wide != nullptr && !(variant & 1); }
// the form of leaf_lookup for a scene: 0 reference layout, 16 + n dense top over n levels of 8x8x8 nodes, -1 any other wide split
inline int tree_form(int variant, const SceneView& S) {
    if (!use_wide(variant, S)) return 0;
    int tree = S.wide_nlev <= 4 ? 16 + S.wide_nlev - 1 : -1;
    for (int i = 1; i < S.wide_nlev; i++)
        if (S.wide_bits[i] != 3) tree = -1;
    return tree;
}
inline size_t stack_lds_bytes(const SceneView& S, int block) {
    bool need = !S.world_bvh_empty || !S.actor_bvh_empty;
    int entries = S.bvh_stack_entries > 0 && S.bvh_stack_entries < kBvhStackEntries ? S.bvh_stack_entries : kBvhStackEntries;
    return need ? (size_t)entries * block * sizeof(int) : 0;
}

}  // namespace chunky
