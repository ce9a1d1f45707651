// render_fallback.hip — the kernels behind render_pool: render_waves (round 1's wave-scheduled kernel: paths bound to lanes,
// pixels shared by groups of lanes; CHUNKY_OPT_KERNEL bit 3, and the fallback where render_pool does not apply — entity BVHs
// that cannot be re-laid out, draw depths above 65535) and render_lanes (one lane owns one pixel for all the passes of a
// launch, walking the path of K/rayTracer.cl:93-107 as written; CHUNKY_OPT_KERNEL bit 1).  Bit-identical to render_pool.
//
// Compiled with -ffp-contract=off (see rt_device.hpp).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "path_state.hpp"

namespace chunky {

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
// What the leader lane of a pixel group keeps for its group (G > 1) — a property of the lane, not of a path.
struct GroupCtl {
    unsigned cur : 1;              // the open pixel passes are issued from
    unsigned exhausted : 1;        // the pixel queue is empty
    unsigned serial_counter : 23;  // serial of the last pixel opened
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
    float f1, f2;
    float t1 = box_quick_far(as_float(a[1]), as_float(a[2]), as_float(a[3]), as_float(a[4]), as_float(a[5]),
                             as_float(a[6]), L.o, L.inv, f1);
    float t2 = box_quick_far(as_float(b[1]), as_float(b[2]), as_float(b[3]), as_float(b[4]), as_float(b[5]),
                             as_float(b[6]), L.o, L.inv, f2);
    bool miss1 = (t1 != t1) || t1 > limit;
    bool miss2 = (t2 != t2) || t2 > limit;
    if (S.bvh_cull) {  // CHUNKY_OPT_BVH_CULL_BEHIND
        miss1 |= f1 < 0;
        miss2 |= f2 < 0;
    }
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
        // (L.pass is an 8-bit field: compare before counting, a launch of 256 passes ends at index 255 — counting first wrapped it to 0
        // and the pixel never ended: found by the fuzz of round 4, which was the first to give this kernel 256 passes per launch)
        if ((int)L.pass + 1 >= n_passes) {
            float* px = res + 3 * (size_t)L.gid;
            px[0] = L.mean.x;
            px[1] = L.mean.y;
            px[2] = L.mean.z;
            need_pixel = true;
        } else {
            L.pass += 1;
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
constexpr int kHandoverBatch = 8;
// parked radiances per open pixel (a pass is issued only inside fold + ring): two per lane of the group; the rings
// of a workgroup take 18 KB of LDS either way, which leaves room for five workgroups per CU
constexpr int ring_size(int group) { return 2 * group; }
struct GroupLds {
    float4* rad;  // [2][kRing] {r, g, b, tag}
    int* hdr;     // [2][8]  {gid, fold, issue, serial, mean.x, mean.y, mean.z, -}
};
enum : int { H_GID = 0, H_FOLD = 1, H_ISSUE = 2, H_SERIAL = 3, H_MEAN = 4 };

template <int TREE, int G, int RING = ring_size(G)>
DEV int next_sample(const SceneView& S, const CameraView& C, const ShardView& T, WaveArgPtr A, PixelPool& pool,
                    LaneState& L, GroupCtl& ctl, GroupLds lds, int st) {
    constexpr int kRing = RING;
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
    return ST_SETUP;
}

// BVH = false compiles the entity-BVH state out (scenes whose two BVHs are the empty sentinel).
template <int TREE, int G, bool BVH = false>
// Five workgroups per CU (96 VGPRs; a march step waits on one or two dependent tree reads, and the fifth wave per
// SIMD fills that time: +5 % over four); the entity-BVH kernels need ~125 registers and stay at four.
__global__ void __launch_bounds__(256, (BVH ? 4 : 5)) render_waves(WaveArgs unused_by_name) {
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
        const int n_octree = n_march > n_block ? (n_march > n_shade ? n_march : n_shade) : (n_block > n_shade ? n_block : n_shade);
        if (!idle && BVH && n_bvh > 0 && n_bvh >= n_octree && n_bvh >= n_leaf) {
            const SceneView S = arg_copy(&fresh_args()->S);
            // keep visiting nodes while the BVH walk holds the majority; a walk only leaves to LEAF or SHADE
            int nv, nl, ns;
            do {
                if (st == ST_BVH) st = bvh_phase(S, L, stack);
                nv = count_lanes(st == ST_BVH);
                nl = count_lanes(st == ST_LEAF);
                ns = count_lanes(st == ST_SHADE);
            } while (nv > 0 && nv >= nl && nv >= n_march && nv >= n_block && nv >= ns);
        } else if (!idle && BVH && n_leaf > 0 && n_leaf >= n_octree) {
            const SceneView S = arg_copy(&fresh_args()->S);
            if (st == ST_LEAF) st = leaf_phase(S, L, stack);
        } else if (!idle && n_march * kWMarch >= n_block * kWBlock && n_march * kWMarch >= n_shade * kWShade) {
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
                const unsigned edge = world_edge(Sm);
                do {
                    LaneMask cand, live;
                    march_step<TREE>(Sm, Om, L, marching, cand, live, data, level, edge);
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
            }
        } else if (!idle && n_block * kWBlock >= n_shade * kWShade) {
            const SceneView S = arg_copy(&fresh_args()->S);
            if (st == ST_BLOCK) st = block_phase<TREE, END>(S, L);
        } else {
            WaveArgPtr A = fresh_args();
            const SceneView S = arg_copy(&A->S);
            const RenderOpts O = arg_copy(&A->O);
            if (st == ST_SHADE) st = shade_phase<TREE, BVH>(S, O, L, stack);
            // G > 1: hand-over rounds cost ~300 instructions; wait until a few finished paths share one
            if (idle || count_lanes(st == ST_NEXT) >= (G == 1 ? 1 : kHandoverBatch)) {
                const CameraView C = arg_copy(&A->C);
                const ShardView T = arg_copy(&A->T);
                if (G == 1)
                    st = next_sample_single<TREE>(S, C, T, A, pool, L, st, first_round);
                else
                    st = next_sample<TREE, G>(S, C, T, A, pool, L, ctl, glds, st);
            }
            if (st == ST_SETUP) st = trace_setup<END>(S, L);
            first_round = false;
        }
    }
}

// variant bit 1 set = render_lanes; otherwise render_waves with variant bits 4-5 forcing its lanes per pixel (1 / 8 / 16).
// The fallback kernels exist for the reference octree layout (0) and the generic wide tree (-1) only.
hipError_t launch_fallback(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, const ShardView& T,
                           const PassSeeds& P, float* res, int* work_counter, hipStream_t stream, KernelChoice* chosen) {
    const bool has_bvh = !S.world_bvh_empty || !S.actor_bvh_empty;
    const int block = 256;
    const int tree = use_wide(variant, S) ? -1 : 0;
    if (!(variant & 2) && work_counter) {
        // wave-scheduled persistent kernel: one resident grid, lanes pull pixels from a counter
        int n_cu = 0;
        if (hipError_t e = current_device_cus(&n_cu)) return e;
        const size_t stack = stack_lds_bytes(S, block);
        size_t lds = stack;
        // lanes per pixel (see next_sample): 8; 16 or 32 when this GPU owns few pixels (multi-GPU tile split: the fewer
        // pixels per group, the longer the tail of the launch)
        int group = T.n_local < (3 << 17) ? 32 : (T.n_local < (3 << 18) ? 16 : 8);
        while (group > 8 && P.n < 2 * group) group /= 2;  // a group needs a few passes per lane to stay busy
        if (P.n < 2 * group) group = 1;
        switch ((variant >> 4) & 3) {  // variant bits 4-5 force the group size (tests cover all three)
            case 1: group = 1; break;
            case 2: group = 8; break;
            case 3: group = 16; break;
            default: break;
        }
        if (group == 32 && has_bvh) group = 16;
        typedef void (*Kernel)(WaveArgs);
        Kernel k;
#define WAVES_KERNEL(G) (has_bvh ? (tree == 0 ? render_waves<0, G, true> : render_waves<-1, G, true>) \
                                 : (tree == 0 ? render_waves<0, G, false> : render_waves<-1, G, false>))
        switch (group) {
            case 1: k = WAVES_KERNEL(1); break;
            case 16: k = WAVES_KERNEL(16); break;
            case 32: k = tree == 0 ? render_waves<0, 32, false> : render_waves<-1, 32, false>; break;
            default: group = 8; k = WAVES_KERNEL(8); break;
        }
#undef WAVES_KERNEL
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
    int grid = (T.n_local + block - 1) / block;
    if (grid <= 0 || P.n <= 0) return hipSuccess;
    if (chosen) *chosen = KernelChoice{tree, 0, has_bvh ? 1 : 0, grid, -1, 0};
    if (tree != 0)
        hipLaunchKernelGGL(render_lanes<-1>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O, T, P, res);
    else
        hipLaunchKernelGGL(render_lanes<0>, dim3(grid), dim3(block), stack_lds_bytes(S, block), stream, S, C, O, T, P, res);
    return hipGetLastError();
}

}  // namespace chunky
