// render_pool.hip — the hot path: render_pool (a persistent grid whose waves each run a pool of 64 + K path state machines
// phase by phase — MARCH / BLOCK / SHADE, plus the entity-BVH walk — voted over the pool, samples claimed per XCD) and
// fold_kernel (the running mean of K/rayTracer.cl:109-112 over the staged samples, in pass order); the read-back exchange
// of a multi-GPU group; launch_render.  DESIGN.md section 5 describes the kernel.
//
// Compiled with -ffp-contract=off (see rt_device.hpp).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "path_state.hpp"
#include "pool_walk.hpp"

#ifndef CHUNKY_STAGING_STORE
#define CHUNKY_STAGING_STORE 0  // how finished samples reach the staging array: 0 = non-temporal (measured best), 1-3: see the deposit site
#endif

namespace chunky {

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
// A lane (or parked slot) that holds no path and wants a sample is "fresh": state ST_SHADE — the SHADE branch serves it — with
// the path depth 255 (kFreshDepth; launch_pool sends max_depth above 254 to the other kernels).  The states a path can be in
// between two phase executions are then 0 MARCH, 1 BLOCK, 2 SHADE, 3 DONE (+ ST_BVH / ST_LEAF with entity BVHs, + ST_MODEL where full
// cubes and model blocks are tested in phases of their own: SORT): the state IS its
// class, the census is one compare per class, and the vote and the swap need no classification (round 6: a fresh path used to be
// a state of its own, 12, and every iteration classified both state vectors through a chain of compares and branches).
constexpr unsigned kFreshDepth = 255u;

struct PoolLds {
    uint4* park;  // [WORDS][K]: 16-byte word g of slot s at park[g * K + s] (consecutive lanes, consecutive 16 bytes)
    int* tags;    // [K] state of the path parked in slot s
    int* list;    // [K] scratch: the slots taking part in a swap, by rank
};

template <bool BVH>
DEV int phase_class(int st) {
    return BVH && st >= ST_BVH ? 5 : st;  // (ST_MARCH 0, ST_BLOCK 1, ST_SHADE 2, ST_DONE 3; the entity walk's states are one class)
}

// A parked path is WORDS 16-byte words.  6 words without entity BVHs (1/d and the distance marched share their registers, and so
// their word, with the hit's colour and emittance: pool_pack; the march-step count shares word 0 with the flags —
// launch_pool sends draw depths above 65535 to render_waves — and the candidate block takes the place of the BVH cursor's
// word); 8 with them; 9 for the extended integrator.  The flag bits sit where LaneState's bit-fields have them.
template <int WORDS>
DEV void pool_pack(const LaneState& L, uint4 (&v)[WORDS]) {
    constexpr bool SHORT = WORDS <= 7;         // no entity BVHs: the march-step count shares word 0 with the flags, the candidate block takes the cursor's place
    constexpr int H = SHORT ? 5 : 6;  // first of the two words of the main record
    const unsigned misc = (unsigned)L.depth | ((unsigned)L.shadow << 8) | ((unsigned)L.oct_hit << 9) | ((unsigned)L.trace_hit << 10) |
                          ((unsigned)L.cand_level << 11) | ((unsigned)L.bvh_which << 15) |
                          (SHORT ? (unsigned)L.steps << 16 : ((unsigned)L.pid << 16) | ((unsigned)L.tkind << 24) | ((unsigned)L.after_nee << 26));
    if (WORDS > 8) v[WORDS - 1] = make_uint4(__float_as_uint(L.pend.x), __float_as_uint(L.pend.y), __float_as_uint(L.pend.z), (unsigned)L.h.spec);
    v[0] = make_uint4((unsigned)L.sidx, L.rng, misc, SHORT ? (unsigned)L.cand_data : (unsigned)L.steps);
    v[1] = make_uint4(__float_as_uint(L.radiance.x), __float_as_uint(L.radiance.y), __float_as_uint(L.radiance.z), __float_as_uint(L.throughput.x));
    v[2] = make_uint4(__float_as_uint(L.throughput.y), __float_as_uint(L.throughput.z), __float_as_uint(L.o.x), __float_as_uint(L.o.y));
    v[3] = make_uint4(__float_as_uint(L.o.z), __float_as_uint(L.d.x), __float_as_uint(L.d.y), __float_as_uint(L.d.z));
    if (WORDS == 6) {
        // Six words: the march's values (1/d, the distance marched) and the hit's colour and emittance are never alive together — from the
        // start of a trace to its end colour and emittance are dead (a hit rewrites them before SHADE reads them; a shadow ray's
        // emittance, |dot(sun direction, normal)|, is evaluated by shade_phase where it is read), from the end of a trace to the start of
        // the next one 1/d and the distance marched are dead (trace_setup sets both) — so the kernel keeps a hit's colour and emittance
        // in the registers of 1/d and the distance marched between the block test that hit and SHADE (hit_to_march_registers /
        // march_registers_to_hit), and a parked path has no word for them.
        v[4] = make_uint4(__float_as_uint(L.inv.x), __float_as_uint(L.inv.y), __float_as_uint(L.inv.z), __float_as_uint(L.dist_march));
        v[5] = make_uint4(__float_as_uint(L.h.distance), __float_as_uint(L.h.normal.x), __float_as_uint(L.h.normal.y), __float_as_uint(L.h.normal.z));
        return;
    }
    v[4] = make_uint4(__float_as_uint(L.inv.x), __float_as_uint(L.inv.y), __float_as_uint(L.inv.z), __float_as_uint(L.dist_march));
    if (WORDS > 7) v[5] = make_uint4((unsigned)L.bvh_cur, (unsigned)L.bvh_top, __float_as_uint(L.bvh_dist), (unsigned)L.cand_data);
    v[H] = make_uint4(__float_as_uint(L.h.distance), __float_as_uint(L.h.normal.x), __float_as_uint(L.h.normal.y), __float_as_uint(L.h.normal.z));
    v[H + 1] = make_uint4(__float_as_uint(L.h.color.x), __float_as_uint(L.h.color.y), __float_as_uint(L.h.color.z), __float_as_uint(L.h.emittance));
}
template <int WORDS>
DEV void pool_unpack(LaneState& L, const uint4 (&v)[WORDS]) {
    constexpr bool SHORT = WORDS <= 7;
    constexpr int H = SHORT ? 5 : 6;
    if (WORDS > 8) {
        L.pend = mk3(__uint_as_float(v[WORDS - 1].x), __uint_as_float(v[WORDS - 1].y), __uint_as_float(v[WORDS - 1].z));
        L.h.spec = (int)v[WORDS - 1].w;
    }
    L.sidx = (int)v[0].x; L.rng = v[0].y;
    L.depth = v[0].z & 0xFFu; L.shadow = (v[0].z >> 8) & 1u; L.oct_hit = (v[0].z >> 9) & 1u; L.trace_hit = (v[0].z >> 10) & 1u;
    L.cand_level = (v[0].z >> 11) & 15u; L.bvh_which = (v[0].z >> 15) & 1u;
    if (SHORT) {
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
    if (WORDS == 6) {
        L.h.distance = __uint_as_float(v[5].x);
        L.h.normal = mk3(__uint_as_float(v[5].y), __uint_as_float(v[5].z), __uint_as_float(v[5].w));
        return;
    }
    L.h.distance = __uint_as_float(v[H].x);
    L.h.normal = mk3(__uint_as_float(v[H].y), __uint_as_float(v[H].z), __uint_as_float(v[H].w));
    L.h.color = f4{__uint_as_float(v[H + 1].x), __uint_as_float(v[H + 1].y), __uint_as_float(v[H + 1].z), 0.0f};
    L.h.emittance = __uint_as_float(v[H + 1].w);
}

// The six-word record's register sharing (pool_pack): a block test that ended the trace leaves the hit's colour and emittance where
// 1/d and the distance marched were, SHADE takes them back.  (After a shadow ray's hit, or a trace that ended in the march, what moves
// is dead on both sides.)
DEV void hit_to_march_registers(LaneState& L) {
    L.inv = mk3(L.h.color.x, L.h.color.y, L.h.color.z);
    L.dist_march = L.h.emittance;
}
DEV void march_registers_to_hit(LaneState& L) {
    L.h.color = f4{L.inv.x, L.inv.y, L.inv.z, 0.0f};
    L.h.emittance = L.dist_march;
}

// LDS traffic between the lanes of ONE wave: the hardware executes a wave's LDS instructions in order; the fence keeps
// the compiler from moving accesses of different lanes to the same word across it.
DEV void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Lanes whose path does not wait for phase X trade it for a parked path that does (as many as both sides have).
// Lanes that hold nothing any more (ST_DONE) give their place up first.  Returns the number of swaps.
//
// The state tags of the parked paths live in registers (`ptag` of lane j = tag of slot j).  Partners find each other by
// rank through two small LDS arrays — slot and tag of the r-th parked path that comes in, tag of the r-th lane path that
// goes out — written by everybody first and read after ONE fence; then each swapping lane reads its partner's eight
// 16-byte groups in one burst and writes its own over them.
template <int K, int WORDS = 8>
DEV int pool_swap(PoolLds P, LaneState& L, int& st, int& ptag, int X, int lane) {
    constexpr bool BVH = WORDS == 8 || WORDS == 9;  // (the extended integrator's 9-word record carries the walk's fields too; 6 / 7 words: no entity BVHs)
    // who trades, as lane masks in scalar registers (mask algebra on the scalar unit; the vector unit only compares)
    constexpr LaneMask kSlots = K >= 64 ? ~0ull : ((1ull << K) - 1ull);            // lane j < K speaks for parked slot j
    const LaneMask m_done = __ballot(st == ST_DONE);
    const LaneMask m_goes = __ballot(phase_class<BVH>(st) != X);                   // (ST_DONE is a class of its own: included)
    const LaneMask m_in = __ballot(phase_class<BVH>(ptag) == X) & kSlots;
    const int n_out = (int)__popcll(m_goes), n_in = (int)__popcll(m_in);
    const int n = n_out < n_in ? n_out : n_in;
    if (n == 0) return 0;  // wave-uniform
    const int r_in = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_in >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_in, 0u));
    int r_out;
    if (m_done == 0) {  // (wave-uniform) the usual case: lanes go out in lane order
        r_out = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_goes >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_goes, 0u));
    } else {            // the last moments of a launch: lanes that hold nothing any more give their place up first
        const bool done = st == ST_DONE;
        const LaneMask m_mine = done ? m_done : (m_goes & ~m_done);
        r_out = (done ? 0 : (int)__popcll(m_done)) + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_mine >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_mine, 0u));
    }
    const bool comes = in_mask(m_in) && r_in < n, goes = in_mask(m_goes) && r_out < n;
    if (comes) P.list[r_in] = lane | (ptag << 8);  // slot and tag of the r-th path that comes in
    if (goes) P.tags[r_out] = st;                  // tag of the r-th path that goes out
    wave_lds_fence();
    if (comes) ptag = P.tags[r_in];
    if (goes) {
        const int e = P.list[r_out];
        const int s = e & 0xFF;
        uint4 mine[WORDS], theirs[WORDS];
        pool_pack<WORDS>(L, mine);
        // one LDS exchange per 8 bytes: the parked record and the lane's registers trade places in place
#pragma unroll
        for (int g = 0; g < WORDS; g++) {
            unsigned long long* q = (unsigned long long*)&P.park[g * K + s];
            const unsigned long long lo = __hip_atomic_exchange(q, (unsigned long long)mine[g].x | ((unsigned long long)mine[g].y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            const unsigned long long hi = __hip_atomic_exchange(q + 1, (unsigned long long)mine[g].z | ((unsigned long long)mine[g].w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            theirs[g] = make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
        }
        st = e >> 8;
        pool_unpack<WORDS>(L, theirs);
    }
    wave_lds_fence();  // the next swap's readers (other lanes) come after these writes
    return n;
}
// Tunables of the pool kernel (each measured on the bench; DESIGN.md section 5 has the sweeps).  The -D overrides exist
// for tuning builds (tools/variants.sh) only.
#ifndef CHUNKY_POOL_WAVES
#define CHUNKY_POOL_WAVES 6   // waves per SIMD: a march step waits on two dependent tree reads, every wave counts
#endif
#ifndef CHUNKY_POOL_BVH_WAVES
#define CHUNKY_POOL_BVH_WAVES 5
#endif
#ifndef CHUNKY_POOL_WORDS
#define CHUNKY_POOL_WORDS 6   // 16-byte words of a parked path without entity BVHs (7 in tuning builds: 1/d and colour side by side)
#endif
#ifndef CHUNKY_POOL_PARK
#define CHUNKY_POOL_PARK (CHUNKY_POOL_WORDS == 6 ? 64 : 56)   // paths parked per wave (LDS: WORDS x 16 + 8 bytes each; 6 x 4 waves x 64 x 104 B fill 156 of 160 KB)
#endif
#ifndef CHUNKY_POOL_REFILL
#define CHUNKY_POOL_REFILL 20 // leave the march loop to refill once this many lanes are free and parked marchers exist (round 6, with the new leave rule: 16 / 20 / 24 = 7 504 / 7 526 / 7 510 headline, 4 170 / 4 140 / 4 075 indoor)
#endif
#ifndef CHUNKY_WALK_LEAVE
#define CHUNKY_WALK_LEAVE 24  // leave the entity-BVH walk once this many lanes have finished theirs
#endif
#ifndef CHUNKY_STAY_FEW_PARKED
#define CHUNKY_STAY_FEW_PARKED 12  // "hardly any marcher parked" (see the march's leave rule)
#endif
#ifndef CHUNKY_STAY_LONGER
#define CHUNKY_STAY_LONGER 16      // ... how many lanes emptier the march then runs before the wave leaves it
#endif
constexpr int kPoolPark = CHUNKY_POOL_PARK, kPoolRefill = CHUNKY_POOL_REFILL, kWalkLeave = CHUNKY_WALK_LEAVE;
#ifndef CHUNKY_STAY_LONGER_BVH
#define CHUNKY_STAY_LONGER_BVH 6   // ... in the kernels with entity BVHs (16 / 10 / 6 / 0 lanes: entities 225.5 / 227 / 225.6 / 223, the city with its entities 851 / 898 / 921 / 914)
#endif
constexpr int kStayFewParked = CHUNKY_STAY_FEW_PARKED, kStayLonger = CHUNKY_STAY_LONGER, kStayLongerBvh = CHUNKY_STAY_LONGER_BVH;
#ifndef CHUNKY_W_MODEL
#define CHUNKY_W_MODEL 4
#endif
#ifndef CHUNKY_MODEL_FIRE
#define CHUNKY_MODEL_FIRE 560  // model blocks are tested once (how many wait) x (iterations since they last were) reaches this
#endif
constexpr int kWModel = CHUNKY_W_MODEL, kModelFire = CHUNKY_MODEL_FIRE;
constexpr int kWWalk = 1;          // vote weight of the walk against kWMarch / kWBlock / kWShade = 4: the walkers are the pool's standing crowd
constexpr int kSampleBatch = 256;  // sample indices a wave claims per atomic (measured: 64 -20 %, 128 -5 %, 512 -0.1 %, 1024 -1.3 %)

// Samples are handed out per XCD.  Each of the eight XCDs has its own L2, and workgroup b of a launch runs on XCD b % 8 (read
// from the hardware: HW_REG_XCC_ID).  The launch's samples — tile-major, so a contiguous range is a stripe of the image with
// all its passes — are cut into kXcdRanges equal ranges with a counter each; a wave starts in a range of the XCD it runs on
// and, when that has run dry (sky stripes finish long before terrain stripes), goes on with the range that follows, where its
// neighbours already are.  The waves that share an L2 — and the paths that share a wave's pool — thus work on neighbouring
// tiles of one stripe, rays of one kind, while the chip as a whole is spread over the image.  Which wave renders a sample
// has no influence on its value.  Measured on the bench (one counter: 5.55 Gsamples/s): eight stripes 5.92; going on with
// the fullest range instead of the next 5.77; tile groups dealt round-robin to the XCDs instead of stripes: groups of 1-8
// tiles -0.6 ... +0.9 %, 16-240 tiles +3 %.
constexpr int kXcdRanges = 8;  // one range per XCD (16 / 32 / 64 ranges measured -0.3 / -0.9 / -1.7 %)
constexpr int kXcdCounters = 64;  // the range counters sit at work_counter[64 ...] (behind the claim counter and the 24 profile words)
DEV int xcd_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return (int)(x & 7u);
}
// the range a wave starts in: the XCD's ranges are consecutive, its workgroups take them in turn
DEV int xcd_first_range() { return xcd_id(); }
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
// The march loop of one entry into MARCH: steps while `stay` lanes or more are marching.  Who found a candidate accumulates in
// `to_block`; `data` / `level` are the leaf every lane looked at last.
template <int TREE, bool GUARD, bool STATS>
DEV void march_loop(const SceneView& Sm, const RenderOpts& Om, LaneState& L, LaneMask& marching, LaneMask& to_block, int& data, int& level,
                    int& nm, int stay, const LaneMask* far_masks, unsigned long long* prof, int* entry = nullptr) {
    const unsigned edge = world_edge(Sm);
    do {
        if (STATS) {
            prof[0] += 1;
            prof[1] += (unsigned long long)nm;
        }
        LaneMask cand, live;
        march_step<TREE, GUARD>(Sm, Om, L, marching, cand, live, data, level, edge, far_masks, entry);
        to_block |= cand;
        marching = live & ~cand;
        nm = __popcll(marching);
    } while (nm >= stay);
}

// SORT: full cubes and model blocks are tested in phases of their own (on the re-laid-out tree, whose leaf entries say which a block is);
// launch_pool picks it for scenes with many model blocks
template <int TREE, int K, bool STATS, bool BVH = false, bool EXT = false, bool SORT = false>
__global__ void __launch_bounds__(256, ((STATS || EXT) ? 4 : (BVH ? CHUNKY_POOL_BVH_WAVES : CHUNKY_POOL_WAVES))) render_pool(WaveArgs unused_by_name) {
    constexpr int WORDS = EXT ? 9 : (BVH ? 8 : CHUNKY_POOL_WORDS);  // 16-byte words of a parked path (pool_pack)
    constexpr int END = BVH ? ST_TRACED : ST_SHADE;  // where a lane goes when the octree part of a trace ends
    // candidates sorted into full cubes (ST_BLOCK) and model blocks (ST_MODEL); not instantiated with entity BVHs or the extended
    // integrator (their pools are small: measured -4 % / -2 %)
    constexpr bool SPLIT = SORT;
    static_assert(!SORT || (TREE != 0 && !BVH && !EXT), "sorted block tests: the plain kernel on the re-laid-out tree only");
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
        // every parked slot starts fresh (depth 255 in its flag word); with entity BVHs it also owns a to-visit stack
        if (lane < K) P.park[lane] = make_uint4(0u, 0u, kFreshDepth | (WORDS <= 7 ? 0u : (unsigned)(64 + lane) << 16), 0u);
    }
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
    L.depth = kFreshDepth;
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
    XcdClaim claim{xcd_first_range(), 0u, 0u};
    int ranges_tried = 0;  // ranges this wave has found empty
    int st = ST_SHADE;  // fresh (L.depth == kFreshDepth): the first SHADE execution hands out samples
    int ptag = lane < K ? ST_SHADE : ST_DONE;
    wave_lds_fence();
    // the pool's census: paths waiting for each phase, in lanes and parked (taken at the END of an iteration, so that the loop has
    // one exit, at its head: a break in mid-loop makes the compiler define every loop-carried scalar on the exit path, with
    // v_readfirstlane of nothing, in every iteration)
    int c_march = 0, c_block = 0, c_shade = 0, c_bvh = 0, c_leaf = 0, c_model = 0;
    int model_age = 0;  // iterations since model blocks were last tested
    auto census = [&]() {
        if (BVH && __ballot(st == ST_TRACED)) {  // octree part of some traces just ended: entity BVHs next
            const SceneView S = arg_copy(&fresh_args()->S);
            if (st == ST_TRACED) st = rbvh_begin(S, L);
        }
        c_march = count_lanes(st == ST_MARCH) + count_lanes(ptag == ST_MARCH);
        c_block = count_lanes(st == ST_BLOCK) + count_lanes(ptag == ST_BLOCK);
        c_shade = count_lanes(st == ST_SHADE) + count_lanes(ptag == ST_SHADE);  // (fresh paths included)
        c_model = SPLIT ? count_lanes(st == ST_MODEL) + count_lanes(ptag == ST_MODEL) : 0;
        c_bvh = BVH ? count_lanes(st == ST_BVH) + count_lanes(ptag == ST_BVH) : 0;
        c_leaf = BVH ? count_lanes(st == ST_LEAF) + count_lanes(ptag == ST_LEAF) : 0;
    };
    census();
    while ((c_march | c_block | c_shade | c_bvh | c_leaf | c_model) != 0) {  // until every lane and every slot is ST_DONE
        // at most 64 paths run at once; among phases that can fill the wave SHADE and BLOCK go first (they feed the march)
        const int v_march = (c_march < 64 ? c_march : 64) * kWMarch, v_block = (c_block < 64 ? c_block : 64) * kWBlock,
                  v_shade = (c_shade < 64 ? c_shade : 64) * kWShade;
        // (written as integer arithmetic: as a chain of ?: on wave-uniform bools the compiler routes the choice through a VGPR)
        int X = (int)((unsigned)(v_march - v_block - 1) >> 31);  // 1 when v_block >= v_march, else 0
        int v_best = v_block > v_march ? v_block : v_march;
        if (v_shade >= v_best) {
            X = 2;
            v_best = v_shade;
        }
        if (SPLIT) {
            // Model blocks are a phase of their own: their tests cost three times the cube test, and the wave pays for them whenever ONE
            // lane has a model block.  How long they wait for company: a batch of T paths costs one execution per T arrivals, and
            // T / 2 slots of the pool while it gathers — the sum is least at T ~ sqrt(arrival rate), and with the rate estimated as
            // (waiting) / (iterations since the last execution) that is: run once waiting x iterations reaches a constant.  Where model
            // blocks are common (the city: 7 arrive per iteration) a wave's worth gathers first; where they are rare (the indoor room:
            // one in 40 iterations) three or four go together.
            const int v_model = (c_model < 64 ? c_model : 64) * kWModel;
            model_age += 1;
            if (v_model > v_best || c_model * model_age >= kModelFire) {
                X = ST_MODEL;
                v_best = v_model > v_best ? v_model : v_best;
                model_age = 0;
            }
        }
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
            if (SPLIT) {
                const int other_m = c_model < 64 ? c_model : 64;
                other = other > other_m ? other : other_m;
            }
            // (with entity BVHs the walkers are not counted: they are the pool's standing crowd and wait in any case)
            int stay = (other + nm + 1) >> 1;
            // ... and longer still when hardly any marcher is parked (fewer than kStayFewParked): leaving then means a swap round and a phase
            // that cannot be refilled afterwards, so the march goes on kStayLonger lanes emptier before it hands over.  Measured on the final
            // loop (round 6, after the iteration overhead fell): threshold x lanes 12 x 16 = 7 510 / 3 930 / 4 080 Msamples/s on headline / city /
            // indoor against 7 250 / 3 740 / 4 065 without; unconditional (-12 lanes) 7 430 / 3 880 / 4 010 — a scene whose pool is full of
            // marchers (the indoor room) is better off leaving early and refilling.
            // (with entity BVHs the pool is small and the walkers are its standing crowd: 6 lanes — 16 there costs the city with its entities 7 %)
            if (parked_march < kStayFewParked) stay -= BVH ? kStayLongerBvh : kStayLonger;
            if (K > 0 && parked_march >= kPoolRefill && stay < 65 - kPoolRefill) stay = 65 - kPoolRefill;
            if (stay < 1) stay = 1;
            LaneMask marching = entered, to_block = 0;
            // the exit-plane selectors (1.0 where the ray runs towards +axis) as three registers for the length of the loop — they
            // are not part of a parked path; as three lane masks they cost three v_cndmask per step (round 6: +0.9 %)
            L.far = far_of(L.inv);
            const LaneMask* far_masks = nullptr;
            int data, level, entry = 0;
            // a direction component that is exactly -0 (inv = -inf) is the one case in which the leaf exit has to guard against a
            // NaN (leaf_exit_distance): as good as never does a marching lane of the wave have one, and the loop then runs
            // without the three guards (+0.7 % on the bench)
            const float ninf = -rt_inf();
            const bool guard = ((__ballot(L.inv.x == ninf) | __ballot(L.inv.y == ninf) | __ballot(L.inv.z == ninf)) & entered) != 0;  // (masks on the scalar unit)
            if (guard)
                march_loop<TREE, true, STATS>(Sm, Om, L, marching, to_block, data, level, nm, stay, far_masks, prof, SPLIT ? &entry : nullptr);
            else
                march_loop<TREE, false, STATS>(Sm, Om, L, marching, to_block, data, level, nm, stay, far_masks, prof, SPLIT ? &entry : nullptr);
            const bool found = in_mask(to_block);
            L.cand_data = found ? data : L.cand_data;
            L.cand_level = found ? level : L.cand_level;
            const int st_found = SPLIT && (entry & 0x2000000) ? ST_MODEL : ST_BLOCK;  // bit 25 of a leaf entry: a model block (widetree.hpp kWideKindLow)
            st = found ? st_found : (in_mask(entered & ~marching & ~to_block) ? END : st);
            if (STATS) {
                prof[0] -= 1;
                prof[1] -= (unsigned long long)n_exec;
            }
        } else if (X == 1) {
            n_exec = count_lanes(st == ST_BLOCK);
            const SceneView S = arg_copy(&fresh_args()->S);
            if (st == ST_BLOCK) {
                st = block_phase<TREE, END, false, SPLIT ? kBlockCubes : kBlockAny>(S, L);
                if (WORDS == 6 && st == END) hit_to_march_registers(L);
            }
        } else if (SPLIT && X == ST_MODEL) {
            n_exec = count_lanes(st == ST_MODEL);
            if (STATS) parts.t[8] += (unsigned long long)n_exec + (1ull << 40);  // value 22 of the profile: lanes, and executions in bits 40 up
            const SceneView S = arg_copy(&fresh_args()->S);
            asm volatile("; chunky-mark models");  // (comments in the compiled kernel: tools/isa_scratch.py finds the model blocks' phase by them)
            if (st == ST_MODEL) {
                st = block_phase<TREE, END, false, kBlockModels>(S, L);
                if (WORDS == 6 && st == END) hit_to_march_registers(L);
            }
            asm volatile("; chunky-mark models-end");
        } else if (BVH && X == 5) {
            // The walk: inner-node visits and triangle tests are one step function (rwalk_step: the same four 16-byte reads
            // from one array or the other), so a walker is ST_BVH throughout and the loop below counts that one state.  The
            // wave stays while enough walkers remain, or until enough lanes are free for a refill from parked walkers.
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
            n_exec = count_lanes(st == ST_SHADE);
            WaveArgPtr A = fresh_args();
            const SceneView S = arg_copy(&A->S);
            const RenderOpts O = arg_copy(&A->O);
            const bool served = st == ST_SHADE;
            bool fresh = served && L.depth == kFreshDepth;  // holds no path: wants a sample
            if (served && !fresh) {
                if (WORDS == 6) march_registers_to_hit(L);
                st = EXT ? shade_phase_ext<TREE, BVH>(S, O, L) : shade_phase<TREE, BVH, STATS>(S, O, L, stack, &parts);
            }
            part_begin<STATS>(&parts);
            if (st == ST_NEXT) {  // the path is finished: its radiance waits in the staging array for fold_kernel
                // streamed past the caches (nt): written once, read once by fold_kernel; the L2 stays with the tree
                float* __restrict__ out = A->staging + 3 * (size_t)(unsigned)L.sidx;
#if CHUNKY_STAGING_STORE == 1  // tuning builds (tools/variants.sh): plain stores, merged by the L2 (-2.3 % on the bench)
                out[0] = L.radiance.x, out[1] = L.radiance.y, out[2] = L.radiance.z;
#elif CHUNKY_STAGING_STORE == 2  // device-scope stores (sc1): written through the L2
                __hip_atomic_store(out, L.radiance.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(out + 1, L.radiance.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(out + 2, L.radiance.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif CHUNKY_STAGING_STORE == 3  // system-scope stores (sc0 sc1)
                __hip_atomic_store(out, L.radiance.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(out + 1, L.radiance.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(out + 2, L.radiance.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
                __builtin_nontemporal_store(L.radiance.x, out);
                __builtin_nontemporal_store(L.radiance.y, out + 1);
                __builtin_nontemporal_store(L.radiance.z, out + 2);
#endif
                fresh = true;
            }
            part_end<STATS>(&parts, PT_DEPOSIT);
            // ---- new samples (K/rayTracer.cl:55-91).  Sample index = (tile of kSampleTile pixel slots, pass, slot in tile): a
            //      tile gets all its passes before the next tile starts, so the paths in flight on the whole GPU cover a few
            //      thousand neighbouring pixels — a part of the scene that stays in the 4 MB L2s (pass-major order spread
            //      them over a third of the image: L2 hit rate 91 %, 66 GB of fabric reads per launch instead of 4) ----
            const bool need = fresh;
            const unsigned sidx = xcd_claim(A->Q.next + kXcdCounters, claim, ranges_tried, need, A->xcd_stripe, A->n_samples);  // convergent
            if (need && sidx != kClaimNone) {
                const unsigned n_samples = A->n_samples;
                if (sidx >= n_samples) {
                    st = ST_DONE;
                    fresh = false;
                } else {
                    const CameraView C = arg_copy(&A->C);
                    const ShardView T = arg_copy(&A->T);
                    // sidx = ((tile * sub-blocks per tile + sub-block) * passes + pass) * kSubBlock + slot in the sub-block
                    const unsigned per_sub = (unsigned)A->P.n * (unsigned)kSubBlock;
                    const unsigned sub = fast_quotient(sidx, arg_copy(&A->div_sub)), rem = sidx - sub * per_sub;  // sub = tile * (kSampleTile / kSubBlock) + sub-block
                    const unsigned pass = rem / (unsigned)kSubBlock;
                    const int slot = (int)(sub * (unsigned)kSubBlock + (rem & (unsigned)(kSubBlock - 1)));
                    const SlotPixel px = pool_slot_pixel(T, C.width, C.height, slot, arg_copy(&A->div_bw));
                    const int gid = px.gid;
                    if (gid < C.width * C.height) {  // else: a padding slot, nothing to render (the lane claims again)
                        const int* seeds_dev = A->seeds_dev;  // (wave-uniform: launches longer than P.seed holds)
                        unsigned rng = (unsigned)(seeds_dev ? seeds_dev[pass] : A->P.seed[pass]) + (unsigned)gid;
                        rt_pcg_next(&rng);
                        const RayOD pr = primary_ray(C, gid, rng, false, px.x, px.y);
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
                        fresh = false;
                    }
                }
            }
            part_end<STATS>(&parts, PT_NEWSAMPLE);
            if (fresh) {  // no sample this time (the tail of a batch, a padding slot): it asks again at the next SHADE
                st = ST_SHADE;
                L.depth = kFreshDepth;
            }
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
            if (X == ST_MODEL) parts.t[9] += dt;  // the model blocks' phase: value 23 of the profile
            if (X == 5) {  // the entity-BVH walk is profiled with BLOCK
                prof[3] += 1;
                prof[4] += (unsigned long long)n_exec;
                prof[5] += dt;
            }
        }
        census();
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
        // (plain loads: a thread's consecutive passes share cache lines)
        mean = (mean * (float)spp + p[(size_t)k * (3 * kSubBlock)]) / (float)(spp + 1);
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
// The exchange by ncclReduce (capi.hip group_gather_reduce) sums zero-padded framebuffers: this clears the pixels shard T does
// not own — same ownership rule as shard_gid / pool_slot_gid (runs of T.tile pixel indices, or 16 x 16 blocks, dealt round-robin).
__global__ void __launch_bounds__(256) clear_foreign_kernel(ShardView T, int width, int height, float* __restrict__ fb) {
    const int gid = (int)(blockIdx.x * 256 + threadIdx.x);
    if (gid >= width * height || T.world == 1) return;
    int unit;
    if (T.tile != 0) {
        unit = gid / T.tile;
    } else {
        const int y = gid / width, x = gid - y * width;
        unit = (y >> kTileLog) * ((width + kTileEdge - 1) >> kTileLog) + (x >> kTileLog);
    }
    if (unit % T.world == T.rank) return;
    float* a = fb + 3 * (size_t)gid;
    a[0] = 0.0f; a[1] = 0.0f; a[2] = 0.0f;
}
// ------------------------------------------------------------------------------------ launchers
// render_pool + fold_kernel.  variant bits 6-7 pick the parked paths per wave: 0 = 64 (default), 1 = none, 2 = 32 (test rigs:
// the generic tree form only).
static hipError_t launch_pool(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, const ShardView& T,
                              const PassSeeds& P, float* res, int* work_counter, hipStream_t stream, KernelChoice* chosen,
                              float* staging, const int* seeds_dev) {
    const int block = 256;
    int n_cu = 0;
    if (hipError_t e = current_device_cus(&n_cu)) return e;
    if (T.n_local <= 0 || P.n <= 0) return hipSuccess;
    const bool stats = (variant & 4) != 0;
    int tree = tree_form(variant, S);
    const bool bvh = !S.world_bvh_empty || !S.actor_bvh_empty;
    int depth = 0;  // entries per to-visit stack (the reference reserves 64, K/bvh.h:38; here: the height of the taller BVH + 1)
    if (bvh) depth = S.bvh_stack_entries > 0 && S.bvh_stack_entries < kBvhStackEntries ? S.bvh_stack_entries : kBvhStackEntries;
    int park = kPoolPark;
    switch ((variant >> 6) & 3) {
        case 1: park = 0; break;
        case 2: park = 32; break;
        default: break;
    }
    typedef void (*Kernel)(WaveArgs);
    Kernel k;
    const bool ext = opts_extended(O);  // EXPERIMENTAL light-transport options: their own instantiations
    // Block tests sorted by kind (full cubes / model blocks in phases of their own): for scenes in which model blocks are common
    // (S.sort_blocks, capi.hip scene_view) — the city gains 2 – 3 %, a scene with hardly any pays 1 % for the bookkeeping; variant bit 8
    // forces it on, bit 9 off (tests and A/B runs); the plain kernel at its full pool only
    const bool sorted = ((variant & 256) || S.sort_blocks) && !(variant & 512) && !bvh && !ext && tree != 0 && S.block_info != nullptr;
    bool sorted_ran = false;
    int words = bvh ? 8 : CHUNKY_POOL_WORDS;
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
        if (stats) {
            if (tree != 17) tree = -1;
            park = 16;
            k = tree == 17 ? render_pool<17, 16, true, true> : render_pool<-1, 16, true, true>;
#ifdef CHUNKY_POOL_BVH_PARK  // tuning builds (tools/variants.sh): a fixed pool size for the entity kernel, e.g. 8 parked paths at six waves
        } else if (true) {
            if (tree != 17) tree = -1;
            park = CHUNKY_POOL_BVH_PARK;
            k = tree == 17 ? render_pool<17, CHUNKY_POOL_BVH_PARK, false, true> : render_pool<-1, CHUNKY_POOL_BVH_PARK, false, true>;
#endif
        } else if (park == 32) {
            k = tree == 17 ? render_pool<17, 32, false, true> : (tree == 18 ? render_pool<18, 32, false, true> : render_pool<-1, 32, false, true>);
        } else {
            k = tree == 17 ? render_pool<17, 16, false, true> : (tree == 18 ? render_pool<18, 16, false, true> : render_pool<-1, 16, false, true>);
        }
    } else if (stats) {
        if (tree != 17) tree = -1;
        park = kPoolPark;
        if (sorted) k = tree == 17 ? render_pool<17, kPoolPark, true, false, false, true> : render_pool<-1, kPoolPark, true, false, false, true>;
        else k = tree == 17 ? render_pool<17, kPoolPark, true> : render_pool<-1, kPoolPark, true>;
        sorted_ran = sorted;
    } else if (park != kPoolPark) {
        if (tree != 0) tree = -1;
        if (park == 0) k = tree == 0 ? render_pool<0, 0, false> : render_pool<-1, 0, false>;
        else k = tree == 0 ? render_pool<0, 32, false> : render_pool<-1, 32, false>;
    } else if (sorted) {
        switch (tree) {
            case 16: k = render_pool<16, kPoolPark, false, false, false, true>; break;
            case 17: k = render_pool<17, kPoolPark, false, false, false, true>; break;
            case 18: k = render_pool<18, kPoolPark, false, false, false, true>; break;
            case 19: k = render_pool<19, kPoolPark, false, false, false, true>; break;
            default: tree = -1; k = render_pool<-1, kPoolPark, false, false, false, true>; break;
        }
        sorted_ran = true;
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
    if (chosen) *chosen = KernelChoice{tree, 1, bvh ? 1 : 0, grid, park, ext ? 1 : 0, sorted_ran ? 1 : 0};
    e = hipMemsetAsync(work_counter, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(work_counter + kXcdCounters, 0, kXcdRanges * sizeof(int), stream);  // the per-XCD sample ranges (xcd_claim)
    if (e != hipSuccess) return e;
    WaveArgs A{S, C, O, T, P, WorkQueue{work_counter}, res, (unsigned long long*)(work_counter + 2), (unsigned)depth, staging, (unsigned)n_samples,
               (unsigned)((n_tiles + kXcdRanges - 1) / kXcdRanges * kSampleTile * P.n), fast_div((unsigned)P.n * (unsigned)kSubBlock),
               fast_div((unsigned)((C.width + kTileEdge - 1) >> kTileLog)), seeds_dev};
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, stream, A);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long long threads = 3ll * n_tiles * kSampleTile;
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, (const float*)staging, res, T,
                       C.width * C.height, C.width, n_tiles * kSampleTile, P.n, P.first_spp);
    return hipGetLastError();
}
bool pool_kernel_applies(int variant, const SceneView& S, const RenderOpts& O, bool have_queue_and_staging) {
    const bool any_bvh = !S.world_bvh_empty || !S.actor_bvh_empty;
    // render_pool: always without entity BVHs; with them when they could be re-laid out (rt_device.hpp bvh_rec / tri_rec)
    // (its 6-word parked record counts march steps in 16 bits: a larger draw depth runs render_waves)
    const bool steps_fit = (any_bvh || opts_extended(O) || O.draw_depth <= 65535) && O.max_depth <= 254;  // (path depth 255 marks a fresh path)
    return !(variant & 2) && !(variant & 8) && have_queue_and_staging && steps_fit &&
           (!any_bvh || (S.bvh_rec && S.tri_rec && S.mat8 && !(variant & 1)));
}
hipError_t launch_render(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, const ShardView& T,
                         const PassSeeds& P, float* res, int* work_counter, hipStream_t stream, KernelChoice* chosen,
                         float* staging, const int* seeds_dev) {
    if (pool_kernel_applies(variant, S, O, work_counter && staging)) {
        if (P.n > kMaxPoolPasses || (P.n > kMaxPassesPerLaunch && !seeds_dev)) return hipErrorInvalidValue;
        return launch_pool(variant, S, C, O, T, P, res, work_counter, stream, chosen, staging, P.n > kMaxPassesPerLaunch ? seeds_dev : nullptr);
    }
    if (P.n > kMaxPassesPerLaunch) return hipErrorInvalidValue;
    if (T.world != 1 && T.tile == 0) {
        // shards of 16 x 16 blocks are mapped by render_pool only: the other kernels take the rank's pixels as a list
        if (!T.list) return hipErrorNotSupported;
        ShardView F = T;
        F.n_local = T.n_list;
        return launch_fallback(variant, S, C, O, F, P, res, work_counter, stream, chosen);
    }
    return launch_fallback(variant, S, C, O, T, P, res, work_counter, stream, chosen);
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

hipError_t launch_clear_foreign(const ShardView& T, int width, int height, float* fb, hipStream_t stream) {
    const long long n = (long long)width * height;
    if (n <= 0 || T.world == 1) return hipSuccess;
    hipLaunchKernelGGL(clear_foreign_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, T, width, height, fb);
    return hipGetLastError();
}

}  // namespace chunky
