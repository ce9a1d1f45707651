// pool_walk.hpp — the entity-BVH walk of render_pool on the aligned node / triangle records (rt_device.hpp): shared by
// render_pool.hip and the helper self test in aux_kernels.hip.
#pragma once
#include "path_state.hpp"

namespace chunky {

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
// The first four words of the record a walker is at.  Byte offset of the record off S.bvh_rec:
DEV unsigned rwalk_record_at(const SceneView& S, int cur) {
    const int lref = -1 - cur, tri = lref >> 6;
    // nodes and triangles share one allocation: a 32-bit byte offset off one scalar base, whichever the walker is at
    return cur >= 0 ? (unsigned)cur << 6 : S.tri_off + (unsigned)tri * 80u;
}
struct WalkWords {
    int4 r0, r1, r2, r3;
};
#ifndef CHUNKY_WALK_PAIR_FETCH
#define CHUNKY_WALK_PAIR_FETCH 0  // tuning builds: 1 = neighbouring lanes fetch each other's records half and half (rwalk_fetch_pairs)
#endif
#if CHUNKY_WALK_PAIR_FETCH
// Experiment (EXPERIMENTS.md 4.8): the four reads of a step, issued so that the two lanes of a pair read 32 CONTIGUOUS bytes of
// one record in one instruction (one L1 tag look-up per pair instead of one per lane: the walk runs at the L1s' look-up rate) —
// reads 0 / 1 fetch the even lane's record, the even lane its words 0 and 2, the odd lane words 1 and 3; reads 2 / 3 the odd lane's
// record the same way — and the halves are then exchanged inside the pair (DPP quad_perm 1,0,3,2).  Every lane ends up with the
// same four words it used to read by itself.  Called by ALL lanes of the wave (a lane that is not walking still fetches for its
// partner); `walking` = this lane has a walker.
DEV int dpp_pair_swap(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }  // quad_perm [1,0,3,2]
DEV int dpp_pair_even(int v) { return __builtin_amdgcn_mov_dpp(v, 0xA0, 0xF, 0xF, true); }  // quad_perm [0,0,2,2]
DEV int dpp_pair_odd(int v) { return __builtin_amdgcn_mov_dpp(v, 0xF5, 0xF, 0xF, true); }   // quad_perm [1,1,3,3]
DEV int4 dpp_pair_swap4(int4 v) { return make_int4(dpp_pair_swap(v.x), dpp_pair_swap(v.y), dpp_pair_swap(v.z), dpp_pair_swap(v.w)); }
DEV WalkWords rwalk_fetch_pairs(const SceneView& S, const LaneState& L, bool walking, int lane) {
    const unsigned at = walking ? rwalk_record_at(S, L.bvh_cur) : 0u;
    const int odd = lane & 1;
    const unsigned at_e = (unsigned)dpp_pair_even((int)at), at_o = (unsigned)dpp_pair_odd((int)at);
    const bool act_e = dpp_pair_even(walking ? 1 : 0) != 0, act_o = dpp_pair_odd(walking ? 1 : 0) != 0;
    int4 x0 = make_int4(0, 0, 0, 0), x1 = x0, y0 = x0, y1 = x0;
    if (act_e) {  // the even lane's record: words (0, 1) in one instruction, (2, 3) in the next
        const int4* __restrict__ p = (const int4*)((const char*)S.bvh_rec + at_e) + odd;
        x0 = p[0];
        x1 = p[2];
    }
    if (act_o) {  // the odd lane's record
        const int4* __restrict__ p = (const int4*)((const char*)S.bvh_rec + at_o) + odd;
        y0 = p[0];
        y1 = p[2];
    }
    // the even lane keeps x (its words 0, 2) and hands over y (the partner's words 0, 2); the odd lane the other way round
    const int4 send0 = odd ? x0 : y0, send1 = odd ? x1 : y1;
    const int4 recv0 = dpp_pair_swap4(send0), recv1 = dpp_pair_swap4(send1);
    WalkWords w;
    w.r0 = odd ? recv0 : x0;
    w.r1 = odd ? y0 : recv0;
    w.r2 = odd ? recv1 : x1;
    w.r3 = odd ? y1 : recv1;
    return w;
}
#endif
DEV WalkWords rwalk_fetch(const SceneView& S, const LaneState& L) {
    const int4* __restrict__ p = (const int4*)((const char*)S.bvh_rec + rwalk_record_at(S, L.bvh_cur));
    WalkWords w;
    w.r0 = p[0], w.r1 = p[1], w.r2 = p[2];
    w.r3 = p[3];  // (of a triangle only a hit needs this one; fetching it for inner nodes only measured 2 % slower)
    return w;
}
DEV int rwalk_apply(const SceneView& S, LaneState& L, PathStacks K, const WalkWords& W);
DEV int rwalk_step(const SceneView& S, LaneState& L, PathStacks K) { return rwalk_apply(S, L, K, rwalk_fetch(S, L)); }
DEV int rwalk_apply(const SceneView& S, LaneState& L, PathStacks K, const WalkWords& W) {
    const int cur = L.bvh_cur;
    const bool inner = cur >= 0;
    const int lref = -1 - cur, tri = lref >> 6, left = lref & 63;
    const int4 r0 = W.r0, r1 = W.r1, r2 = W.r2, r3 = W.r3;
    const float limit = L.shadow ? L.bvh_dist : L.h.distance;
    if (inner) {
        const int first = r0.x, second = r0.y;
        float f1, f2;
        const float t1 = box_quick_far(as_float(r1.x), as_float(r1.y), as_float(r1.z), as_float(r1.w), as_float(r2.x), as_float(r2.y), L.o, L.inv, f1);
        const float t2 = box_quick_far(as_float(r2.z), as_float(r2.w), as_float(r3.x), as_float(r3.y), as_float(r3.z), as_float(r3.w), L.o, L.inv, f2);
        bool miss1 = (t1 != t1) || t1 > limit;
        bool miss2 = (t2 != t2) || t2 > limit;
        if (S.bvh_cull) {  // (extension, wave-uniform) a child entirely behind the origin counts as missed
            miss1 |= f1 < 0;
            miss2 |= f2 < 0;
        }
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
                        const int4 r4 = ((const int4*)((const char*)S.bvh_rec + rwalk_record_at(S, cur)))[4];
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

}  // namespace chunky
