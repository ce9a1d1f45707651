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
    // (L.bvh_top counts in units of the stacks' stride, K.paths: entry e of stack p sits at base[e * paths + p], so a push or a
    // pop is one add and one LDS access — as a plain height the index was a 64-bit multiply-add per access)
    L.bvh_top -= K.paths;
    L.bvh_cur = K.base[(unsigned)L.bvh_top + (unsigned)L.pid];
    return ST_BVH;
}
// The first four words of the record a walker is at.  Byte offset of the record off S.bvh_rec:
DEV unsigned rwalk_record_at(const SceneView& S, int cur) {
    // nodes (64 bytes) and triangles (80) share one allocation: a 32-bit byte offset off one scalar base, whichever the walker is
    // at — both offsets by shifts and adds (a v_mul_lo_u32 runs at a quarter of the rate), the choice a select, no branch
    const unsigned tri = (unsigned)~cur >> 6;  // -1 - cur = ~cur
    unsigned t5, at_tri;  // tri * 80 + tri_off as two v_lshl_add_u32 (written as C the optimiser folds them back into the multiply)
    asm("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(t5) : "v"(tri));
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(at_tri) : "v"(t5), "s"(S.tri_off));
    const unsigned at_node = (unsigned)cur << 6;
    return cur >= 0 ? at_node : at_tri;
}
struct WalkWords {
    int4 r0, r1, r2, r3;
};
// (Fetched by pairs of lanes — the two lanes of a pair reading 32 contiguous bytes of one record per instruction and exchanging the
// halves by DPP, half the L1 tag look-ups — the walk is 9 % SLOWER: EXPERIMENTS.md 4.8.)
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
            K.base[(unsigned)L.bvh_top + (unsigned)L.pid] = go_first ? second : first;
            L.bvh_top += K.paths;
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
