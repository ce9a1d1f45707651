/* port.c — plain-C restatement of the reference's per-sample path (CPU oracle, "port" kind).
 *
 * TEST INFRASTRUCTURE ONLY.  Used by tests/ as the checker, by __graft_entry__.smoke() and by
 * bench.py's cpu_baseline leg.  Never linked into or called from chunkyclplugin_amd/.
 *
 * Pinning: this file is validated bit-for-bit against oracle/_ref (the reference kernel itself,
 * compiled in place for x86-64) by tests/test_oracle_pinning.py in the build container, and
 * against the committed outputs of that reference build under tests/golden/ everywhere else.
 *
 * Every function cites the reference lines it follows; K/ = /root/reference/src/main/opencl/
 * kernel/include/.  Arithmetic contract: IEEE binary32 (binary64 at the reference's double
 * literal sites), one rounding per source-level operation (-ffp-contract=off), OpenCL builtins
 * as defined in chunkyclplugin_amd/csrc/rt_math.h.
 *
 * It also counts the reference algorithm's access stream (BASELINE.md section 4 "algorithmic
 * bytes per sample"), which the roofline figure in bench.py is computed from.
 */
#define _GNU_SOURCE /* sched_setaffinity, CPU_SET */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef __linux__
#include <sched.h>
#include <unistd.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../chunkyclplugin_amd/csrc/rt_math.h"
#include "oracle_scene.h"

#define EPS 0.000005f   /* K/constants.h:4 */
#define OFFSET 0.0001f  /* K/constants.h:5 */
#define ANY_TYPE 0x7FFFFFFE

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
#ifndef PORT_LIBM
static inline float dot3(v3 a, v3 b) { return rt_dot3(a.x, a.y, a.z, b.x, b.y, b.z); }
static inline v3 cross3(v3 a, v3 b) {
    return V3(rt_cross_c(a.y, b.z, a.z, b.y), rt_cross_c(a.z, b.x, a.x, b.z), rt_cross_c(a.x, b.y, a.y, b.x));
}
static inline v3 normalize3(v3 a) { return scale3(a, rt_rlen3(a.x, a.y, a.z)); }
#else
/* PORT_LIBM (make port_libm -> libchunky_port_libm.so): this restatement on a SECOND platform layer — glibc's sinf / cosf /
 * asinf / acosf / atan2f and unfused dot / cross / normalize, exactly the definitions oracle/ref_shim.cpp gives the compiled
 * reference under REF_SHIM_LIBM (:53-125).  tests/test_platform_layer.py demands port_libm == ref_libm BIT FOR BIT: the
 * restatement's LOGIC is then pinned to the reference's under builtins that do not come from rt_math.h. */
#include <math.h>
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline v3 normalize3(v3 a) {
    float len = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return V3(a.x / len, a.y / len, a.z / len);
}
static inline void port_libm_sincos(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
#define rt_sincos port_libm_sincos
#define rt_sin sinf
#define rt_cos cosf
#define rt_asin asinf
#define rt_acos acosf
#define rt_atan2 atan2f
#endif

/* ---------------------------------------------------------------- access-stream counters --- */
typedef struct {
    int64_t samples, traces, steps, node, block, model_hdr, aabb, quad, mat, texel, bvh_inner,
        leaf_hdr, tri, sky, hits;
} Counters;
#define N_COUNTERS 15
static Counters g_total;
static int g_count_enabled = 0;
static _Thread_local Counters* t_ctr = 0;
#define COUNT(field, n) do { if (t_ctr) t_ctr->field += (n); } while (0)

void port_counters_enable(int on) { g_count_enabled = on; }
void port_counters_reset(void) { memset(&g_total, 0, sizeof g_total); }
void port_counters_read(int64_t* out) { memcpy(out, &g_total, sizeof g_total); }
static void counters_merge(const Counters* c) {
    const int64_t* s = (const int64_t*)c;
    int64_t* d = (int64_t*)&g_total;
    for (int i = 0; i < N_COUNTERS; i++) {
#ifdef _OPENMP
#pragma omp atomic
#endif
        d[i] += s[i];
    }
}

/* ------------------------------------------------------------------------- path state ------ */
/* K/wavefront.h:6-51 — Pixel, Ray and IntersectionRecord flattened; `ray` is shared between a
 * record and its shadow copy exactly like the reference's Ray* aliasing (wavefront.h:66). */
typedef struct {
    v3 color, throughput;        /* Pixel */
    v3 origin, direction;        /* Ray */
    int ray_material, ray_depth; /* Ray.material stays 0 (wavefront.h:34) */
} Path;
typedef struct {
    float distance;
    int material;
    v3 normal, point;
    v4 color;
    float emittance;
    int spec; /* material word 5 (spec | metal << 8 | rough << 16, PackedMaterial.java:69-71): read by the extensions only */
} Record;

typedef struct { /* K/sky.h:9-17 */
    int flags, texture_size, texture;
    float intensity;
    v3 su, sv, sw;
} Sun;

/* ------------------------------------------------------------------------- primitives ------ */
typedef struct { float xmin, xmax, ymin, ymax, zmin, zmax; } Box;

/* K/primitives.h:30-48 */
static float box_quick(const Box* b, v3 o, v3 inv) {
    float t1x = (b->xmin - o.x) * inv.x, t1y = (b->ymin - o.y) * inv.y, t1z = (b->zmin - o.z) * inv.z;
    float t2x = (b->xmax - o.x) * inv.x, t2y = (b->ymax - o.y) * inv.y, t2z = (b->zmax - o.z) * inv.z;
    float tmin = rt_fmax(rt_fmin(t1x, t2x), rt_fmax(rt_fmin(t1y, t2y), rt_fmin(t1z, t2z)));
    float tmax = rt_fmin(rt_fmax(t1x, t2x), rt_fmin(rt_fmax(t1y, t2y), rt_fmax(t1z, t2z)));
    return (tmax < tmin) ? rt_nan() : tmin;
}

/* K/primitives.h:52-61 */
static float box_exit(const Box* b, v3 o, v3 inv) {
    float t1x = (b->xmin - o.x) * inv.x, t1y = (b->ymin - o.y) * inv.y, t1z = (b->zmin - o.z) * inv.z;
    float t2x = (b->xmax - o.x) * inv.x, t2y = (b->ymax - o.y) * inv.y, t2z = (b->zmax - o.z) * inv.z;
    return rt_fmin(rt_fmax(t1x, t2x), rt_fmin(rt_fmax(t1y, t2y), rt_fmax(t1z, t2z)));
}

/* K/primitives.h:66-112 (unit = 0) and :117-162 (map2 = 1).  Entry face by the exact-equality
 * chain, last match wins.  `dir` is whatever the caller passes (block.h:52 passes the march
 * position, Appendix B#1). */
static float box_full(const Box* b, v3 o, v3 dir, v3 inv, v3* normal, float* u, float* v, int map2) {
    float t1x = (b->xmin - o.x) * inv.x, t1y = (b->ymin - o.y) * inv.y, t1z = (b->zmin - o.z) * inv.z;
    float t2x = (b->xmax - o.x) * inv.x, t2y = (b->ymax - o.y) * inv.y, t2z = (b->zmax - o.z) * inv.z;
    float tmin = rt_fmax(rt_fmin(t1x, t2x), rt_fmax(rt_fmin(t1y, t2y), rt_fmin(t1z, t2z)));
    float tmax = rt_fmin(rt_fmax(t1x, t2x), rt_fmin(rt_fmax(t1y, t2y), rt_fmax(t1z, t2z)));
    if (tmax < tmin) return rt_nan();
    v3 p = add3(o, scale3(dir, tmin)); /* origin + tmin * dir */
    if (!map2) {
        float dx = 1 / (b->xmax - b->xmin), dy = 1 / (b->ymax - b->ymin), dz = 1 / (b->zmax - b->zmin);
        if (t1x == tmin) { *u = 1 - (p.z - b->zmin) * dz; *v = (p.y - b->ymin) * dy; *normal = V3(-1, 0, 0); }
        if (t2x == tmin) { *u = (p.z - b->zmin) * dz; *v = (p.y - b->ymin) * dy; *normal = V3(1, 0, 0); }
        if (t1y == tmin) { *u = (p.x - b->xmin) * dx; *v = 1 - (p.z - b->zmin) * dz; *normal = V3(0, -1, 0); }
        if (t2y == tmin) { *u = (p.x - b->xmin) * dx; *v = (p.z - b->zmin) * dz; *normal = V3(0, 1, 0); }
        if (t1z == tmin) { *u = (p.x - b->xmin) * dx; *v = (p.y - b->ymin) * dy; *normal = V3(0, 0, -1); }
        if (t2z == tmin) { *u = 1 - (p.x - b->xmin) * dx; *v = (p.y - b->ymin) * dy; *normal = V3(0, 0, 1); }
    } else {
        if (t1x == tmin) { *u = p.z; *v = p.y; *normal = V3(-1, 0, 0); }
        if (t2x == tmin) { *u = 1 - p.z; *v = p.y; *normal = V3(1, 0, 0); }
        if (t1y == tmin) { *u = p.x; *v = p.z; *normal = V3(0, -1, 0); }
        if (t2y == tmin) { *u = p.x; *v = 1 - p.z; *normal = V3(0, 1, 0); }
        if (t1z == tmin) { *u = 1 - p.x; *v = p.y; *normal = V3(0, 0, -1); }
        if (t2z == tmin) { *u = p.x; *v = p.y; *normal = V3(0, 0, 1); }
    }
    return tmin;
}

/* ------------------------------------------------------------------ atlas + materials ------ */
static inline v4 scale4(v4 a, float s) { v4 r = {a.x * s, a.y * s, a.z * s, a.w * s}; return r; }
static inline v4 mul4(v4 a, v4 b) { v4 r = {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; return r; }

/* K/utils.h:6-14 — note /256 */
static v4 color_from_argb(unsigned argb) {
    v4 c;
    c.w = (float)((argb >> 24) & 0xFF) / 256.0f;
    c.x = (float)((argb >> 16) & 0xFF) / 256.0f;
    c.y = (float)((argb >> 8) & 0xFF) / 256.0f;
    c.z = (float)(argb & 0xFF) / 256.0f;
    return c;
}

/* K/textureAtlas.h:10-28 + the image-array read contract of rt_math.h */
static v4 atlas_read_uv(const OracleScene* s, float u, float v, int location, int size) {
    int width = (size >> 16) & 0xFFFF, height = size & 0xFFFF;
    v = 1 - v;
    int x = rt_clampi((int)((u - EPS) * width), 0, width - 1);
    int y = rt_clampi((int)((v - EPS) * height), 0, height - 1);
    x += ((location >> 22) & 0x1FF) * 16;
    y += ((location >> 13) & 0x1FF) * 16;
    int d = location & 0x7FFFF;
    x = rt_clampi(x, 0, s->atlas_w - 1);
    y = rt_clampi(y, 0, s->atlas_h - 1);
    d = rt_clampi(d, 0, s->atlas_layers - 1);
    const uint8_t* t = s->atlas + 4 * (((size_t)d * s->atlas_h + y) * s->atlas_w + x);
    COUNT(texel, 1);
    v4 c = {rt_unorm8(t[0]), rt_unorm8(t[1]), rt_unorm8(t[2]), rt_unorm8(t[3])};
    return c;
}

/* K/material.h:31-40 + :42-82 */
static int material_sample(const OracleScene* s, int material, Record* rec, float u, float v) {
    const int32_t* m = s->material_palette + material;
    unsigned flags = m[0], tint = m[1], tex_size = m[2], color_w = m[3], ne = m[4];
    COUNT(mat, 1);
    v4 color = (flags & 4) ? atlas_read_uv(s, u, v, (int)color_w, (int)tex_size) : color_from_argb(color_w);
    if (!(color.w > EPS)) return 0;
    rec->color = color;
    switch (tint >> 24) {
        case 0xFF: rec->color = mul4(rec->color, color_from_argb(tint)); break;
        case 1: rec->color = mul4(rec->color, color_from_argb(0xFF71A74Du)); break;
        case 2: rec->color = mul4(rec->color, color_from_argb(0xFF8EB971u)); break;
        case 3: rec->color = mul4(rec->color, color_from_argb(0xFF3F76E4u)); break;
        default: break;
    }
    if (flags & 2)
        rec->emittance = atlas_read_uv(s, u, v, (int)ne, (int)tex_size).w;
    else
        rec->emittance = (float)((ne & 0xFF) / 255.0); /* double site, material.h:79 */
    rec->spec = m[5];
    return 1;
}

/* ------------------------------------------------------------------------ block models ----- */
/* K/primitives.h:200-260.  +z faces leave `mat` unset in the reference (primitives.h:209-234,
 * SURVEY.md Appendix B#9).  LLVM resolves that undef to the EAST material (the dead first store
 * to `mat` is removed, then select(x==1, me, undef) folds to me) — observed in oracle/_ref at -O2
 * and pinned by tests/test_oracle_pinning.py::test_aabb_plus_z_face — so +z = east material,
 * flags 0 is the definition used here and in the HIP kernels. */
static float textured_box(const int32_t* model, float best, v3 o, v3 dir, v3 inv, v3* normal, float* u,
                          float* v, int* material) {
    Box b;
    memcpy(&b, model, 24);
    int flags_all = model[6];
    v3 n;
    float tu, tv;
    float dist = box_full(&b, o, dir, inv, &n, &tu, &tv, 1);
    if (rt_isnan(dist) || dist >= best || dist < -EPS) return rt_nan();
    int mat = model[8], flags = 0;
    if (n.z == -1) { mat = model[7]; flags = flags_all; }
    if (n.x == 1) { mat = model[8]; flags = flags_all >> 4; }
    if (n.z == -1) { mat = model[9]; flags = flags_all >> 8; }
    if (n.x == -1) { mat = model[10]; flags = flags_all >> 12; }
    if (n.y == 1) { mat = model[11]; flags = flags_all >> 16; }
    if (n.y == -1) { mat = model[12]; flags = flags_all >> 20; }
    if (flags & 8) return rt_nan();
    if (flags & 4) tu = 1 - tu;
    if (flags & 2) tv = 1 - tv;
    if (flags & 1) { float t = tu; tu = tv; tv = t; }
    *material = mat;
    *normal = n;
    *u = tu;
    *v = tv;
    return dist;
}

/* K/primitives.h:274-319 */
static float quad_hit(const int32_t* q, float best, v3 o, v3 dir, v3* normal, float* u, float* v) {
    float f[13];
    memcpy(f, q, sizeof f);
    v3 qo = V3(f[0], f[1], f[2]), xv = V3(f[3], f[4], f[5]), yv = V3(f[6], f[7], f[8]);
    v3 n = normalize3(cross3(xv, yv));
    float denom = dot3(dir, n);
    if (denom < -EPS) {
        float t = -(dot3(o, n) - dot3(n, qo)) / denom;
        if (t > -EPS && t < best) {
            v3 pt = sub3(add3(o, scale3(dir, t)), qo);
            float uu = dot3(pt, xv) / dot3(xv, xv);
            float vv = dot3(pt, yv) / dot3(yv, yv);
            if (uu >= 0 && uu <= 1 && vv >= 0 && vv <= 1) {
                *u = f[9] + (uu * f[10]);
                *v = f[11] + (vv * f[12]);
                *normal = n;
                return t;
            }
        }
    }
    return rt_nan();
}

/* K/block.h:30-118 */
static float intersect_block(const OracleScene* s, int block, int bx, int by, int bz, Record* rec, v3 pos,
                             v3 dir, v3 inv) {
    if (block == ANY_TYPE) return rt_nan();
    int type = s->block_palette[block], ptr = s->block_palette[block + 1];
    COUNT(block, 1);
    v3 no = sub3(sub3(pos, scale3(dir, OFFSET)), V3((float)bx, (float)by, (float)bz));
    v3 normal = V3(0, 0, 0);
    float u = 0, v = 0;
    switch (type) {
        case 1: {
            Box unit = {0, 1, 0, 1, 0, 1};
            float dist = box_full(&unit, no, pos, inv, &normal, &u, &v, 0); /* dir := pos, block.h:52 */
            if (rt_isnan(dist)) return rt_nan();
            rec->normal = normal;
            return material_sample(s, ptr, rec, u, v) ? dist - OFFSET : rt_nan();
        }
        case 2: {
            int hit = 0, material = 0;
            float dist = rt_inf();
            int boxes = s->aabb_models[ptr];
            COUNT(model_hdr, 1);
            for (int i = 0; i < boxes; i++) {
                COUNT(aabb, 1);
                float t = textured_box(s->aabb_models + ptr + 1 + i * 13, dist, no, dir, inv, &normal, &u, &v, &material);
                if (!rt_isnan(t) && material_sample(s, material, rec, u, v)) {
                    rec->normal = normal;
                    dist = t;
                    hit = 1;
                }
            }
            return hit ? dist : rt_nan();
        }
        case 3: {
            int hit = 0;
            float dist = rt_inf();
            int quads = s->quad_models[ptr];
            COUNT(model_hdr, 1);
            for (int i = 0; i < quads; i++) {
                const int32_t* q = s->quad_models + ptr + 1 + i * 15;
                COUNT(quad, 1);
                float t = quad_hit(q, dist, no, dir, &normal, &u, &v);
                if (!rt_isnan(t) && material_sample(s, q[13], rec, u, v)) {
                    rec->normal = normal;
                    dist = t;
                    hit = 1;
                }
            }
            return hit ? dist : rt_nan();
        }
        default: return rt_nan();
    }
}

/* --------------------------------------------------------------------------- octree -------- */
static inline int ifloor(float x) { return (int)rt_floor(x); } /* K/utils.h:16-19 */

/* K/octree.h:41-109 */
static int octree_intersect(const OracleScene* s, const Path* p, Record* rec, int draw_depth) {
    const int32_t* tree = s->octree;
    int depth = s->octree_depth;
    v3 o = p->origin, d = p->direction;
    float dist_march = 0;
    v3 inv = V3(1 / d.x, 1 / d.y, 1 / d.z);
    v3 off = scale3(d, OFFSET);
    int lx = ifloor(o.x) >> depth, ly = ifloor(o.y) >> depth, lz = ifloor(o.z) >> depth;
    if ((lx != 0) | (ly != 0) | (lz != 0)) {
        float size = (float)(1 << depth);
        Box world = {0, size, 0, size, 0, size};
        float dist = box_quick(&world, o, inv);
        if (rt_isnan(dist) || dist < 0) return 0;
        dist_march += dist + OFFSET;
    }
    for (int i = 0; i < draw_depth; i++) {
        if (dist_march > rec->distance) return 0;
        v3 pos = add3(o, scale3(d, dist_march));
        v3 po = add3(pos, off);
        int bx = ifloor(po.x), by = ifloor(po.y), bz = ifloor(po.z);
        if (((bx >> depth) != 0) | ((by >> depth) != 0) | ((bz >> depth) != 0)) return 0;
        COUNT(steps, 1);
        int level = depth;
        int data = tree[0];
        COUNT(node, 1);
        while (data > 0) {
            level--;
            data = tree[data + ((((bx >> level) & 1) << 2) | (((by >> level) & 1) << 1) | ((bz >> level) & 1))];
            COUNT(node, 1);
        }
        data = -data;
        lx = bx >> level; ly = by >> level; lz = bz >> level;
        if (data != p->ray_material) {
            float dist = intersect_block(s, data, bx, by, bz, rec, pos, d, inv);
            if (!rt_isnan(dist)) {
                rec->distance = dist_march + dist;
                rec->material = data;
                return 1;
            }
        }
        Box leaf = {(float)(lx << level), (float)((lx + 1) << level), (float)(ly << level),
                    (float)((ly + 1) << level), (float)(lz << level), (float)((lz + 1) << level)};
        dist_march += box_exit(&leaf, po, inv) + OFFSET;
    }
    return 0;
}

/* ----------------------------------------------------------------------------- BVH --------- */
/* K/primitives.h:335-409 */
static float triangle_hit(const int32_t* t, float best, v3 o, v3 dir, v3* normal, float* u, float* v, int* material) {
    float f[19];
    memcpy(f, t + 1, sizeof f);
    int flags = t[0];
    v3 e1 = V3(f[0], f[1], f[2]), e2 = V3(f[3], f[4], f[5]), to = V3(f[6], f[7], f[8]);
    v3 pvec = cross3(dir, e2);
    float det = dot3(e1, pvec);
    if ((flags >> 8) & 1) {
        if (det > -EPS && det < EPS) return rt_nan();
    } else if (det > -EPS) {
        return rt_nan();
    }
    float recip = 1 / det;
    v3 tvec = sub3(o, to);
    float uu = dot3(tvec, pvec) * recip;
    if (uu < 0 || uu > 1) return rt_nan();
    v3 qvec = cross3(tvec, e1);
    float vv = dot3(dir, qvec) * recip;
    if (vv < 0 || (uu + vv) > 1) return rt_nan();
    float tt = dot3(e2, qvec) * recip;
    if (tt > EPS && tt < best) {
        float w = 1 - uu - vv;
        *u = f[12] * uu + f[14] * vv + f[16] * w;
        *v = f[13] * uu + f[15] * vv + f[17] * w;
        *normal = V3(f[9], f[10], f[11]);
        *material = t[19];
        return tt;
    }
    return rt_nan();
}

static inline float as_f(int32_t i) { return rt_u2f((unsigned)i); }

/* EXTENSION (CHUNKY_OPT_BVH_CULL_BEHIND, default off): a child whose box lies entirely behind the ray origin — AABB_exit
 * (K/primitives.h:52-61) negative — counts as missed.  The reference walks such boxes (its quick test only asks whether the
 * ray's LINE pierces the box before the current hit); this is the specification of the option, not of the reference. */
static int g_bvh_cull = 0;
void port_set_bvh_cull(int on) { g_bvh_cull = on; }

/* K/bvh.h:22-113 */
static int bvh_intersect(const OracleScene* s, const int32_t* bvh, const Path* p, Record* rec) {
    if (bvh[0] == 0 && rt_isnan(as_f(bvh[1])) && rt_isnan(as_f(bvh[2])) && rt_isnan(as_f(bvh[3])) &&
        rt_isnan(as_f(bvh[4])) && rt_isnan(as_f(bvh[5])) && rt_isnan(as_f(bvh[6])))
        return 0;
    const int32_t* trigs = s->bvh_trigs;
    int hit = 0, to_visit = 0, cur = 0;
    int stack[64];
    v3 o = p->origin, d = p->direction;
    v3 inv = V3(1 / d.x, 1 / d.y, 1 / d.z);
    for (;;) {
        int head = bvh[cur];
        if (head <= 0) {
            int prim = -head;
            int n = trigs[prim];
            COUNT(leaf_hdr, 1);
            for (int i = 0; i < n; i++) {
                v3 normal;
                float u, v;
                int material;
                COUNT(tri, 1);
                float dist = triangle_hit(trigs + prim + 1 + 20 * i, rec->distance, o, d, &normal, &u, &v, &material);
                if (!rt_isnan(dist) && material_sample(s, material, rec, u, v)) {
                    rec->normal = normal;
                    rec->distance = dist;
                    hit = 1;
                }
            }
            if (to_visit == 0) break;
            cur = stack[--to_visit];
        } else {
            int second = head;
            Box b1, b2;
            memcpy(&b1, bvh + cur + 7 + 1, 24);
            memcpy(&b2, bvh + second + 1, 24);
            COUNT(bvh_inner, 1);
            float t1 = box_quick(&b1, o, inv);
            float t2 = box_quick(&b2, o, inv);
            int miss1 = rt_isnan(t1) || t1 > rec->distance;
            int miss2 = rt_isnan(t2) || t2 > rec->distance;
            if (g_bvh_cull) {
                if (box_exit(&b1, o, inv) < 0) miss1 = 1;
                if (box_exit(&b2, o, inv) < 0) miss2 = 1;
            }
            if (miss1) {
                if (miss2) {
                    if (to_visit == 0) break;
                    cur = stack[--to_visit];
                } else {
                    cur = second;
                }
            } else if (miss2) {
                cur += 7;
            } else if (t1 < t2) {
                stack[to_visit++] = second;
                cur += 7;
            } else {
                stack[to_visit++] = cur + 7;
                cur = second;
            }
        }
    }
    return hit;
}

/* K/kernel.h:14-24 */
static int closest_intersect(const OracleScene* s, const Path* p, Record* rec, int draw_depth) {
    COUNT(traces, 1);
    int hit = 0;
    hit |= octree_intersect(s, p, rec, draw_depth);
    hit |= bvh_intersect(s, s->world_bvh, p, rec);
    hit |= bvh_intersect(s, s->actor_bvh, p, rec);
    if (hit) rec->point = add3(p->origin, scale3(p->direction, rec->distance - OFFSET));
    return hit;
}

/* ----------------------------------------------------------------------- sun and sky ------- */
/* K/sky.h:19-40 */
static Sun sun_new(const int32_t* data) {
    Sun sun;
    sun.flags = data[0];
    sun.texture_size = data[1];
    sun.texture = data[2];
    sun.intensity = as_f(data[3]);
    float phi = as_f(data[4]), theta = as_f(data[5]);
    float r = rt_fabs(rt_cos(phi));
    sun.sw = V3(rt_cos(theta) * r, rt_sin(phi), rt_sin(theta) * r);
    sun.su = (rt_fabs(sun.sw.x) > 0.1f) ? V3(0, 1, 0) : V3(1, 0, 0);
    sun.sv = normalize3(cross3(sun.sw, sun.su));
    sun.su = cross3(sun.sv, sun.sw);
    return sun;
}

/* K/sky.h:97-106 + the linear / mirrored-repeat sampler contract of rt_math.h */
static void sky_intersect(const OracleScene* s, const Path* p, Record* rec) {
    v3 d = p->direction;
    float theta = rt_atan2(d.z, d.x);
    theta /= RT_PI_F * 2;
    theta = rt_fmod1(rt_fmod1(theta) + 1);
    float phi = (rt_asin(rt_clamp(d.y, -1.0f, 1.0f)) + RT_PI_2_F) * RT_1_PI_F;
    int i0, i1, j0, j1;
    float a, b;
    rt_mirror_linear(theta, s->sky_w, &i0, &i1, &a);
    rt_mirror_linear(phi, s->sky_h, &j0, &j1, &b);
    const uint8_t* t00 = s->sky + 4 * ((size_t)j0 * s->sky_w + i0);
    const uint8_t* t10 = s->sky + 4 * ((size_t)j0 * s->sky_w + i1);
    const uint8_t* t01 = s->sky + 4 * ((size_t)j1 * s->sky_w + i0);
    const uint8_t* t11 = s->sky + 4 * ((size_t)j1 * s->sky_w + i1);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float c[4];
    for (int k = 0; k < 4; k++)
        c[k] = w00 * rt_unorm8(t00[k]) + w10 * rt_unorm8(t10[k]) + w01 * rt_unorm8(t01[k]) + w11 * rt_unorm8(t11[k]);
    COUNT(sky, 1);
    rec->color.x = c[0] * s->sky_intensity;
    rec->color.y = c[1] * s->sky_intensity;
    rec->color.z = c[2] * s->sky_intensity;
    rec->color.w = c[3] * s->sky_intensity;
}

/* K/sky.h:42-66 */
static void sun_intersect(const OracleScene* s, const Sun* sun, const Path* p, Record* rec) {
    v3 d = p->direction;
    if (!(sun->flags & 1) || dot3(d, sun->sw) < 0.5f) return;
    float radius = 0.03f; /* `float radius = 0.03;` */
    float width = radius * 4;
    float width2 = width * 2;
    float a = RT_PI_2_F - rt_acos(dot3(d, sun->su)) + width;
    if (a >= 0 && a < width2) {
        float b = RT_PI_2_F - rt_acos(dot3(d, sun->sv)) + width;
        if (b >= 0 && b < width2) {
            v4 c = atlas_read_uv(s, a / width2, b / width2, sun->texture, sun->texture_size);
            c = scale4(c, sun->intensity);
            rec->color.x += c.x; rec->color.y += c.y; rec->color.z += c.z; rec->color.w += c.w;
        }
    }
}

/* K/sky.h:68-93 — note direction = u * v (component-wise), then += w */
static int sun_sample_direction(const Sun* sun, Path* p, Record* rec, unsigned* state) {
    if (!(sun->flags & 1)) return 0;
    float radius_cos = rt_cos(0.03f);
    float x1 = rt_pcg_float(state), x2 = rt_pcg_float(state);
    float cos_a = 1 - x1 + x1 * radius_cos;
    float sin_a = rt_sqrt(1 - cos_a * cos_a);
    float phi = 2 * RT_PI_F * x2;
    float sp, cp;
    rt_sincos(phi, &sp, &cp);
    v3 u = scale3(sun->su, cp * sin_a);
    v3 v = scale3(sun->sv, sp * sin_a);
    v3 w = scale3(sun->sw, cos_a);
    v3 dir = mul3(u, v);
    dir = add3(dir, w);
    dir = normalize3(dir);
    p->direction = dir;
    rec->emittance = rt_fabs(dot3(dir, rec->normal));
    return 1;
}

/* K/kernel.h:26-31 */
static void intersect_sky(const OracleScene* s, const Sun* sun, Path* p, Record* rec) {
    sky_intersect(s, p, rec);
    sun_intersect(s, sun, p, rec);
    v3 c = V3(rec->color.x, rec->color.y, rec->color.z);
    p->color = add3(p->color, scale3(mul3(c, p->throughput), rec->emittance));
}

/* K/kernel.h:33-44 */
static void apply_ray_color(Path* p, Record* rec, float emitter_scale) {
    p->origin = rec->point;
    v3 c = V3(rec->color.x, rec->color.y, rec->color.z);
    p->throughput = mul3(p->throughput, c);
    v3 e = scale3(c, rec->emittance * emitter_scale);
    p->color = add3(p->color, mul3(e, p->throughput));
}

/* K/kernel.h:46-98 */
static int next_path(Path* p, Record* rec, unsigned* state, int max_depth) {
    p->origin = rec->point;
    float x1 = rt_pcg_float(state), x2 = rt_pcg_float(state);
    float r = rt_sqrt(x1);
    float theta = 2 * RT_PI_F * x2;
    float st, ct;
    rt_sincos(theta, &st, &ct);
    float tx = r * ct, ty = r * st, tz = rt_sqrt(1 - x1);
    v3 n = rec->normal;
    float xx, xy, xz = 0;
    if ((double)rt_fabs(n.x) > 0.1) { xx = 0; xy = 1; } else { xx = 1; xy = 0; } /* double compare, kernel.h:66 */
    float ux = xy * n.z - xz * n.y;
    float uy = xz * n.x - xx * n.z;
    float uz = xx * n.y - xy * n.x;
    r = 1 / rt_sqrt(ux * ux + uy * uy + uz * uz);
    ux *= r; uy *= r; uz *= r;
    float vx = uy * n.z - uz * n.y;
    float vy = uz * n.x - ux * n.z;
    float vz = ux * n.y - uy * n.x;
    p->direction.x = ux * tx + vx * ty + n.x * tz;
    p->direction.y = uy * tx + vy * ty + n.y * tz;
    p->direction.z = uz * tx + vz * ty + n.z * tz;
    p->origin = add3(p->origin, scale3(p->direction, OFFSET));
    p->ray_depth += 1;
    rec->distance = rt_inf();
    return p->ray_depth < max_depth;
}

/* ------------------------------------------------------------------------- camera ---------- */
/* K/rayTracer.cl:55-91 + K/camera.h:8-32.  `normalize_dir` is the preview kernel's extra
 * normalize (rayTracer.cl:186). */
static void primary_ray(const OracleScene* s, int gid, unsigned* state, Path* p, int normalize_dir) {
    if (s->projector_type != -1) {
        const float* cs = s->camera_settings;
        float half_width = (float)(s->width / (2.0 * s->height));
        float inv_height = (float)(1.0 / s->height);
        float x = -half_width + ((float)(gid % s->width) + rt_pcg_float(state)) * inv_height;
        float y = (float)(-0.5 + (double)(((float)(gid / s->width) + rt_pcg_float(state)) * inv_height));
        float aperture = cs[12], subject = cs[13], fov_tan = cs[14];
        v3 o = V3(0, 0, 0);
        v3 d = V3(fov_tan * x, fov_tan * y, 1.0f);
        if (aperture > 0) {
            d = scale3(d, subject / d.z);
            float r = rt_sqrt(rt_pcg_float(state)) * aperture;
            float theta = (float)((double)(rt_pcg_float(state) * RT_PI_F) * 2.0);
            float st, ct;
            rt_sincos(theta, &st, &ct);
            float rx = ct * r, ry = st * r;
            d = sub3(d, V3(rx, ry, 0));
            o = add3(o, V3(rx, ry, 0));
        }
        if (normalize_dir) d = normalize3(d);
        v3 m1 = V3(cs[3], cs[4], cs[5]), m2 = V3(cs[6], cs[7], cs[8]), m3 = V3(cs[9], cs[10], cs[11]);
        p->direction = V3(dot3(m1, d), dot3(m2, d), dot3(m3, d));
        p->origin = add3(V3(dot3(m1, o), dot3(m2, o), dot3(m3, o)), V3(cs[0], cs[1], cs[2]));
    } else {
        const float* r = s->camera_settings + (size_t)gid * 6;
        p->origin = V3(r[0], r[1], r[2]);
        p->direction = V3(r[3], r[4], r[5]);
    }
}

static void path_init(Path* p, Record* rec) {
    p->color = V3(0, 0, 0);
    p->throughput = V3(1, 1, 1);
    p->origin = V3(0, 0, 0);
    p->direction = V3(0, 0, 0);
    p->ray_material = 0;
    p->ray_depth = 0;
    memset(rec, 0, sizeof *rec);
    rec->distance = rt_inf();
}

static void put_hit(OracleHit* h, int hit, const Record* r) {
    h->hit = hit;
    h->material = r->material;
    h->distance = r->distance;
    h->normal[0] = r->normal.x; h->normal[1] = r->normal.y; h->normal[2] = r->normal.z;
    h->color[0] = r->color.x; h->color[1] = r->color.y; h->color[2] = r->color.z; h->color[3] = r->color.w;
    h->emittance = r->emittance;
    h->point[0] = r->point.x; h->point[1] = r->point.y; h->point[2] = r->point.z;
}

/* ---- helper-level known answers: the restatement's counterparts of the reference helpers oracle/ref_shim.cpp ref_helpers
 * drives, on the same rows (row layout and `which` numbering: see there; tests/golden/helpers.npz holds the reference's
 * answers). */
#define HELPER_IN 32
#define HELPER_OUT 12
static inline int f2i(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float i2f(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline v3 ld3(const float* p) { return V3(p[0], p[1], p[2]); }
static inline v3 rcp3v(v3 d) { return V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
void port_helpers(const OracleScene* s, int which, int n, const float* in_rows, float* out_rows) {
    const Sun sun = sun_new(s->sun);
    for (int r = 0; r < n; r++) {
        const float* in = in_rows + (size_t)r * HELPER_IN;
        float* out = out_rows + (size_t)r * HELPER_OUT;
        for (int k = 0; k < HELPER_OUT; k++) out[k] = 0;
        Path p;
        Record rec;
        path_init(&p, &rec);
        v3 nrm = V3(0, 0, 0);
        float u = 0, v = 0;
        switch (which) {
            case 0: case 1: {
                Box b = {in[0], in[1], in[2], in[3], in[4], in[5]};
                out[0] = which == 0 ? box_quick(&b, ld3(in + 6), rcp3v(ld3(in + 9))) : box_exit(&b, ld3(in + 6), rcp3v(ld3(in + 9)));
                break;
            }
            case 2: case 3: {
                Box unit = {0, 1, 0, 1, 0, 1}, b = {in[0], in[1], in[2], in[3], in[4], in[5]};
                out[0] = which == 2 ? box_full(&unit, ld3(in + 6), ld3(in + 9), rcp3v(ld3(in + 12)), &nrm, &u, &v, 0)
                                    : box_full(&b, ld3(in + 6), ld3(in + 9), rcp3v(ld3(in + 9)), &nrm, &u, &v, 1);
                out[1] = nrm.x; out[2] = nrm.y; out[3] = nrm.z; out[4] = u; out[5] = v;
                break;
            }
            case 4: {
                out[0] = intersect_block(s, f2i(in[0]), (int)in[1], (int)in[2], (int)in[3], &rec, ld3(in + 4), ld3(in + 7), rcp3v(ld3(in + 7)));
                out[1] = rec.normal.x; out[2] = rec.normal.y; out[3] = rec.normal.z;
                out[4] = rec.color.x; out[5] = rec.color.y; out[6] = rec.color.z; out[7] = rec.color.w;
                out[8] = rec.emittance;
                break;
            }
            case 6: {
                int32_t tri[20];
                memcpy(tri, in, sizeof tri);
                int mat = 0;
                out[0] = triangle_hit(tri, in[26], ld3(in + 20), ld3(in + 23), &nrm, &u, &v, &mat);
                out[1] = nrm.x; out[2] = nrm.y; out[3] = nrm.z; out[4] = u; out[5] = v; out[6] = i2f(mat);
                break;
            }
            case 7: {
                unsigned state = (unsigned)f2i(in[0]);
                rec.normal = ld3(in + 1);
                sun_sample_direction(&sun, &p, &rec, &state);
                out[0] = p.direction.x; out[1] = p.direction.y; out[2] = p.direction.z; out[3] = rec.emittance; out[4] = i2f((int)state);
                break;
            }
            case 8: {
                p.direction = ld3(in);
                rec.color.x = in[3]; rec.color.y = in[4]; rec.color.z = in[5]; rec.color.w = in[6];
                const v4 before = rec.color;
                sun_intersect(s, &sun, &p, &rec);
                out[0] = rec.color.x; out[1] = rec.color.y; out[2] = rec.color.z; out[3] = rec.color.w;
                out[4] = memcmp(&before, &rec.color, sizeof before) ? 1.0f : 0.0f;  /* (the restatement returns nothing: "added" stands in) */
                break;
            }
            case 9: {
                p.direction = ld3(in);
                sky_intersect(s, &p, &rec);
                out[0] = rec.color.x; out[1] = rec.color.y; out[2] = rec.color.z; out[3] = rec.color.w;
                break;
            }
            case 10: {
                unsigned state = (unsigned)f2i(in[0]);
                rec.normal = ld3(in + 1);
                rec.point = ld3(in + 4);
                next_path(&p, &rec, &state, 5);
                out[0] = p.direction.x; out[1] = p.direction.y; out[2] = p.direction.z;
                out[3] = p.origin.x; out[4] = p.origin.y; out[5] = p.origin.z; out[6] = i2f((int)state);
                break;
            }
            case 11: {
                const v4 c = atlas_read_uv(s, in[0], in[1], f2i(in[2]), f2i(in[3]));
                out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
                break;
            }
            case 12: {
                out[0] = material_sample(s, f2i(in[0]), &rec, in[1], in[2]) ? 1.0f : 0.0f;
                out[1] = rec.color.x; out[2] = rec.color.y; out[3] = rec.color.z; out[4] = rec.color.w;
                out[5] = rec.emittance;
                break;
            }
            case 14: case 15: {
                p.origin = ld3(in);
                p.direction = ld3(in + 3);
                if (which == 15) rec.distance = in[6];
                const int hit = which == 14 ? octree_intersect(s, &p, &rec, 256) : bvh_intersect(s, s->world_bvh, &p, &rec);
                int o = 0;
                out[o++] = hit ? 1.0f : 0.0f;
                out[o++] = rec.distance;
                if (which == 14) out[o++] = i2f(rec.material);
                out[o++] = rec.normal.x; out[o++] = rec.normal.y; out[o++] = rec.normal.z;
                out[o++] = rec.color.x; out[o++] = rec.color.y; out[o++] = rec.color.z; out[o++] = rec.color.w;
                out[o++] = rec.emittance;
                break;
            }
            default: break;
        }
    }
}

/* The three constants of the render loop (K/rayTracer.cl:94 drawDepth 256, :99 emitter factor 13, :107 ray depth 5).
 * The HIP library exposes them as options (CHUNKY_OPT_DRAW_DEPTH / _EMITTER_SCALE / _MAX_DEPTH); the checker follows
 * so that non-default values can be compared too.  Defaults = the reference. */
static int g_draw_depth = 256, g_max_depth = 5;
static float g_emitter_scale = 13.0f;
void port_set_options(int draw_depth, int max_depth, float emitter_scale) {
    g_draw_depth = draw_depth;
    g_max_depth = max_depth;
    g_emitter_scale = emitter_scale;
}

/* One sample: K/rayTracer.cl:40-107.  Returns pixel.color; optionally records each trace. */
static v3 trace_sample(const OracleScene* s, const Sun* sun, int seed, int gid, OracleHit* hits, int* n_hits) {
    Path p;
    Record rec;
    path_init(&p, &rec);
    unsigned state = (unsigned)seed + (unsigned)gid;
    rt_pcg_next(&state);
    primary_ray(s, gid, &state, &p, 0);
    int n = 0;
    COUNT(samples, 1);
    do {
        int hit = closest_intersect(s, &p, &rec, g_draw_depth);
        if (hits) put_hit(&hits[n++], hit, &rec);
        if (!hit) {
            rec.emittance = 1;
            intersect_sky(s, sun, &p, &rec);
            break;
        }
        COUNT(hits, 1);
        apply_ray_color(&p, &rec, g_emitter_scale);
        if (sun_sample_direction(sun, &p, &rec, &state)) {
            Record shadow = rec; /* IntersectionRecord_copy, wavefront.h:64-78 */
            shadow.point = rec.normal; /* the copy's dead `point = normal` (wavefront.h:73) */
            int sh = closest_intersect(s, &p, &shadow, g_draw_depth);
            if (hits) put_hit(&hits[n++], sh, &shadow);
            if (!sh) intersect_sky(s, sun, &p, &shadow);
        }
    } while (next_path(&p, &rec, &state, g_max_depth));
    if (n_hits) *n_hits = n;
    return p.color;
}

/* ------------------------------------------------------------------------- extensions ------ */
/* SURVEY.md section 8 row f2 / DESIGN.md section 9: light-transport options the north star names and the reference does
 * not have.  EXPERIMENTAL — there is no reference implementation to pin them to; this restatement IS their
 * specification, the HIP kernels must reproduce it bit for bit, and tests/test_extensions.py checks it analytically
 * (energy conservation, mirror limit, NEE on/off means).  With every option at its default the reference path above runs.
 *
 *   sun_sampling  -1 as the reference (sample the sun iff PackedSun flag bit 0, K/sky.h:69), 0 never, 1 always — Chunky's
 *                 sunEnabled, separated from drawTexture (which keeps gating the sun DISC, K/sky.h:45)
 *   emitters      1 as the reference (emittance x emitter_scale at every hit, K/kernel.h:39-43), 0 Chunky's emittersEnabled = false
 *   bsdf          1: material word 5 = spec | metal << 8 | rough << 16 (PackedMaterial.java:69-71, loaded and ignored by
 *                 K/material.h:38).  With probability max(spec, metal) / 255 a hit reflects specularly: throughput *=
 *                 lerp(1, colour, metal); direction = reflect(d, n), blended with a cosine sample by `rough`; no sun / emitter
 *                 sampling at that vertex.  Otherwise the diffuse path of the reference.
 *   nee           1: at every diffuse vertex one emitter block (uniform over the scene's emitter list), one of its six
 *                 faces and a point on it are drawn; if the faces see each other and the segment is free, its radiance
 *                 colour^2 x emittance x emitter_scale (what the implicit path adds on hitting it, K/kernel.h:39-43) is
 *                 added with the area-measure weight cos cos / (pi d^2) x 6 N A.  The bounce ray that follows does not count
 *                 an emitter it hits again; the last vertex of a path (whose bounce ray is never traced) samples none.
 * Random draws per vertex, in this order: [1: specular?  only if max(spec, metal) > 0] then specular: [2 if rough > 0];
 * diffuse: [2 sun if on] [4 emitter if on] 2 bounce. */
typedef struct {
    int sun_sampling, emitters, bsdf, nee;
} Ext;
static Ext g_ext = {-1, 1, 0, 0};
void port_set_ext(int sun_sampling, int emitters, int bsdf, int nee) {
    g_ext.sun_sampling = sun_sampling;
    g_ext.emitters = emitters;
    g_ext.bsdf = bsdf;
    g_ext.nee = nee;
}
static int ext_active(void) { return g_ext.sun_sampling != -1 || g_ext.emitters != 1 || g_ext.bsdf != 0 || g_ext.nee != 0; }

/* The emitter list: every octree leaf whose block is a full cube (model type 1) with a non-zero emittance byte and no
 * emittance texture, in pre-order (children in index order), as {x, y, z, level << 25 | block pointer}: one box of edge
 * 2^level per leaf.  Returns the number found (may exceed cap; only cap are written). */
static int list_emitters(const OracleScene* s, int node, int x, int y, int z, int level, int32_t* out, int cap, int n) {
    int v = s->octree[node];
    if (v > 0) {
        for (int c = 0; c < 8; c++) {
            int h = 1 << (level - 1);
            n = list_emitters(s, v + c, x + ((c >> 2) & 1) * h, y + ((c >> 1) & 1) * h, z + (c & 1) * h, level - 1, out, cap, n);
        }
        return n;
    }
    int block = -v;
    if (block == 0 || block == ANY_TYPE) return n;
    if (s->block_palette[block] != 1) return n;
    const int32_t* m = s->material_palette + s->block_palette[block + 1];
    if ((m[0] & 2) || (m[4] & 0xFF) == 0) return n;
    if (n < cap) {
        out[4 * n] = x; out[4 * n + 1] = y; out[4 * n + 2] = z;
        out[4 * n + 3] = (level << 25) | block;
    }
    return n + 1;
}
int port_list_emitters(const OracleScene* s, int32_t* out4, int cap) {
    return list_emitters(s, 0, 0, 0, 0, s->octree_depth, out4, cap, 0);
}
static const int32_t* g_emitters = 0;
static int g_n_emitters = 0;
void port_use_emitters(const int32_t* list4, int n) { g_emitters = list4; g_n_emitters = n; }

/* cosine-weighted direction about n from two draws: the direction part of nextPath (K/kernel.h:52-90) */
static v3 cosine_direction(v3 n, float x1, float x2) {
    float r = rt_sqrt(x1);
    float theta = 2 * RT_PI_F * x2;
    float st, ct;
    rt_sincos(theta, &st, &ct);
    float tx = r * ct, ty = r * st, tz = rt_sqrt(1 - x1);
    float xx, xy, xz = 0;
    if ((double)rt_fabs(n.x) > 0.1) { xx = 0; xy = 1; } else { xx = 1; xy = 0; }
    float ux = xy * n.z - xz * n.y;
    float uy = xz * n.x - xx * n.z;
    float uz = xx * n.y - xy * n.x;
    r = 1 / rt_sqrt(ux * ux + uy * uy + uz * uz);
    ux *= r; uy *= r; uz *= r;
    float vx = uy * n.z - uz * n.y;
    float vy = uz * n.x - ux * n.z;
    float vz = ux * n.y - uy * n.x;
    return V3(ux * tx + vx * ty + n.x * tz, uy * tx + vy * ty + n.y * tz, uz * tx + vz * ty + n.z * tz);
}

static v3 trace_sample_ext(const OracleScene* s, const Sun* sun, const Ext* x, int seed, int gid) {
    Path p;
    Record rec;
    path_init(&p, &rec);
    unsigned state = (unsigned)seed + (unsigned)gid;
    rt_pcg_next(&state);
    primary_ray(s, gid, &state, &p, 0);
    COUNT(samples, 1);
    int after_nee = 0; /* the vertex before sampled the emitters: one found by the bounce ray is not counted twice */
    for (;;) {
        int hit = closest_intersect(s, &p, &rec, g_draw_depth);
        if (!hit) {
            rec.emittance = 1;
            intersect_sky(s, sun, &p, &rec);
            break;
        }
        COUNT(hits, 1);
        const v3 base = p.throughput;
        const v3 c = V3(rec.color.x, rec.color.y, rec.color.z);
        const v3 thr_d = mul3(base, c);
        const v3 point = rec.point, n = rec.normal;
        if (x->emitters && !after_nee)
            p.color = add3(p.color, mul3(scale3(c, rec.emittance * g_emitter_scale), thr_d)); /* K/kernel.h:39-43 */
        after_nee = 0;
        int specular = 0;
        float metal = 0, rough = 0;
        if (x->bsdf) {
            const float spec = rt_unorm8((unsigned)rec.spec & 0xFF);
            metal = rt_unorm8(((unsigned)rec.spec >> 8) & 0xFF);
            rough = rt_unorm8(((unsigned)rec.spec >> 16) & 0xFF);
            const float ps = rt_fmax(spec, metal);
            if (ps > 0) specular = rt_pcg_float(&state) < ps;
        }
        if (specular) {
            p.throughput = V3(base.x * (c.x * metal + (1 - metal)), base.y * (c.y * metal + (1 - metal)), base.z * (c.z * metal + (1 - metal)));
            const v3 d = p.direction;
            v3 refl = sub3(d, scale3(n, 2 * dot3(d, n)));
            if (rough > 0) {
                float x1 = rt_pcg_float(&state), x2 = rt_pcg_float(&state);
                v3 dd = cosine_direction(n, x1, x2);
                refl = normalize3(add3(scale3(dd, rough), scale3(refl, 1 - rough)));
                float rn = dot3(refl, n);
                if (rn < 0) refl = sub3(refl, scale3(n, 2 * rn));
            }
            p.direction = refl;
            p.origin = add3(point, scale3(refl, OFFSET));
        } else {
            p.throughput = thr_d;
            p.origin = point;
            const int sun_on = x->sun_sampling < 0 ? (sun->flags & 1) : x->sun_sampling;
            if (sun_on) { /* Sun_sampleDirection + shadow trace, K/sky.h:68-93, K/rayTracer.cl:101-106 */
                Sun on = *sun;
                on.flags |= 1;
                sun_sample_direction(&on, &p, &rec, &state);
                Record shadow = rec;
                shadow.point = rec.normal;
                if (!closest_intersect(s, &p, &shadow, g_draw_depth)) intersect_sky(s, sun, &p, &shadow);
            }
            /* not at the last vertex: the bounce ray that would find the same light implicitly is never traced there */
            if (x->nee && x->emitters && g_n_emitters > 0 && p.ray_depth + 1 < g_max_depth) {
                const float xk = rt_pcg_float(&state), xf = rt_pcg_float(&state), xu = rt_pcg_float(&state), xv = rt_pcg_float(&state);
                int k = (int)(xk * (float)g_n_emitters);
                if (k > g_n_emitters - 1) k = g_n_emitters - 1;
                int face = (int)(xf * 6.0f);
                if (face > 5) face = 5;
                const int32_t* em = g_emitters + 4 * k;
                const int level = (em[3] >> 25) & 15, block = em[3] & 0x1FFFFFF;
                const float size = (float)(1 << level);
                const float a = xu * size, b = xv * size;
                const float fa = a - rt_floor(a), fb = b - rt_floor(b);
                const float ex = (float)em[0], ey = (float)em[1], ez = (float)em[2];
                v3 pe, nf;
                float tu, tv;
                switch (face) {
                    case 0: pe = V3(ex, ey + a, ez + b); nf = V3(-1, 0, 0); tu = 1 - fb; tv = fa; break;
                    case 1: pe = V3(ex + size, ey + a, ez + b); nf = V3(1, 0, 0); tu = fb; tv = fa; break;
                    case 2: pe = V3(ex + a, ey, ez + b); nf = V3(0, -1, 0); tu = fa; tv = 1 - fb; break;
                    case 3: pe = V3(ex + a, ey + size, ez + b); nf = V3(0, 1, 0); tu = fa; tv = fb; break;
                    case 4: pe = V3(ex + a, ey + b, ez); nf = V3(0, 0, -1); tu = fa; tv = fb; break;
                    default: pe = V3(ex + a, ey + b, ez + size); nf = V3(0, 0, 1); tu = 1 - fa; tv = fb; break;
                }
                const v3 l = sub3(pe, point);
                const float d2 = dot3(l, l);
                const float dist = rt_sqrt(d2);
                const v3 dir = scale3(l, 1 / dist);
                const float cs = dot3(dir, n), cl = -dot3(dir, nf);
                if (cs > 0 && cl > 0 && dist > 0.002f) {
                    Record er;
                    memset(&er, 0, sizeof er);
                    if (material_sample(s, s->block_palette[block + 1], &er, tu, tv) && er.emittance > 0) {
                        Path q = p;
                        q.origin = point;
                        q.direction = dir;
                        Record sh = rec;
                        sh.distance = dist - 0.001f; /* anything nearer than the emitter's face hides it */
                        if (!closest_intersect(s, &q, &sh, g_draw_depth)) {
                            const float w = (cs * cl) / (RT_PI_F * d2) * (6.0f * (float)g_n_emitters * (size * size));
                            const v3 ce = V3(er.color.x, er.color.y, er.color.z);
                            const v3 le = mul3(ce, scale3(ce, er.emittance * g_emitter_scale));
                            p.color = add3(p.color, mul3(thr_d, scale3(le, w)));
                        }
                    }
                }
                after_nee = 1;
            }
            float x1 = rt_pcg_float(&state), x2 = rt_pcg_float(&state);
            p.direction = cosine_direction(n, x1, x2);
            p.origin = add3(point, scale3(p.direction, OFFSET));
        }
        p.ray_depth += 1;
        rec.distance = rt_inf();
        if (!(p.ray_depth < g_max_depth)) break;
    }
    return p.color;
}

/* ------------------------------------------------------------------------ entry points ----- */
/* Host pass loop of OpenClPathTracingRenderer.java:102-144 around the accumulate of
 * K/rayTracer.cl:109-112: pass k uses seeds[k] and bufferSpp = first_spp + k. */
int port_render_passes(const OracleScene* s, const int32_t* seeds, int n_passes, int first_spp,
                       int64_t gid_begin, int64_t gid_end, float* res, int threads) {
    Sun sun = sun_new(s->sun);
    if (threads < 1) threads = 1;
    for (int k = 0; k < n_passes; k++) {
        int seed = seeds[k], spp = first_spp + k;
#pragma omp parallel num_threads(threads)
        {
            Counters local;
            memset(&local, 0, sizeof local);
            t_ctr = g_count_enabled ? &local : 0;
            const Sun my_sun = sun;  /* per-worker copies: see port_render_gids */
            const OracleScene my_scene = *s;
#pragma omp for schedule(dynamic, 256)
            for (int64_t gid = gid_begin; gid < gid_end; gid++) {
                v3 c = ext_active() ? trace_sample_ext(&my_scene, &my_sun, &g_ext, seed, (int)gid) : trace_sample(&my_scene, &my_sun, seed, (int)gid, 0, 0);
                float* px = res + 3 * gid;
                px[0] = (px[0] * spp + c.x) / (spp + 1);
                px[1] = (px[1] * spp + c.y) / (spp + 1);
                px[2] = (px[2] * spp + c.z) / (spp + 1);
            }
            if (g_count_enabled) counters_merge(&local);
            t_ctr = 0;
        }
    }
    return 0;
}

/* Timing runs (bench.py's cpu_baseline leg, tools/cpu_sweep.py): for the duration of one parallel region worker t binds itself
 * to the t-th CPU the calling thread may use (Linux numbers the first hardware thread of every core before the second ones: up
 * to the core count that is one worker per core), and gets its previous mask back at the end.  Without it the leg depends on
 * where the scheduler happens to put (and keeps moving) the workers: on the 8-CPU build VM four unpinned workers ran at a
 * quarter of the pinned rate.  Off by default. */
static int g_pin_threads = 0;
void port_set_pinning(int on) { g_pin_threads = on; }
#if defined(__linux__) && defined(_OPENMP)
typedef struct { int n; int cpu[1024]; } PinPlan;
static void pin_plan(PinPlan* plan) {  /* by the calling thread, before the region */
    plan->n = 0;
    if (!g_pin_threads) return;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    for (int c = 0; c < CPU_SETSIZE && plan->n < 1024; c++)
        if (CPU_ISSET(c, &allowed)) plan->cpu[plan->n++] = c;
}
static int pin_enter(const PinPlan* plan, cpu_set_t* saved) {  /* by every worker */
    if (plan->n == 0 || sched_getaffinity(0, sizeof *saved, saved) != 0) return 0;
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(plan->cpu[omp_get_thread_num() % plan->n], &one);
    return sched_setaffinity(0, sizeof one, &one) == 0;
}
static void pin_leave(int pinned, const cpu_set_t* saved) {
    if (pinned) (void)sched_setaffinity(0, sizeof *saved, saved);
}
#define PIN_PLAN() PinPlan pin_plan_; pin_plan(&pin_plan_)
#define PIN_ENTER() cpu_set_t pin_saved_; const int pinned_ = pin_enter(&pin_plan_, &pin_saved_)
#define PIN_LEAVE() pin_leave(pinned_, &pin_saved_)
#else
#define PIN_PLAN() do {} while (0)
#define PIN_ENTER() do {} while (0)
#define PIN_LEAVE() do {} while (0)
#endif

/* Same, over an explicit list of pixel indices (bench.py samples whole rows of a large image).  The running mean of a pixel
 * is kept in registers over its passes and written once (the same recurrence in the same order: bit-identical), so threads
 * working on neighbouring pixels do not trade cache lines of `res` pass by pass. */
int port_render_gids(const OracleScene* s, const int32_t* seeds, int n_passes, int first_spp, const int32_t* gids,
                     int64_t n_gids, float* res, int threads) {
    Sun sun = sun_new(s->sun);
    if (threads < 1) threads = 1;
    PIN_PLAN();
#pragma omp parallel num_threads(threads)
    {
        Counters local;
        memset(&local, 0, sizeof local);
        t_ctr = g_count_enabled ? &local : 0;
        PIN_ENTER();
        /* every worker reads its OWN copies of what the samples read all the time: `sun` and `*s` live in the calling thread's
         * stack frame / the caller's memory, next to words that thread keeps writing while it works — on a 2-socket, 16-CCD
         * host that one shared line capped the whole leg at ~14 Msamples/s whatever the thread count (profiles/r04_cpu_sweep) */
        const Sun my_sun = sun;
        const OracleScene my_scene = *s;
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < n_gids; i++) {
            int gid = gids[i];
            float* px = res + 3 * (int64_t)gid;
            float m0 = px[0], m1 = px[1], m2 = px[2];
            for (int k = 0; k < n_passes; k++) {
                int spp = first_spp + k;
                v3 c = ext_active() ? trace_sample_ext(&my_scene, &my_sun, &g_ext, seeds[k], gid) : trace_sample(&my_scene, &my_sun, seeds[k], gid, 0, 0);
                m0 = (m0 * spp + c.x) / (spp + 1);
                m1 = (m1 * spp + c.y) / (spp + 1);
                m2 = (m2 * spp + c.z) / (spp + 1);
            }
            px[0] = m0;
            px[1] = m1;
            px[2] = m2;
        }
        PIN_LEAVE();
        if (g_count_enabled) counters_merge(&local);
        t_ctr = 0;
    }
    return 0;
}

int port_trace_records(const OracleScene* s, int seed, int gid, OracleHit* out, float* radiance) {
    Sun sun = sun_new(s->sun);
    int n = 0;
    v3 c = trace_sample(s, &sun, seed, gid, out, &n);
    radiance[0] = c.x; radiance[1] = c.y; radiance[2] = c.z;
    return n;
}

/* K/rayTracer.cl:115-217 */
int port_preview(const OracleScene* s, int32_t* argb, int threads) {
    Sun sun = sun_new(s->sun);
    int W = s->width, H = s->height;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
    for (int gid = 0; gid < W * H; gid++) {
        int px = gid % W, py = gid / W;
        if ((px == W / 2 && (py >= H / 2 - 5 && py <= H / 2 + 5)) ||
            (py == H / 2 && (px >= W / 2 - 5 && px <= W / 2 + 5))) {
            argb[gid] = (int32_t)0xFFFFFFFFu;
            continue;
        }
        Path p;
        Record rec;
        path_init(&p, &rec);
        unsigned state = 0;
        rt_pcg_next(&state);
        primary_ray(s, gid, &state, &p, 1);
        if (closest_intersect(s, &p, &rec, 256)) {
            float shading = dot3(rec.normal, V3(0.25f, 0.866f, 0.433f));
            shading = rt_fmax(0.3f, shading);
            rec.color = scale4(rec.color, shading);
        } else {
            rec.emittance = 1;
            intersect_sky(s, &sun, &p, &rec);
        }
        float r = rt_sqrt(rec.color.x), g = rt_sqrt(rec.color.y), b = rt_sqrt(rec.color.z);
        int ri = ifloor(rt_clamp(r * 255.0f, 0.0f, 255.0f));
        int gi = ifloor(rt_clamp(g * 255.0f, 0.0f, 255.0f));
        int bi = ifloor(rt_clamp(b * 255.0f, 0.0f, 255.0f));
        argb[gid] = (int32_t)(0xFF000000u | ((unsigned)ri << 16) | ((unsigned)gi << 8) | (unsigned)bi);
    }
    return 0;
}

void port_math(int which, int n, const float* a, const float* b, float* out) {
    for (int i = 0; i < n; i++) {
        switch (which) {
            case 0: out[i] = rt_sin(a[i]); break;
            case 1: out[i] = rt_cos(a[i]); break;
            case 2: out[i] = rt_asin(a[i]); break;
            case 3: out[i] = rt_acos(a[i]); break;
            case 4: out[i] = rt_atan2(a[i], b[i]); break;
            case 5: out[i] = rt_fmod1(a[i]); break;
            case 6: out[i] = rt_fmin(a[i], b[i]); break;
            case 7: out[i] = rt_fmax(a[i], b[i]); break;
            case 8: out[i] = rt_sqrt(a[i]); break;
            case 9: out[i] = a[i] / b[i]; break;
            default: out[i] = 0;
        }
    }
}

/* The geometric builtins and the sky sampler as this restatement evaluates them (rt_math.h), for tests/test_rt_math.py:
 * which 0 dot -> out[0]; 1 cross -> out[0..2]; 2 normalize(a) -> out[0..2]; 3 fmod(a[0], b[0]) for ANY b (the kernel only
 * uses b = 1; the OpenCL shim of oracle/_ref falls back to C fmodf otherwise) -> out[0]. */
void port_geom(int which, int n, const float* a3, const float* b3, float* out3) {
    for (int i = 0; i < n; i++) {
        const float* a = a3 + 3 * i;
        const float* b = b3 + 3 * i;
        float* o = out3 + 3 * i;
        o[0] = o[1] = o[2] = 0;
        if (which == 0) {
            o[0] = rt_dot3(a[0], a[1], a[2], b[0], b[1], b[2]);
        } else if (which == 1) {
            v3 c = cross3(V3(a[0], a[1], a[2]), V3(b[0], b[1], b[2]));
            o[0] = c.x; o[1] = c.y; o[2] = c.z;
        } else if (which == 2) {
            v3 c = normalize3(V3(a[0], a[1], a[2]));
            o[0] = c.x; o[1] = c.y; o[2] = c.z;
        } else if (which == 3) {
            o[0] = b[0] == 1.0f ? rt_fmod1(a[0]) : fmodf(a[0], b[0]);
        }
    }
}
/* CLK_NORMALIZED_COORDS_TRUE | CLK_ADDRESS_MIRRORED_REPEAT | CLK_FILTER_LINEAR on an RGBA8 image (K/sky.h:95-105): the
 * indices / weight of rt_mirror_linear along one axis, and the filtered texel at (s, t). */
void port_mirror_linear(int n, const float* s, int w, int32_t* i0, int32_t* i1, float* a) {
    for (int i = 0; i < n; i++) {
        int j0, j1;
        rt_mirror_linear(s[i], w, &j0, &j1, &a[i]);
        i0[i] = j0;
        i1[i] = j1;
    }
}
void port_sample_linear(int n, const float* st, const uint8_t* rgba, int w, int h, float* out4) {
    for (int i = 0; i < n; i++) {
        int i0, i1, j0, j1;
        float a, b;
        rt_mirror_linear(st[2 * i], w, &i0, &i1, &a);
        rt_mirror_linear(st[2 * i + 1], h, &j0, &j1, &b);
        const uint8_t* t00 = rgba + 4 * ((size_t)j0 * w + i0);
        const uint8_t* t10 = rgba + 4 * ((size_t)j0 * w + i1);
        const uint8_t* t01 = rgba + 4 * ((size_t)j1 * w + i0);
        const uint8_t* t11 = rgba + 4 * ((size_t)j1 * w + i1);
        float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
        for (int k = 0; k < 4; k++)
            out4[4 * i + k] = w00 * rt_unorm8(t00[k]) + w10 * rt_unorm8(t10[k]) + w01 * rt_unorm8(t01[k]) + w11 * rt_unorm8(t11[k]);
    }
}

/* Exhaustive check behind the threshold table of the HIP tone-map kernel: over every float with bit pattern in [lo, hi] the
 * output byte of the GAMMA tail — min(255, (uint)(pow(c, 1/2.2) * 255 + 0.5)), post_processing_filter.cl:24-27, rgba.h:9-14 —
 * never decreases as c grows, and changes exactly at `thresholds[k]` (the smallest float whose byte is >= k).  Returns the
 * number of bit patterns that break either. */
static unsigned gamma_byte_of(float c) {
    const float f = rt_pow(c, (float)(1.0 / 2.2)) * 255.0f + 0.5f;
    const unsigned u = !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xFFFFFFFFu : (unsigned)f);
    return u > 255u ? 255u : u;
}
int64_t port_gamma_scan(uint32_t lo, uint32_t hi, const float* thresholds, int threads) {
    int64_t bad = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(+ : bad)
    for (int64_t chunk = lo >> 16; chunk <= (int64_t)(hi >> 16); chunk++) {
        uint32_t b0 = (uint32_t)chunk << 16, b1 = b0 | 0xFFFFu;
        if (b0 < lo) b0 = lo;
        if (b1 > hi) b1 = hi;
        float c;
        uint32_t prev_bits = b0 ? b0 - 1 : 0;
        memcpy(&c, &prev_bits, 4);
        unsigned prev = gamma_byte_of(c);
        for (uint64_t b = b0; b <= b1; b++) {
            const uint32_t bits = (uint32_t)b;
            memcpy(&c, &bits, 4);
            const unsigned k = gamma_byte_of(c);
            unsigned by_table = 0; /* number of thresholds T[1..255] <= c, by bisection */
            {
                int a = 0, z = 255; /* T[a] <= c < T[z + 1] */
                while (a < z) {
                    const int m = (a + z + 1) >> 1;
                    if (c >= thresholds[m]) a = m; else z = m - 1;
                }
                by_table = (unsigned)a;
            }
            if (k < prev || k != by_table) bad++;
            prev = k;
        }
    }
    return bad;
}

/* ------------------------------------------------------------------------------ tone map ----
 * `filter`, tonemap/include/post_processing_filter.cl:5-51, with fp64 present (double.h:19-21):
 * per channel c = (float)input * exposure, then the curve selected by `type`, then color_to_argb
 * (rgba.h:6-17).  The (uint) conversion of rgba.h:9-14 is undefined for NaN / negative / >= 2^32 in
 * OpenCL; this restatement (and the HIP kernel) saturates there. */
static uint32_t filter_to_uint(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
static float filter_channel(float c, int type) {
    switch (type) {
        case 0: /* :24-27 */
            return rt_pow(c, (float)(1.0 / 2.2));
        case 1: /* :28-32 */
            c = rt_fmax(0.0f, c - 0.004f);
            return (c * (6.2f * c + 0.5f)) / (c * (6.2f * c + 1.7f) + 0.06f);
        case 2: /* :33-38 */
            c = (c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f);
            c = rt_clamp(c, 0.0f, 1.0f);
            return rt_pow(c, (float)(1.0 / 2.2));
        case 3: { /* :39-44 */
            c *= 16;
            c = ((c * (0.15f * c + 0.10f * 0.50f) + 0.20f * 0.02f) / (c * (0.15f * c + 0.50f) + 0.20f * 0.30f)) - 0.02f / 0.30f;
            c /= (((11.2f * (0.15f * 11.2f + 0.10f * 0.50f) + 0.20f * 0.02f) / (11.2f * (0.15f * 11.2f + 0.50f) + 0.20f * 0.30f)) - 0.02f / 0.30f);
            return c;
        }
        default: return c;
    }
}
void port_filter(int n, int width, int height, float exposure, const uint64_t* input, uint32_t* res, int type) {
    (void)width;
    (void)height;
    for (int gid = 0; gid < n; gid++) {
        uint32_t ch[3];
        for (int i = 0; i < 3; i++) {
            double d;
            memcpy(&d, &input[3 * (size_t)gid + i], 8);
            float c = filter_channel((float)d * exposure, type);
            uint32_t u = filter_to_uint(c * 255.0f + 0.5f);
            ch[i] = u > 255u ? 255u : u;
        }
        res[gid] = (255u << 24) | (ch[0] << 16) | (ch[1] << 8) | ch[2];
    }
}
void port_pow(int n, const float* a, const float* b, float* out) {
    for (int i = 0; i < n; i++) out[i] = rt_pow(a[i], b[i]);
}
