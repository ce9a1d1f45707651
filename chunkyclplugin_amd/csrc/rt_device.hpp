// rt_device.hpp — gfx950 device building blocks of the Chunky path tracer.
//
// Each helper states which lines of the reference kernel
// (/root/reference/src/main/opencl/kernel/include/, "K/") define the arithmetic it must reproduce
// bit for bit.  The arithmetic contract (IEEE binary32, binary64 at the reference's double-literal
// sites, no contraction, OpenCL builtins as defined in rt_math.h) is SURVEY.md Appendix E; this
// translation unit is compiled with -ffp-contract=off.
//
// Nothing here mirrors the reference's control flow: the helpers are leaf computations that the
// kernels in render_pool.hip / render_fallback.hip schedule per wavefront.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_math.h"

namespace chunky {

constexpr float kEps = 0.000005f;    // K/constants.h:4
constexpr float kOffset = 0.0001f;   // K/constants.h:5
constexpr int kAnyType = 0x7FFFFFFE; // K/block.h:32

struct f3 {
    float x, y, z;
};
struct f4 {
    float x, y, z, w;
};
#define DEV __device__ __forceinline__

DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
DEV f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
DEV f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
DEV float dot(f3 a, f3 b) { return rt_dot3(a.x, a.y, a.z, b.x, b.y, b.z); }
DEV f3 cross(f3 a, f3 b) {
    return f3{rt_cross_c(a.y, b.z, a.z, b.y), rt_cross_c(a.z, b.x, a.x, b.z), rt_cross_c(a.x, b.y, a.y, b.x)};
}
DEV f3 normalize(f3 a) { return a * rt_rlen3(a.x, a.y, a.z); }
DEV f3 rcp3(f3 a) { return f3{1.0f / a.x, 1.0f / a.y, 1.0f / a.z}; }
DEV float as_float(int i) { return __int_as_float(i); }

// ---------------------------------------------------------------------------------------------
// Device view of one packed scene (wire formats: SURVEY.md Appendix A).  Passed by value as a
// kernel argument so every field is wave-uniform (SGPR-resident).
struct SceneView {
    const int* __restrict__ octree;   // K/rayTracer.cl:16
    const int* __restrict__ blocks;   // :18
    const int* __restrict__ quads;    // :19
    const int* __restrict__ aabbs;    // :20
    const int* __restrict__ world_bvh;  // :22
    const int* __restrict__ actor_bvh;  // :23
    const int* __restrict__ trigs;    // :24
    const uint32_t* __restrict__ atlas;  // :26  RGBA8 texels as little-endian uint32, [layer][y][x]
    const int* __restrict__ materials;   // :27
    const float4* __restrict__ sky;      // :29  [y][x], UNORM8 texels already converted (rt_unorm8) at upload
    int octree_depth;
    int atlas_w, atlas_h, atlas_layers;
    int sky_w, sky_h;
    float sky_intensity;              // :30
    // Sun_new (K/sky.h:19-40) evaluated once on the host with the same rt_math.h
    int sun_flags, sun_tex_size, sun_tex;
    float sun_intensity;
    f3 su, sv, sw;
    float sun_radius_cos;             // cos(0.03f), K/sky.h:73
    int world_bvh_empty, actor_bvh_empty;  // K/bvh.h:23-32 sentinel, tested at upload
    int bvh_stack_entries;                 // height of the taller BVH + 1 (the reference reserves 64, K/bvh.h:38)
    // wide re-layout of the octree (widetree.hpp); null when it could not be built
    const uint32_t* __restrict__ wide;
    int wide_nlev;
    int wide_shift[6], wide_bits[6];
    // per block (index = block pointer / 2) 8 ints: {modelType, modelPointer, material words 0..4 of a
    // full cube (K/material.h:31-40), 0} — the block-palette and material-palette reads of
    // K/block.h:36-49 as ONE 32-byte load; built at upload, null when a palette is missing
    const int4* __restrict__ block_info;
    int sort_blocks;   // model blocks are common in this world: render_pool tests full cubes and model blocks in phases of their own (capi.hip scene_view)
    int n_block_ints;  // ints in the block palette: a leaf whose block pointer lies beyond it never intersects (the wide tree marks such leaves at upload)
    // per quad, at the quad's own int offset in `quads`: {normal xyz, dot(normal, origin), |xv|^2, |yv|^2} — the
    // ray-independent part of K/primitives.h:262-276, evaluated once at upload with this same rt_math.h; null = compute
    const float* __restrict__ quad_aux;
    // 16-byte-aligned re-layouts built at upload (capi.hip rebuild_derived); block_info word 7 of a model block =
    // first record << 8 | primitive count, 0 = none (the packed palettes are read as they are):
    //   mat8      per material two words {flags, tint, textureSize, color} {normal_emittance, word 5, 0, 0}
    //   aabb_rec  per box three words {xmin, xmax, ymin, ymax} {zmin, zmax, flags, E} {S, W, T, B} (materials = mat8 indices)
    //   quad_rec  per quad six words {o, dot(n, o)} {xv, |xv|^2} {yv, |yv|^2} {uv} {n, emittance byte | word 5 << 8} {flags, tint, textureSize, color}
    const int4* __restrict__ mat8;
    const int4* __restrict__ aabb_rec;
    const int4* __restrict__ quad_rec;
    // Both entity BVHs re-laid out (capi.hip build_bvh_records), null when they could not be:
    //   bvh_rec  per inner node four words {ref A, ref B, 0, 0} {A: xmin, xmax, ymin, ymax} {A: zmin, zmax, B: xmin, xmax}
    //            {B: ymin, ymax, zmin, zmax} — A is the child that follows the node (K/bvh.h:73), B the one it points at
    //   tri_rec  per triangle five words {e1, flags} {e2, material (mat8 index)} {o, t1.u} {n, t1.v} {t2.u, t2.v, t3.u, t3.v}
    // A reference is an inner record's index, or -1 - (first triangle record << 6 | count) for a leaf.
    const int4* __restrict__ bvh_rec;
    const int4* __restrict__ tri_rec;  // = bvh_rec + tri_off bytes: one allocation, so either kind of record is a 32-bit offset off one base
    unsigned tri_off;
    int world_root, actor_root;  // reference of each BVH's root
    // emitter next-event estimation (extension): every emitter leaf of the octree as {x, y, z, level << 25 | block pointer},
    // in pre-order — the list of oracle/port.c port_list_emitters
    const int4* __restrict__ emitters;
    int n_emitters;
    // CHUNKY_OPT_BVH_CULL_BEHIND (extension, default 0 = the reference's walk): children whose box lies entirely behind the ray
    // origin count as missed.  The reference's quick test (K/primitives.h:30-48) has no such exit and walks them.
    int bvh_cull;
};

struct CameraView {
    const float* __restrict__ rays;  // projector -1: W*H*6 floats (K/camera.h:8-11)
    int projector_type;
    float pos[3], m[9];              // ClCamera.java:42-52
    float aperture, subject_distance, fov_tan;
    int width, height;
    float half_width, inv_height;    // K/rayTracer.cl:66-67 (double sites, rounded once on the host)
};

struct RenderOpts {
    int draw_depth;       // 256, K/rayTracer.cl:94
    int max_depth;        // 5,   K/rayTracer.cl:107
    float emitter_scale;  // 13,  K/rayTracer.cl:99
    // EXPERIMENTAL light-transport options (specification: oracle/port.c trace_sample_ext).  The
    // defaults are the reference's behaviour and select the reference kernels.
    int sun_sampling;     // -1 as the reference (PackedSun flag bit 0), 0 never, 1 always
    int emitters;         // 1 as the reference; 0 = emittersEnabled false
    int bsdf;             // 1: specular / metal / roughness from material word 5
    int nee;              // 1: emitter next-event estimation
    int bvh_cull;         // 1: CHUNKY_OPT_BVH_CULL_BEHIND (travels to the kernels in SceneView::bvh_cull)
};
__host__ __device__ inline bool opts_extended(const RenderOpts& O) { return O.sun_sampling != -1 || O.emitters != 1 || O.bsdf != 0 || O.nee != 0; }

// Per-path state: Pixel + Ray + IntersectionRecord of K/wavefront.h:6-51, flattened.
struct Hit {
    float distance;
    int material;
    f3 normal;
    f4 color;
    float emittance;
    int spec;  // material word 5 (spec | metal << 8 | rough << 16): only the extended integrator reads it
};

// ---------------------------------------------------------------------------------------------
// slab tests — K/primitives.h:30-61
struct Slabs {
    float t1x, t1y, t1z, t2x, t2y, t2z;
};
DEV Slabs slabs(float x0, float x1, float y0, float y1, float z0, float z1, f3 o, f3 inv) {
    return Slabs{(x0 - o.x) * inv.x, (y0 - o.y) * inv.y, (z0 - o.z) * inv.z,
                 (x1 - o.x) * inv.x, (y1 - o.y) * inv.y, (z1 - o.z) * inv.z};
}
DEV float slab_near(const Slabs& s) {
    return rt_fmax(rt_fmin(s.t1x, s.t2x), rt_fmax(rt_fmin(s.t1y, s.t2y), rt_fmin(s.t1z, s.t2z)));
}
DEV float slab_far(const Slabs& s) {
    return rt_fmin(rt_fmax(s.t1x, s.t2x), rt_fmin(rt_fmax(s.t1y, s.t2y), rt_fmax(s.t1z, s.t2z)));
}
// AABB_quick_intersect: NaN = miss
DEV float box_quick(float x0, float x1, float y0, float y1, float z0, float z1, f3 o, f3 inv) {
    Slabs s = slabs(x0, x1, y0, y1, z0, z1, o, inv);
    float tn = slab_near(s), tf = slab_far(s);
    return (tf < tn) ? rt_nan() : tn;
}
// the same test, also handing out where the ray's line leaves the box (CHUNKY_OPT_BVH_CULL_BEHIND: a box whose far end lies
// behind the origin cannot hold a hit)
DEV float box_quick_far(float x0, float x1, float y0, float y1, float z0, float z1, f3 o, f3 inv, float& tf) {
    Slabs s = slabs(x0, x1, y0, y1, z0, z1, o, inv);
    float tn = slab_near(s);
    tf = slab_far(s);
    return (tf < tn) ? rt_nan() : tn;
}
// AABB_exit
DEV float box_exit(float x0, float x1, float y0, float y1, float z0, float z1, f3 o, f3 inv) {
    return slab_far(slabs(x0, x1, y0, y1, z0, z1, o, inv));
}

// ---------------------------------------------------------------------------------------------
// K/utils.h:6-14 (divide by 256, exact)
DEV f4 color_from_argb(unsigned argb) {
    const float k = 1.0f / 256.0f;
    return f4{(float)((argb >> 16) & 0xFF) * k, (float)((argb >> 8) & 0xFF) * k, (float)(argb & 0xFF) * k,
              (float)((argb >> 24) & 0xFF) * k};
}
DEV f4 unpack_unorm8(uint32_t t) {
    return f4{rt_unorm8(t & 0xFF), rt_unorm8((t >> 8) & 0xFF), rt_unorm8((t >> 16) & 0xFF), rt_unorm8(t >> 24)};
}

// K/textureAtlas.h:10-28: nearest, unnormalised, clamp-to-edge, layer clamped
DEV uint32_t atlas_texel(const SceneView& S, float u, float v, int location, int size) {
    int width = (size >> 16) & 0xFFFF, height = size & 0xFFFF;
    v = 1 - v;
    int x = rt_clampi((int)((u - kEps) * (float)width), 0, width - 1);
    int y = rt_clampi((int)((v - kEps) * (float)height), 0, height - 1);
    x += ((location >> 22) & 0x1FF) * 16;
    y += ((location >> 13) & 0x1FF) * 16;
    int d = location & 0x7FFFF;
    x = rt_clampi(x, 0, S.atlas_w - 1);
    y = rt_clampi(y, 0, S.atlas_h - 1);
    d = rt_clampi(d, 0, S.atlas_layers - 1);
    return S.atlas[((size_t)d * S.atlas_h + y) * S.atlas_w + x];
}

// (ne & 0xFF) / 255.0 evaluated in double and rounded to float (K/material.h:79) has 256 possible
// results: a table replaces the f64 divide sequence on the hit path.
__device__ __constant__ const float kEmittanceLut[256] = {
#include "emit_lut.inc"
};

// K/material.h:31-82.  `shade` = false skips the writes that only matter to the main record
// (shadow rays need the accept/reject decision only).
DEV bool material_eval(const SceneView& S, unsigned flags, unsigned tint, unsigned tex_size, unsigned color_w,
                       unsigned ne, float u, float v, Hit& h, int m5 = 0);
DEV bool material_sample(const SceneView& S, int material, float u, float v, Hit& h) {
    const int* m = S.materials + material;
    return material_eval(S, m[0], m[1], m[2], m[3], m[4], u, v, h, m[5]);
}
DEV bool material_eval(const SceneView& S, unsigned flags, unsigned tint, unsigned tex_size, unsigned color_w,
                       unsigned ne, float u, float v, Hit& h, int m5) {
    f4 c = (flags & 4) ? unpack_unorm8(atlas_texel(S, u, v, (int)color_w, (int)tex_size)) : color_from_argb(color_w);
    if (!(c.w > kEps)) return false;
    unsigned tt = tint >> 24;
    if (tt == 0xFF || tt == 1 || tt == 2 || tt == 3) {
        unsigned tc = tt == 0xFF ? tint : (tt == 1 ? 0xFF71A74Du : (tt == 2 ? 0xFF8EB971u : 0xFF3F76E4u));
        f4 t = color_from_argb(tc);
        c = f4{c.x * t.x, c.y * t.y, c.z * t.z, c.w * t.w};
    }
    h.color = c;
    if (flags & 2)
        h.emittance = unpack_unorm8(atlas_texel(S, u, v, (int)ne, (int)tex_size)).w;
    else
        h.emittance = kEmittanceLut[ne & 0xFF];  // double site K/material.h:79, tabulated
    h.spec = m5;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Entry face + UV of a slab hit: the equality chain of K/primitives.h:86-109 / :136-159, last
// match wins.  p = origin + tmin * dir as the caller defines dir.
struct Face {
    f3 n;
    float u, v;
};
DEV Face face_unit(const Slabs& s, float tmin, f3 p) {  // AABB_full_intersect on the unit cube
    // d = 1/(max-min) = 1 exactly for the unit cube, (p - 0) * 1 = p exactly
    Face f{mk3(0, 0, 0), 0, 0};
    if (s.t1x == tmin) f = Face{mk3(-1, 0, 0), 1 - p.z, p.y};
    if (s.t2x == tmin) f = Face{mk3(1, 0, 0), p.z, p.y};
    if (s.t1y == tmin) f = Face{mk3(0, -1, 0), p.x, 1 - p.z};
    if (s.t2y == tmin) f = Face{mk3(0, 1, 0), p.x, p.z};
    if (s.t1z == tmin) f = Face{mk3(0, 0, -1), p.x, p.y};
    if (s.t2z == tmin) f = Face{mk3(0, 0, 1), 1 - p.x, p.y};
    return f;
}
DEV Face face_map2(const Slabs& s, float tmin, f3 p) {  // AABB_full_intersect_map_2
    Face f{mk3(0, 0, 0), 0, 0};
    if (s.t1x == tmin) f = Face{mk3(-1, 0, 0), p.z, p.y};
    if (s.t2x == tmin) f = Face{mk3(1, 0, 0), 1 - p.z, p.y};
    if (s.t1y == tmin) f = Face{mk3(0, -1, 0), p.x, p.z};
    if (s.t2y == tmin) f = Face{mk3(0, 1, 0), p.x, 1 - p.z};
    if (s.t1z == tmin) f = Face{mk3(0, 0, -1), 1 - p.x, p.y};
    if (s.t2z == tmin) f = Face{mk3(0, 0, 1), p.x, p.y};
    return f;
}

// Full cube (block model type 1) — K/block.h:48-65 with K/primitives.h:66-112.
// `no` = march position minus d*OFFSET minus the block corner; the reference passes the march
// position `pos` where a direction is expected (K/block.h:52), so the UV point is no + tmin*pos.
// The material arrives as its 5 words (read from the material palette, or inline in block_info).
DEV float cube_hit(const SceneView& S, unsigned m0, unsigned m1, unsigned m2, unsigned m3, unsigned m4, f3 no, f3 pos,
                   f3 inv, Hit& h, int m5 = 0) {
    Slabs s = slabs(0, 1, 0, 1, 0, 1, no, inv);
    float tn = slab_near(s), tf = slab_far(s);
    if (tf < tn) return rt_nan();
    Face f = face_unit(s, tn, no + pos * tn);
    h.normal = f.n;  // written before the material test (K/block.h:59-60)
    return material_eval(S, m0, m1, m2, m3, m4, f.u, f.v, h, m5) ? tn - kOffset : rt_nan();
}

// AABB model (type 2) — K/block.h:66-91, K/primitives.h:165-260
DEV float aabb_model_hit(const SceneView& S, int ptr, f3 no, f3 dir, f3 inv, Hit& h) {
    const int* __restrict__ model = S.aabbs + ptr;
    int boxes = model[0];
    float best = rt_inf();
    bool hit = false;
    for (int i = 0; i < boxes; i++) {
        const int* b = model + 1 + i * 13;
        Slabs s = slabs(as_float(b[0]), as_float(b[1]), as_float(b[2]), as_float(b[3]), as_float(b[4]),
                        as_float(b[5]), no, inv);
        float tn = slab_near(s), tf = slab_far(s);
        if (tf < tn) continue;
        if (tn != tn || tn >= best || tn < -kEps) continue;
        Face f = face_map2(s, tn, no + dir * tn);
        int fl_all = b[6];
        // face -> material / flags: K/primitives.h:209-234 (both z tests are "== -1"; +z keeps the
        // east material with flags 0, see oracle/port.c textured_box)
        int mat = b[8], fl = 0;
        if (f.n.x == 1) { mat = b[8]; fl = fl_all >> 4; }
        if (f.n.z == -1) { mat = b[9]; fl = fl_all >> 8; }
        if (f.n.x == -1) { mat = b[10]; fl = fl_all >> 12; }
        if (f.n.y == 1) { mat = b[11]; fl = fl_all >> 16; }
        if (f.n.y == -1) { mat = b[12]; fl = fl_all >> 20; }
        if (fl & 8) continue;
        float u = f.u, v = f.v;
        if (fl & 4) u = 1 - u;
        if (fl & 2) v = 1 - v;
        if (fl & 1) { float t = u; u = v; v = t; }
        if (material_sample(S, mat, u, v, h)) {
            h.normal = f.n;
            best = tn;
            hit = true;
        }
    }
    return hit ? best : rt_nan();
}

// Quad model (type 3) — K/block.h:92-116, K/primitives.h:263-319
DEV float quad_model_hit(const SceneView& S, int ptr, f3 no, f3 dir, Hit& h) {
    const int* __restrict__ model = S.quads + ptr;
    int quads = model[0];
    float best = rt_inf();
    bool hit = false;
    for (int i = 0; i < quads; i++) {
        const int* q = model + 1 + i * 15;
        f3 qo = mk3(as_float(q[0]), as_float(q[1]), as_float(q[2]));
        f3 xv = mk3(as_float(q[3]), as_float(q[4]), as_float(q[5]));
        f3 yv = mk3(as_float(q[6]), as_float(q[7]), as_float(q[8]));
        f3 n;
        float n_qo, xx, yy;
        if (S.quad_aux) {
            const float* a = S.quad_aux + (q - S.quads);
            n = mk3(a[0], a[1], a[2]);
            n_qo = a[3];
            xx = a[4];
            yy = a[5];
        } else {
            n = normalize(cross(xv, yv));
            n_qo = dot(n, qo);
            xx = dot(xv, xv);
            yy = dot(yv, yv);
        }
        float denom = dot(dir, n);
        if (!(denom < -kEps)) continue;
        float t = -(dot(no, n) - n_qo) / denom;
        if (!(t > -kEps && t < best)) continue;
        f3 pt = (no + dir * t) - qo;
        float u = dot(pt, xv) / xx;
        float v = dot(pt, yv) / yy;
        if (!(u >= 0 && u <= 1 && v >= 0 && v <= 1)) continue;
        float tu = as_float(q[9]) + (u * as_float(q[10]));
        float tv = as_float(q[11]) + (v * as_float(q[12]));
        if (material_sample(S, q[13], tu, tv, h)) {
            h.normal = n;
            best = t;
            hit = true;
        }
    }
    return hit ? best : rt_nan();
}

DEV bool material_sample8(const SceneView& S, int m8, float u, float v, Hit& h) {
    const int4 a = S.mat8[m8], b = S.mat8[m8 + 1];
    return material_eval(S, a.x, a.y, a.z, a.w, b.x, u, v, h, b.y);
}

// aabb_model_hit / quad_model_hit on the aligned records (same arithmetic, same order; rec = first record << 8 | count)
DEV float aabb_model_hit_rec(const SceneView& S, int rec, f3 no, f3 dir, f3 inv, Hit& h) {
    const int boxes = rec & 0xFF, first = (int)((unsigned)rec >> 8);
    float best = rt_inf();
    bool hit = false;
    for (int i = 0; i < boxes; i++) {
        const int4 r0 = S.aabb_rec[(first + i) * 3], r1 = S.aabb_rec[(first + i) * 3 + 1], r2 = S.aabb_rec[(first + i) * 3 + 2];
        Slabs s = slabs(as_float(r0.x), as_float(r0.y), as_float(r0.z), as_float(r0.w), as_float(r1.x), as_float(r1.y), no, inv);
        float tn = slab_near(s), tf = slab_far(s);
        if (tf < tn) continue;
        if (tn != tn || tn >= best || tn < -kEps) continue;
        Face f = face_map2(s, tn, no + dir * tn);
        const int fl_all = r1.z;
        int mat = r1.w, fl = 0;  // +z keeps the east material with flags 0 (see aabb_model_hit)
        if (f.n.x == 1) { mat = r1.w; fl = fl_all >> 4; }
        if (f.n.z == -1) { mat = r2.x; fl = fl_all >> 8; }
        if (f.n.x == -1) { mat = r2.y; fl = fl_all >> 12; }
        if (f.n.y == 1) { mat = r2.z; fl = fl_all >> 16; }
        if (f.n.y == -1) { mat = r2.w; fl = fl_all >> 20; }
        if (fl & 8) continue;
        float u = f.u, v = f.v;
        if (fl & 4) u = 1 - u;
        if (fl & 2) v = 1 - v;
        if (fl & 1) { float t = u; u = v; v = t; }
        if (material_sample8(S, mat, u, v, h)) {
            h.normal = f.n;
            best = tn;
            hit = true;
        }
    }
    return hit ? best : rt_nan();
}

DEV float quad_model_hit_rec(const SceneView& S, int rec, f3 no, f3 dir, Hit& h) {
    const int quads = rec & 0xFF, first = (int)((unsigned)rec >> 8);
    float best = rt_inf();
    bool hit = false;
    for (int i = 0; i < quads; i++) {
        const int4* __restrict__ q = S.quad_rec + (first + i) * 6;
        const int4 r0 = q[0], r1 = q[1], r2 = q[2], r4 = q[4];
        const f3 qo = mk3(as_float(r0.x), as_float(r0.y), as_float(r0.z));
        const f3 xv = mk3(as_float(r1.x), as_float(r1.y), as_float(r1.z));
        const f3 yv = mk3(as_float(r2.x), as_float(r2.y), as_float(r2.z));
        const f3 n = mk3(as_float(r4.x), as_float(r4.y), as_float(r4.z));
        const float n_qo = as_float(r0.w), xx = as_float(r1.w), yy = as_float(r2.w);
        float denom = dot(dir, n);
        if (!(denom < -kEps)) continue;
        float t = -(dot(no, n) - n_qo) / denom;
        if (!(t > -kEps && t < best)) continue;
        f3 pt = (no + dir * t) - qo;
        float u = dot(pt, xv) / xx;
        float v = dot(pt, yv) / yy;
        if (!(u >= 0 && u <= 1 && v >= 0 && v <= 1)) continue;
        const int4 r3 = q[3], r5 = q[5];
        float tu = as_float(r3.x) + (u * as_float(r3.y));
        float tv = as_float(r3.z) + (v * as_float(r3.w));
        if (material_eval(S, r5.x, r5.y, r5.z, r5.w, r4.w, tu, tv, h, (int)((unsigned)r4.w >> 8))) {  // r4.w = emittance byte | word 5 << 8
            h.normal = n;
            best = t;
            hit = true;
        }
    }
    return hit ? best : rt_nan();
}

// BlockPalette_intersectBlock — K/block.h:30-118.  bx/by/bz = integer cell of the march point.
// KINDS: kBlockAny every block type; kBlockCubes / kBlockModels for callers that sorted their candidates into full cubes and model
// blocks (render_pool, by the model bit of the re-laid-out tree's leaf entries — widetree.cpp annotate_wide_tree — and tests them in
// phases of their own): the other kind's code is not part of that instantiation, a block of the other kind (there is none: the
// entries are annotated from the same palette; block_info marks a malformed block as a type that never hits) does not hit.  Both
// need block_info (capi.hip builds it for every scene).
enum : int { kBlockAny = 0, kBlockCubes = 1, kBlockModels = 2 };
template <int KINDS = kBlockAny>
DEV float block_hit(const SceneView& S, int block, int bx, int by, int bz, f3 pos, f3 dir, f3 inv, Hit& h) {
    if (KINDS == kBlockAny && block == kAnyType) return rt_nan();
    f3 no = (pos - dir * kOffset) - mk3((float)bx, (float)by, (float)bz);
    int type, ptr;
    if (KINDS != kBlockAny || S.block_info) {
        const int4 a = S.block_info[(unsigned)block], b = S.block_info[(unsigned)block + 1u];
        type = a.x;
        ptr = a.y;
        if (KINDS != kBlockModels && type == 1) return cube_hit(S, a.z, a.w, b.x, b.y, b.z, no, pos, inv, h, b.w);  // word 7 of a cube: material word 5
        if (KINDS == kBlockCubes) return rt_nan();
        if (b.w != 0) return type == 2 ? aabb_model_hit_rec(S, b.w, no, dir, inv, h) : quad_model_hit_rec(S, b.w, no, dir, h);
    } else {
        type = S.blocks[block];
        ptr = S.blocks[block + 1];
        if (type == 1) {
            const int* m = S.materials + ptr;
            return cube_hit(S, m[0], m[1], m[2], m[3], m[4], no, pos, inv, h, m[5]);
        }
    }
    switch (type) {
        case 2: return aabb_model_hit(S, ptr, no, dir, inv, h);
        case 3: return quad_model_hit(S, ptr, no, dir, h);
        default: return rt_nan();
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle — K/primitives.h:321-409 (Moller-Trumbore, single-sided = det < -EPS)
DEV float triangle_hit(const int* __restrict__ t, float best, f3 o, f3 dir, f3& n, float& u, float& v, int& mat) {
    int flags = t[0];
    f3 e1 = mk3(as_float(t[1]), as_float(t[2]), as_float(t[3]));
    f3 e2 = mk3(as_float(t[4]), as_float(t[5]), as_float(t[6]));
    f3 to = mk3(as_float(t[7]), as_float(t[8]), as_float(t[9]));
    f3 pvec = cross(dir, e2);
    float det = dot(e1, pvec);
    if ((flags >> 8) & 1) {
        if (det > -kEps && det < kEps) return rt_nan();
    } else if (det > -kEps) {
        return rt_nan();
    }
    float recip = 1 / det;
    f3 tvec = o - to;
    float uu = dot(tvec, pvec) * recip;
    if (uu < 0 || uu > 1) return rt_nan();
    f3 qvec = cross(tvec, e1);
    float vv = dot(dir, qvec) * recip;
    if (vv < 0 || (uu + vv) > 1) return rt_nan();
    float tt = dot(e2, qvec) * recip;
    if (tt > kEps && tt < best) {
        float w = 1 - uu - vv;
        u = as_float(t[13]) * uu + as_float(t[15]) * vv + as_float(t[17]) * w;
        v = as_float(t[14]) * uu + as_float(t[16]) * vv + as_float(t[18]) * w;
        n = mk3(as_float(t[10]), as_float(t[11]), as_float(t[12]));
        mat = t[19];
        return tt;
    }
    return rt_nan();
}

// Bvh_intersect — K/bvh.h:22-113.  `stack` is this lane's 64-entry to-visit stack; the caller
// chooses where it lives (LDS column or scratch).
template <typename Stack>
DEV bool bvh_hit(const SceneView& S, const int* __restrict__ bvh, f3 o, f3 d, Hit& h, Stack& stack) {
    const int* __restrict__ trigs = S.trigs;
    bool hit = false;
    int to_visit = 0, cur = 0;
    f3 inv = rcp3(d);
    for (;;) {
        int head = bvh[cur];
        if (head <= 0) {
            int prim = -head;
            int n = trigs[prim];
            for (int i = 0; i < n; i++) {
                f3 nn;
                float u, v;
                int mat;
                float dist = triangle_hit(trigs + prim + 1 + 20 * i, h.distance, o, d, nn, u, v, mat);
                if (dist == dist && material_sample(S, mat, u, v, h)) {
                    h.normal = nn;
                    h.distance = dist;
                    hit = true;
                }
            }
            if (to_visit == 0) break;
            cur = stack.pop(--to_visit);
        } else {
            int second = head;
            const int* a = bvh + cur + 7;
            const int* b = bvh + second;
            float f1, f2;
            float t1 = box_quick_far(as_float(a[1]), as_float(a[2]), as_float(a[3]), as_float(a[4]), as_float(a[5]),
                                     as_float(a[6]), o, inv, f1);
            float t2 = box_quick_far(as_float(b[1]), as_float(b[2]), as_float(b[3]), as_float(b[4]), as_float(b[5]),
                                     as_float(b[6]), o, inv, f2);
            bool miss1 = (t1 != t1) || t1 > h.distance;
            bool miss2 = (t2 != t2) || t2 > h.distance;
            if (S.bvh_cull) {
                miss1 |= f1 < 0;
                miss2 |= f2 < 0;
            }
            if (miss1) {
                if (miss2) {
                    if (to_visit == 0) break;
                    cur = stack.pop(--to_visit);
                } else {
                    cur = second;
                }
            } else if (miss2) {
                cur += 7;
            } else if (t1 < t2) {
                stack.push(to_visit++, second);
                cur += 7;
            } else {
                stack.push(to_visit++, cur + 7);
                cur = second;
            }
        }
    }
    return hit;
}

// ---------------------------------------------------------------------------------------------
// Sky_intersect — K/sky.h:95-106 with the linear / mirrored-repeat sampler contract of rt_math.h, in two halves: the
// four texel reads are issued by sky_fetch, their weighted sum is taken by sky_blend — intersectSky puts the sun-disc
// test (two acos and a dependent texel read of its own) between them, so the two round trips to memory overlap.
struct SkyTexels {
    float4 t00, t10, t01, t11;
    float a, b;
};
DEV SkyTexels sky_fetch(const SceneView& S, f3 d) {
    float theta = rt_atan2(d.z, d.x);
    theta = theta / (RT_PI_F * 2);
    theta = rt_fmod1(rt_fmod1(theta) + 1);
    float phi = (rt_asin(rt_clamp(d.y, -1.0f, 1.0f)) + RT_PI_2_F) * RT_1_PI_F;
    int i0, i1, j0, j1;
    SkyTexels k;
    rt_mirror_linear(theta, S.sky_w, &i0, &i1, &k.a);
    rt_mirror_linear(phi, S.sky_h, &j0, &j1, &k.b);
    k.t00 = S.sky[j0 * S.sky_w + i0];
    k.t10 = S.sky[j0 * S.sky_w + i1];
    k.t01 = S.sky[j1 * S.sky_w + i0];
    k.t11 = S.sky[j1 * S.sky_w + i1];
    return k;
}
DEV f4 sky_blend(const SceneView& S, const SkyTexels& q) {
    const float a = q.a, b = q.b;
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float k = S.sky_intensity;
    // the alpha channel is sampled by the reference too but never read again (K/kernel.h:30)
    return f4{(w00 * q.t00.x + w10 * q.t10.x + w01 * q.t01.x + w11 * q.t11.x) * k,
              (w00 * q.t00.y + w10 * q.t10.y + w01 * q.t01.y + w11 * q.t11.y) * k,
              (w00 * q.t00.z + w10 * q.t10.z + w01 * q.t01.z + w11 * q.t11.z) * k, 0.0f};
}
DEV f4 sky_color(const SceneView& S, f3 d) { return sky_blend(S, sky_fetch(S, d)); }

// Sun_intersect — K/sky.h:42-66: adds the sun-disc texel (x intensity) to c
DEV void sun_disc(const SceneView& S, f3 d, f4& c) {
    if (!(S.sun_flags & 1) || dot(d, S.sw) < 0.5f) return;  // (c + 0 would turn a -0 channel into +0)
    const float radius = 0.03f;
    const float width = radius * 4;
    const float width2 = width * 2;
    float a = RT_PI_2_F - rt_acos(dot(d, S.su)) + width;
    if (a >= 0 && a < width2) {
        float b = RT_PI_2_F - rt_acos(dot(d, S.sv)) + width;
        if (b >= 0 && b < width2) {
            f4 t = unpack_unorm8(atlas_texel(S, a / width2, b / width2, S.sun_tex, S.sun_tex_size));
            c.x += t.x * S.sun_intensity;
            c.y += t.y * S.sun_intensity;
            c.z += t.z * S.sun_intensity;
        }
    }
}

// intersectSky — K/kernel.h:26-31: returns color.xyz * throughput * emittance
DEV f3 sky_radiance(const SceneView& S, f3 d, f3 throughput, float emittance) {
    const SkyTexels q = sky_fetch(S, d);
    // the disc's texel, fetched while the sky's four are on their way; `in_disc` keeps "+= texel" apart from "no add"
    bool in_disc = false;
    f3 add = mk3(0, 0, 0);
    // The disc test of K/sky.h:42-66 asks 0 <= pi/2 - acos(x) + 0.12 < 0.24, i.e. |asin(x)| <= 0.12, |x| <= 0.1197: a direction with
    // |x| > 0.125 fails it by 0.005 — thousands of ulps of rt_acos — so such lanes skip the acos (its square root and polynomial
    // ran whenever ANY lane of a SHADE execution looked within 60 degrees of the sun); the outcome of no lane changes.
    const float xu = dot(d, S.su);
    if ((S.sun_flags & 1) && !(dot(d, S.sw) < 0.5f) && rt_fabs(xu) <= 0.125f) {
        const float radius = 0.03f;
        const float width = radius * 4;
        const float width2 = width * 2;
        float a = RT_PI_2_F - rt_acos(xu) + width;
        if (a >= 0 && a < width2) {
            float b = RT_PI_2_F - rt_acos(dot(d, S.sv)) + width;
            if (b >= 0 && b < width2) {
                f4 t = unpack_unorm8(atlas_texel(S, a / width2, b / width2, S.sun_tex, S.sun_tex_size));
                add = mk3(t.x * S.sun_intensity, t.y * S.sun_intensity, t.z * S.sun_intensity);
                in_disc = true;
            }
        }
    }
    f4 c = sky_blend(S, q);
    if (in_disc) {
        c.x += add.x;
        c.y += add.y;
        c.z += add.z;
    }
    return (mk3(c.x, c.y, c.z) * throughput) * emittance;
}

// Sun_sampleDirection — K/sky.h:68-93 (direction = u * v component-wise, then += w)
DEV f3 sun_sample(const SceneView& S, unsigned& rng) {
    float x1 = rt_pcg_float(&rng), x2 = rt_pcg_float(&rng);
    float cos_a = 1 - x1 + x1 * S.sun_radius_cos;
    float sin_a = rt_sqrt(1 - cos_a * cos_a);
    float phi = 2 * RT_PI_F * x2;
    float sp, cp;
    rt_sincos(phi, &sp, &cp);
    f3 u = S.su * (cp * sin_a);
    f3 v = S.sv * (sp * sin_a);
    f3 w = S.sw * cos_a;
    return normalize((u * v) + w);
}

// nextPath — K/kernel.h:46-98: cosine-weighted bounce about n (double compare at :66)
DEV f3 diffuse_bounce(f3 n, unsigned& rng) {
    float x1 = rt_pcg_float(&rng), x2 = rt_pcg_float(&rng);
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

// Primary ray — K/rayTracer.cl:55-91, K/camera.h:8-32.  `unit_dir` = the preview kernel's extra
// normalize (K/rayTracer.cl:186).
struct RayOD {
    f3 o, d;
};
// Returned by value through scalars (ox..dz each assigned once per branch): out-parameters written
// in both branches end up as a select between two addresses, which keeps them in scratch memory.
// (px, py) = (gid % width, gid / width): callers that know the pixel's column and row pass them and save the two divisions
DEV RayOD primary_ray(const CameraView& C, int gid, unsigned& rng, bool unit_dir, int px, int py) {
    float ox, oy, oz, dx, dy, dz;
    if (C.projector_type != -1) {
        float x = -C.half_width + ((float)px + rt_pcg_float(&rng)) * C.inv_height;
        float y = (float)(-0.5 + (double)(((float)py + rt_pcg_float(&rng)) * C.inv_height));
        f3 lo = mk3(0, 0, 0);
        f3 ld = mk3(C.fov_tan * x, C.fov_tan * y, 1.0f);
        if (C.aperture > 0) {
            ld = ld * (C.subject_distance / ld.z);
            float r = rt_sqrt(rt_pcg_float(&rng)) * C.aperture;
            float theta = (float)((double)(rt_pcg_float(&rng) * RT_PI_F) * 2.0);
            float st, ct;
            rt_sincos(theta, &st, &ct);
            float rx = ct * r, ry = st * r;
            ld = ld - mk3(rx, ry, 0);
            lo = lo + mk3(rx, ry, 0);
        }
        if (unit_dir) ld = normalize(ld);
        f3 m1 = mk3(C.m[0], C.m[1], C.m[2]), m2 = mk3(C.m[3], C.m[4], C.m[5]), m3 = mk3(C.m[6], C.m[7], C.m[8]);
        dx = dot(m1, ld);
        dy = dot(m2, ld);
        dz = dot(m3, ld);
        ox = dot(m1, lo) + C.pos[0];
        oy = dot(m2, lo) + C.pos[1];
        oz = dot(m3, lo) + C.pos[2];
    } else {
        const float* r = C.rays + (size_t)gid * 6;
        ox = r[0];
        oy = r[1];
        oz = r[2];
        dx = r[3];
        dy = r[4];
        dz = r[5];
    }
    return RayOD{mk3(ox, oy, oz), mk3(dx, dy, dz)};
}
DEV RayOD primary_ray(const CameraView& C, int gid, unsigned& rng, bool unit_dir) {
    return primary_ray(C, gid, rng, unit_dir, gid % C.width, gid / C.width);
}

}  // namespace chunky
