/* ref_shim.cpp — host runtime that lets the REFERENCE kernel run on x86-64.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/): never linked into or called by the product path.
 *
 * oracle/Makefile compiles /root/reference/src/main/opencl/kernel/include/rayTracer.cl IN PLACE
 * (nothing is copied into this repo) with ROCm clang as OpenCL C 1.2 for x86-64, using the
 * reference's own flags (-cl-std=CL1.2 -Werror, KernelLoader.java:52) plus -ffp-contract=off.
 * That object leaves 27 OpenCL builtins undefined; this file defines them (Itanium-mangled names
 * via asm labels) on top of chunkyclplugin_amd/csrc/rt_math.h — the same definitions the HIP
 * kernels use — and adds plain-pointer extern "C" drivers so tests can call the reference from
 * ctypes.  Result: oracle/_ref/libchunky_ref.so (git-ignored; built only where /root/reference
 * exists).
 *
 * Struct mirrors below restate only the memory layout of the reference's private structs
 * (wavefront.h:6-56, sky.h:9-17) so the drivers can read hit records back.
 */
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

#include "../chunkyclplugin_amd/csrc/rt_math.h"
#include "oracle_scene.h"

typedef float cl_float2 __attribute__((ext_vector_type(2)));
typedef float cl_float3 __attribute__((ext_vector_type(3)));
typedef float cl_float4 __attribute__((ext_vector_type(4)));
typedef int cl_int2 __attribute__((ext_vector_type(2)));
typedef int cl_int4 __attribute__((ext_vector_type(4)));

/* ------------------------------------------------------------------ images + samplers ---- */
struct ShimImage {
    const uint8_t* rgba;
    int w, h, layers;
};

static thread_local size_t tls_gid = 0;

#define OCL(name) __asm__(name)

extern "C" {
size_t ocl_get_global_id(unsigned) OCL("_Z13get_global_idj");
size_t ocl_get_global_id(unsigned) { return tls_gid; }

/* sampler_t is an opaque pointer on x86; keep the initializer bits in it */
void* ocl_translate_sampler(int init) OCL("__translate_sampler_initializer");
void* ocl_translate_sampler(int init) { return (void*)(intptr_t)init; }

/* REF_SHIM_LIBM (make ref_libm): the same reference object linked against ANOTHER conforming platform — glibc's
 * correctly-rounded-ish transcendentals, unfused dot / cross / normalize — for the tolerance study of
 * tools/tolerance_study.py: how far two conforming OpenCL platforms drift apart on whole images (north_star: 1e-5). */
#ifdef REF_SHIM_LIBM
#include <math.h>
#define ALT(native, ours) (native)
#else
#define ALT(native, ours) (ours)
#endif
float ocl_cos(float x) OCL("_Z3cosf");
float ocl_cos(float x) { return ALT(cosf(x), rt_cos(x)); }
float ocl_sin(float x) OCL("_Z3sinf");
float ocl_sin(float x) { return ALT(sinf(x), rt_sin(x)); }
float ocl_acos(float x) OCL("_Z4acosf");
float ocl_acos(float x) { return ALT(acosf(x), rt_acos(x)); }
float ocl_asin(float x) OCL("_Z4asinf");
float ocl_asin(float x) { return ALT(asinf(x), rt_asin(x)); }
float ocl_fabs(float x) OCL("_Z4fabsf");
float ocl_fabs(float x) { return rt_fabs(x); }
float ocl_sqrt(float x) OCL("_Z4sqrtf");
float ocl_sqrt(float x) { return rt_sqrt(x); }
cl_float4 ocl_sqrt4(cl_float4 v) OCL("_Z4sqrtDv4_f");
cl_float4 ocl_sqrt4(cl_float4 v) {
    return (cl_float4){rt_sqrt(v.x), rt_sqrt(v.y), rt_sqrt(v.z), rt_sqrt(v.w)};
}
float ocl_atan2(float y, float x) OCL("_Z5atan2ff");
float ocl_atan2(float y, float x) { return ALT(atan2f(y, x), rt_atan2(y, x)); }
float ocl_fmod(float x, float y) OCL("_Z4fmodff");
float ocl_fmod(float x, float y) {
    /* the reference only ever calls fmod(., 1) (sky.h:102) */
    if (y == 1.0f) return rt_fmod1(x);
    return __builtin_fmodf(x, y);
}
float ocl_fmax(float a, float b) OCL("_Z4fmaxff");
float ocl_fmax(float a, float b) { return rt_fmax(a, b); }
float ocl_fmin(float a, float b) OCL("_Z4fminff");
float ocl_fmin(float a, float b) { return rt_fmin(a, b); }
cl_float3 ocl_fmax3(cl_float3 a, cl_float3 b) OCL("_Z4fmaxDv3_fS_");
cl_float3 ocl_fmax3(cl_float3 a, cl_float3 b) {
    return (cl_float3){rt_fmax(a.x, b.x), rt_fmax(a.y, b.y), rt_fmax(a.z, b.z)};
}
cl_float3 ocl_fmin3(cl_float3 a, cl_float3 b) OCL("_Z4fminDv3_fS_");
cl_float3 ocl_fmin3(cl_float3 a, cl_float3 b) {
    return (cl_float3){rt_fmin(a.x, b.x), rt_fmin(a.y, b.y), rt_fmin(a.z, b.z)};
}
float ocl_clampf(float x, float lo, float hi) OCL("_Z5clampfff");
float ocl_clampf(float x, float lo, float hi) { return rt_clamp(x, lo, hi); }
int ocl_clampi(int x, int lo, int hi) OCL("_Z5clampiii");
int ocl_clampi(int x, int lo, int hi) { return rt_clampi(x, lo, hi); }
cl_float3 ocl_clamp3(cl_float3 v, float lo, float hi) OCL("_Z5clampDv3_fff");
cl_float3 ocl_clamp3(cl_float3 v, float lo, float hi) {
    return (cl_float3){rt_clamp(v.x, lo, hi), rt_clamp(v.y, lo, hi), rt_clamp(v.z, lo, hi)};
}
int ocl_isnan(float x) OCL("_Z5isnanf");
int ocl_isnan(float x) { return rt_isnan(x); }
float ocl_dot(cl_float3 a, cl_float3 b) OCL("_Z3dotDv3_fS_");
float ocl_dot(cl_float3 a, cl_float3 b) { return ALT(a.x * b.x + a.y * b.y + a.z * b.z, rt_dot3(a.x, a.y, a.z, b.x, b.y, b.z)); }
cl_float3 ocl_cross(cl_float3 a, cl_float3 b) OCL("_Z5crossDv3_fS_");
cl_float3 ocl_cross(cl_float3 a, cl_float3 b) {
#ifdef REF_SHIM_LIBM
    return (cl_float3){a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
#else
    return (cl_float3){rt_cross_c(a.y, b.z, a.z, b.y), rt_cross_c(a.z, b.x, a.x, b.z),
                       rt_cross_c(a.x, b.y, a.y, b.x)};
#endif
}
cl_float3 ocl_normalize(cl_float3 v) OCL("_Z9normalizeDv3_f");
cl_float3 ocl_normalize(cl_float3 v) {
#ifdef REF_SHIM_LIBM
    float len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return (cl_float3){v.x / len, v.y / len, v.z / len};
#else
    float r = rt_rlen3(v.x, v.y, v.z);
    return (cl_float3){v.x * r, v.y * r, v.z * r};
#endif
}
cl_float3 ocl_floor3(cl_float3 v) OCL("_Z5floorDv3_f");
cl_float3 ocl_floor3(cl_float3 v) { return (cl_float3){rt_floor(v.x), rt_floor(v.y), rt_floor(v.z)}; }
cl_float3 ocl_vload3(size_t off, const float* p) OCL("_Z6vload3mPU8CLglobalKf");
cl_float3 ocl_vload3(size_t off, const float* p) {
    return (cl_float3){p[3 * off], p[3 * off + 1], p[3 * off + 2]};
}
void ocl_vstore3(cl_float3 v, size_t off, float* p) OCL("_Z7vstore3Dv3_fmPU8CLglobalf");
void ocl_vstore3(cl_float3 v, size_t off, float* p) {
    p[3 * off] = v.x;
    p[3 * off + 1] = v.y;
    p[3 * off + 2] = v.z;
}

/* ---- builtins used only by the tone-map kernel (tonemap/include/post_processing_filter.cl) ---- */
typedef unsigned cl_uint4 __attribute__((ext_vector_type(4)));
cl_float3 ocl_pow3(cl_float3 a, cl_float3 b) OCL("_Z3powDv3_fS_");
cl_float3 ocl_pow3(cl_float3 a, cl_float3 b) { return (cl_float3){rt_pow(a.x, b.x), rt_pow(a.y, b.y), rt_pow(a.z, b.z)}; }
cl_float3 ocl_clamp3v(cl_float3 v, cl_float3 lo, cl_float3 hi) OCL("_Z5clampDv3_fS_S_");
cl_float3 ocl_clamp3v(cl_float3 v, cl_float3 lo, cl_float3 hi) {
    return (cl_float3){rt_clamp(v.x, lo.x, hi.x), rt_clamp(v.y, lo.y, hi.y), rt_clamp(v.z, lo.z, hi.z)};
}
cl_uint4 ocl_clampu4(cl_uint4 v, cl_uint4 lo, cl_uint4 hi) OCL("_Z5clampDv4_jS_S_");
cl_uint4 ocl_clampu4(cl_uint4 v, cl_uint4 lo, cl_uint4 hi) {
    cl_uint4 r;
    for (int i = 0; i < 4; i++) r[i] = v[i] < lo[i] ? lo[i] : (v[i] > hi[i] ? hi[i] : v[i]);
    return r;
}
cl_float3 ocl_vload3p(size_t off, const float* p) OCL("_Z6vload3mPU9CLprivateKf");
cl_float3 ocl_vload3p(size_t off, const float* p) { return (cl_float3){p[3 * off], p[3 * off + 1], p[3 * off + 2]}; }

/* atlas: CLK_NORMALIZED_COORDS_FALSE | CLK_ADDRESS_CLAMP_TO_EDGE | CLK_FILTER_NEAREST, integer
 * coordinates (textureAtlas.h:8,15); array layer clamped to [0, layers-1] (OpenCL 1.2 8.4). */
cl_float4 ocl_read_image_array(const ShimImage* img, void* smp, cl_int4 c)
    OCL("_Z11read_imagef20ocl_image2d_array_ro11ocl_samplerDv4_i");
cl_float4 ocl_read_image_array(const ShimImage* img, void*, cl_int4 c) {
    int x = rt_clampi(c.x, 0, img->w - 1);
    int y = rt_clampi(c.y, 0, img->h - 1);
    int l = rt_clampi(c.z, 0, img->layers - 1);
    const uint8_t* t = img->rgba + 4 * (((size_t)l * img->h + y) * img->w + x);
    return (cl_float4){rt_unorm8(t[0]), rt_unorm8(t[1]), rt_unorm8(t[2]), rt_unorm8(t[3])};
}

/* sky: CLK_NORMALIZED_COORDS_TRUE | CLK_ADDRESS_MIRRORED_REPEAT | CLK_FILTER_LINEAR (sky.h:95) */
cl_float4 ocl_read_image_2d(const ShimImage* img, void* smp, cl_float2 c)
    OCL("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_f");
cl_float4 ocl_read_image_2d(const ShimImage* img, void*, cl_float2 c) {
    int i0, i1, j0, j1;
    float a, b;
    rt_mirror_linear(c.x, img->w, &i0, &i1, &a);
    rt_mirror_linear(c.y, img->h, &j0, &j1, &b);
    const uint8_t* t00 = img->rgba + 4 * ((size_t)j0 * img->w + i0);
    const uint8_t* t10 = img->rgba + 4 * ((size_t)j0 * img->w + i1);
    const uint8_t* t01 = img->rgba + 4 * ((size_t)j1 * img->w + i0);
    const uint8_t* t11 = img->rgba + 4 * ((size_t)j1 * img->w + i1);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    cl_float4 r;
    for (int k = 0; k < 4; k++) {
        r[k] = w00 * rt_unorm8(t00[k]) + w10 * rt_unorm8(t10[k]) + w01 * rt_unorm8(t01[k]) +
               w11 * rt_unorm8(t11[k]);
    }
    return r;
}
} /* extern "C" builtins */

/* ------------------------------------------------ reference entry points (defined in rt .o) -- */
struct RefPixel {          /* wavefront.h:6-11   48 B */
    int index;
    cl_float3 color;
    cl_float3 throughput;
};
struct RefRay {            /* wavefront.h:21-29  64 B */
    RefPixel* pixel;
    cl_float3 origin;
    cl_float3 direction;
    int material;
    int rayDepth;
};
struct RefRecord {         /* wavefront.h:39-51  96 B */
    RefPixel* pixel;
    RefRay* ray;
    float distance;
    int material;
    cl_float3 normal;
    cl_float3 point;
    cl_float4 color;
    float emittance;
};
struct RefSun {            /* sky.h:9-17         64 B */
    int flags, textureSize, texture;
    float intensity;
    cl_float3 su, sv, sw;
};
struct RefOctree { const int* data; int depth; };                     /* octree.h:11-14 */
struct RefMatPalette { const int* palette; };                         /* material.h:12-14 */
struct RefBvh { const int* bvh; const int* trigs; RefMatPalette* mp; };  /* bvh.h:8-12 */
struct RefBlockPalette { const int* blocks; const int* quads; const int* aabbs; RefMatPalette* mp; }; /* block.h:14-19 */

static_assert(sizeof(RefPixel) == 48 && sizeof(RefRay) == 64 && sizeof(RefRecord) == 96 &&
              sizeof(RefSun) == 64, "struct mirror out of sync with the OpenCL compile");

extern "C" {
void render(const int*, const float*, const int*, const int*, const int*, const int*, const int*,
            const int*, const int*, const int*, const ShimImage*, const int*, const ShimImage*,
            const float*, const int*, const int*, const int*, const int*, const int*, float*);
void preview(const int*, const float*, const int*, const int*, const int*, const int*, const int*,
             const int*, const int*, const int*, const ShimImage*, const int*, const ShimImage*,
             const float*, const int*, const int*, const int*, int*);
RefSun Sun_new(const int*);
bool closestIntersect(RefRecord*, RefOctree*, RefBlockPalette*, const ShimImage*, int, RefBvh*, RefBvh*);
void intersectSky(RefRecord*, const ShimImage*, RefSun*, const ShimImage*, float);
void applyRayColor(RefRecord*, float);
bool Sun_sampleDirection(RefSun*, RefRecord*, unsigned*);
bool nextPath(RefRecord*, unsigned*, int);
RefRecord IntersectionRecord_copy(RefRecord*);
unsigned Random_nextState(unsigned*);
float Random_nextFloat(unsigned*);
void Camera_pinHole(float, float, unsigned*, cl_float3*, cl_float3*, const float*);
}

static void parallel_for(int64_t begin, int64_t end, int threads, void (*fn)(int64_t, void*), void* ctx) {
    if (threads <= 1) {
        for (int64_t i = begin; i < end; i++) fn(i, ctx);
        return;
    }
    std::atomic<int64_t> next(begin);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([&]() {
            for (;;) {
                int64_t b = next.fetch_add(256);
                if (b >= end) break;
                int64_t e = b + 256 < end ? b + 256 : end;
                for (int64_t i = b; i < e; i++) fn(i, ctx);
            }
        });
    }
    for (auto& th : pool) th.join();
}

struct RenderCtx {
    const OracleScene* sc;
    ShimImage atlas, sky;
    int seed, spp;
    float* res;
    int* argb;
};

static void render_one(int64_t gid, void* p) {
    RenderCtx* c = (RenderCtx*)p;
    const OracleScene* s = c->sc;
    tls_gid = (size_t)gid;
    render(&s->projector_type, s->camera_settings, &s->octree_depth, s->octree, s->block_palette,
           s->quad_models, s->aabb_models, s->world_bvh, s->actor_bvh, s->bvh_trigs, &c->atlas,
           s->material_palette, &c->sky, &s->sky_intensity, s->sun, &c->seed, &c->spp, &s->width,
           &s->height, c->res);
}
static void preview_one(int64_t gid, void* p) {
    RenderCtx* c = (RenderCtx*)p;
    const OracleScene* s = c->sc;
    tls_gid = (size_t)gid;
    preview(&s->projector_type, s->camera_settings, &s->octree_depth, s->octree, s->block_palette,
            s->quad_models, s->aabb_models, s->world_bvh, s->actor_bvh, s->bvh_trigs, &c->atlas,
            s->material_palette, &c->sky, &s->sky_intensity, s->sun, &s->width, &s->height, c->argb);
}

static void make_ctx(RenderCtx* c, const OracleScene* s) {
    c->sc = s;
    c->atlas = ShimImage{s->atlas, s->atlas_w, s->atlas_h, s->atlas_layers};
    c->sky = ShimImage{s->sky, s->sky_w, s->sky_h, 1};
}

extern "C" {

/* One launch of the reference `render` kernel per pass over gids [gid_begin, gid_end)
 * (host loop of OpenClPathTracingRenderer.java:102-144: seed_k, bufferSpp = first_spp + k). */
int ref_render_passes(const OracleScene* s, const int32_t* seeds, int n_passes, int first_spp,
                      int64_t gid_begin, int64_t gid_end, float* res, int threads) {
    RenderCtx c;
    make_ctx(&c, s);
    c.res = res;
    for (int k = 0; k < n_passes; k++) {
        c.seed = seeds[k];
        c.spp = first_spp + k;
        parallel_for(gid_begin, gid_end, threads, render_one, &c);
    }
    return 0;
}

int ref_preview(const OracleScene* s, int32_t* argb, int threads) {
    RenderCtx c;
    make_ctx(&c, s);
    c.argb = argb;
    parallel_for(0, (int64_t)s->width * s->height, threads, preview_one, &c);
    return 0;
}

/* Re-drive the loop of rayTracer.cl:93-107 for one (gid, seed) with the reference's exported
 * helpers, recording every closestIntersect outcome (main and shadow traces, in call order).
 * Returns the number of traces; *radiance = pixel.color at the end. */
int ref_trace_records(const OracleScene* s, int seed, int gid, OracleHit* out, float* radiance) {
    ShimImage atlas{s->atlas, s->atlas_w, s->atlas_h, s->atlas_layers};
    ShimImage sky{s->sky, s->sky_w, s->sky_h, 1};
    RefPixel pixel;
    pixel.index = gid;
    pixel.color = (cl_float3){0, 0, 0};
    pixel.throughput = (cl_float3){1, 1, 1};
    RefRay ray;
    std::memset(&ray, 0, sizeof ray);
    ray.pixel = &pixel;
    RefRecord rec;
    std::memset(&rec, 0, sizeof rec);
    rec.pixel = &pixel;
    rec.ray = &ray;
    rec.distance = rt_inf();
    RefMatPalette mp{s->material_palette};
    RefOctree oct{s->octree, s->octree_depth};
    RefBvh wb{s->world_bvh, s->bvh_trigs, &mp}, ab{s->actor_bvh, s->bvh_trigs, &mp};
    RefBlockPalette bp{s->block_palette, s->quad_models, s->aabb_models, &mp};
    RefSun sun = Sun_new(s->sun);

    unsigned state = (unsigned)seed + (unsigned)gid;
    Random_nextState(&state);
    if (s->projector_type != -1) {
        const float* cs = s->camera_settings;
        /* film coordinates exactly as rayTracer.cl:66-69 (double sites) */
        float halfWidth = (float)(s->width / (2.0 * s->height));
        float invHeight = (float)(1.0 / s->height);
        float x = (float)(-halfWidth + ((gid % s->width) + Random_nextFloat(&state)) * invHeight);
        float y = (float)(-0.5 + ((gid / s->width) + Random_nextFloat(&state)) * invHeight);
        cl_float3 o, d;
        Camera_pinHole(x, y, &state, &o, &d, cs + 12);
        cl_float3 m1{cs[3], cs[4], cs[5]}, m2{cs[6], cs[7], cs[8]}, m3{cs[9], cs[10], cs[11]};
        ray.direction = (cl_float3){ocl_dot(m1, d), ocl_dot(m2, d), ocl_dot(m3, d)};
        ray.origin = (cl_float3){ocl_dot(m1, o), ocl_dot(m2, o), ocl_dot(m3, o)};
        ray.origin += (cl_float3){cs[0], cs[1], cs[2]};
    } else {
        const float* r = s->camera_settings + (size_t)gid * 6;
        ray.origin = (cl_float3){r[0], r[1], r[2]};
        ray.direction = (cl_float3){r[3], r[4], r[5]};
    }
    int n = 0;
    auto put = [&](bool hit, const RefRecord& r) {
        OracleHit& h = out[n++];
        h.hit = hit;
        h.material = r.material;
        h.distance = r.distance;
        for (int k = 0; k < 3; k++) { h.normal[k] = r.normal[k]; h.point[k] = r.point[k]; }
        for (int k = 0; k < 4; k++) h.color[k] = r.color[k];
        h.emittance = r.emittance;
    };
    do {
        bool hit = closestIntersect(&rec, &oct, &bp, &atlas, 256, &wb, &ab);
        put(hit, rec);
        if (!hit) {
            rec.emittance = 1;
            intersectSky(&rec, &atlas, &sun, &sky, s->sky_intensity);
            break;
        }
        applyRayColor(&rec, 13.0f);
        if (Sun_sampleDirection(&sun, &rec, &state)) {
            RefRecord sr = IntersectionRecord_copy(&rec);
            bool sh = closestIntersect(&sr, &oct, &bp, &atlas, 256, &wb, &ab);
            put(sh, sr);
            if (!sh) intersectSky(&sr, &atlas, &sun, &sky, s->sky_intensity);
        }
    } while (nextPath(&rec, &state, 5));
    radiance[0] = pixel.color.x;
    radiance[1] = pixel.color.y;
    radiance[2] = pixel.color.z;
    return n;
}

/* ---- helper-level known answers: the reference object's own exported helpers, one call per input row ----
 * (tests/golden/generate.py write_helpers -> tests/golden/helpers.npz; oracle/port.c port_helpers and the device's
 * chunky_selftest_helpers evaluate their counterparts on the same rows).  A row is HELPER_IN floats in, HELPER_OUT floats
 * out; ints travel as their bit patterns.  which:
 *   0 AABB_quick_intersect   in: box[0:6] o[6:9] d[9:12] (invDir = 1/d)                 out: t
 *   1 AABB_exit              same                                                       out: t
 *   2 AABB_full_intersect    unit box; o[6:9] dir[9:12] invDir = 1/in[12:15]            out: t n[3] uv[2]
 *   3 AABB_full_intersect_map_2  box o dir (invDir = 1/dir)                             out: t n[3] uv[2]
 *   4 BlockPalette_intersectBlock  block[0] cell[1:4] (ints) pos[4:7] dir[7:10]         out: t n[3] color[4] emittance
 *   6 Triangle_new + Triangle_intersect  tri[0:20] o[20:23] dir[23:26] distance[26]     out: t n[3] uv[2] material
 *   7 Sun_sampleDirection    state[0] normal[1:4]                                       out: dir[3] emittance state
 *   8 Sun_intersect          d[0:3] color[3:7]                                          out: color[4] hit
 *   9 Sky_intersect          d[0:3]                                                     out: color[4]
 *  10 nextPath               state[0] normal[1:4] point[4:7]                            out: dir[3] origin[3] state
 *  11 Atlas_read_uv          u v location size                                          out: color[4]
 *  12 Material_get + Material_sample  material[0] u v                                   out: ok color[4] emittance
 *  14 Octree_octreeIntersect o[0:3] d[3:6]                                              out: hit distance material n[3] color[4] emittance
 *  15 Bvh_intersect (world)  o[0:3] d[3:6] distance[6]                                  out: hit distance n[3] color[4] emittance */
#define HELPER_IN 32
#define HELPER_OUT 12
} /* extern "C" */
typedef int cl_int3v __attribute__((ext_vector_type(3)));
struct RefAABB { float xmin, xmax, ymin, ymax, zmin, zmax; };              /* primitives.h:8-15 */
struct RefTriangle { int flags; cl_float3 e1, e2, o, n; cl_float2 t1, t2, t3; int material; };  /* primitives.h:323-333 */
struct RefMaterial { unsigned w[6]; };                                     /* material.h:20-27 */
static_assert(sizeof(RefAABB) == 24 && sizeof(RefTriangle) == 112 && sizeof(RefMaterial) == 24, "struct mirror out of sync");
extern "C" {
RefAABB AABB_new(float, float, float, float, float, float);
float AABB_quick_intersect(RefAABB*, cl_float3, cl_float3);
float AABB_exit(RefAABB*, cl_float3, cl_float3);
float AABB_full_intersect(RefAABB*, cl_float3, cl_float3, cl_float3, cl_float3*, cl_float2*);
float AABB_full_intersect_map_2(RefAABB*, cl_float3, cl_float3, cl_float3, cl_float3*, cl_float2*);
float BlockPalette_intersectBlock(RefBlockPalette*, int, cl_int3v, RefRecord*, cl_float3, cl_float3, cl_float3, const ShimImage*);
RefTriangle Triangle_new(const int*, int);
float Triangle_intersect(RefTriangle*, float, cl_float3, cl_float3, cl_float3*, cl_float2*, int*);
bool Sun_intersect(RefSun*, RefRecord*, const ShimImage*);
void Sky_intersect(RefRecord*, const ShimImage*, float);
cl_float4 Atlas_read_uv(float, float, int, int, const ShimImage*);
RefMaterial Material_get(RefMatPalette*, int);
bool Material_sample(RefMaterial*, const ShimImage*, RefRecord*, cl_float2);
bool Octree_octreeIntersect(RefOctree*, RefRecord*, RefBlockPalette*, const ShimImage*, int);
bool Bvh_intersect(RefBvh*, RefRecord*, const ShimImage*);

static inline int f2i(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float i2f(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline cl_float3 ld3(const float* p) { return (cl_float3){p[0], p[1], p[2]}; }
static inline cl_float3 rcp3v(cl_float3 d) { return (cl_float3){1.0f / d.x, 1.0f / d.y, 1.0f / d.z}; }

void ref_helpers(const OracleScene* s, int which, int n, const float* in_rows, float* out_rows) {
    ShimImage atlas{s->atlas, s->atlas_w, s->atlas_h, s->atlas_layers};
    ShimImage sky{s->sky, s->sky_w, s->sky_h, 1};
    RefMatPalette mp{s->material_palette};
    RefOctree oct{s->octree, s->octree_depth};
    RefBvh wb{s->world_bvh, s->bvh_trigs, &mp};
    RefBlockPalette bp{s->block_palette, s->quad_models, s->aabb_models, &mp};
    RefSun sun = Sun_new(s->sun);
    for (int r = 0; r < n; r++) {
        const float* in = in_rows + (size_t)r * HELPER_IN;
        float* out = out_rows + (size_t)r * HELPER_OUT;
        for (int k = 0; k < HELPER_OUT; k++) out[k] = 0;
        RefPixel pixel;
        pixel.index = 0;
        pixel.color = (cl_float3){0, 0, 0};
        pixel.throughput = (cl_float3){1, 1, 1};
        RefRay ray;
        std::memset(&ray, 0, sizeof ray);
        ray.pixel = &pixel;
        RefRecord rec;
        std::memset(&rec, 0, sizeof rec);
        rec.pixel = &pixel;
        rec.ray = &ray;
        rec.distance = rt_inf();
        cl_float3 nrm = {0, 0, 0};
        cl_float2 uv = {0, 0};
        switch (which) {
            case 0: case 1: {
                RefAABB b = AABB_new(in[0], in[1], in[2], in[3], in[4], in[5]);
                out[0] = which == 0 ? AABB_quick_intersect(&b, ld3(in + 6), rcp3v(ld3(in + 9))) : AABB_exit(&b, ld3(in + 6), rcp3v(ld3(in + 9)));
                break;
            }
            case 2: case 3: {
                RefAABB b = which == 2 ? AABB_new(0, 1, 0, 1, 0, 1) : AABB_new(in[0], in[1], in[2], in[3], in[4], in[5]);
                out[0] = which == 2 ? AABB_full_intersect(&b, ld3(in + 6), ld3(in + 9), rcp3v(ld3(in + 12)), &nrm, &uv)
                                    : AABB_full_intersect_map_2(&b, ld3(in + 6), ld3(in + 9), rcp3v(ld3(in + 9)), &nrm, &uv);
                out[1] = nrm.x; out[2] = nrm.y; out[3] = nrm.z; out[4] = uv.x; out[5] = uv.y;
                break;
            }
            case 4: {
                ray.direction = ld3(in + 7);
                cl_int3v cell = {(int)in[1], (int)in[2], (int)in[3]};
                out[0] = BlockPalette_intersectBlock(&bp, f2i(in[0]), cell, &rec, ld3(in + 4), ld3(in + 7), rcp3v(ld3(in + 7)), &atlas);
                out[1] = rec.normal.x; out[2] = rec.normal.y; out[3] = rec.normal.z;
                for (int k = 0; k < 4; k++) out[4 + k] = rec.color[k];
                out[8] = rec.emittance;
                break;
            }
            case 6: {
                int tri[20];
                std::memcpy(tri, in, sizeof tri);
                RefTriangle t = Triangle_new(tri, 0);
                int mat = 0;
                out[0] = Triangle_intersect(&t, in[26], ld3(in + 20), ld3(in + 23), &nrm, &uv, &mat);
                out[1] = nrm.x; out[2] = nrm.y; out[3] = nrm.z; out[4] = uv.x; out[5] = uv.y; out[6] = i2f(mat);
                break;
            }
            case 7: {
                unsigned state = (unsigned)f2i(in[0]);
                rec.normal = ld3(in + 1);
                Sun_sampleDirection(&sun, &rec, &state);
                out[0] = ray.direction.x; out[1] = ray.direction.y; out[2] = ray.direction.z; out[3] = rec.emittance; out[4] = i2f((int)state);
                break;
            }
            case 8: {
                ray.direction = ld3(in);
                rec.color = (cl_float4){in[3], in[4], in[5], in[6]};
                const bool hit = Sun_intersect(&sun, &rec, &atlas);
                for (int k = 0; k < 4; k++) out[k] = rec.color[k];
                out[4] = hit ? 1.0f : 0.0f;
                break;
            }
            case 9: {
                ray.direction = ld3(in);
                Sky_intersect(&rec, &sky, s->sky_intensity);
                for (int k = 0; k < 4; k++) out[k] = rec.color[k];
                break;
            }
            case 10: {
                unsigned state = (unsigned)f2i(in[0]);
                rec.normal = ld3(in + 1);
                rec.point = ld3(in + 4);
                nextPath(&rec, &state, 5);
                out[0] = ray.direction.x; out[1] = ray.direction.y; out[2] = ray.direction.z;
                out[3] = ray.origin.x; out[4] = ray.origin.y; out[5] = ray.origin.z; out[6] = i2f((int)state);
                break;
            }
            case 11: {
                cl_float4 c = Atlas_read_uv(in[0], in[1], f2i(in[2]), f2i(in[3]), &atlas);
                for (int k = 0; k < 4; k++) out[k] = c[k];
                break;
            }
            case 12: {
                RefMaterial m = Material_get(&mp, f2i(in[0]));
                const bool ok = Material_sample(&m, &atlas, &rec, (cl_float2){in[1], in[2]});
                out[0] = ok ? 1.0f : 0.0f;
                for (int k = 0; k < 4; k++) out[1 + k] = rec.color[k];
                out[5] = rec.emittance;
                break;
            }
            case 14: case 15: {
                ray.origin = ld3(in);
                ray.direction = ld3(in + 3);
                if (which == 15) rec.distance = in[6];
                const bool hit = which == 14 ? Octree_octreeIntersect(&oct, &rec, &bp, &atlas, 256) : Bvh_intersect(&wb, &rec, &atlas);
                int o = 0;
                out[o++] = hit ? 1.0f : 0.0f;
                out[o++] = rec.distance;
                if (which == 14) out[o++] = i2f(rec.material);
                out[o++] = rec.normal.x; out[o++] = rec.normal.y; out[o++] = rec.normal.z;
                for (int k = 0; k < 4; k++) out[o++] = rec.color[k];
                out[o++] = rec.emittance;
                break;
            }
            default: break;
        }
    }
}

/* ---- known-answer helpers ---- */
void ref_pcg_stream(unsigned state, int n, unsigned* states, float* floats) {
    for (int i = 0; i < n; i++) {
        unsigned s2 = state;
        floats[i] = Random_nextFloat(&s2);
        states[i] = Random_nextState(&state);
    }
}
void ref_sun_basis(const int32_t* sun_data, float* out9) {
    RefSun s = Sun_new(sun_data);
    for (int k = 0; k < 3; k++) { out9[k] = s.su[k]; out9[3 + k] = s.sv[k]; out9[6 + k] = s.sw[k]; }
}
/* builtins as the reference sees them (for the rt_math accuracy tests) */
void ref_math(int which, int n, const float* a, const float* b, float* out) {
    for (int i = 0; i < n; i++) {
        switch (which) {
            case 0: out[i] = ocl_sin(a[i]); break;
            case 1: out[i] = ocl_cos(a[i]); break;
            case 2: out[i] = ocl_asin(a[i]); break;
            case 3: out[i] = ocl_acos(a[i]); break;
            case 4: out[i] = ocl_atan2(a[i], b[i]); break;
            case 5: out[i] = ocl_fmod(a[i], 1.0f); break;
            case 6: out[i] = ocl_fmin(a[i], b[i]); break;
            case 7: out[i] = ocl_fmax(a[i], b[i]); break;
            case 8: out[i] = ocl_sqrt(a[i]); break;
            case 9: out[i] = a[i] / b[i]; break;
            default: out[i] = 0;
        }
    }
}
} /* extern "C" */

/* the reference `filter` kernel (tonemap/include/post_processing_filter.cl:5-51), one call per work-item */
extern "C" void filter(int width, int height, float exposure, const unsigned long* input, unsigned* res, int type);
extern "C" void ref_filter(int n, int width, int height, float exposure, const uint64_t* input, uint32_t* res, int type) {
    for (int i = 0; i < n; i++) {
        tls_gid = (size_t)i;
        filter(width, height, exposure, (const unsigned long*)input, res, type);
    }
}
extern "C" void ref_pow(int n, const float* a, const float* b, float* out) {
    for (int i = 0; i < n; i++) {
        cl_float3 r = ocl_pow3((cl_float3){a[i], a[i], a[i]}, (cl_float3){b[i], b[i], b[i]});
        out[i] = r.x;
    }
}
