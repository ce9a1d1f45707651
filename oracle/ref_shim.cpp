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
