/* oracle_scene.h — plain-pointer description of one packed scene, shared by the two CPU checkers
 * (oracle/_ref: the reference kernel itself compiled for x86-64; oracle/port.c: the C restatement).
 * TEST INFRASTRUCTURE ONLY: nothing under chunkyclplugin_amd/ may include or link this.
 * Field meaning = the 20 arguments of the reference `render` kernel
 * (/root/reference/src/main/opencl/kernel/include/rayTracer.cl:11-37) with the two OpenCL images
 * replaced by flat RGBA8 arrays + dimensions. */
#ifndef CHUNKY_ORACLE_SCENE_H
#define CHUNKY_ORACLE_SCENE_H
#include <stdint.h>

typedef struct OracleScene {
    int32_t projector_type;        /* rayTracer.cl:12  (0 pinhole, -1 pre-generated rays) */
    const float* camera_settings;  /* rayTracer.cl:13  15 floats, or W*H*6 */
    int32_t octree_depth;          /* rayTracer.cl:15 */
    const int32_t* octree;         /* rayTracer.cl:16 */
    const int32_t* block_palette;  /* rayTracer.cl:18 */
    const int32_t* quad_models;    /* rayTracer.cl:19 */
    const int32_t* aabb_models;    /* rayTracer.cl:20 */
    const int32_t* world_bvh;      /* rayTracer.cl:22 */
    const int32_t* actor_bvh;      /* rayTracer.cl:23 */
    const int32_t* bvh_trigs;      /* rayTracer.cl:24 */
    const uint8_t* atlas;          /* rayTracer.cl:26  RGBA8, [layer][y][x][4] */
    int32_t atlas_w, atlas_h, atlas_layers;
    const int32_t* material_palette; /* rayTracer.cl:27 */
    const uint8_t* sky;            /* rayTracer.cl:29  RGBA8, [y][x][4] */
    int32_t sky_w, sky_h;
    float sky_intensity;           /* rayTracer.cl:30 */
    const int32_t* sun;            /* rayTracer.cl:31  6 ints */
    int32_t width, height;         /* rayTracer.cl:35-36 */
} OracleScene;

/* One record per trace of a path, written by *_trace_records (layout shared with the tests). */
typedef struct OracleHit {
    int32_t hit;        /* closestIntersect result */
    int32_t material;   /* record.material (block pointer; octree hits only) */
    float distance;
    float normal[3];
    float color[4];
    float emittance;
    float point[3];
} OracleHit;

#define ORACLE_MAX_TRACES 10 /* 5 path segments x (main + shadow) */

#endif
