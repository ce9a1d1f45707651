/* chunky_hip.h — C ABI of the MI355X-native Chunky path-tracing hot path.
 *
 * This is the boundary a Chunky plugin binds instead of JOCL/OpenCL: plain pointers and sizes,
 * no C++ or torch types.  Each entry point names the reference interface it replaces
 * (paths relative to /root/reference/src/main/java/dev/thatredox/chunkynative/, "J/", and
 * /root/reference/src/main/opencl/kernel/include/, "K/").  INTEGRATION.md shows the JNI stub a
 * maintainer adds on the Java side.
 *
 * Conventions:
 *   - every function returns 0 on success, a negative chunky_status on failure; the message is
 *     available from chunky_last_error() on the calling thread.  Nothing aborts the process.
 *   - host arrays are COPIED before the call returns (the reference creates every buffer with
 *     CL_MEM_COPY_HOST_PTR / blocking writes: J/opencl/util/ClIntBuffer.java:23-25); the caller
 *     keeps ownership.  Zero-length int arrays are legal (ClIntBuffer.java:15-18).
 *   - handles may be destroyed from any thread, at most once (the reference frees from a GC
 *     cleaner thread: J/util/NativeCleaner.java:45-53).  Calls on one context are serialised by
 *     an internal mutex (the reference's renderLock, J/opencl/OpenClPathTracingRenderer.java:56).
 *   - there is no CPU fallback: without a HIP device chunky_init fails with CHUNKY_E_NO_DEVICE.
 */
#ifndef CHUNKY_HIP_H
#define CHUNKY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum chunky_status {
    CHUNKY_OK = 0,
    CHUNKY_E_INVALID = -1,    /* bad argument / handle */
    CHUNKY_E_NO_DEVICE = -2,  /* no HIP device, or device index out of range */
    CHUNKY_E_HIP = -3,        /* a HIP runtime call failed (message has the HIP error string) */
    CHUNKY_E_STATE = -4,      /* call sequence error (e.g. render before a scene is complete) */
    CHUNKY_E_ABORTED = -5     /* the postRender callback asked to stop */
} chunky_status;

typedef struct chunky_ctx chunky_ctx;        /* one GPU: J/opencl/renderer/RendererInstance.java */
typedef struct chunky_scene chunky_scene;    /* device copies of the packed scene: J/opencl/renderer/ClSceneLoader.java */
typedef struct chunky_render chunky_render;  /* one render target + camera: OpenClPathTracingRenderer.render locals */

/* ---- device (replaces RendererInstance.java:31-110: platform/device enumeration, context, queue) */
int chunky_device_count(void);
int chunky_device_name(int device, char* buf, int buf_len);
int chunky_init(int device, chunky_ctx** out);
int chunky_shutdown(chunky_ctx* ctx);
/* Several GPUs behind ONE context, in one process (SURVEY.md section 8b "init(device_ids[], n)", 8e; the reference holds one
 * cl_context per JVM, RendererInstance.java:74-101, and a Chunky JVM is one process).  The handle is an ordinary chunky_ctx:
 * every scene / render / filter entry point accepts it.  A scene created on it is replicated on every member (each chunky_scene_set_*
 * uploads to all of them); a render target created on it is cut into blocks of 16 x 16 pixels dealt round-robin to the members
 * (member i renders blocks b with b % n == i, `gid` stays the global pixel index, so seeds and results are those of one GPU);
 * chunky_render_passes enqueues every member's share and returns; chunky_render_read is the one exchange per read-back: each
 * member packs the blocks it owns and sends them to member 0 (one grouped RCCL send / receive over xGMI, or peer copies where
 * RCCL is not available: chunky_group_transport; 1/n of the image per member) where they are scattered into the image, which is
 * then read back — bit for bit the one-GPU image.  `devices` may name a device more than once (members then share it: how the path is tested on a 1-GPU box).
 * chunky_shutdown destroys the members. */
int chunky_group_create(const int* devices, int n, chunky_ctx** out);
/* Members of a context: 1 for chunky_init's, n for chunky_group_create's. */
int chunky_group_size(chunky_ctx* ctx);
/* Device index of member i. */
int chunky_group_device(chunky_ctx* ctx, int i);
/* How each member's share of a read-back reaches member 0, decided once in chunky_group_create (no reference counterpart; the
 * reference is single-device): out[i] = CHUNKY_PEER_LOCAL (member 0 itself, or a member on member 0's device), CHUNKY_PEER_DIRECT
 * (hipDeviceEnablePeerAccess succeeded: hipMemcpyPeerAsync writes member 0's memory over xGMI), CHUNKY_PEER_STAGED (the devices
 * report no peer access: the runtime stages the copy through the host), or -(hipError_t) when enabling it failed (the copy still
 * works, staged).  n = room in out, at least chunky_group_size(ctx). */
#define CHUNKY_PEER_LOCAL 0
#define CHUNKY_PEER_DIRECT 1
#define CHUNKY_PEER_STAGED 2
int chunky_group_peer_status(chunky_ctx* ctx, int* out, int n);
/* What carries the one exchange per read-back of a group (no reference counterpart: one device, one queue,
 * RendererInstance.java:74-101; SURVEY.md section 8e "a single RCCL reduce of per-tile radiance over xGMI").
 * chunky_group_create binds RCCL at run time (dlopen: the collective library is not a link dependency; CHUNKY_RCCL_LIB names
 * the file, else librccl.so.1) and opens one communicator over the members (ncclCommInitAll) when they are distinct devices:
 *   CHUNKY_TRANSPORT_RCCL_SENDRECV  each member packs the blocks it owns and ncclSend's them, member 0 posts the matching
 *                                   ncclRecv's — ONE grouped RCCL operation per read-back, 1/n of the image per member —
 *                                   and scatters them into the image (the default when the communicator exists);
 *   CHUNKY_TRANSPORT_RCCL_REDUCE    ONE ncclReduce(sum) of the members' zero-padded framebuffers onto member 0, as SURVEY
 *                                   words it: every member moves the whole image, x + 0 = x keeps it bit-identical;
 *   CHUNKY_TRANSPORT_PEER_COPY      the packed blocks travel by hipMemcpyPeerAsync (chunky_group_peer_status says how):
 *                                   the fallback when RCCL cannot be bound, the communicator cannot be created (members
 *                                   sharing a device), or an RCCL call fails later — the render survives and `detail` says why.
 * All three leave the same bytes in the pixels the group's members own (with an outer chunky_render_set_shard split the REDUCE
 * form also zeroes the pixels of member 0's image that no member owns; the other two leave them alone).
 * First contact is checked, not assumed: before RCCL becomes a group's transport, chunky_group_create sends a known pattern from
 * every member to member 0 through the new communicators and compares the bytes; an error, a wrong byte or an exchange that does
 * not finish makes the group start on peer copies, with the reason in `detail`.  No exchange waits in the driver behind an RCCL
 * kernel: the library polls the members' streams and the communicators' asynchronous errors, and after 30 s without completion
 * (or on any RCCL error) it calls ncclCommAbort FIRST — the one call that ends a collective whose peer or link died — then drains
 * the streams and repeats that read-back, and every later one, on peer copies.
 * chunky_group_transport reports the transport the next read-back will use and a human-readable detail (library file and
 * version, or the reason for the fallback); chunky_group_set_transport picks one (CHUNKY_E_STATE when it needs a communicator
 * that does not exist).  On a chunky_init context: PEER_COPY, nothing to exchange.
 * Environment: the shipping library reads ONE variable, CHUNKY_RCCL_LIB (a deployment's own librccl file).  Everything else —
 * the initial transport, the one-GPU rigs of tests/test_gpu_rccl_transport.py (CHUNKY_GROUP_TRANSPORT, CHUNKY_GROUP_SELF_EXCHANGE,
 * CHUNKY_GROUP_NO_PROBE, CHUNKY_GROUP_TIMEOUT_MS, CHUNKY_RCCL_TRY_SHARED) and the tuning overrides of the octree / BVH re-layouts —
 * exists only in a build with -DCHUNKY_TUNING (chunkyclplugin_amd/native.py build_tuning), never in what a JVM loads. */
#define CHUNKY_TRANSPORT_PEER_COPY 0
#define CHUNKY_TRANSPORT_RCCL_SENDRECV 1
#define CHUNKY_TRANSPORT_RCCL_REDUCE 2
int chunky_group_transport(chunky_ctx* ctx, int* transport, char* detail, int detail_len);
int chunky_group_set_transport(chunky_ctx* ctx, int transport);
const char* chunky_last_error(void);
/* Library identity: "chunky-hip <version> gfx950". */
const char* chunky_version(void);

/* ---- scene upload (replaces the clCreateBuffer/clCreateImage sites of ClSceneLoader.java:52-150,
 *      ClIntBuffer.java:14-26, ClTextureLoader.java:50-67, ClSky.java:28-61) */
int chunky_scene_create(chunky_ctx* ctx, chunky_scene** out);
int chunky_scene_destroy(chunky_scene* scene);

/* References inside the scene data (the reference kernel follows them unchecked: clCreateBuffer copies ints, nothing validates them):
 *   - octree branch values and BVH node links are checked on upload (CHUNKY_E_INVALID: outside the array, cyclic, deeper than 64);
 *   - an octree leaf whose block pointer lies beyond the block palette renders as air;
 *   - a block whose model pointer, primitive count or material pointers leave their palettes never intersects, like a block of
 *     unknown model type (K/block.h:44-47);
 *   - an entity BVH whose leaves leave the triangle palette, or whose triangles' materials leave the material palette, makes
 *     chunky_render_passes / _run / _preview fail with CHUNKY_E_INVALID.
 * Well-formed data is unaffected.  The palettes may arrive in any order; the checks run when a render call first sees them together. */
/* octreeData after the leaf remap of ClSceneLoader.java:52-63 + octreeDepth (getOctreeData/getOctreeDepth) */
int chunky_scene_set_octree(chunky_scene* scene, const int32_t* tree, int64_t n_ints, int depth);
/* Same, from Chunky's raw PackedOctree.treeData + blockMapping: performs the remap
 * `i>0 || -i>=len ? i : -blockMapping[-i]` of ClSceneLoader.java:56-58 on the way in (a22). */
int chunky_scene_load_octree(chunky_scene* scene, const int32_t* tree_data, int64_t n_ints, int depth,
                             const int32_t* block_mapping, int64_t n_mapping);

typedef enum chunky_palette {
    CHUNKY_PALETTE_BLOCK = 0,    /* getBlockPalette():    2 ints/block      (PackedBlock.java:80-85) */
    CHUNKY_PALETTE_MATERIAL = 1, /* getMaterialPalette(): 6 ints/material   (PackedMaterial.java:89-100) */
    CHUNKY_PALETTE_AABB = 2,     /* getAabbPalette():     1+13n ints/model  (PackedAabbModel.java:41-47) */
    CHUNKY_PALETTE_QUAD = 3,     /* getQuadPalette():     1+15n ints/model  (PackedQuadModel.java:39-45) */
    CHUNKY_PALETTE_TRIG = 4      /* getTrigPalette():     1+20n ints/leaf   (PackedTriangleModel.java:28-34) */
} chunky_palette;
int chunky_scene_set_palette(chunky_scene* scene, int kind, const int32_t* data, int64_t n_ints);

typedef enum chunky_bvh { CHUNKY_BVH_WORLD = 0, CHUNKY_BVH_ACTOR = 1 } chunky_bvh;
/* getWorldBvh()/getActorBvh(): 7 ints/node (PackedBvhNode.java:16-31); {0, NaN x 6} = empty */
int chunky_scene_set_bvh(chunky_scene* scene, int which, const int32_t* nodes, int64_t n_ints);

/* Texture atlas (getTexturePalette().getAtlas()): RGBA8, [layer][y][x][4].  gfx950 has no image
 * instructions, so the atlas is a flat buffer; width/height need only cover the occupied tiles
 * (the reference allocates 8192x8192 per layer, ClTextureLoader.java:50-58; locations are
 * identical).  set_atlas allocates (and zero-fills when rgba == NULL); write_atlas_tile mirrors the
 * per-texture clEnqueueWriteImage of ClTextureLoader.java:61-66. */
int chunky_scene_set_atlas(chunky_scene* scene, const uint8_t* rgba, int width, int height, int layers);
int chunky_scene_write_atlas_tile(chunky_scene* scene, int x, int y, int layer, int w, int h, const uint8_t* rgba);
/* getSky(): skyTexture RGBA8 [h][w][4] + skyIntensity (ClSky.java:28-30,43-61) */
int chunky_scene_set_sky(chunky_scene* scene, const uint8_t* rgba, int width, int height, float intensity);
/* The emitter list CHUNKY_OPT_EMITTER_NEE samples (no reference counterpart): every octree leaf whose block is a full cube
 * with a non-zero emittance byte, in pre-order, 4 ints each {x, y, z, level << 25 | block pointer}.  *count receives the
 * number of emitters; at most `cap` are written. */
int chunky_scene_emitters(chunky_scene* scene, int32_t* out4, int32_t cap, int32_t* count);
/* getSun(): flags, textureSize, textureLocation, intensity, altitude, azimuth (PackedSun.java:32-41) */
int chunky_scene_set_sun(chunky_scene* scene, const int32_t sun[6]);

/* ---- render target (replaces the buffers and the launch of OpenClPathTracingRenderer.java:67-141) */
int chunky_render_create(chunky_ctx* ctx, chunky_scene* scene, int width, int height, chunky_render** out);
int chunky_render_destroy(chunky_render* r);
/* ClCamera (ClCamera.java:33-70): projector_type 0 = pinhole with 15 floats; -1 = pre-generated rays,
 * width*height*6 floats (ClCamera.java:72-105); any other value is rejected. */
int chunky_render_set_camera(chunky_render* r, int projector_type, const float* settings, int64_t n_floats);

typedef enum chunky_option {
    CHUNKY_OPT_DRAW_DEPTH = 0,      /* int, default 256  (K/rayTracer.cl:94) */
    CHUNKY_OPT_MAX_DEPTH = 1,       /* int >= 1, default 5 (K/rayTracer.cl:107) */
    CHUNKY_OPT_EMITTER_SCALE = 2,   /* float bits, default 13.0f (K/rayTracer.cl:99) */
    CHUNKY_OPT_KERNEL = 3,          /* int: kernel variant, 0 = default (the pool kernel); bit 0 reference octree layout, bit 1
                                     * one lane per path (render_lanes), bit 2 phase profile (pool kernel), bit 3 the
                                     * fallback kernel render_waves instead of the pool kernel, bits 4-5 (render_waves)
                                     * lanes per pixel 1/8/16, bits 6-7 (pool kernel) paths parked per wave none/32 instead
                                     * of 64, bit 8 / bit 9 (pool kernel) full cubes and model blocks tested in phases of their own: always / never
                                     * (default: where model blocks are common, from 3 % of the world's leaves on) (all bit-identical) */
    /* EXPERIMENTAL light-transport extensions (the reference has none of them — it gates sun
     * sampling on drawTexture, PackedSun.java:16 / K/sky.h:69, ignores emittersEnabled, and loads material word 5 without
     * using it, K/material.h:38).  Specification: oracle/port.c trace_sample_ext; DESIGN.md section 9.  The defaults are the
     * reference's behaviour and run the reference kernels. */
    CHUNKY_OPT_SUN_SAMPLING = 4,    /* int: -1 as the reference (PackedSun flag bit 0), 0 never, 1 always (Chunky sunEnabled) */
    CHUNKY_OPT_EMITTERS = 5,        /* int: 1 (default) / 0 = Chunky emittersEnabled false */
    CHUNKY_OPT_BSDF = 6,            /* int: 0 (default) / 1 = specular, metalness, roughness from material word 5 (PackedMaterial.java:69-71) */
    CHUNKY_OPT_EMITTER_NEE = 7,     /* int: 0 (default) / 1 = next-event estimation towards the scene's emitter blocks */
    /* EXTENSION, not the reference's walk: 1 = in the entity-BVH traversal a child whose box lies entirely BEHIND the ray origin
     * (the far end of the slab test is negative) counts as missed.  The reference's quick test (K/primitives.h:30-48, used by
     * K/bvh.h:72-85) has no such exit: it descends into every box the ray's LINE pierces, and about half of its node visits and
     * triangle tests are spent behind the origin (EXPERIMENTS.md 4.4).  A triangle there can only be "hit" when rounding noise
     * carries its barycentric test across, and that DOES happen: with entity boxes on the block grid and a bounce origin within a
     * few ulps of a box edge, the reference accepts a hit the culled walk never tests — 40 of 10^8 adversarial traces
     * (EXPERIMENTS.md 5.3, tests/test_bvh_cull.py).  Whole frames are usually identical to the reference's, single pixels
     * occasionally not: hence an option, and 0 the default.  Specification: oracle/port.c with port_set_bvh_cull(1). */
    CHUNKY_OPT_BVH_CULL_BEHIND = 8  /* int: 0 (default) / 1 */
} chunky_option;
int chunky_render_set_option(chunky_render* r, int option, int32_t value);

/* Multi-GPU image-tile ownership (no reference counterpart — the reference is single-device):
 * tile = 0: the image is cut into blocks of 16 x 16 pixels (row-major over blocks, edge blocks
 * partial), block b belongs to rank b % world — the shape the pool kernel renders in, so a share
 * runs as fast per pixel as the whole image (other kernels refuse it); tile > 0: pixel indices are
 * cut into runs of `tile` consecutive gids, run t belongs to rank t % world (every kernel).
 * A rank renders only its tiles; every other pixel of its buffer stays 0, so a SUM reduce over
 * ranks (one RCCL collective per read-back) reproduces the 1-GPU image bit for bit. */
int chunky_render_set_shard(chunky_render* r, int rank, int world, int tile);
/* (On a group's render target the members split the caller's share again: member i of n renders as rank + world * i of world * n.) */
/* Use a caller-owned device buffer (3*width*height floats) as the framebuffer, e.g. a torch tensor
 * that torch.distributed reduces over RCCL.  NULL returns to the internal buffer.
 * Ordering contract: the library runs on its own non-blocking stream and only synchronises THAT stream here.  The caller
 * must have completed every write of its own to the buffer (e.g. torch.zeros: torch.cuda.synchronize() first) before the
 * next chunky_render_passes, and must call chunky_render_sync before it reads or reduces the buffer on another stream. */
int chunky_render_set_device_buffer(chunky_render* r, void* device_ptr);
int chunky_render_device_buffer(chunky_render* r, void** device_ptr);
/* Zero the device framebuffer (the reference uploads a zeroed passBuffer, OpenClPathTracingRenderer.java:61,71). */
int chunky_render_reset(chunky_render* r);

/* Enqueue n passes: pass k uses seeds[k] as *randomSeed and first_buffer_spp + k as *bufferSpp
 * (OpenClPathTracingRenderer.java:106-141; n = 1 reproduces the reference's one launch per spp).
 * Asynchronous; chunky_render_sync / chunky_render_read wait. */
int chunky_render_passes(chunky_render* r, const int32_t* seeds, int n, int first_buffer_spp);
int chunky_render_sync(chunky_render* r);
/* Blocking read-back of the running-mean buffer, 3*width*height floats (clEnqueueReadBuffer,
 * OpenClPathTracingRenderer.java:164-166). */
int chunky_render_read(chunky_render* r, float* out, int64_t n_floats);
/* The exchange of chunky_render_read without the copy to the host: waits for the queued passes; on a group it then gathers
 * every member's blocks into member 0's device buffer (chunky_render_device_buffer), which holds the whole image afterwards.
 * On a single-device target it is chunky_render_sync. */
int chunky_render_gather(chunky_render* r);
/* Device time of the render kernels enqueued since the last call, from HIP events on the stream the
 * kernels run on: total milliseconds and number of launches. */
int chunky_render_kernel_time(chunky_render* r, float* total_ms, int* launches);
/* (On a group: the largest member total — the members run concurrently — and member 0's launch count.) */

/* Which kernel instantiation the most recent chunky_render_passes launch ran (no reference counterpart; the parity
 * tests assert it, so that a comparison with the oracle is a comparison of the kernel that is timed): out8 =
 * {tree form: 0 reference octree layout (K/octree.h:81-89), -1 generic wide tree, 16 + n dense top node over n levels
 * of 8x8x8 nodes; lanes per pixel (0 = one lane per pixel for the whole launch); entity-BVH phases present (K/bvh.h:22-113);
 * workgroups launched; paths parked per wave (pool kernel; -1 = the grouped kernel); extended integrator; the most passes one
 * launch of this target carries (256, fewer when the staged samples of a launch would not fit: chunky_render_passes cuts longer
 * requests into launches of that many); 1 when the launch tested full cubes and model blocks in phases of their own}.  On a group:
 * member 0's launch. */
int chunky_render_kernel_info(chunky_render* r, int32_t out8[8]);

/* Profile of the wave-scheduled kernel, filled only while CHUNKY_OPT_KERNEL has bit 2 set: for each of
 * the phases MARCH, BLOCK, SHADE the number of wave-level executions, the lanes active in them and
 * the shader cycles spent (s_memtime), summed over all waves since the last reset (9 values), then
 * the sum and the maximum of the wave lifetimes and the number of waves (3 values), then the executions
 * and cycles of the pixel/pass hand-over part of SHADE (2 values), then ten cycle sums of parts of SHADE
 * (sky lookup, direction sampling, trace setup, then the hand-over's deposit, fold, pixel opening, pass hand-out,
 * new sample; then, of the pool kernel's phase for model blocks — where it tests them apart from the full cubes, CHUNKY_OPT_KERNEL
 * bit 8 — lanes (executions in bits 40 up) and cycles): 24 values in all. */
int chunky_render_phase_stats(chunky_render* r, uint64_t* out24, int reset);

/* Preview kernel (K/rayTracer.cl:115-217; OpenClPreviewRenderer.java:47-115): width*height ARGB ints. */
int chunky_render_preview(chunky_render* r, int32_t* argb_out);

/* Parity instrument: for each gid, the outcome of every closestIntersect of one sample (main and
 * shadow traces in call order) and the sample's radiance. */
typedef struct chunky_hit_record {
    int32_t hit;       /* closestIntersect result */
    int32_t material;  /* record.material: block-palette pointer of the octree hit */
    float distance;
    float normal[3];
    float color[4];
    float emittance;
    float point[3];
} chunky_hit_record;
#define CHUNKY_MAX_TRACES 10
int chunky_render_trace_records(chunky_render* r, int32_t seed, const int32_t* gids, int n,
                                chunky_hit_record* records /* n*CHUNKY_MAX_TRACES */, int32_t* counts /* n */,
                                float* radiance /* 3n */);

/* ---- host pass loop (replaces OpenClPathTracingRenderer.render, J/opencl/OpenClPathTracingRenderer.java:54-191):
 * seeds from java.util.Random(0).nextInt(), bufferSpp restarting at 0 after each read-back, merge
 * sample = (sample*sampSpp + pass*passSpp) / (sampSpp+passSpp) in double (:167-173).
 * `sample_buffer` is Chunky's double[3*W*H]; `*scene_spp` its scene.spp (in/out).  `post_render`
 * (may be NULL) is polled at least every 100 ms and at every merge; non-zero stops the loop
 * (:153-157,163).  `merge_interval` = passes per read-back (the reference uses 1024, :158). */
typedef int (*chunky_post_render_fn)(void* user);
int chunky_render_run(chunky_render* r, double* sample_buffer, int32_t* scene_spp, int32_t target_spp,
                      int32_t merge_interval, chunky_post_render_fn post_render, void* user);

/* The same loop with every hook the reference's loop has (any pointer may be NULL):
 *   post_render       BooleanSupplier postRender: polled at least every 100 ms between launches and before every merge;
 *                     non-zero stops the loop (OpenClPathTracingRenderer.java:153-157,163,181).
 *   progress          after every launch, with the new scene.spp (the reference increments scene.spp per pass, :144).
 *   merged            after every merge into sample_buffer, with the spp the sample buffer now holds: the place of
 *                     scene.postProcessFrame + manager.redrawScreen (:172-177).  sample_buffer is complete and not
 *                     touched by the library while the callback runs.
 *   save_event        the two conditions that force a merge (:150): returns 1 when isSaveEvent(manager.getSnapshotControl(),
 *                     scene, spp) holds (:193-195) — a snapshot or render dump is due when the scene reaches `spp` — and 2 when
 *                     only scene.shouldFinalizeBuffer() does.  The loop asks for every spp the next launch would cover and cuts
 *                     the launch there, merges at once (:151,162-178); after a 1 it polls post_render once more after `merged`
 *                     (the reference makes that poll for real save events only, :179-182), after a 2 it does not.
 *   poll_gate         consulted before the TIMED poll only (the reference's `!manager.shouldFinalize()`, :154): zero skips that
 *                     poll.  The polls before a merge (:163) and after a save event (:181) are unconditional, as in the reference.
 *   regenerate_camera between launches, for projections other than pinhole: the reference re-generates the jittered
 *                     camera-ray table on a worker while passes run (:146-148, ClCamera.java:72-104).  The hook may
 *                     call chunky_render_set_camera (from this or any other thread: the context mutex is the
 *                     reference's renderLock) to install a fresh table; passes already queued finish with the old one.
 * chunky_render_run(..., post_render, user) is chunky_render_run_ex with only post_render set.
 * struct_size = sizeof(chunky_run_callbacks) as the CALLER was compiled: members are appended over time, and the library reads
 * only those the caller's struct holds (a host built against an older header keeps working; 0 or a size that cuts a member in
 * half is CHUNKY_E_INVALID). */
typedef struct chunky_run_callbacks {
    size_t struct_size;
    int (*post_render)(void* user);
    void (*progress)(void* user, int32_t scene_spp);
    void (*merged)(void* user, int32_t sample_spp);
    int (*save_event)(void* user, int32_t spp);
    void (*regenerate_camera)(void* user);
    void* user;
    int (*poll_gate)(void* user);
} chunky_run_callbacks;
int chunky_render_run_ex(chunky_render* r, double* sample_buffer, int32_t* scene_spp, int32_t target_spp,
                         int32_t merge_interval, const chunky_run_callbacks* callbacks);
/* The seed stream itself: first n values of new java.util.Random(seed).nextInt(). */
int chunky_java_random_ints(int64_t seed, int32_t* out, int n);

/* ---- tone mapping: the `filter` kernel (tonemap/include/post_processing_filter.cl:5-51) ----------------
 * Replaces GpuPostProcessingFilter.processFrame (GpuPostProcessingFilter.java:40-65): `input` is Chunky's
 * sample buffer, 3 doubles (R, G, B) per pixel; `argb_out` receives width*height words 0xFFRRGGBB.  `exposure`
 * is narrowed to float as the reference does (:53).  type: 0 GAMMA, 1 TONEMAP1, 2 ACES ("TONEMAP2"), 3 HABLE
 * ("TONEMAP3") (ImposterCombinationGpuPostProcessingFilter.java:11-15); any other value applies the exposure
 * only, like the reference's switch without a default.  Blocking; copies in and out (CL_MEM_COPY_HOST_PTR +
 * blocking read in the reference). */
#define CHUNKY_FILTER_GAMMA 0
#define CHUNKY_FILTER_TONEMAP1 1
#define CHUNKY_FILTER_ACES 2
#define CHUNKY_FILTER_HABLE 3
int chunky_filter_frame(chunky_ctx* ctx, int width, int height, double exposure, const double* input,
                        int32_t* argb_out, int type);
/* Host-side instrument (no device needed): the 256 thresholds the GAMMA / ACES curves are evaluated with — T[k] = the smallest
 * float c >= 0 whose output byte min(255, (uint)(pow(c, 1/2.2) * 255 + 0.5)) is >= k (post_processing_filter.cl:24-27,
 * rgba.h:9-14).  The byte is a monotone step function of c, so comparing c with these is the same function as evaluating
 * pow; tests/test_filter.py checks that over every float. */
int chunky_filter_gamma_thresholds(float* out256);
/* Same kernel on buffers already in device memory (`d_input`: 3*n_pixels doubles, `d_argb`: n_pixels words),
 * enqueued `repeat` times on the context's stream and waited for; *kernel_ms (may be NULL) receives the mean
 * device time of one launch from HIP events on that stream. */
int chunky_filter_frame_device(chunky_ctx* ctx, int64_t n_pixels, float exposure, const void* d_input, void* d_argb,
                               int type, int repeat, float* kernel_ms);

/* ---- host-side verification hook for the octree re-layout done at upload (no device needed):
 * builds the wide tree of chunkyclplugin_amd/csrc/widetree.hpp from `tree` and looks n cells up in
 * it, returning for each the block pointer (K/octree.h:88) and the leaf level.  level_bits == NULL
 * uses the default split.  *n_entries receives the size of the re-laid-out array. */
int chunky_widetree_lookup(const int32_t* tree, int64_t n_ints, int depth, const int32_t* level_bits, int n_levels,
                           const int32_t* xyz /* 3n */, int n, int32_t* data_out, int32_t* level_out, int64_t* n_entries);

/* ---- self test: evaluate the rt_math.h contract on the device (bit-compared with the host by tests) */
int chunky_selftest_math(chunky_ctx* ctx, int which, int n, const float* a, const float* b, float* out);

/* ---- self test: helper-level known answers (no reference counterpart as an entry point).
 * Evaluates, one call per input row, the device functions the kernels are made of — the counterparts of the reference's
 * helpers K/primitives.h:30-162 (AABB_quick_intersect 0, AABB_exit 1, AABB_full_intersect 2, AABB_full_intersect_map_2 3),
 * K/block.h:30-118 (BlockPalette_intersectBlock 4: cube, AABB-model and quad-model blocks with their materials),
 * K/primitives.h:335-409 (Triangle_new + Triangle_intersect 6), K/sky.h:42-106 (Sun_sampleDirection 7, Sun_intersect 8,
 * Sky_intersect 9), K/kernel.h:46-98 (nextPath 10), K/textureAtlas.h:18-28 (Atlas_read_uv 11), K/material.h:31-82
 * (Material_get + Material_sample 12), K/octree.h:41-109 (Octree_octreeIntersect 14; `tree` 0 = the reference layout, 1 = the
 * wide tree in the form the render kernels pick, reported in *tree_used), K/bvh.h:22-113 (Bvh_intersect on the world BVH: 15
 * on the packed arrays, 18 as the pool kernel walks its aligned records) — on the scene's own palettes, atlas, sky and sun.
 * Rows are 32 floats in and 12 out (ints as their bit patterns); the layouts are listed in oracle/ref_shim.cpp ref_helpers,
 * which produced tests/golden/helpers.npz from the reference object itself.  A parity failure then names a function, not a pixel. */
int chunky_selftest_helpers(chunky_scene* scene, int which, int tree, int n, const float* in_rows, float* out_rows, int32_t* tree_used);

/* ---- self test of the tone map's byte estimate (no reference counterpart): `count` consecutive float bit patterns from
 * first_bits through the fast path of the GAMMA (curve 0) or ACES (curve 2) filter (hardware log2 / exp2 / reciprocal
 * estimate, settled against the threshold table — ACES: after the correctly rounded division — only near a step) and
 * through the reference's arithmetic (post_processing_filter.cl:33-38 for ACES) followed by a plain search of
 * chunky_filter_gamma_thresholds' table; *mismatches = values whose bytes differ (must be 0), *worst_estimate = the
 * furthest an estimate strayed beyond its byte's interval (may be NULL). */
int chunky_selftest_gamma_scan(chunky_ctx* ctx, int curve, uint32_t first_bits, uint64_t count, uint64_t* mismatches, float* worst_estimate);

#ifdef __cplusplus
}
#endif
#endif /* CHUNKY_HIP_H */
