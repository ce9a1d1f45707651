// kernels.hpp — host-visible launch interface of the kernel translation units (render_pool.hip, render_fallback.hip,
// aux_kernels.hip, filter.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace chunky {

struct SceneView;
struct CameraView;
struct RenderOpts;

constexpr int kMaxTraces = 10;        // 5 path segments x (main + shadow trace)
constexpr int kBvhStackEntries = 64;  // K/bvh.h:38

// Image-tile ownership of this process: tiles of `tile` consecutive pixel
// indices dealt round-robin over `world` ranks; n_local = pixel slots owned by `rank`.
struct ShardView {
    int rank, world, tile, n_local;
    // block shards (tile 0) under a kernel that cannot map 16 x 16 blocks itself (launch_fallback): the rank's pixels, in
    // block order, as an explicit list on the device (capi.hip block_pixel_list); null otherwise
    const int* list = nullptr;
    int n_list = 0;
};

// Layout shared with include/chunky_hip.h (chunky_hit_record) and oracle/oracle_scene.h.
struct HitRecord {
    int32_t hit;
    int32_t material;
    float distance;
    float normal[3];
    float color[4];
    float emittance;
    float point[3];
};

// Seeds of the passes one launch runs, passed by value in the kernel-argument segment (wave-uniform
// scalar loads, no per-launch host->device copy): pass k uses seed[k] and bufferSpp first_spp + k
// (OpenClPathTracingRenderer.java:106-109).
constexpr int kMaxPassesPerLaunch = 256;  // seeds travel in the kernel-argument segment (4 KB in all); the pass index is 8 bits
// render_pool takes longer launches when the staged samples fit (a share of the image on several GPUs: fewer end-of-launch tails):
// the seeds then travel in device memory (launch_render's seeds_dev)
constexpr int kMaxPoolPasses = 1024;
struct PassSeeds {
    int n, first_spp;
    int seed[kMaxPassesPerLaunch];
};

// What launch_render picked for a launch (chunky_render_kernel_info): the tests assert that the instantiation they
// mean to compare with the oracle is the one that ran.
struct KernelChoice {
    int tree;    // leaf-lookup form: 0 reference layout, -1 generic wide tree, 16 + n dense top over n 3-bit levels
    int group;   // lanes per pixel (render_waves); 0 = render_lanes (one lane per pixel for the whole launch)
    int bvh;     // entity-BVH phases compiled in
    int blocks;  // workgroups launched
    int pool;    // render_pool: paths parked per wave beside the 64 in its lanes; -1 = another kernel
    int ext;     // the extended integrator (CHUNKY_OPT_SUN_SAMPLING / _EMITTERS / _BSDF / _EMITTER_NEE at non-default values)
    int sorted;  // render_pool: full cubes and model blocks tested in phases of their own
};
// render_pool's tiles of 256 pixel slots: the image's 16 x 16-pixel blocks with one rank (edge blocks padded), the rank's own
// blocks or 256-slot runs with several (path_state.hpp pool_slot_gid)
inline long long pool_tile_count(const ShardView& T, int width, int height) {
    if (T.world != 1) return ((long long)T.n_local + 255) / 256;
    return (long long)((width + 15) / 16) * ((height + 15) / 16);
}
// staging floats render_pool needs for a launch of n passes over this rank's tiles
inline size_t staging_floats(const ShardView& T, int width, int height, int n_passes) {
    return 3 * (size_t)pool_tile_count(T, width, height) * 256 * (size_t)n_passes;
}

// compute units of the CURRENT device (cached per device index: members of a group and threads of a host may sit on different GPUs)
inline hipError_t current_device_cus(int* n_cu) {
    static std::atomic<int> cached[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool slot = dev >= 0 && dev < 64;
    int n = slot ? cached[dev].load(std::memory_order_relaxed) : 0;
    if (n == 0) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) return e;
        n = prop.multiProcessorCount;
        if (slot) cached[dev].store(n, std::memory_order_relaxed);
    }
    *n_cu = n;
    return hipSuccess;
}
// true when launch_render will run render_pool for this scene / option set (else launch_fallback: render_waves, render_lanes)
bool pool_kernel_applies(int variant, const SceneView& S, const RenderOpts& O, bool have_queue_and_staging);
// P.n <= kMaxPassesPerLaunch with the seeds in P.seed, or (render_pool only) up to kMaxPoolPasses with the seeds in seeds_dev
hipError_t launch_render(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, const ShardView& T,
                         const PassSeeds& P, float* res, int* work_counter, hipStream_t stream,
                         KernelChoice* chosen = nullptr, float* staging = nullptr, const int* seeds_dev = nullptr);
// the kernels behind render_pool (render_fallback.hip): render_waves, render_lanes
hipError_t launch_fallback(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, const ShardView& T,
                           const PassSeeds& P, float* res, int* work_counter, hipStream_t stream, KernelChoice* chosen);
// read-back exchange of a multi-GPU group: pack = true copies the pixels of this shard's slots from the image `fb` to `packed`
// (3 floats per slot, padding slots skipped), pack = false scatters them back into an image
hipError_t launch_gather(bool pack, const ShardView& T, int width, int height, float* fb, float* packed, hipStream_t stream);
// ... and, for the exchange by ncclReduce, zeroes every pixel of `fb` that shard T does not own
hipError_t launch_clear_foreign(const ShardView& T, int width, int height, float* fb, hipStream_t stream);
hipError_t launch_trace_records(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, int seed,
                                const int* gids_dev, int n, HitRecord* out, int* counts, float* radiance,
                                hipStream_t stream);
hipError_t launch_preview(int variant, const SceneView& S, const CameraView& C, const RenderOpts& O, int* argb,
                          hipStream_t stream);
// thresholds: 256 floats on the device (capi.hip gamma_thresholds) or null = evaluate pow per channel
hipError_t launch_filter(long long n_pixels, float exposure, const double* in, unsigned* out, int type, hipStream_t stream,
                         const float* thresholds = nullptr);
// tone map self test: gamma_bytes3 (curve 0) / aces_bytes3 (curve 2) against the reference's arithmetic and a search of the
// threshold table, for `count` consecutive float bit patterns
hipError_t launch_gamma_scan(unsigned first, unsigned long long count, int curve, const float* thresholds, unsigned long long* mismatches,
                             float* worst, hipStream_t stream);
// helper-level known answers (aux_kernels.hip helpers_selftest_kernel): rows of 32 floats in, 12 out
hipError_t launch_helpers_selftest(const SceneView& S, int which, int tree, int n, const float* in, float* out, int* tree_used, hipStream_t stream);
hipError_t launch_math_selftest(int which, int n, const float* a, const float* b, float* out, hipStream_t stream);

}  // namespace chunky
