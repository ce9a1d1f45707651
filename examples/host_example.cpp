// examples/host_example.cpp — a host for the C ABI with nothing else in it: no Python, no torch, no HIP headers.
//
//   g++ -std=c++17 -Iinclude examples/host_example.cpp -o host_example -Lchunkyclplugin_amd -lchunky_hip
//       -Wl,-rpath,$PWD/chunkyclplugin_amd -Wl,--allow-shlib-undefined        (one command line)
//   ./host_example scene.raw out.f64 <target spp> [merge interval] [device ...]
//
// It does what a Chunky plugin does through JNI: uploads the packed scene arrays the scene loader holds
// (ClSceneLoader.java:34-150), sets the camera (ClCamera.java:42-52), and runs the renderer's pass loop
// (OpenClPathTracingRenderer.java:95-184: seeds of java.util.Random(0), a read-back and a double-precision merge every
// `merge interval` passes) into Chunky's sample buffer, which it writes to out.f64 (3 x W x H doubles).  With more than one
// device after the merge interval the same calls go to a group context (chunky_group_create): the image's 16x16-pixel blocks
// are rendered side by side.  The scene file is the flat dump of chunkyclplugin_amd.scenes.save_raw (records of
// {name[16], dtype, ndim, dims[4], payload}).  tests/test_host_example.py compares out.f64 with the oracle's run of the same loop.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "chunky_hip.h"

struct Array {
    int dtype = 0;  // 0 int32, 1 uint8, 2 float32, 3 float64
    std::vector<int64_t> dims;
    std::vector<unsigned char> bytes;
    int64_t count() const {
        int64_t n = 1;
        for (int64_t d : dims) n *= d;
        return n;
    }
    const int32_t* i32() const { return reinterpret_cast<const int32_t*>(bytes.data()); }
    const float* f32() const { return reinterpret_cast<const float*>(bytes.data()); }
};

static bool read_scene(const char* path, std::map<std::string, Array>* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "CHKSCN01", 8) != 0) {
        fclose(f);
        return false;
    }
    for (;;) {
        char name[16];
        if (fread(name, 1, 16, f) != 16) break;
        int32_t head[2];
        int64_t dims[4];
        if (fread(head, 4, 2, f) != 2 || fread(dims, 8, 4, f) != 4 || head[1] < 0 || head[1] > 4) {
            fclose(f);
            return false;
        }
        Array a;
        a.dtype = head[0];
        a.dims.assign(dims, dims + head[1]);
        static const int width[4] = {4, 1, 4, 8};
        if (a.dtype < 0 || a.dtype > 3) {
            fclose(f);
            return false;
        }
        a.bytes.resize((size_t)a.count() * width[a.dtype]);
        if (!a.bytes.empty() && fread(a.bytes.data(), 1, a.bytes.size(), f) != a.bytes.size()) {
            fclose(f);
            return false;
        }
        (*out)[std::string(name, strnlen(name, 16))] = std::move(a);
    }
    fclose(f);
    return true;
}

#define TRY(call)                                                                        \
    do {                                                                                 \
        if (int rc_ = (call)) {                                                          \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, chunky_last_error());           \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s scene.raw out.f64 <target spp> [merge interval] [device ...]\n", argv[0]);
        return 1;
    }
    std::map<std::string, Array> sc;
    if (!read_scene(argv[1], &sc)) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 1;
    }
    for (const char* need : {"meta", "octree", "block_palette", "material_palette", "aabb_models", "quad_models", "world_bvh", "actor_bvh",
                             "bvh_trigs", "atlas", "sky", "sky_intensity", "sun", "camera"})
        if (!sc.count(need)) {
            fprintf(stderr, "%s: no array '%s'\n", argv[1], need);
            return 1;
        }
    const int target_spp = atoi(argv[3]);
    const int merge_interval = argc > 4 ? atoi(argv[4]) : 1024;
    std::vector<int> devices;
    for (int i = 5; i < argc; i++) devices.push_back(atoi(argv[i]));
    if (devices.empty()) devices.push_back(0);

    chunky_ctx* ctx = nullptr;
    if (devices.size() == 1)
        TRY(chunky_init(devices[0], &ctx));
    else
        TRY(chunky_group_create(devices.data(), (int)devices.size(), &ctx));
    char name[128] = "";
    chunky_device_name(devices[0], name, sizeof name);

    // the scene loader's arrays (meta = {octree depth, projector type, width, height})
    const int32_t* meta = sc["meta"].i32();
    const int depth = meta[0], projector = meta[1], width = meta[2], height = meta[3];
    chunky_scene* scene = nullptr;
    TRY(chunky_scene_create(ctx, &scene));
    TRY(chunky_scene_set_octree(scene, sc["octree"].i32(), sc["octree"].count(), depth));
    const char* palettes[5] = {"block_palette", "material_palette", "aabb_models", "quad_models", "bvh_trigs"};  // CHUNKY_PALETTE_* order
    for (int kind = 0; kind < 5; kind++) TRY(chunky_scene_set_palette(scene, kind, sc[palettes[kind]].i32(), sc[palettes[kind]].count()));
    TRY(chunky_scene_set_bvh(scene, 0, sc["world_bvh"].i32(), sc["world_bvh"].count()));
    TRY(chunky_scene_set_bvh(scene, 1, sc["actor_bvh"].i32(), sc["actor_bvh"].count()));
    const Array& atlas = sc["atlas"];  // [layers][H][W][4]
    TRY(chunky_scene_set_atlas(scene, atlas.bytes.data(), (int)atlas.dims[2], (int)atlas.dims[1], (int)atlas.dims[0]));
    const Array& sky = sc["sky"];  // [H][W][4]
    TRY(chunky_scene_set_sky(scene, sky.bytes.data(), (int)sky.dims[1], (int)sky.dims[0], sc["sky_intensity"].f32()[0]));
    TRY(chunky_scene_set_sun(scene, sc["sun"].i32()));

    chunky_render* r = nullptr;
    TRY(chunky_render_create(ctx, scene, width, height, &r));
    TRY(chunky_render_set_camera(r, projector, sc["camera"].f32(), sc["camera"].count()));

    // the renderer's loop: Chunky's sample buffer and scene.spp
    std::vector<double> samples((size_t)3 * width * height, 0.0);
    int32_t spp = 0;
    const auto t0 = std::chrono::steady_clock::now();
    TRY(chunky_render_run(r, samples.data(), &spp, target_spp, merge_interval, nullptr, nullptr));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    float kernel_ms = 0;
    int launches = 0;
    chunky_render_kernel_time(r, &kernel_ms, &launches);

    FILE* f = fopen(argv[2], "wb");
    if (!f || fwrite(samples.data(), 8, samples.size(), f) != samples.size()) {
        fprintf(stderr, "cannot write %s\n", argv[2]);
        return 1;
    }
    fclose(f);
    // what carried the read-back exchange of a group (RCCL called from inside the library, or its peer-copy fallback, and why)
    int transport = 0;
    char detail[256] = "";
    TRY(chunky_group_transport(ctx, &transport, detail, sizeof detail));
    for (char* p = detail; *p; p++)
        if (*p == '"' || *p == '\\') *p = '\'';
    printf("{\"device\": \"%s\", \"members\": %d, \"size\": [%d, %d], \"spp\": %d, \"seconds\": %.4f, \"Msamples_per_s\": %.2f, \"launches\": %d, \"kernel_ms\": %.3f, "
           "\"transport\": %d, \"transport_detail\": \"%s\"}\n",
           name, chunky_group_size(ctx), width, height, spp, dt, (double)width * height * spp / dt / 1e6, launches, kernel_ms, transport, detail);
    TRY(chunky_render_destroy(r));
    TRY(chunky_scene_destroy(scene));
    TRY(chunky_shutdown(ctx));
    return 0;
}
