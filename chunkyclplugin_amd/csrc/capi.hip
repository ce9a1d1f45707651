// capi.hip — implementation of include/chunky_hip.h: device context, scene upload, render target
// and the host pass loop (the C++ counterpart of the reference's Java host side,
// J/opencl/renderer/{RendererInstance,ClSceneLoader}.java and J/opencl/OpenClPathTracingRenderer.java).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/chunky_hip.h"
#include "kernels.hpp"
#include "rccl_dyn.hpp"
#include "rt_device.hpp"
#include "widetree.hpp"

using namespace chunky;

static_assert(sizeof(chunky_hit_record) == sizeof(HitRecord), "record layouts must agree");
static_assert(CHUNKY_MAX_TRACES == kMaxTraces, "trace capacity must agree");

// placement of the entity-BVH records (relayout_bvh_records): records in the breadth-first top, records per treelet; 0 = off
#ifndef CHUNKY_BVH_TOP_RECORDS
#define CHUNKY_BVH_TOP_RECORDS 0
#endif
#ifndef CHUNKY_BVH_TREELET_RECORDS
#define CHUNKY_BVH_TREELET_RECORDS 0
#endif

// ------------------------------------------------------------------------------------ errors
static thread_local std::string tls_error;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    tls_error = buf;
    return code;
}
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(CHUNKY_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------ context
struct chunky_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::recursive_mutex mu;  // the reference's renderLock
    std::string name;
    void* gamma_table = nullptr;  // 256 floats: the byte thresholds of the GAMMA / ACES tone maps (gamma_thresholds)
    // chunky_group_create: one member context per GPU; this object then only carries the lock and fans calls out
    std::vector<chunky_ctx*> members;
    std::vector<int> peer_status;  // per member: how its read-back copies reach member 0 (chunky_group_peer_status)
    // the read-back exchange of a group (group_gather): one RCCL communicator per member when the collective library could be
    // bound and the members are distinct devices, else empty — `transport` says what the next read-back will use and
    // `transport_detail` why (chunky_group_transport)
    std::vector<ncclComm_t> comms;
    int transport = CHUNKY_TRANSPORT_PEER_COPY;
    std::string transport_detail = "single device: no exchange";
    bool self_exchange = false;  // test rigs (tuning builds, CHUNKY_GROUP_SELF_EXCHANGE=1): member 0's own blocks travel through the exchange too
    int exchange_timeout_ms = 30000;  // how long an RCCL exchange may stay unfinished before its communicators are aborted (group_wait)
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t upload(const void* src, size_t n, hipStream_t s) {
        if (n != bytes || !p) {
            release();
            hipError_t e = hipMalloc(&p, n ? n : 4);
            if (e != hipSuccess) return e;
            bytes = n;
        }
        if (n == 0) return hipMemsetAsync(p, 0, 4, s);
        hipError_t e = hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(s);  // caller may reuse its array on return (COPY_HOST_PTR)
    }
};

struct chunky_scene {
    chunky_ctx* ctx = nullptr;
    DevBuf octree, blocks, materials, aabbs, quads, trigs, world_bvh, actor_bvh, atlas, sky, wide, block_info, quad_aux;
    DevBuf mat8, aabb_rec, quad_rec;                   // 16-byte-aligned re-layouts of the palettes (rt_device.hpp)
    DevBuf bvh_rec;                                    // both entity BVHs as 64-byte inner nodes, then their triangles as 80-byte records
    size_t tri_off = 0;                                // byte offset of the first triangle record in bvh_rec
    DevBuf emitters;                                   // emitter next-event estimation: {x, y, z, level << 25 | block} per emitter leaf
    std::vector<int32_t> host_octree, host_emitters;
    bool emitters_dirty = true;
    std::vector<int32_t> host_trigs, host_world_bvh, host_actor_bvh;
    int world_root = 0, actor_root = 0;                // first reference of each BVH in bvh_rec / tri_rec (rt_device.hpp)
    bool bvh_dirty = false;
    std::vector<int32_t> host_blocks, host_materials, host_aabbs, host_quads;  // kept to rebuild what is derived from them
    bool derived_dirty = false;                        // block_info, quad_aux, mat8, aabb_rec, quad_rec
    int model_leaf_permille = 0;                       // octree leaves that are model blocks, per thousand leaves that can be hit (scene_view)
    WideTree wide_meta;  // host copy kept so the kind bits can follow the block palette; nlev == 0 when absent
    bool wide_dirty = false;
    int octree_depth = -1;
    int atlas_w = 0, atlas_h = 0, atlas_layers = 0;
    int sky_w = 0, sky_h = 0;
    float sky_intensity = 0;
    int sun[6] = {0, 0, 0, 0, 0, 0};
    bool have_sun = false, world_empty = true, actor_empty = true;
    bool have_world = false, have_actor = false;
    int world_height = 0, actor_height = 0;  // inner-node levels: bounds the to-visit stack (K/bvh.h:38 uses 64)
    int refs = 1;  // owner + render targets
    std::vector<chunky_scene*> replicas;  // on a group: the scene's copy on every member (this object holds no data)
};

struct chunky_render {
    chunky_ctx* ctx = nullptr;
    chunky_scene* scene = nullptr;
    int width = 0, height = 0;
    CameraView cam{};
    bool have_camera = false;
    DevBuf rays;
    RenderOpts opts{256, 5, 13.0f, -1, 1, 0, 0};
    int kernel_variant = 0;
    ShardView shard{0, 1, 256, 0};
    DevBuf own_fb, work_counter;
    DevBuf staging;  // render_pool: one launch's samples, [tile of 256 slots][pass][slot][3] floats
    DevBuf block_list;  // block shards under a kernel without the block mapping: this rank's pixels (ShardView::list)
    float* fb = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;  // timing brackets of enqueued launches
    std::vector<hipEvent_t> free_events;
    float timed_ms = 0;
    int timed_launches = 0;
    KernelChoice last_choice{0, 0, 0, 0, -1, 0};  // what the most recent launch ran (chunky_render_kernel_info)
    int launch_cap = 0;  // most passes one launch carries here (staging size); 0 = not determined yet
    int launch_cap_most = 0;  // ... determined for launches of at most this many passes (kMaxPassesPerLaunch / kMaxPoolPasses)
    DevBuf seed_buf;      // seeds of a launch longer than the kernel-argument segment holds (render_pool)
    int reserve_passes = 0;  // the pass loop is about to climb to launches of this many passes: size the staging array once
    struct SeedSlot {        // pinned host copies of long launches' seeds on their way to seed_buf
        int32_t* host = nullptr;
        hipEvent_t copied = nullptr;
    };
    static constexpr int kSeedSlots = 4;
    SeedSlot seed_ring[kSeedSlots];
    unsigned seed_next = 0;
    // on a group: one target per member (this object holds no device data), the caller's share of the image, and the buffers
    // of the read-back exchange: gather_send[i] on member i's device, gather_recv[i] on member 0's
    std::vector<chunky_render*> parts;
    ShardView outer{0, 1, 0, 0};
    std::vector<DevBuf> gather_send, gather_recv;
    ~chunky_render() {
        for (auto& p : pending) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        for (auto e : free_events) (void)hipEventDestroy(e);
        for (SeedSlot& s : seed_ring) {
            if (s.copied) (void)hipEventDestroy(s.copied);
            if (s.host) (void)hipHostFree(s.host);
        }
    }
};

constexpr size_t kStagingBytes = (size_t)8 << 30;  // 8 GiB: 1920x1080 x 256 passes is 6.4 GB (of 288)

// The pixels of a block shard (tile 0) in block order, padding left out: what launch_fallback's kernels render when
// render_pool does not apply to a sharded target (same ownership rule as pool_slot_gid, so the gather finds every pixel).
static std::vector<int32_t> block_pixel_list(int width, int height, const ShardView& t) {
    std::vector<int32_t> out;
    const int bw = (width + 15) / 16, bh = (height + 15) / 16;
    for (int64_t b = t.rank; b < (int64_t)bw * bh; b += t.world) {
        const int bx = (int)(b % bw) * 16, by = (int)(b / bw) * 16;
        for (int y = by; y < by + 16 && y < height; y++)
            for (int x = bx; x < bx + 16 && x < width; x++) out.push_back(y * width + x);
    }
    return out;
}

static int n_local_slots(int width, int height, const ShardView& t) {
    const int n_pixels = width * height;
    if (t.world == 1) return n_pixels;
    if (t.tile == 0) {  // 16 x 16 blocks rank, rank + world, ... of the image, edge blocks padded
        const int n_blocks = ((width + 15) / 16) * ((height + 15) / 16);
        const int mine = (n_blocks - t.rank + t.world - 1) / t.world;
        return mine > 0 ? mine * 256 : 0;
    }
    int n_tiles = (n_pixels + t.tile - 1) / t.tile;
    int mine = (n_tiles - t.rank + t.world - 1) / t.world;  // tiles rank, rank+world, ...
    return mine > 0 ? mine * t.tile : 0;
}

// ------------------------------------------------------------------------------------ device
extern "C" const char* chunky_last_error(void) { return tls_error.c_str(); }
// 0.4: chunky_run_callbacks carries its size (an ABI change), chunky_group_peer_status, CHUNKY_OPT_BVH_CULL_BEHIND
// 0.5: chunky_group_transport / chunky_group_set_transport (the group's read-back exchange through RCCL, bound at run time)
extern "C" const char* chunky_version(void) { return "chunky-hip 0.5 gfx950"; }

extern "C" int chunky_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int chunky_device_name(int device, char* buf, int buf_len) {
    if (!buf || buf_len <= 0) return fail(CHUNKY_E_INVALID, "chunky_device_name: no buffer");
    hipDeviceProp_t prop;
    if (device < 0 || device >= chunky_device_count()) return fail(CHUNKY_E_NO_DEVICE, "no HIP device %d", device);
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    // (some boxes report an empty marketing name)
    snprintf(buf, buf_len, "%s (%s, %d CUs)", prop.name[0] ? prop.name : "AMD GPU", prop.gcnArchName, prop.multiProcessorCount);
    return CHUNKY_OK;
}

extern "C" int chunky_init(int device, chunky_ctx** out) {
    if (!out) return fail(CHUNKY_E_INVALID, "chunky_init: out is NULL");
    *out = nullptr;
    int n = chunky_device_count();
    if (n <= 0) return fail(CHUNKY_E_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(CHUNKY_E_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<chunky_ctx> c(new chunky_ctx);
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    char nm[256];
    if (chunky_device_name(device, nm, sizeof nm) == CHUNKY_OK) c->name = nm;
    *out = c.release();
    return CHUNKY_OK;
}

// "rccl 2.22.3 (librccl.so.1), 8 ranks, grouped send/recv of the owned blocks"
static std::string rccl_detail(const chunky_ctx* g, int transport) {
    const RcclApi& api = rccl_api();
    char buf[256];
    snprintf(buf, sizeof buf, "rccl %d.%d.%d (%s), %zu rank(s), %s", api.version / 10000, (api.version / 100) % 100, api.version % 100,
             api.where.c_str(), g->comms.size(),
             transport == CHUNKY_TRANSPORT_RCCL_REDUCE ? "ncclReduce(sum) of the zero-padded framebuffers onto member 0"
             : transport == CHUNKY_TRANSPORT_RCCL_SENDRECV ? "grouped ncclSend / ncclRecv of the owned blocks to member 0"
                                                            : "communicator open, peer copies selected");
    return buf;
}

// Gives up the communicators (after an RCCL failure, or at shutdown): later read-backs use peer copies.
static void group_close_rccl(chunky_ctx* g, bool abort) {
    const RcclApi& api = rccl_api();
    for (size_t i = 0; i < g->comms.size(); i++) {
        (void)hipSetDevice(g->members[i]->device);
        if (g->comms[i]) (void)(abort ? api.CommAbort(g->comms[i]) : api.CommDestroy(g->comms[i]));
    }
    g->comms.clear();
    (void)hipGetLastError();
}

// Waits until every stream of `streams` (on `devices`) has drained — WITHOUT blocking in the driver: an RCCL kernel whose peer or
// link died never completes, hipStreamSynchronize would then never return, and the one call that unblocks such a kernel,
// ncclCommAbort, could never be reached.  Polls hipStreamQuery and the communicators' asynchronous errors; returns CHUNKY_OK, or
// CHUNKY_E_HIP with the reason (an error RCCL noticed by itself, or the deadline) — the caller then aborts the communicators FIRST
// and only then synchronises.
static int group_wait(chunky_ctx* g, const std::vector<int>& devices, const std::vector<hipStream_t>& streams) {
    const RcclApi& api = rccl_api();
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<char> done(streams.size(), 0);
    if (g->exchange_timeout_ms == 0)  // (rigs: every exchange counts as hung, whether or not its kernels are still running)
        return fail(CHUNKY_E_HIP, "RCCL exchange unfinished after 0 ms (a hung collective: dead peer or link?)");
    for (unsigned spin = 0;; spin++) {
        bool all = true;
        for (size_t i = 0; i < streams.size(); i++) {
            if (done[i]) continue;
            (void)hipSetDevice(devices[i]);
            const hipError_t q = hipStreamQuery(streams[i]);
            if (q == hipSuccess) {
                done[i] = 1;
            } else if (q == hipErrorNotReady) {
                all = false;
                (void)hipGetLastError();
            } else {
                return fail(CHUNKY_E_HIP, "hipStreamQuery on device %d: %s", devices[i], hipGetErrorString(q));
            }
        }
        if ((spin & 15u) == 0u || all)  // a failure the communicator noticed by itself (a dead link, a dead peer)
            for (size_t i = 0; i < g->comms.size(); i++) {
                ncclResult_t async = ncclSuccess;
                if (api.CommGetAsyncError(g->comms[i], &async) == ncclSuccess && async != ncclSuccess)
                    return fail(CHUNKY_E_HIP, "RCCL communicator of member %zu: %s", i, api.str(async));
            }
        if (all) return CHUNKY_OK;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > (double)g->exchange_timeout_ms)
            return fail(CHUNKY_E_HIP, "RCCL exchange unfinished after %d ms (a hung collective: dead peer or link?)", g->exchange_timeout_ms);
        if (spin > 64) std::this_thread::sleep_for(std::chrono::microseconds(spin > 4096 ? 500 : 20));
    }
}

// First contact: ONE grouped send / receive of a known pattern from every member to member 0 through the communicators just
// created, under group_wait's deadline, and the bytes compared on the host.  RCCL stays the group's transport only if this
// machine, this process and this library file demonstrably move the right bytes; anything else — an error code, a hang, a
// wrong byte — is found HERE, at group creation, where the answer is "peer copies, and chunky_group_transport says why",
// not in the middle of a render.
// (`send` / `recv` belong to the caller: if the probe ends in a hung exchange they must outlive the abort — hipFree would wait
// for the stuck kernel)
static int group_probe_rccl(chunky_ctx* g, std::vector<DevBuf>& send, std::vector<DevBuf>& recv) {
    const RcclApi& api = rccl_api();
    const size_t n = g->members.size(), count = 1024;
    send.resize(n);
    recv.resize(n);
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    std::vector<float> host(count);
    for (size_t i = 0; i < n; i++) {
        chunky_ctx* m = g->members[i];
        for (size_t k = 0; k < count; k++) host[k] = (float)(i * 4096 + k + 1);
        HIP_TRY(hipSetDevice(m->device));
        HIP_TRY(send[i].upload(host.data(), count * 4, m->stream));  // (synchronises the member's stream)
        HIP_TRY(hipSetDevice(g->members[0]->device));
        HIP_TRY(hipMalloc(&recv[i].p, count * 4));
        recv[i].bytes = count * 4;
        HIP_TRY(hipMemsetAsync(recv[i].p, 0, count * 4, g->members[0]->stream));
        devices.push_back(m->device);
        streams.push_back(m->stream);
    }
    HIP_TRY(hipSetDevice(g->members[0]->device));
    HIP_TRY(hipStreamSynchronize(g->members[0]->stream));
    ncclResult_t rc = api.GroupStart();
    if (rc != ncclSuccess) return fail(CHUNKY_E_HIP, "probe: ncclGroupStart: %s", api.str(rc));
    ncclResult_t bad = ncclSuccess;
    const char* where = "";
    for (size_t i = 0; i < n && bad == ncclSuccess; i++) {
        (void)hipSetDevice(g->members[i]->device);
        if ((bad = api.Send(send[i].p, count, ncclFloat, 0, g->comms[i], g->members[i]->stream)) != ncclSuccess) where = "ncclSend";
        (void)hipSetDevice(g->members[0]->device);
        if (bad == ncclSuccess && (bad = api.Recv(recv[i].p, count, ncclFloat, (int)i, g->comms[0], g->members[0]->stream)) != ncclSuccess) where = "ncclRecv";
    }
    rc = api.GroupEnd();  // (always: the thread's group must be closed; a partial list is dealt with by the caller's abort)
    if (bad != ncclSuccess) return fail(CHUNKY_E_HIP, "probe: %s: %s", where, api.str(bad));
    if (rc != ncclSuccess) return fail(CHUNKY_E_HIP, "probe: ncclGroupEnd: %s", api.str(rc));
    if (int w = group_wait(g, devices, streams)) return w;
    HIP_TRY(hipSetDevice(g->members[0]->device));
    for (size_t i = 0; i < n; i++) {
        HIP_TRY(hipMemcpy(host.data(), recv[i].p, count * 4, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < count; k++)
            if (host[k] != (float)(i * 4096 + k + 1))
                return fail(CHUNKY_E_HIP, "probe: member %zu's float %zu arrived as %g", i, k, (double)host[k]);
    }
    return CHUNKY_OK;
}

// One RCCL communicator over the members of a group (chunky_group_transport).  Never an error: without it the exchange
// runs on peer copies and transport_detail says why.
static void group_open_rccl(chunky_ctx* g, const int* devices, int n) {
    g->transport = CHUNKY_TRANSPORT_PEER_COPY;
    std::string want;
    bool try_shared = false, probe = true;
#ifdef CHUNKY_TUNING  // rigs of tests/test_gpu_rccl_transport.py and tools/: the shipping library reads none of these
    if (const char* env = getenv("CHUNKY_GROUP_TRANSPORT")) want = env;
    const char* self = getenv("CHUNKY_GROUP_SELF_EXCHANGE");
    g->self_exchange = self && *self && *self != '0';
    try_shared = getenv("CHUNKY_RCCL_TRY_SHARED") != nullptr;
    probe = getenv("CHUNKY_GROUP_NO_PROBE") == nullptr;
    if (const char* t = getenv("CHUNKY_GROUP_TIMEOUT_MS")) {
        const int v = atoi(t);
        if (v >= 0 && v <= 600000) g->exchange_timeout_ms = v;
    }
#endif
    if (want == "peer") {
        g->transport_detail = "peer copies: asked for by the environment";
        return;
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j] && !try_shared) {
                // (ncclCommInitAll refuses a device list with duplicates; CHUNKY_RCCL_TRY_SHARED lets the tests watch it do so)
                char buf[128];
                snprintf(buf, sizeof buf, "peer copies: members %d and %d share device %d (one RCCL rank per device)", j, i, devices[i]);
                g->transport_detail = buf;
                return;
            }
    const RcclApi& api = rccl_api();
    if (!api.usable()) {
        g->transport_detail = "peer copies: " + api.error;
        return;
    }
    g->comms.assign((size_t)n, nullptr);
    const ncclResult_t rc = api.CommInitAll(g->comms.data(), n, devices);
    if (rc != ncclSuccess) {
        g->comms.clear();
        (void)hipGetLastError();
        g->transport_detail = std::string("peer copies: ncclCommInitAll: ") + api.str(rc);
        return;
    }
    std::vector<DevBuf> probe_send, probe_recv;  // freed at the end of this function: after the abort and the drain below
    if (probe && group_probe_rccl(g, probe_send, probe_recv) != CHUNKY_OK) {
        const std::string why = tls_error;
        group_close_rccl(g, true);  // abort first (a hung probe kernel is unblocked by nothing else), then drain
        for (chunky_ctx* m : g->members) {
            (void)hipSetDevice(m->device);
            (void)hipStreamSynchronize(m->stream);
        }
        (void)hipGetLastError();
        g->transport_detail = "peer copies: RCCL failed its first exchange (" + why + ")";
        return;
    }
    g->transport = want == "rccl-reduce" ? CHUNKY_TRANSPORT_RCCL_REDUCE : CHUNKY_TRANSPORT_RCCL_SENDRECV;
    g->transport_detail = rccl_detail(g, g->transport) + (probe ? "; first exchange verified" : "");
}

extern "C" int chunky_group_create(const int* devices, int n, chunky_ctx** out) {
    if (!out) return fail(CHUNKY_E_INVALID, "chunky_group_create: out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return fail(CHUNKY_E_INVALID, "chunky_group_create: 1..64 devices");
    std::unique_ptr<chunky_ctx> g(new chunky_ctx);
    for (int i = 0; i < n; i++) {
        chunky_ctx* m = nullptr;
        if (int rc = chunky_init(devices[i], &m)) {
            const std::string why = tls_error;
            for (chunky_ctx* c : g->members) (void)chunky_shutdown(c);
            return fail(rc, "chunky_group_create: member %d: %s", i, why.c_str());
        }
        g->members.push_back(m);
    }
    g->device = g->members[0]->device;
    g->name = g->members[0]->name;
    // the read-back exchange copies member i's blocks into member 0's memory: direct (xGMI) where peer access exists, staged
    // by the runtime where it does not — failing to enable it is not an error
    g->peer_status.assign((size_t)n, CHUNKY_PEER_LOCAL);
    for (int i = 1; i < n; i++) {
        if (devices[i] == devices[0]) continue;
        int can = 0;
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipDeviceCanAccessPeer(&can, devices[i], devices[0]);
        if (e == hipSuccess && !can) {
            g->peer_status[(size_t)i] = CHUNKY_PEER_STAGED;
        } else if (e == hipSuccess) {
            e = hipDeviceEnablePeerAccess(devices[0], 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
            g->peer_status[(size_t)i] = e == hipSuccess ? CHUNKY_PEER_DIRECT : -(int)e;
        } else {
            g->peer_status[(size_t)i] = -(int)e;
        }
        (void)hipGetLastError();
    }
    group_open_rccl(g.get(), devices, n);
    (void)hipSetDevice(g->device);
    *out = g.release();
    return CHUNKY_OK;
}

extern "C" int chunky_group_transport(chunky_ctx* ctx, int* transport, char* detail, int detail_len) {
    if (!ctx || !transport) return fail(CHUNKY_E_INVALID, "chunky_group_transport: NULL argument");
    std::lock_guard<std::recursive_mutex> g(ctx->mu);
    *transport = ctx->transport;
    if (detail && detail_len > 0) snprintf(detail, (size_t)detail_len, "%s", ctx->transport_detail.c_str());
    return CHUNKY_OK;
}

extern "C" int chunky_group_set_transport(chunky_ctx* ctx, int transport) {
    if (!ctx) return fail(CHUNKY_E_INVALID, "chunky_group_set_transport: NULL context");
    if (transport != CHUNKY_TRANSPORT_PEER_COPY && transport != CHUNKY_TRANSPORT_RCCL_SENDRECV && transport != CHUNKY_TRANSPORT_RCCL_REDUCE)
        return fail(CHUNKY_E_INVALID, "chunky_group_set_transport: unknown transport %d", transport);
    std::lock_guard<std::recursive_mutex> g(ctx->mu);
    if (transport == ctx->transport) return CHUNKY_OK;
    if (transport != CHUNKY_TRANSPORT_PEER_COPY && ctx->comms.empty())
        return fail(CHUNKY_E_STATE, "chunky_group_set_transport: no RCCL communicator (%s)", ctx->transport_detail.c_str());
    ctx->transport = transport;
    if (!ctx->comms.empty()) ctx->transport_detail = rccl_detail(ctx, transport);
    return CHUNKY_OK;
}

extern "C" int chunky_group_size(chunky_ctx* ctx) {
    if (!ctx) return fail(CHUNKY_E_INVALID, "chunky_group_size: NULL context");
    return ctx->members.empty() ? 1 : (int)ctx->members.size();
}

extern "C" int chunky_group_peer_status(chunky_ctx* ctx, int* out, int n) {
    if (!ctx || !out || n < chunky_group_size(ctx)) return fail(CHUNKY_E_INVALID, "chunky_group_peer_status: need room for %d members", ctx ? chunky_group_size(ctx) : 0);
    if (ctx->members.empty()) {
        out[0] = CHUNKY_PEER_LOCAL;
        return CHUNKY_OK;
    }
    for (size_t i = 0; i < ctx->members.size(); i++) out[i] = ctx->peer_status[i];
    return CHUNKY_OK;
}

extern "C" int chunky_group_device(chunky_ctx* ctx, int i) {
    if (!ctx || i < 0 || i >= chunky_group_size(ctx)) return fail(CHUNKY_E_INVALID, "chunky_group_device: no member %d", i);
    return ctx->members.empty() ? ctx->device : ctx->members[(size_t)i]->device;
}

extern "C" int chunky_shutdown(chunky_ctx* ctx) {
    if (!ctx) return fail(CHUNKY_E_INVALID, "chunky_shutdown: NULL context");
    if (!ctx->members.empty()) {
        int rc = CHUNKY_OK;
        if (!ctx->comms.empty()) {  // (the members' streams are idle by the contract of shutdown: no render target is left)
            for (chunky_ctx* m : ctx->members) {
                (void)hipSetDevice(m->device);
                (void)hipStreamSynchronize(m->stream);
            }
            group_close_rccl(ctx, false);
        }
        for (chunky_ctx* m : ctx->members)
            if (int e = chunky_shutdown(m)) rc = e;
        delete ctx;
        return rc;
    }
    {
        std::lock_guard<std::recursive_mutex> g(ctx->mu);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->gamma_table) (void)hipFree(ctx->gamma_table);
        (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
    return CHUNKY_OK;
}

// ------------------------------------------------------------------------------------ scene
#define LOCK_SCENE(s)                                                        \
    if (!(s) || !(s)->ctx) return fail(CHUNKY_E_INVALID, "NULL scene");      \
    std::lock_guard<std::recursive_mutex> guard_((s)->ctx->mu);              \
    HIP_TRY(hipSetDevice((s)->ctx->device))

extern "C" int chunky_scene_create(chunky_ctx* ctx, chunky_scene** out) {
    if (!ctx || !out) return fail(CHUNKY_E_INVALID, "chunky_scene_create: NULL argument");
    std::unique_ptr<chunky_scene> s(new chunky_scene);
    s->ctx = ctx;
    for (chunky_ctx* m : ctx->members) {  // a group: one replica per member
        chunky_scene* rep = nullptr;
        if (int rc = chunky_scene_create(m, &rep)) {
            for (chunky_scene* r : s->replicas) (void)chunky_scene_destroy(r);
            return rc;
        }
        s->replicas.push_back(rep);
    }
    *out = s.release();
    return CHUNKY_OK;
}

// A call on a group's scene is the same call on every replica (under the group's lock: the reference's renderLock).
template <class F>
static int each_replica(chunky_scene* s, F call) {
    std::lock_guard<std::recursive_mutex> g(s->ctx->mu);
    for (chunky_scene* m : s->replicas)
        if (int rc = call(m)) return rc;
    return CHUNKY_OK;
}
#define FAN_SCENE(s, expr) \
    if ((s) && !(s)->replicas.empty()) return each_replica((s), [&](chunky_scene* m_) { return expr; })

static void scene_unref(chunky_scene* s) {
    if (--s->refs == 0) delete s;
}

extern "C" int chunky_scene_destroy(chunky_scene* scene) {
    if (scene && !scene->replicas.empty()) {
        const int rc = each_replica(scene, [&](chunky_scene* m_) { return chunky_scene_destroy(m_); });
        std::lock_guard<std::recursive_mutex> g(scene->ctx->mu);
        scene->replicas.clear();
        scene_unref(scene);  // render targets of the group keep the (now empty) shell alive until they are destroyed
        return rc;
    }
    LOCK_SCENE(scene);
    (void)hipStreamSynchronize(scene->ctx->stream);
    scene_unref(scene);
    return CHUNKY_OK;
}

static int check_ints(const int32_t* p, int64_t n, const char* what) {
    if (n < 0 || (n > 0 && !p)) return fail(CHUNKY_E_INVALID, "%s: bad array (n=%lld)", what, (long long)n);
    return CHUNKY_OK;
}

extern "C" int chunky_scene_set_octree(chunky_scene* scene, const int32_t* tree, int64_t n, int depth) {
    FAN_SCENE(scene, chunky_scene_set_octree(m_, tree, n, depth));
    LOCK_SCENE(scene);
    if (int rc = check_ints(tree, n, "set_octree")) return rc;
    if (n < 1) return fail(CHUNKY_E_INVALID, "set_octree: empty tree");
    if (depth < 0 || depth > 30) return fail(CHUNKY_E_INVALID, "set_octree: depth %d out of range", depth);
    // every branch value must address a whole 8-int child group inside the array (K/octree.h:83-87)
    for (int64_t i = 0; i < n; i++) {
        int32_t v = tree[i];
        if (v > 0 && (int64_t)v + 8 > n) return fail(CHUNKY_E_INVALID, "set_octree: node %lld points outside the tree", (long long)i);
    }
    HIP_TRY(scene->octree.upload(tree, (size_t)n * 4, scene->ctx->stream));
    scene->octree_depth = depth;
    scene->host_octree.assign(tree, tree + n);
    scene->emitters_dirty = true;
    // wide re-layout for the fast lookup; scenes it cannot express keep the reference layout only
    scene->wide.release();
    scene->wide_meta = WideTree();
    int bits[kWideMaxLevels];
    int nlev = default_wide_levels(depth, bits);
#ifdef CHUNKY_TUNING
    if (const char* e = getenv("CHUNKY_DEBUG_WIDE_BITS")) {  // experiments: another split, e.g. "4,3,2" (16^3 top node)
        nlev = 0;
        for (const char* q = e; *q && nlev < kWideMaxLevels;) {
            bits[nlev++] = atoi(q);
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
    }
#endif
    const char* why = "";
    WideTree wt;
    if (build_wide_tree(tree, n, depth, bits, nlev, &wt, &why)) {
        scene->wide_meta = std::move(wt);
        scene->wide_dirty = true;  // annotated + uploaded by scene_view once the block palette is known
    }
    return CHUNKY_OK;
}

extern "C" int chunky_scene_load_octree(chunky_scene* scene, const int32_t* tree_data, int64_t n, int depth,
                                        const int32_t* block_mapping, int64_t n_mapping) {
    if (int rc = check_ints(tree_data, n, "load_octree")) return rc;
    if (int rc = check_ints(block_mapping, n_mapping, "load_octree mapping")) return rc;
    std::vector<int32_t> mapped((size_t)n);
    for (int64_t i = 0; i < n; i++) {  // ClSceneLoader.java:56-58
        int32_t v = tree_data[i];
        mapped[(size_t)i] = (v > 0 || -(int64_t)v >= n_mapping) ? v : -block_mapping[-v];
    }
    return chunky_scene_set_octree(scene, mapped.data(), n, depth);
}

extern "C" int chunky_scene_set_palette(chunky_scene* scene, int kind, const int32_t* data, int64_t n) {
    FAN_SCENE(scene, chunky_scene_set_palette(m_, kind, data, n));
    LOCK_SCENE(scene);
    if (int rc = check_ints(data, n, "set_palette")) return rc;
    DevBuf* dst = nullptr;
    switch (kind) {
        case CHUNKY_PALETTE_BLOCK: dst = &scene->blocks; break;
        case CHUNKY_PALETTE_MATERIAL: dst = &scene->materials; break;
        case CHUNKY_PALETTE_AABB: dst = &scene->aabbs; break;
        case CHUNKY_PALETTE_QUAD: dst = &scene->quads; break;
        case CHUNKY_PALETTE_TRIG: dst = &scene->trigs; break;
        default: return fail(CHUNKY_E_INVALID, "set_palette: unknown kind %d", kind);
    }
    HIP_TRY(dst->upload(data, (size_t)n * 4, scene->ctx->stream));
    switch (kind) {
        case CHUNKY_PALETTE_BLOCK: scene->host_blocks.assign(data, data + n); scene->wide_dirty = true; break;
        case CHUNKY_PALETTE_MATERIAL: scene->host_materials.assign(data, data + n); break;
        case CHUNKY_PALETTE_AABB: scene->host_aabbs.assign(data, data + n); break;
        case CHUNKY_PALETTE_QUAD: scene->host_quads.assign(data, data + n); break;
        case CHUNKY_PALETTE_TRIG: scene->host_trigs.assign(data, data + n); break;
        default: break;
    }
    if (kind != CHUNKY_PALETTE_TRIG) scene->derived_dirty = true;  // rebuilt by scene_view before the next launch
    if (kind == CHUNKY_PALETTE_BLOCK || kind == CHUNKY_PALETTE_MATERIAL) scene->emitters_dirty = true;
    if (kind == CHUNKY_PALETTE_TRIG || kind == CHUNKY_PALETTE_MATERIAL) scene->bvh_dirty = true;
    return CHUNKY_OK;
}

extern "C" int chunky_scene_set_bvh(chunky_scene* scene, int which, const int32_t* nodes, int64_t n) {
    FAN_SCENE(scene, chunky_scene_set_bvh(m_, which, nodes, n));
    LOCK_SCENE(scene);
    if (int rc = check_ints(nodes, n, "set_bvh")) return rc;
    if (which != CHUNKY_BVH_WORLD && which != CHUNKY_BVH_ACTOR) return fail(CHUNKY_E_INVALID, "set_bvh: which=%d", which);
    if (n < 7) return fail(CHUNKY_E_INVALID, "set_bvh: a BVH has at least one 7-int node (got %lld ints)", (long long)n);
    bool empty = nodes[0] == 0;  // K/bvh.h:23-32
    for (int k = 1; k <= 6 && empty; k++) {
        float f;
        memcpy(&f, &nodes[k], 4);
        empty = f != f;
    }
    // height of the tree = most entries the to-visit stack can hold; also rejects child links that
    // leave the array or form a cycle (a malformed BVH would hang the traversal)
    int height = 0;
    if (!empty) {
        std::vector<std::pair<int64_t, int>> todo;
        todo.emplace_back(0, 0);
        int64_t visited = 0;
        while (!todo.empty()) {
            auto [at, d] = todo.back();
            todo.pop_back();
            if (at < 0 || at + 7 > n || ++visited > n) return fail(CHUNKY_E_INVALID, "set_bvh: node link outside the array or cyclic");
            if (d > height) height = d;
            const int32_t head = nodes[at];
            if (head > 0) {
                todo.emplace_back(at + 7, d + 1);
                todo.emplace_back((int64_t)head, d + 1);
            }
        }
        if (height > 63) return fail(CHUNKY_E_INVALID, "set_bvh: tree deeper than the reference's 64-entry stack");
    }
    (which == CHUNKY_BVH_WORLD ? scene->world_height : scene->actor_height) = height;
    DevBuf& dst = which == CHUNKY_BVH_WORLD ? scene->world_bvh : scene->actor_bvh;
    HIP_TRY(dst.upload(nodes, (size_t)n * 4, scene->ctx->stream));
    (which == CHUNKY_BVH_WORLD ? scene->host_world_bvh : scene->host_actor_bvh).assign(nodes, nodes + n);
    scene->bvh_dirty = true;
    if (which == CHUNKY_BVH_WORLD) {
        scene->world_empty = empty;
        scene->have_world = true;
    } else {
        scene->actor_empty = empty;
        scene->have_actor = true;
    }
    return CHUNKY_OK;
}

extern "C" int chunky_scene_set_atlas(chunky_scene* scene, const uint8_t* rgba, int w, int h, int layers) {
    FAN_SCENE(scene, chunky_scene_set_atlas(m_, rgba, w, h, layers));
    LOCK_SCENE(scene);
    if (w <= 0 || h <= 0 || layers <= 0) return fail(CHUNKY_E_INVALID, "set_atlas: bad size %dx%dx%d", w, h, layers);
    size_t bytes = (size_t)w * h * layers * 4;
    if (rgba) {
        HIP_TRY(scene->atlas.upload(rgba, bytes, scene->ctx->stream));
    } else {
        scene->atlas.release();
        HIP_TRY(hipMalloc(&scene->atlas.p, bytes));
        scene->atlas.bytes = bytes;
        HIP_TRY(hipMemsetAsync(scene->atlas.p, 0, bytes, scene->ctx->stream));
        HIP_TRY(hipStreamSynchronize(scene->ctx->stream));
    }
    scene->atlas_w = w;
    scene->atlas_h = h;
    scene->atlas_layers = layers;
    return CHUNKY_OK;
}

extern "C" int chunky_scene_write_atlas_tile(chunky_scene* scene, int x, int y, int layer, int w, int h,
                                             const uint8_t* rgba) {
    FAN_SCENE(scene, chunky_scene_write_atlas_tile(m_, x, y, layer, w, h, rgba));
    LOCK_SCENE(scene);
    if (!scene->atlas.p) return fail(CHUNKY_E_STATE, "write_atlas_tile before set_atlas");
    if (!rgba || x < 0 || y < 0 || layer < 0 || w <= 0 || h <= 0 || x + w > scene->atlas_w || y + h > scene->atlas_h ||
        layer >= scene->atlas_layers)
        return fail(CHUNKY_E_INVALID, "write_atlas_tile: region outside the atlas");
    char* base = (char*)scene->atlas.p + (((size_t)layer * scene->atlas_h + y) * scene->atlas_w + x) * 4;
    HIP_TRY(hipMemcpy2DAsync(base, (size_t)scene->atlas_w * 4, rgba, (size_t)w * 4, (size_t)w * 4, h,
                             hipMemcpyHostToDevice, scene->ctx->stream));
    HIP_TRY(hipStreamSynchronize(scene->ctx->stream));
    return CHUNKY_OK;
}

extern "C" int chunky_scene_set_sky(chunky_scene* scene, const uint8_t* rgba, int w, int h, float intensity) {
    FAN_SCENE(scene, chunky_scene_set_sky(m_, rgba, w, h, intensity));
    LOCK_SCENE(scene);
    if (!rgba || w <= 0 || h <= 0) return fail(CHUNKY_E_INVALID, "set_sky: bad texture");
    // texels are converted once here with the same rt_unorm8 the kernels would apply per sample
    std::vector<float> texels((size_t)w * h * 4);
    for (size_t i = 0; i < texels.size(); i++) texels[i] = rt_unorm8(rgba[i]);
    HIP_TRY(scene->sky.upload(texels.data(), texels.size() * 4, scene->ctx->stream));
    scene->sky_w = w;
    scene->sky_h = h;
    scene->sky_intensity = intensity;
    return CHUNKY_OK;
}

static void list_emitters(const chunky_scene* s, std::vector<int32_t>* out);
// The emitter list exists only for CHUNKY_OPT_EMITTER_NEE and chunky_scene_emitters: built on first use after a change.
static int refresh_emitters(chunky_scene* s) {
    if (!s->emitters_dirty) return CHUNKY_OK;
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));  // queued passes may still read the old list
    list_emitters(s, &s->host_emitters);
    s->emitters.release();
    if (!s->host_emitters.empty()) HIP_TRY(s->emitters.upload(s->host_emitters.data(), s->host_emitters.size() * 4, s->ctx->stream));
    s->emitters_dirty = false;
    return CHUNKY_OK;
}
extern "C" int chunky_scene_emitters(chunky_scene* scene, int32_t* out4, int32_t cap, int32_t* count) {
    if (scene && !scene->replicas.empty()) return chunky_scene_emitters(scene->replicas[0], out4, cap, count);  // replicas agree
    LOCK_SCENE(scene);
    if (!count || cap < 0 || (cap > 0 && !out4)) return fail(CHUNKY_E_INVALID, "scene_emitters: bad arguments");
    if (int rc = refresh_emitters(scene)) return rc;
    const int32_t n = (int32_t)(scene->host_emitters.size() / 4);
    *count = n;
    const int32_t give = n < cap ? n : cap;
    if (out4 && give > 0) memcpy(out4, scene->host_emitters.data(), (size_t)give * 16);
    return CHUNKY_OK;
}

extern "C" int chunky_scene_set_sun(chunky_scene* scene, const int32_t sun[6]) {
    FAN_SCENE(scene, chunky_scene_set_sun(m_, sun));
    LOCK_SCENE(scene);
    if (!sun) return fail(CHUNKY_E_INVALID, "set_sun: NULL");
    memcpy(scene->sun, sun, sizeof scene->sun);
    scene->have_sun = true;
    return CHUNKY_OK;
}

static float bits_to_float(int32_t i) {
    float f;
    memcpy(&f, &i, 4);
    return f;
}

// quad_aux (rt_device.hpp): for every quad of every quad model the block palette points at, the
// ray-independent values of K/primitives.h:262-276 — normalize(cross(xv, yv)), dot(n, origin), dot(xv, xv),
// dot(yv, yv) — written at the quad's own int offset.  Same rt_math.h expressions as the kernel, so
// the stored floats are the ones the kernel would compute.  Returns false (no table) when two
// models overlap in a way that would make entries collide, or a pointer leaves the array.
static bool build_quad_aux(const std::vector<int32_t>& B, const std::vector<int32_t>& Q, std::vector<float>* out) {
    out->assign(Q.size(), 0.0f);
    std::vector<int64_t> owner(Q.size(), -1);
    bool any = false;
    for (size_t k = 0; k + 1 < B.size(); k += 2) {
        if (B[k] != 3) continue;
        const int64_t ptr = B[k + 1];
        if (ptr < 0 || (size_t)ptr >= Q.size()) return false;
        const int64_t count = Q[(size_t)ptr];
        if (count < 0 || (size_t)(ptr + 1 + 15 * count) > Q.size()) return false;
        for (int64_t i = 0; i < count; i++) {
            const int64_t q = ptr + 1 + 15 * i;
            for (int w = 0; w < 6; w++) {
                if (owner[(size_t)(q + w)] >= 0 && owner[(size_t)(q + w)] != q) return false;
                owner[(size_t)(q + w)] = q;
            }
            float f[9];
            memcpy(f, &Q[(size_t)q], sizeof f);
            const float cx = rt_cross_c(f[4], f[8], f[5], f[7]), cy = rt_cross_c(f[5], f[6], f[3], f[8]),
                        cz = rt_cross_c(f[3], f[7], f[4], f[6]);
            const float rl = rt_rlen3(cx, cy, cz);
            const float nx = cx * rl, ny = cy * rl, nz = cz * rl;
            float* a = out->data() + q;
            a[0] = nx;
            a[1] = ny;
            a[2] = nz;
            a[3] = rt_dot3(nx, ny, nz, f[0], f[1], f[2]);
            a[4] = rt_dot3(f[3], f[4], f[5], f[3], f[4], f[5]);
            a[5] = rt_dot3(f[6], f[7], f[8], f[6], f[7], f[8]);
            any = true;
        }
    }
    return any;
}

// How common model blocks are in a world: octree leaves whose block is an AABB or quad model (types 2, 3), per thousand leaves that
// can be hit at all.  render_pool tests full cubes and model blocks in phases of their own where that pays: the model tests cost
// three times the cube test and a wave runs them whenever ONE lane of a block test has a model block, but a class more costs every
// iteration of every wave 1 % in bookkeeping.  Measured: the benchmark city (110 per thousand; 58 % of the block tests on its saved
// view) +2.3 ... +2.8 %, the synthetic outdoor world (10) +0.1 ... +0.5 %, the same world 16 times larger -1.5 %, the indoor room
// (0.3) -1 %: sorted from 30 per thousand on.  (CHUNKY_OPT_KERNEL bits 8 / 9 force it on / off.)
constexpr int kSortBlocksPermille = 30;
static int model_leaf_permille(const std::vector<int32_t>& T, const std::vector<int32_t>& B) {
    int64_t cubes = 0, models = 0;
    for (const int32_t v : T) {
        if (v > 0) continue;  // a branch
        const int64_t ptr = -(int64_t)v;
        if (ptr == 0 || ptr + 1 >= (int64_t)B.size()) continue;  // air, ANY_TYPE, a pointer beyond the palette
        const int32_t type = B[(size_t)ptr];
        cubes += type == 1;
        models += type == 2 || type == 3;
    }
    return cubes + models > 0 ? (int)(models * 1000 / (cubes + models)) : 0;
}

// Everything the kernels read that is derived from the four palettes (rt_device.hpp has the layouts):
//   block_info  per block {type, pointer, 5 material words of a full cube, model record}
//   mat8        materials at a 32-byte stride (two 16-byte reads instead of five unaligned dwords)
//   aabb_rec    AABB-model boxes as three 16-byte words each, materials as mat8 indices
//   quad_rec    quad-model quads as six 16-byte words each (the material's five words inline), with the ray-independent values of K/primitives.h:262-276
//               (unit normal, its dot with the origin, |xv|^2, |yv|^2) evaluated here with the kernel's own rt_math.h
// A block whose model cannot be re-laid out (pointer outside its palette, more than 255 primitives, a material pointer
// that is not a whole material) keeps model record 0 and takes the path that reads the packed palettes as they are.
// The host half of rebuild_derived: everything it derives from the four palettes, as plain vectors (no device call: this is the part
// that reads caller-supplied ints, and tests/sanitize/capi_host_fuzz.cpp runs it under AddressSanitizer on hostile palettes).
struct DerivedRecords {
    std::vector<int32_t> info, mat8, aabb_rec, quad_rec;
};
static void derive_records(const std::vector<int32_t>& B, const std::vector<int32_t>& M, const std::vector<int32_t>& A, const std::vector<int32_t>& Q,
                           DerivedRecords* out) {
    const size_t n_blocks = B.size() / 2, n_mats = M.size() / 6;
    std::vector<int32_t>&mat8 = out->mat8, &info = out->info, &aabb_rec = out->aabb_rec, &quad_rec = out->quad_rec;
    mat8.assign(n_mats * 8, 0);
    for (size_t m = 0; m < n_mats; m++)
        for (int w = 0; w < 6; w++) mat8[m * 8 + w] = M[m * 6 + w];  // word 5 (spec | metal | rough) rides in the second word
    auto mat_index = [&](int32_t ptr, int32_t* out) {  // packed material pointer -> index of its first 16-byte word in mat8
        if (ptr < 0 || ptr % 6 != 0 || (size_t)ptr / 6 >= n_mats) return false;
        *out = (ptr / 6) * 2;
        return true;
    };
    info.assign(n_blocks * 8, 0);
    aabb_rec.clear();
    quad_rec.clear();
    std::vector<int64_t> aabb_at(A.size(), -1), quad_at(Q.size(), -1);  // model pointer -> first record (models are shared between blocks)
    for (size_t k = 0; k < n_blocks; k++) {
        int32_t* e = &info[k * 8];
        const int32_t type = B[2 * k], ptr = B[2 * k + 1];
        e[0] = type;
        e[1] = ptr;
        if (type == 1) {
            if (ptr >= 0 && (size_t)ptr + 5 <= M.size()) {
                for (int w = 0; w < 5; w++) e[2 + w] = M[(size_t)ptr + w];
                if ((size_t)ptr + 6 <= M.size()) e[7] = M[(size_t)ptr + 5];  // material word 5 (extensions)
            } else {
                e[0] = 0x7FFFFFFF;  // malformed cube: an unknown model type never hits (K/block.h:44-47)
            }
        } else if (type == 2 || type == 3) {
            // A model whose pointer, primitive count or material pointers leave their palettes would make the kernels read outside
            // device memory (the reference has no such check: its behaviour there is undefined).  Such a block never hits, like
            // an unknown model type (K/block.h:44-47); every well-formed block is untouched by this.
            const std::vector<int32_t>& P = type == 2 ? A : Q;
            const int64_t stride = type == 2 ? 13 : 15;
            bool sound = ptr >= 0 && (size_t)ptr < P.size();
            if (sound) {
                const int64_t count = P[(size_t)ptr];
                sound = count >= 0 && (size_t)(ptr + 1 + stride * count) <= P.size();
                for (int64_t i = 0; sound && i < count; i++) {
                    const int32_t* prim = &P[(size_t)(ptr + 1 + stride * i)];
                    if (type == 2) {
                        for (int w = 1; w < 6 && sound; w++) sound = prim[7 + w] >= 0 && (size_t)prim[7 + w] + 6 <= M.size();  // E, S, W, T, B: the ones that are read
                    } else {
                        sound = prim[13] >= 0 && (size_t)prim[13] + 6 <= M.size();
                    }
                }
            }
            if (!sound) e[0] = 0x7FFFFFFF;
        }
        if (e[0] == 2) {
            const int64_t count = A[(size_t)ptr];
            if (count < 1 || count > 255) continue;
            if (aabb_at[(size_t)ptr] < 0) {
                const int64_t first = (int64_t)aabb_rec.size() / 12;
                bool ok = true;
                std::vector<int32_t> rec((size_t)count * 12);
                for (int64_t i = 0; i < count && ok; i++) {
                    const int32_t* b = &A[(size_t)(ptr + 1 + 13 * i)];
                    int32_t* r = &rec[(size_t)i * 12];
                    for (int w = 0; w < 7; w++) r[w] = b[w];  // six bounds, flags
                    for (int w = 0; w < 5 && ok; w++) ok = mat_index(b[8 + w], &r[7 + w]);  // E, S, W, T, B (N is never read: K/primitives.h:209-234)
                }
                if (!ok) {
                    aabb_at[(size_t)ptr] = -2;
                } else {
                    aabb_at[(size_t)ptr] = first;
                    aabb_rec.insert(aabb_rec.end(), rec.begin(), rec.end());
                }
            }
            if (aabb_at[(size_t)ptr] >= 0 && aabb_at[(size_t)ptr] < (1 << 22)) e[7] = (int32_t)((aabb_at[(size_t)ptr] << 8) | count);
        } else if (e[0] == 3) {
            const int64_t count = Q[(size_t)ptr];
            if (count < 1 || count > 255) continue;
            if (quad_at[(size_t)ptr] < 0) {
                const int64_t first = (int64_t)quad_rec.size() / 24;
                bool ok = true;
                std::vector<int32_t> rec((size_t)count * 24);
                for (int64_t i = 0; i < count && ok; i++) {
                    const int32_t* q = &Q[(size_t)(ptr + 1 + 15 * i)];
                    float f[9];
                    memcpy(f, q, sizeof f);
                    const float cx = rt_cross_c(f[4], f[8], f[5], f[7]), cy = rt_cross_c(f[5], f[6], f[3], f[8]),
                                cz = rt_cross_c(f[3], f[7], f[4], f[6]);
                    const float rl = rt_rlen3(cx, cy, cz);
                    const float nx = cx * rl, ny = cy * rl, nz = cz * rl;
                    const float aux[6] = {nx, ny, nz, rt_dot3(nx, ny, nz, f[0], f[1], f[2]), rt_dot3(f[3], f[4], f[5], f[3], f[4], f[5]),
                                          rt_dot3(f[6], f[7], f[8], f[6], f[7], f[8])};
                    int32_t a[6];
                    memcpy(a, aux, sizeof a);
                    int32_t* r = &rec[(size_t)i * 24];
                    r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = a[3];      // origin, dot(n, origin)
                    r[4] = q[3]; r[5] = q[4]; r[6] = q[5]; r[7] = a[4];      // xv, |xv|^2
                    r[8] = q[6]; r[9] = q[7]; r[10] = q[8]; r[11] = a[5];    // yv, |yv|^2
                    r[12] = q[9]; r[13] = q[10]; r[14] = q[11]; r[15] = q[12];  // uv
                    r[16] = a[0]; r[17] = a[1]; r[18] = a[2];                // unit normal
                    int32_t m8 = 0;
                    ok = mat_index(q[13], &m8);                              // the quad's material, inline: one dependent read less
                    if (ok && (M[(size_t)q[13]] & 2)) ok = false;          // an emittance texture needs the full word: packed path
                    if (ok) {
                        const int32_t* m = &M[(size_t)q[13]];
                        r[19] = (m[4] & 0xFF) | (int32_t)((uint32_t)m[5] << 8);
                        r[20] = m[0]; r[21] = m[1]; r[22] = m[2]; r[23] = m[3];
                    }
                }
                if (!ok) {
                    quad_at[(size_t)ptr] = -2;
                } else {
                    quad_at[(size_t)ptr] = first;
                    quad_rec.insert(quad_rec.end(), rec.begin(), rec.end());
                }
            }
            if (quad_at[(size_t)ptr] >= 0 && quad_at[(size_t)ptr] < (1 << 22)) e[7] = (int32_t)((quad_at[(size_t)ptr] << 8) | count);
        }
    }
}

static int rebuild_derived(chunky_scene* s) {
    const std::vector<int32_t>&B = s->host_blocks, &M = s->host_materials, &A = s->host_aabbs, &Q = s->host_quads;
    hipStream_t st = s->ctx->stream;
    HIP_TRY(hipStreamSynchronize(st));  // queued passes may still read the old copies
    s->block_info.release();
    s->quad_aux.release();
    s->mat8.release();
    s->aabb_rec.release();
    s->quad_rec.release();
    s->derived_dirty = false;
    // (also for an empty block or material palette: block_info then exists with every block marked as one that never hits — a
    // cube's material would lie outside the palette — and render_pool's sorted block tests rely on its existence)
    DerivedRecords d;
    derive_records(B, M, A, Q, &d);
    HIP_TRY(s->block_info.upload(d.info.data(), d.info.size() * 4, st));
    HIP_TRY(s->mat8.upload(d.mat8.data(), d.mat8.size() * 4, st));
    if (!d.aabb_rec.empty()) HIP_TRY(s->aabb_rec.upload(d.aabb_rec.data(), d.aabb_rec.size() * 4, st));
    if (!d.quad_rec.empty()) HIP_TRY(s->quad_rec.upload(d.quad_rec.data(), d.quad_rec.size() * 4, st));
    std::vector<float> aux;  // for quads that kept the packed path
    if (build_quad_aux(B, Q, &aux)) HIP_TRY(s->quad_aux.upload(aux.data(), aux.size() * 4, st));
    return CHUNKY_OK;
}

// The two entity BVHs re-laid out for aligned 16-byte reads (rt_device.hpp has the layouts): every inner node becomes a
// 64-byte record holding BOTH children (their references and boxes — what one visit of K/bvh.h:72-85 reads), every
// triangle an 80-byte record with its material as a mat8 index.  A reference is the index of an inner record, or
// -1 - (first triangle record << 6 | count) for a leaf.  The walk order, the tests and the arithmetic stay the reference's.
// Returns false (no records: the packed arrays are walked as they are) when something does not fit: a leaf of more than
// 63 triangles, a triangle pointer outside the palette, a material pointer that is not a whole material.
// Placement of the inner records of one BVH (records [lo, hi) of bvh_rec, root reference *root): the first `top` records
// breadth-first from the root (the levels every walk crosses, contiguous), then every subtree below that cut as depth-first
// TREELETS of at most `treelet` records, each treelet breadth-first from its own root — a walker that enters a treelet finds
// its next few visits in the same 1–2 KB.  References are renumbered; nothing else changes.
static void relayout_bvh_records(std::vector<int32_t>* bvh_rec, size_t lo, size_t hi, int* root, int top, int treelet) {
    if (hi <= lo || *root < 0) return;
    const size_t n = hi - lo;
    std::vector<int32_t> order;  // order[k] = old index of the record that moves to lo + k
    order.reserve(n);
    auto rec = [&](int32_t idx) { return &(*bvh_rec)[(size_t)idx * 16]; };
    std::vector<int32_t> frontier{*root};
    {   // the top, breadth-first
        size_t head = 0;
        while (head < frontier.size() && order.size() < (size_t)top) {
            const int32_t at = frontier[head++];
            order.push_back(at);
            for (int c = 0; c < 2; c++)
                if (rec(at)[c] >= 0) frontier.push_back(rec(at)[c]);
        }
        frontier.erase(frontier.begin(), frontier.begin() + (ptrdiff_t)head);
    }
    // below the cut: treelets, depth-first (a stack of treelet roots; the first child's treelet follows its parent's)
    std::vector<int32_t> roots(frontier.rbegin(), frontier.rend()), members, next;
    while (!roots.empty()) {
        members.assign(1, roots.back());
        roots.pop_back();
        next.clear();
        for (size_t head = 0; head < members.size(); head++) {
            const int32_t at = members[head];
            order.push_back(at);
            for (int c = 0; c < 2; c++) {
                const int32_t r = rec(at)[c];
                if (r < 0) continue;
                if (members.size() < (size_t)treelet) members.push_back(r); else next.push_back(r);
            }
        }
        roots.insert(roots.end(), next.rbegin(), next.rend());
    }
    if (order.size() != n) return;  // (cannot happen: every record of the range hangs under the root exactly once)
    std::vector<int32_t> where(n), moved(n * 16);
    for (size_t k = 0; k < n; k++) where[(size_t)order[k] - lo] = (int32_t)(lo + k);
    for (size_t k = 0; k < n; k++) {
        const int32_t* from = rec(order[k]);
        int32_t* to = &moved[k * 16];
        std::copy(from, from + 16, to);
        for (int c = 0; c < 2; c++)
            if (to[c] >= 0) to[c] = where[(size_t)to[c] - lo];
    }
    std::copy(moved.begin(), moved.end(), bvh_rec->begin() + (ptrdiff_t)(lo * 16));
    *root = where[(size_t)*root - lo];
}

// Triangle records in the order in which the inner records (as placed) refer to their leaves, so that the leaves under one
// treelet lie together.  A leaf shared by several references moves once.
static void reorder_triangles(std::vector<int32_t>* bvh_rec, std::vector<int32_t>* tri_rec, int* world_root, int* actor_root) {
    const size_t n_tri = tri_rec->size() / 20;
    std::vector<int32_t> moved;
    moved.reserve(tri_rec->size());
    std::vector<int32_t> first_at(n_tri + 1, -1);  // old first triangle of a leaf -> new
    auto move_leaf = [&](int32_t* ref) {
        if (*ref >= 0) return;
        const int32_t l = -1 - *ref, first = l >> 6, count = l & 63;
        if (count == 0 || (size_t)first + (size_t)count > n_tri) return;
        if (first_at[(size_t)first] < 0) {
            first_at[(size_t)first] = (int32_t)(moved.size() / 20);
            moved.insert(moved.end(), tri_rec->begin() + (ptrdiff_t)first * 20, tri_rec->begin() + (ptrdiff_t)(first + count) * 20);
        }
        *ref = -1 - ((first_at[(size_t)first] << 6) | count);
    };
    move_leaf(world_root);
    move_leaf(actor_root);
    for (size_t k = 0; k < bvh_rec->size() / 16; k++) {
        move_leaf(&(*bvh_rec)[k * 16]);
        move_leaf(&(*bvh_rec)[k * 16 + 1]);
    }
    tri_rec->swap(moved);  // every reference now points into `moved`
}

// top / treelet sizes of relayout_bvh_records; CHUNKY_BVH_LAYOUT="top,treelet" overrides them for tuning runs ("0,0" = the
// plain depth-first order of round 2)
static void bvh_layout_params(int* top, int* treelet) {
    *top = CHUNKY_BVH_TOP_RECORDS;
    *treelet = CHUNKY_BVH_TREELET_RECORDS;
#ifdef CHUNKY_TUNING
    if (const char* e = getenv("CHUNKY_BVH_LAYOUT")) {
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 0 && b >= 0) {
            *top = a;
            *treelet = b;
        }
    }
#endif
}

static bool build_bvh_records(const chunky_scene* s, std::vector<int32_t>* bvh_rec, std::vector<int32_t>* tri_rec, int* world_root,
                              int* actor_root) {
    const std::vector<int32_t>&T = s->host_trigs, &M = s->host_materials;
    const size_t n_mats = M.size() / 6;
    std::vector<int64_t> leaf_at(T.size(), -1);  // triangle pointer -> its leaf reference (leaves may be shared)
    auto leaf_ref = [&](int64_t prim, int32_t* ref) {
        if (prim < 0 || (size_t)prim >= T.size()) return false;
        if (leaf_at[(size_t)prim] >= 0) {
            *ref = (int32_t)(-1 - leaf_at[(size_t)prim]);
            return true;
        }
        const int64_t count = T[(size_t)prim];
        if (count < 0 || count > 63 || (size_t)(prim + 1 + 20 * count) > T.size()) return false;
        const int64_t first = (int64_t)tri_rec->size() / 20;
        if (first >= (1 << 24)) return false;
        for (int64_t i = 0; i < count; i++) {
            const int32_t* t = &T[(size_t)(prim + 1 + 20 * i)];
            const int32_t mp = t[19];
            if (mp < 0 || mp % 6 != 0 || (size_t)mp / 6 >= n_mats) return false;
            const int32_t r[20] = {t[1], t[2], t[3], t[0],             // e1, flags
                                   t[4], t[5], t[6], (mp / 6) * 2,     // e2, material (mat8 index)
                                   t[7], t[8], t[9], t[13],            // o, t1.u
                                   t[10], t[11], t[12], t[14],         // n, t1.v
                                   t[15], t[16], t[17], t[18]};        // t2.u, t2.v, t3.u, t3.v
            tri_rec->insert(tri_rec->end(), r, r + 20);
        }
        leaf_at[(size_t)prim] = (first << 6) | count;
        *ref = (int32_t)(-1 - leaf_at[(size_t)prim]);
        return true;
    };
    auto build = [&](const std::vector<int32_t>& N, bool empty, int* root) {
        *root = 0;
        if (empty || N.size() < 7) return true;
        // reference of the node at int offset `at`: inner nodes get records in visiting (depth-first) order
        struct Job { int64_t at; int32_t* slot; };
        std::vector<std::pair<int64_t, int64_t>> todo;  // (node offset, index of the int in bvh_rec that receives its reference)
        int32_t root_ref = 0;
        // iterative: slot index -1 means the root reference
        todo.emplace_back(0, -1);
        int64_t guard = 0;
        while (!todo.empty()) {
            auto [at, slot] = todo.back();
            todo.pop_back();
            if (at < 0 || (size_t)at + 7 > N.size() || ++guard > (int64_t)N.size()) return false;
            const int32_t head = N[(size_t)at];
            int32_t ref;
            if (head <= 0) {
                if (!leaf_ref(-(int64_t)head, &ref)) return false;
            } else {
                const int64_t a = at + 7, b = head;
                if ((size_t)a + 7 > N.size() || b < 0 || (size_t)b + 7 > N.size()) return false;
                const int64_t idx = (int64_t)bvh_rec->size() / 16;
                if (idx >= (1 << 24)) return false;  // (with at most 2^24 triangles: every record within 32-bit byte offsets)
                ref = (int32_t)idx;
                bvh_rec->resize(bvh_rec->size() + 16, 0);
                int32_t* r = &(*bvh_rec)[(size_t)idx * 16];
                for (int w = 0; w < 6; w++) {
                    r[4 + w] = N[(size_t)a + 1 + w];    // first child's box  (words 1, 2.xy)
                    r[10 + w] = N[(size_t)b + 1 + w];   // second child's box (words 2.zw, 3)
                }
                todo.emplace_back(b, idx * 16 + 1);
                todo.emplace_back(a, idx * 16 + 0);
            }
            if (slot < 0) root_ref = ref; else (*bvh_rec)[(size_t)slot] = ref;
        }
        *root = root_ref;
        return true;
    };
    bvh_rec->clear();
    tri_rec->clear();
    if (!build(s->host_world_bvh, s->world_empty, world_root)) return false;
    const size_t world_records = bvh_rec->size() / 16;
    if (!build(s->host_actor_bvh, s->actor_empty, actor_root)) return false;
    // where the records sit (addresses only: the walk's order, tests and arithmetic do not see it)
    int top = 0, treelet = 0;
    bvh_layout_params(&top, &treelet);
    if (treelet > 1) {
        relayout_bvh_records(bvh_rec, 0, world_records, world_root, top, treelet);
        relayout_bvh_records(bvh_rec, world_records, bvh_rec->size() / 16, actor_root, top, treelet);
        reorder_triangles(bvh_rec, tri_rec, world_root, actor_root);
    }
    return true;
}

// Every leaf of an entity BVH has to lie inside the triangle palette, every triangle's material inside the material palette: the
// kernels follow these ints as they are (the reference does too — with hostile data its reads are undefined; here the render call
// is refused instead).  Node links were checked by chunky_scene_set_bvh.
static bool bvh_leaves_sound(const std::vector<int32_t>& N, bool empty, const std::vector<int32_t>& T, const std::vector<int32_t>& M) {
    if (empty || N.size() < 7) return true;
    std::vector<int64_t> todo{0};  // the nodes the walk can reach (first child at +7, second at node[0]: K/bvh.h:72-85)
    size_t visited = 0;
    while (!todo.empty()) {
        const int64_t at = todo.back();
        todo.pop_back();
        if (at < 0 || (size_t)at + 7 > N.size() || ++visited > N.size()) return false;
        const int32_t head = N[(size_t)at];
        if (head > 0) {
            todo.push_back(at + 7);
            todo.push_back((int64_t)head);
            continue;
        }
        const int64_t prim = -(int64_t)head;
        if ((size_t)prim >= T.size()) return false;
        const int64_t count = T[(size_t)prim];
        if (count < 0 || (size_t)(prim + 1 + 20 * count) > T.size()) return false;
        for (int64_t i = 0; i < count; i++) {
            const int32_t mp = T[(size_t)(prim + 20 + 20 * i)];  // word 19 of the triangle
            if (mp < 0 || (size_t)mp + 6 > M.size()) return false;
        }
    }
    return true;
}

// The emitter list of the next-event-estimation extension (same rule and order as oracle/port.c
// port_list_emitters): every octree leaf whose block is a full cube with a non-zero emittance byte and no emittance
// texture, in pre-order (children in index order), as {x, y, z, level << 25 | block pointer}.
static void list_emitters(const chunky_scene* s, std::vector<int32_t>* out) {
    out->clear();
    const std::vector<int32_t>&T = s->host_octree, &B = s->host_blocks, &M = s->host_materials;
    if (T.empty() || s->octree_depth < 0 || s->octree_depth > 15) return;
    struct Item { int64_t node; int x, y, z, level; };
    std::vector<Item> todo{{0, 0, 0, 0, s->octree_depth}};
    // chunky_scene_set_octree only checks that branch values stay inside the array: a tree with a cycle, or with branches
    // below level 0, must not make this walk run or allocate without end — the kernels' walk is bounded by the depth, so
    // here a node at level 0 is a leaf whatever it holds, and no more nodes are visited than the array has
    size_t visited = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (++visited > T.size()) break;
        const int32_t v = T[(size_t)it.node];
        if (v > 0 && it.level > 0) {
            const int h = 1 << (it.level - 1);
            for (int c = 7; c >= 0; c--)  // pushed in reverse: popped in index order
                todo.push_back({(int64_t)v + c, it.x + ((c >> 2) & 1) * h, it.y + ((c >> 1) & 1) * h, it.z + (c & 1) * h, it.level - 1});
            continue;
        }
        if (v > 0) continue;  // a branch below level 0: not a leaf the kernels can reach
        const int64_t block = -(int64_t)v;
        if (block == 0 || block == 0x7FFFFFFE || block + 1 >= (int64_t)B.size() || block >= (1 << 25)) continue;
        if (B[(size_t)block] != 1) continue;
        const int64_t mp = B[(size_t)block + 1];
        if (mp < 0 || (size_t)mp + 6 > M.size()) continue;
        if ((M[(size_t)mp] & 2) || (M[(size_t)mp + 4] & 0xFF) == 0) continue;
        const int32_t rec[4] = {it.x, it.y, it.z, (int32_t)((it.level << 25) | (int32_t)block)};
        out->insert(out->end(), rec, rec + 4);
    }
}

// Assemble the kernel-side view; Sun_new (K/sky.h:19-40) is evaluated here, on the host, with the
// same rt_math.h the device uses.
static int scene_view(chunky_scene* s, SceneView* v, bool want_emitters = false) {
    if (!s->octree.p || s->octree_depth < 0) return fail(CHUNKY_E_STATE, "scene has no octree");
    if (!s->blocks.p || !s->materials.p) return fail(CHUNKY_E_STATE, "scene has no block/material palette");
    if (!s->atlas.p) return fail(CHUNKY_E_STATE, "scene has no texture atlas");
    if (!s->sky.p) return fail(CHUNKY_E_STATE, "scene has no sky texture");
    if (!s->have_sun) return fail(CHUNKY_E_STATE, "scene has no sun");
    if ((!s->world_empty || !s->actor_empty) && !s->trigs.p) return fail(CHUNKY_E_STATE, "scene has a BVH but no triangles");
    v->octree = (const int*)s->octree.p;
    v->blocks = (const int*)s->blocks.p;
    v->quads = (const int*)s->quads.p;
    v->aabbs = (const int*)s->aabbs.p;
    v->world_bvh = (const int*)s->world_bvh.p;
    v->actor_bvh = (const int*)s->actor_bvh.p;
    v->trigs = (const int*)s->trigs.p;
    v->atlas = (const uint32_t*)s->atlas.p;
    v->materials = (const int*)s->materials.p;
    v->sky = (const float4*)s->sky.p;
    v->octree_depth = s->octree_depth;
    v->atlas_w = s->atlas_w;
    v->atlas_h = s->atlas_h;
    v->atlas_layers = s->atlas_layers;
    v->sky_w = s->sky_w;
    v->sky_h = s->sky_h;
    v->sky_intensity = s->sky_intensity;
    v->sun_flags = s->sun[0];
    v->sun_tex_size = s->sun[1];
    v->sun_tex = s->sun[2];
    v->sun_intensity = bits_to_float(s->sun[3]);
    float phi = bits_to_float(s->sun[4]), theta = bits_to_float(s->sun[5]);
    float r = rt_fabs(rt_cos(phi));
    float swx = rt_cos(theta) * r, swy = rt_sin(phi), swz = rt_sin(theta) * r;
    float sux = 1, suy = 0, suz = 0;
    if (rt_fabs(swx) > 0.1f) {
        sux = 0;
        suy = 1;
    }
    // sv = normalize(cross(sw, su)); su = cross(sv, sw)
    float cx = rt_cross_c(swy, suz, swz, suy), cy = rt_cross_c(swz, sux, swx, suz), cz = rt_cross_c(swx, suy, swy, sux);
    float rl = rt_rlen3(cx, cy, cz);
    float svx = cx * rl, svy = cy * rl, svz = cz * rl;
    v->sw = f3{swx, swy, swz};
    v->sv = f3{svx, svy, svz};
    v->su = f3{rt_cross_c(svy, swz, svz, swy), rt_cross_c(svz, swx, svx, swz), rt_cross_c(svx, swy, svy, swx)};
    v->sun_radius_cos = rt_cos(0.03f);
    v->bvh_cull = 0;  // a render target's option: set by its launch sites
    v->world_bvh_empty = (s->world_empty || !s->world_bvh.p) ? 1 : 0;
    v->actor_bvh_empty = (s->actor_empty || !s->actor_bvh.p) ? 1 : 0;
    if (s->wide_dirty && s->wide_meta.nlev > 0) {
        annotate_wide_tree(&s->wide_meta, s->host_blocks.data(), (int64_t)s->host_blocks.size());
        s->model_leaf_permille = model_leaf_permille(s->host_octree, s->host_blocks);
        HIP_TRY(hipStreamSynchronize(s->ctx->stream));  // queued passes may still read the old copy
        HIP_TRY(s->wide.upload(s->wide_meta.data.data(), s->wide_meta.data.size() * 4, s->ctx->stream));
        s->wide_dirty = false;
    }
    if (s->derived_dirty)
        if (int rc = rebuild_derived(s)) return rc;
    if (s->bvh_dirty) {
        HIP_TRY(hipStreamSynchronize(s->ctx->stream));
        s->bvh_rec.release();
        s->tri_off = 0;
        if (!bvh_leaves_sound(s->host_world_bvh, s->world_empty, s->host_trigs, s->host_materials) ||
            !bvh_leaves_sound(s->host_actor_bvh, s->actor_empty, s->host_trigs, s->host_materials))
            return fail(CHUNKY_E_INVALID, "an entity BVH leaf or a triangle's material lies outside its palette");
        std::vector<int32_t> nodes, tris;
        if ((!s->world_empty || !s->actor_empty) && build_bvh_records(s, &nodes, &tris, &s->world_root, &s->actor_root)) {
            if (nodes.empty()) nodes.resize(16, 0);  // both roots are leaves
            tris.resize(tris.size() + 20, 0);        // a step at the end of the last leaf reads one record past it
            // ONE allocation — nodes, then triangles — so the walk addresses either kind of record as a 32-bit byte offset off
            // one scalar base (a walk longer than 2 GiB of records keeps the packed arrays: build_bvh_records' index limits)
            s->tri_off = nodes.size() * 4;
            nodes.insert(nodes.end(), tris.begin(), tris.end());
            HIP_TRY(s->bvh_rec.upload(nodes.data(), nodes.size() * 4, s->ctx->stream));
        }
        s->bvh_dirty = false;
    }
    if (want_emitters)
        if (int rc = refresh_emitters(s)) return rc;
    v->emitters = want_emitters ? (const int4*)s->emitters.p : nullptr;
    v->n_emitters = want_emitters ? (int)(s->host_emitters.size() / 4) : 0;
    v->n_block_ints = (int)(s->host_blocks.size() < 0x7FFFFFFFu ? s->host_blocks.size() : 0x7FFFFFFFu);
    v->sort_blocks = s->model_leaf_permille >= kSortBlocksPermille ? 1 : 0;
    v->bvh_rec = (const int4*)s->bvh_rec.p;
    v->tri_rec = s->bvh_rec.p ? (const int4*)((const char*)s->bvh_rec.p + s->tri_off) : nullptr;
    v->tri_off = (unsigned)s->tri_off;
    v->world_root = s->world_root;
    v->actor_root = s->actor_root;
    v->quad_aux = (const float*)s->quad_aux.p;
    v->bvh_stack_entries = (s->world_height > s->actor_height ? s->world_height : s->actor_height) + 1;
    v->block_info = (const int4*)s->block_info.p;
    v->mat8 = (const int4*)s->mat8.p;
    v->aabb_rec = (const int4*)s->aabb_rec.p;
    v->quad_rec = (const int4*)s->quad_rec.p;
    v->wide = s->wide_meta.nlev > 0 ? (const uint32_t*)s->wide.p : nullptr;
    v->wide_nlev = s->wide_meta.nlev;
    for (int i = 0; i < 6; i++) {
        v->wide_shift[i] = s->wide_meta.shift[i];
        v->wide_bits[i] = s->wide_meta.bits[i];
    }
    return CHUNKY_OK;
}

// ------------------------------------------------------------------------------------ render
#define LOCK_RENDER(r)                                                       \
    if (!(r) || !(r)->ctx) return fail(CHUNKY_E_INVALID, "NULL render");     \
    std::lock_guard<std::recursive_mutex> guard_((r)->ctx->mu);              \
    HIP_TRY(hipSetDevice((r)->ctx->device))

// A call on a group's render target is the same call on every member's part.
template <class F>
static int each_part(chunky_render* r, F call) {
    std::lock_guard<std::recursive_mutex> g(r->ctx->mu);
    for (chunky_render* m : r->parts)
        if (int rc = call(m)) return rc;
    return CHUNKY_OK;
}
#define FAN_RENDER(r, expr) \
    if ((r) && !(r)->parts.empty()) return each_part((r), [&](chunky_render* m_) { return expr; })

// member i of n renders rank + world * i of world * n of the image (rank / world: the caller's own share, chunky_render_set_shard:
// its tiles are t = rank (mod world); dealing them round-robin to n members gives member i the tiles t = rank + world * i (mod world * n))
static int group_apply_shards(chunky_render* r) {
    const int n = (int)r->parts.size();
    for (int i = 0; i < n; i++)
        if (int rc = chunky_render_set_shard(r->parts[(size_t)i], r->outer.rank + r->outer.world * i, r->outer.world * n, r->outer.tile)) return rc;
    return CHUNKY_OK;
}

extern "C" int chunky_render_create(chunky_ctx* ctx, chunky_scene* scene, int width, int height, chunky_render** out) {
    if (!ctx || !scene || !out) return fail(CHUNKY_E_INVALID, "chunky_render_create: NULL argument");
    if (scene->ctx != ctx) return fail(CHUNKY_E_INVALID, "scene belongs to another context");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (1 << 30))
        return fail(CHUNKY_E_INVALID, "bad image size %dx%d", width, height);
    if (!ctx->members.empty()) {
        std::lock_guard<std::recursive_mutex> g(ctx->mu);
        if (scene->replicas.size() != ctx->members.size()) return fail(CHUNKY_E_STATE, "chunky_render_create: the scene has been destroyed");
        std::unique_ptr<chunky_render> r(new chunky_render);
        r->ctx = ctx;
        r->scene = scene;
        r->width = width;
        r->height = height;
        int rc = CHUNKY_OK;
        for (size_t i = 0; i < ctx->members.size() && rc == CHUNKY_OK; i++) {
            chunky_render* part = nullptr;
            rc = chunky_render_create(ctx->members[i], scene->replicas[i], width, height, &part);
            if (rc == CHUNKY_OK) r->parts.push_back(part);
        }
        if (rc == CHUNKY_OK) rc = group_apply_shards(r.get());
        if (rc != CHUNKY_OK) {
            for (chunky_render* part : r->parts) (void)chunky_render_destroy(part);
            return rc;
        }
        r->gather_send.resize(ctx->members.size());
        r->gather_recv.resize(ctx->members.size());
        scene->refs++;
        *out = r.release();
        return CHUNKY_OK;
    }
    std::lock_guard<std::recursive_mutex> g(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    std::unique_ptr<chunky_render> r(new chunky_render);
    r->ctx = ctx;
    r->scene = scene;
    r->width = width;
    r->height = height;
    size_t bytes = (size_t)width * height * 3 * sizeof(float);
    HIP_TRY(hipMalloc(&r->own_fb.p, bytes));
    r->own_fb.bytes = bytes;
    r->fb = (float*)r->own_fb.p;
    HIP_TRY(hipMemsetAsync(r->fb, 0, bytes, ctx->stream));
    // [0] the sample / pixel queue, [2..49] the phase profile, [64..127] render_pool's range counters (render_pool.hip xcd_claim)
    HIP_TRY(hipMalloc(&r->work_counter.p, 512));
    r->work_counter.bytes = 512;
    HIP_TRY(hipMemsetAsync(r->work_counter.p, 0, 512, ctx->stream));
    r->shard = ShardView{0, 1, 256, width * height};
    scene->refs++;
    *out = r.release();
    return CHUNKY_OK;
}

extern "C" int chunky_render_destroy(chunky_render* r) {
    if (r && !r->parts.empty()) {
        int rc = each_part(r, [&](chunky_render* m_) { return chunky_render_destroy(m_); });
        {
            std::lock_guard<std::recursive_mutex> g(r->ctx->mu);
            for (size_t i = 0; i < r->gather_send.size(); i++) {  // each buffer is freed on the device it lives on
                (void)hipSetDevice(r->ctx->members[i]->device);
                r->gather_send[i].release();
                (void)hipSetDevice(r->ctx->members[0]->device);
                r->gather_recv[i].release();
            }
            scene_unref(r->scene);
        }
        delete r;
        return rc;
    }
    LOCK_RENDER(r);
    (void)hipStreamSynchronize(r->ctx->stream);
    scene_unref(r->scene);
    delete r;
    return CHUNKY_OK;
}

extern "C" int chunky_render_set_camera(chunky_render* r, int projector_type, const float* settings, int64_t n) {
    FAN_RENDER(r, chunky_render_set_camera(m_, projector_type, settings, n));
    LOCK_RENDER(r);
    if (!settings) return fail(CHUNKY_E_INVALID, "set_camera: NULL settings");
    CameraView& c = r->cam;
    c.width = r->width;
    c.height = r->height;
    c.half_width = (float)(r->width / (2.0 * r->height));  // K/rayTracer.cl:66
    c.inv_height = (float)(1.0 / r->height);               // K/rayTracer.cl:67
    if (projector_type == 0) {
        if (n != 15) return fail(CHUNKY_E_INVALID, "set_camera: pinhole needs 15 floats, got %lld", (long long)n);
        memcpy(c.pos, settings, 12);
        memcpy(c.m, settings + 3, 36);
        c.aperture = settings[12];
        c.subject_distance = settings[13];
        c.fov_tan = settings[14];
        c.rays = nullptr;
    } else if (projector_type == -1) {
        int64_t need = (int64_t)r->width * r->height * 6;
        if (n != need) return fail(CHUNKY_E_INVALID, "set_camera: pre-generated rays need %lld floats, got %lld", (long long)need, (long long)n);
        HIP_TRY(hipStreamSynchronize(r->ctx->stream));  // rays may still be read by queued passes
        HIP_TRY(r->rays.upload(settings, (size_t)n * 4, r->ctx->stream));
        c.rays = (const float*)r->rays.p;
    } else {
        return fail(CHUNKY_E_INVALID, "set_camera: projector type %d is not supported (0 or -1)", projector_type);
    }
    c.projector_type = projector_type;
    r->have_camera = true;
    return CHUNKY_OK;
}

extern "C" int chunky_render_set_option(chunky_render* r, int option, int32_t value) {
    FAN_RENDER(r, chunky_render_set_option(m_, option, value));
    LOCK_RENDER(r);
    switch (option) {
        case CHUNKY_OPT_DRAW_DEPTH:
            if (value < 0) return fail(CHUNKY_E_INVALID, "draw depth must be >= 0");
            r->opts.draw_depth = value;
            break;
        case CHUNKY_OPT_MAX_DEPTH:
            if (value < 1 || value > 255) return fail(CHUNKY_E_INVALID, "max depth must be in 1..255");
            r->opts.max_depth = value;
            break;
        case CHUNKY_OPT_EMITTER_SCALE: r->opts.emitter_scale = bits_to_float(value); break;
        case CHUNKY_OPT_KERNEL: r->kernel_variant = value; break;
        case CHUNKY_OPT_SUN_SAMPLING:
            if (value < -1 || value > 1) return fail(CHUNKY_E_INVALID, "sun sampling: -1 (as the reference), 0 or 1");
            r->opts.sun_sampling = value;
            break;
        case CHUNKY_OPT_EMITTERS:
            if (value != 0 && value != 1) return fail(CHUNKY_E_INVALID, "emitters: 0 or 1");
            r->opts.emitters = value;
            break;
        case CHUNKY_OPT_BSDF:
            if (value != 0 && value != 1) return fail(CHUNKY_E_INVALID, "bsdf: 0 or 1");
            r->opts.bsdf = value;
            break;
        case CHUNKY_OPT_EMITTER_NEE:
            if (value != 0 && value != 1) return fail(CHUNKY_E_INVALID, "emitter NEE: 0 or 1");
            r->opts.nee = value;
            break;
        case CHUNKY_OPT_BVH_CULL_BEHIND:
            if (value != 0 && value != 1) return fail(CHUNKY_E_INVALID, "BVH cull: 0 or 1");
            r->opts.bvh_cull = value;
            break;
        default: return fail(CHUNKY_E_INVALID, "unknown option %d", option);
    }
    return CHUNKY_OK;
}

extern "C" int chunky_render_set_shard(chunky_render* r, int rank, int world, int tile) {
    if (r && !r->parts.empty()) {
        if (world < 1 || rank < 0 || rank >= world || tile < 0) return fail(CHUNKY_E_INVALID, "set_shard: rank %d / world %d / tile %d", rank, world, tile);
        std::lock_guard<std::recursive_mutex> g(r->ctx->mu);
        r->outer = ShardView{rank, world, tile, 0};
        return group_apply_shards(r);
    }
    LOCK_RENDER(r);
    if (world < 1 || rank < 0 || rank >= world || tile < 0) return fail(CHUNKY_E_INVALID, "set_shard: rank %d / world %d / tile %d", rank, world, tile);
    ShardView t{rank, world, tile, 0};
    t.n_local = n_local_slots(r->width, r->height, t);
    if (r->shard.list) HIP_TRY(hipStreamSynchronize(r->ctx->stream));  // queued launches may still read the old list
    r->block_list.release();
    r->shard = t;
    r->launch_cap = 0;  // the share changed: so does what a launch can stage
    return CHUNKY_OK;
}

extern "C" int chunky_render_set_device_buffer(chunky_render* r, void* device_ptr) {
    if (r && !r->parts.empty()) return chunky_render_set_device_buffer(r->parts[0], device_ptr);  // the image lives on member 0
    LOCK_RENDER(r);
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    r->fb = device_ptr ? (float*)device_ptr : (float*)r->own_fb.p;
    return CHUNKY_OK;
}

extern "C" int chunky_render_device_buffer(chunky_render* r, void** device_ptr) {
    if (r && !r->parts.empty()) return chunky_render_device_buffer(r->parts[0], device_ptr);
    LOCK_RENDER(r);
    if (!device_ptr) return fail(CHUNKY_E_INVALID, "NULL out pointer");
    *device_ptr = r->fb;
    return CHUNKY_OK;
}

extern "C" int chunky_render_reset(chunky_render* r) {
    FAN_RENDER(r, chunky_render_reset(m_));
    LOCK_RENDER(r);
    HIP_TRY(hipMemsetAsync(r->fb, 0, (size_t)r->width * r->height * 3 * sizeof(float), r->ctx->stream));
    return CHUNKY_OK;
}

static hipError_t get_event(chunky_render* r, hipEvent_t* e) {
    if (!r->free_events.empty()) {
        *e = r->free_events.back();
        r->free_events.pop_back();
        return hipSuccess;
    }
    return hipEventCreate(e);
}

static int collect_timing(chunky_render* r) {
    for (auto& p : r->pending) {
        float ms = 0;
        HIP_TRY(hipEventSynchronize(p.second));
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        r->timed_ms += ms;
        r->timed_launches += 1;
        r->free_events.push_back(p.first);
        r->free_events.push_back(p.second);
    }
    r->pending.clear();
    return CHUNKY_OK;
}

// The most passes one launch of this target carries: render_pool stages every sample of a launch (12 bytes each) — at most
// kStagingBytes of it, fewer than 2^31 samples, at most `most` passes (kMaxPassesPerLaunch: the seeds fit the kernel-argument
// segment; kMaxPoolPasses for render_pool, which reads longer launches' seeds from device memory — a share of the image on several
// GPUs then pays the end-of-launch tail once per 1024 passes instead of four times).  Sized by the tiles THIS rank renders.
static int launch_pass_cap(const chunky_render* r, size_t budget, int most = kMaxPassesPerLaunch) {
    const int64_t n_slots = (int64_t)(staging_floats(r->shard, r->width, r->height, 1) / 3);  // padded tiles
    if (n_slots <= 0) return most;
    int64_t cap = (int64_t)(budget / 12) / n_slots;
    const int64_t cap31 = ((int64_t)1 << 31) / n_slots - 1;
    if (cap > cap31) cap = cap31;
    if (cap > most) cap = most;
    return cap < 1 ? 1 : (int)cap;
}

extern "C" int chunky_render_passes(chunky_render* r, const int32_t* seeds, int n, int first_buffer_spp) {
    FAN_RENDER(r, chunky_render_passes(m_, seeds, n, first_buffer_spp));  // asynchronous on every member: the shares run side by side
    LOCK_RENDER(r);
    if (n < 0 || (n > 0 && !seeds) || first_buffer_spp < 0) return fail(CHUNKY_E_INVALID, "render_passes: bad arguments");
    if (!r->have_camera) return fail(CHUNKY_E_STATE, "render_passes before set_camera");
    SceneView S;
    if (int rc = scene_view(r->scene, &S, r->opts.nee != 0)) return rc;
    S.bvh_cull = r->opts.bvh_cull;
    if (opts_extended(r->opts)) {  // the extensions exist in render_pool only
        const bool bvh = !S.world_bvh_empty || !S.actor_bvh_empty;
        if ((r->kernel_variant & (1 | 2 | 4 | 8)) || (bvh && !(S.bvh_rec && S.tri_rec && S.mat8)))
            return fail(CHUNKY_E_STATE, "the extended light-transport options need the default kernel (CHUNKY_OPT_KERNEL 0)");
    }
    if (r->pending.size() > 4096)
        if (int rc = collect_timing(r)) return rc;
    if (r->shard.n_local <= 0) return CHUNKY_OK;  // this rank (or group member) owns no tile of so small an image: nothing to render
    if (r->shard.world != 1 && r->shard.tile == 0 && !r->shard.list &&
        !pool_kernel_applies(r->kernel_variant, S, r->opts, r->work_counter.p != nullptr)) {
        // a share of 16 x 16 blocks (every group member has one) and a scene / option set render_pool does not take: the
        // fallback kernels render the same pixels from a list
        const std::vector<int32_t> px = block_pixel_list(r->width, r->height, r->shard);
        if (px.empty()) return CHUNKY_OK;
        HIP_TRY(r->block_list.upload(px.data(), px.size() * 4, r->ctx->stream));
        r->shard.list = (const int*)r->block_list.p;
        r->shard.n_list = (int)px.size();
    }
    // render_pool takes up to kMaxPoolPasses per launch, the other kernels what the kernel-argument segment holds
    const int most = pool_kernel_applies(r->kernel_variant, S, r->opts, r->work_counter.p != nullptr) ? kMaxPoolPasses : kMaxPassesPerLaunch;
    if (r->launch_cap <= 0 || r->launch_cap_most != most) {
        r->launch_cap = launch_pass_cap(r, kStagingBytes, most);
        r->launch_cap_most = most;
    }
    for (int done = 0; done < n;) {
        PassSeeds ps;
        ps.n = (n - done) < r->launch_cap ? (n - done) : r->launch_cap;
        size_t need = staging_floats(r->shard, r->width, r->height, ps.n) * sizeof(float);
        if (r->staging.bytes < need) {  // grows to the largest launch seen; launches on the stream are ordered, so it is reused
            HIP_TRY(hipStreamSynchronize(r->ctx->stream));
            r->staging.release();
            // chunky_render_run_ex climbs 1, 8, 64 ... passes per launch: one allocation for where it is going, not four
            int ahead = r->reserve_passes < r->launch_cap ? r->reserve_passes : r->launch_cap;
            if (ahead > kMaxPassesPerLaunch) ahead = kMaxPassesPerLaunch;  // (the pass loop's own launches stop there)
            if (ahead > ps.n) {
                size_t want = staging_floats(r->shard, r->width, r->height, ahead) * sizeof(float);
                size_t free_b = 0, total_b = 0;  // never more than half of what the device has left: other targets and members live there too
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b / 2) want = 0;
                (void)hipGetLastError();
                if (want >= need && hipMalloc(&r->staging.p, want) == hipSuccess) {
                    r->staging.bytes = want;
                } else {
                    (void)hipGetLastError();
                    r->staging.p = nullptr;
                }
            }
        }
        if (r->staging.bytes < need) {
            // memory is short: shorter launches instead of a failed render (each halving halves the array)
            while (hipMalloc(&r->staging.p, need) != hipSuccess) {
                (void)hipGetLastError();
                r->staging.p = nullptr;
                if (ps.n == 1) return fail(CHUNKY_E_HIP, "render_passes: cannot allocate %zu bytes for one pass of staged samples", need);
                ps.n = (ps.n + 1) / 2;
                r->launch_cap = ps.n;
                need = staging_floats(r->shard, r->width, r->height, ps.n) * sizeof(float);
            }
            r->staging.bytes = need;
        }
        ps.first_spp = first_buffer_spp + done;
        const int* seeds_dev = nullptr;
        if (ps.n <= kMaxPassesPerLaunch) {
            memcpy(ps.seed, seeds + done, (size_t)ps.n * 4);
        } else {  // a long launch: its seeds go to device memory, in stream order behind the launch that read the buffer last
            if (!r->seed_buf.p) {
                HIP_TRY(hipMalloc(&r->seed_buf.p, (size_t)kMaxPoolPasses * 4));
                r->seed_buf.bytes = (size_t)kMaxPoolPasses * 4;
            }
            // from a pinned slot of the target's own (the caller may reuse or free `seeds` as soon as this call returns — the JNI
            // glue releases the Java array — and a copy out of pageable memory is only safe if the runtime happens to stage it)
            chunky_render::SeedSlot& slot = r->seed_ring[r->seed_next++ % chunky_render::kSeedSlots];
            if (!slot.host) {  // the event first: a slot is only ever seen with both or with neither
                if (!slot.copied) HIP_TRY(hipEventCreateWithFlags(&slot.copied, hipEventDisableTiming));
                HIP_TRY(hipHostMalloc((void**)&slot.host, (size_t)kMaxPoolPasses * 4, hipHostMallocDefault));
            } else {
                HIP_TRY(hipEventSynchronize(slot.copied));  // the copy that read this slot last (kSeedSlots launches ago)
            }
            memcpy(slot.host, seeds + done, (size_t)ps.n * 4);
            HIP_TRY(hipMemcpyAsync(r->seed_buf.p, slot.host, (size_t)ps.n * 4, hipMemcpyHostToDevice, r->ctx->stream));
            HIP_TRY(hipEventRecord(slot.copied, r->ctx->stream));
            seeds_dev = (const int*)r->seed_buf.p;
        }
        hipEvent_t e0, e1;
        HIP_TRY(get_event(r, &e0));
        HIP_TRY(get_event(r, &e1));
        HIP_TRY(hipEventRecord(e0, r->ctx->stream));
        HIP_TRY(launch_render(r->kernel_variant, S, r->cam, r->opts, r->shard, ps, r->fb, (int*)r->work_counter.p, r->ctx->stream,
                              &r->last_choice, (float*)r->staging.p, seeds_dev));
        HIP_TRY(hipEventRecord(e1, r->ctx->stream));
        r->pending.emplace_back(e0, e1);
        done += ps.n;
    }
    return CHUNKY_OK;
}

extern "C" int chunky_render_sync(chunky_render* r) {
    FAN_RENDER(r, chunky_render_sync(m_));
    LOCK_RENDER(r);
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    return CHUNKY_OK;
}

// The one exchange per read-back of a group: every member but the first packs the pixels of the
// blocks it owns (3 floats each, in the order of its pixel slots), they travel into member 0's memory, and member 0
// scatters them into the image.  Blocks are disjoint, so this is the "reduce of per-tile radiance" with 1/n of the bytes
// per member and no arithmetic: the image is bit for bit what one GPU renders.  What carries them is the group's
// transport (chunky_group_transport): ONE grouped RCCL send / receive, or peer copies; CHUNKY_TRANSPORT_RCCL_REDUCE is
// the literal form instead — one ncclReduce(sum) of the zero-padded framebuffers.
static int gather_buffers(chunky_render* r, size_t i, size_t bytes) {
    chunky_render* pi = r->parts[i];
    if (r->gather_recv[i].bytes < bytes) {
        HIP_TRY(hipSetDevice(r->parts[0]->ctx->device));
        r->gather_recv[i].release();
        HIP_TRY(hipMalloc(&r->gather_recv[i].p, bytes));
        r->gather_recv[i].bytes = bytes;
    }
    HIP_TRY(hipSetDevice(pi->ctx->device));
    if (r->gather_send[i].bytes < bytes) {
        r->gather_send[i].release();
        HIP_TRY(hipMalloc(&r->gather_send[i].p, bytes));
        r->gather_send[i].bytes = bytes;
    }
    return CHUNKY_OK;
}
// member 0 scatters what arrived and the host waits for it
static int gather_scatter(chunky_render* r, size_t first) {
    chunky_render* p0 = r->parts[0];
    std::lock_guard<std::recursive_mutex> g0(p0->ctx->mu);
    HIP_TRY(hipSetDevice(p0->ctx->device));
    for (size_t i = first; i < r->parts.size(); i++)
        if (r->parts[i]->shard.n_local > 0)
            HIP_TRY(launch_gather(false, r->parts[i]->shard, p0->width, p0->height, p0->fb, (float*)r->gather_recv[i].p, p0->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p0->ctx->stream));
    return CHUNKY_OK;
}

static int group_gather_peer(chunky_render* r) {
    const int dev0 = r->parts[0]->ctx->device;
    const size_t n = r->parts.size();
    for (size_t i = 1; i < n; i++) {
        chunky_render* pi = r->parts[i];
        std::lock_guard<std::recursive_mutex> gi(pi->ctx->mu);
        const size_t bytes = (size_t)pi->shard.n_local * 3 * sizeof(float);
        if (bytes == 0) continue;
        if (int rc = gather_buffers(r, i, bytes)) return rc;
        // on member i's stream, behind its queued passes: pack, then the copy across
        HIP_TRY(launch_gather(true, pi->shard, pi->width, pi->height, pi->fb, (float*)r->gather_send[i].p, pi->ctx->stream));
        if (pi->ctx->device == dev0)
            HIP_TRY(hipMemcpyAsync(r->gather_recv[i].p, r->gather_send[i].p, bytes, hipMemcpyDeviceToDevice, pi->ctx->stream));
        else
            HIP_TRY(hipMemcpyPeerAsync(r->gather_recv[i].p, dev0, r->gather_send[i].p, pi->ctx->device, bytes, pi->ctx->stream));
    }
    for (size_t i = 1; i < n; i++) {  // the members work side by side; the host waits for each in turn
        HIP_TRY(hipSetDevice(r->parts[i]->ctx->device));
        HIP_TRY(hipStreamSynchronize(r->parts[i]->ctx->stream));
    }
    return gather_scatter(r, 1);
}

// (inside an open ncclGroupStart: the thread's group has to be closed whatever happened — what was queued up to there may be
// a partial list, e.g. a Send whose Recv was never posted; the caller, group_gather, ABORTS the communicators before it waits
// for any stream, which is what unblocks such a kernel)
#define RCCL_TRY(expr)                                                                           \
    do {                                                                                         \
        const ncclResult_t e_ = (expr);                                                          \
        if (e_ != ncclSuccess) {                                                                 \
            if (in_group) (void)api.GroupEnd();                                                  \
            return fail(CHUNKY_E_HIP, "%s: %s", #expr, api.str(e_));                             \
        }                                                                                        \
    } while (0)

// Every member's stream drained of the passes queued on it (plain blocking waits: nothing of RCCL is on the streams yet), so
// that the deadline of the exchange that follows measures the exchange and not a long render before it.
static int group_drain_passes(chunky_render* r, std::vector<int>* devices, std::vector<hipStream_t>* streams) {
    for (chunky_render* part : r->parts) {
        HIP_TRY(hipSetDevice(part->ctx->device));
        HIP_TRY(hipStreamSynchronize(part->ctx->stream));
        devices->push_back(part->ctx->device);
        streams->push_back(part->ctx->stream);
    }
    return CHUNKY_OK;
}

// ONE grouped RCCL operation: member i's ncclSend of its packed blocks on its own stream (behind the pack kernel), member 0's
// matching ncclRecv's on its stream (ahead of the scatter kernels).  Every call that can fail for reasons of its own — buffer
// allocation, the pack launches, selecting a device — happens BEFORE ncclGroupStart.
static int group_gather_sendrecv(chunky_render* r) {
    const RcclApi& api = rccl_api();
    chunky_ctx* g = r->ctx;
    chunky_render* p0 = r->parts[0];
    const size_t n = r->parts.size(), first = g->self_exchange ? 0 : 1;
    bool in_group = false;
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    if (int rc = group_drain_passes(r, &devices, &streams)) return rc;
    for (size_t i = first; i < n; i++) {
        chunky_render* pi = r->parts[i];
        std::lock_guard<std::recursive_mutex> gi(pi->ctx->mu);
        const size_t bytes = (size_t)pi->shard.n_local * 3 * sizeof(float);
        if (bytes == 0) continue;
        if (int rc = gather_buffers(r, i, bytes)) return rc;
        HIP_TRY(launch_gather(true, pi->shard, pi->width, pi->height, pi->fb, (float*)r->gather_send[i].p, pi->ctx->stream));
    }
    RCCL_TRY(api.GroupStart());
    in_group = true;
    for (size_t i = first; i < n; i++) {
        chunky_render* pi = r->parts[i];
        const size_t count = (size_t)pi->shard.n_local * 3;
        if (count == 0) continue;
        (void)hipSetDevice(pi->ctx->device);  // (selected successfully a moment ago, in group_drain_passes)
        RCCL_TRY(api.Send(r->gather_send[i].p, count, ncclFloat, 0, g->comms[i], pi->ctx->stream));
        (void)hipSetDevice(p0->ctx->device);
        RCCL_TRY(api.Recv(r->gather_recv[i].p, count, ncclFloat, (int)i, g->comms[0], p0->ctx->stream));
    }
    in_group = false;
    RCCL_TRY(api.GroupEnd());
    if (int rc = group_wait(g, devices, streams)) return rc;  // the sends and the receives are complete, or the deadline has passed
    return gather_scatter(r, first);
}

// The literal form: every member clears what it does not own (after chunky_render_set_shard on a live render a member may still
// hold pixels of its old share; member 0 holds the blocks earlier read-backs left there), every framebuffer is then zero outside
// its member's own blocks, and ONE ncclReduce(sum) onto member 0 assembles the image in place.  (Pixels NO member owns — the
// other ranks' when the group itself is one rank of an outer chunky_render_set_shard split — are zero afterwards; the other two
// transports leave them as they were.)
static int group_gather_reduce(chunky_render* r) {
    const RcclApi& api = rccl_api();
    chunky_ctx* g = r->ctx;
    chunky_render* p0 = r->parts[0];
    const size_t n = r->parts.size();
    const size_t count = (size_t)p0->width * p0->height * 3;
    bool in_group = false;
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    if (int rc = group_drain_passes(r, &devices, &streams)) return rc;
    for (size_t i = 0; i < n; i++) {
        chunky_render* pi = r->parts[i];
        std::lock_guard<std::recursive_mutex> gi(pi->ctx->mu);
        HIP_TRY(hipSetDevice(pi->ctx->device));
        HIP_TRY(launch_clear_foreign(pi->shard, pi->width, pi->height, pi->fb, pi->ctx->stream));
    }
    RCCL_TRY(api.GroupStart());
    in_group = true;
    for (size_t i = 0; i < n; i++) {
        chunky_render* pi = r->parts[i];
        (void)hipSetDevice(pi->ctx->device);
        RCCL_TRY(api.Reduce(pi->fb, pi->fb, count, ncclFloat, ncclSum, 0, g->comms[i], pi->ctx->stream));
    }
    in_group = false;
    RCCL_TRY(api.GroupEnd());
    if (int rc = group_wait(g, devices, streams)) return rc;
    HIP_TRY(hipSetDevice(p0->ctx->device));
    return CHUNKY_OK;
}
#undef RCCL_TRY

static int group_gather(chunky_render* r) {
    chunky_ctx* g = r->ctx;
    if (g->transport != CHUNKY_TRANSPORT_PEER_COPY && !g->comms.empty()) {
        const int rc = g->transport == CHUNKY_TRANSPORT_RCCL_REDUCE ? group_gather_reduce(r) : group_gather_sendrecv(r);
        if (rc == CHUNKY_OK) return rc;
        // An RCCL call failed: the render must not be lost with it.  The members' own blocks are intact (the exchange only
        // ever writes buffers of its own, and — the reduce — pixels of member 0's image that member 0 does not own), so the
        // same read-back runs again on peer copies, and so does every later one; chunky_group_transport says why.
        // ABORT FIRST: if an RCCL kernel sits unfinished on a member's stream (a dead peer, or the partial list of a call that
        // failed inside ncclGroupStart), only ncclCommAbort ends it — a stream wait before the abort would never return.
        const std::string why = tls_error;
        group_close_rccl(g, true);
        for (chunky_render* part : r->parts) {
            (void)hipSetDevice(part->ctx->device);
            (void)hipStreamSynchronize(part->ctx->stream);
        }
        (void)hipGetLastError();
        g->transport = CHUNKY_TRANSPORT_PEER_COPY;
        g->transport_detail = "peer copies: " + why;
    }
    return group_gather_peer(r);
}

extern "C" int chunky_render_gather(chunky_render* r) {
    if (r && !r->parts.empty()) {
        std::lock_guard<std::recursive_mutex> g(r->ctx->mu);
        return group_gather(r);
    }
    return chunky_render_sync(r);
}

extern "C" int chunky_render_read(chunky_render* r, float* out, int64_t n) {
    if (r && !r->parts.empty()) {
        std::lock_guard<std::recursive_mutex> g(r->ctx->mu);
        const int64_t need = (int64_t)r->width * r->height * 3;
        if (!out || n != need) return fail(CHUNKY_E_INVALID, "render_read: need %lld floats, got %lld", (long long)need, (long long)n);
        if (int rc = group_gather(r)) return rc;
        return chunky_render_read(r->parts[0], out, n);
    }
    LOCK_RENDER(r);
    int64_t need = (int64_t)r->width * r->height * 3;
    if (!out || n != need) return fail(CHUNKY_E_INVALID, "render_read: need %lld floats, got %lld", (long long)need, (long long)n);
    HIP_TRY(hipMemcpyAsync(out, r->fb, (size_t)n * 4, hipMemcpyDeviceToHost, r->ctx->stream));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    return CHUNKY_OK;
}

extern "C" int chunky_render_kernel_time(chunky_render* r, float* total_ms, int* launches) {
    if (r && !r->parts.empty()) {  // the members run side by side: the slowest one's total, member 0's launch count
        std::lock_guard<std::recursive_mutex> g(r->ctx->mu);
        float worst = 0;
        for (size_t i = 0; i < r->parts.size(); i++) {
            float ms = 0;
            int n = 0;
            if (int rc = chunky_render_kernel_time(r->parts[i], &ms, &n)) return rc;
            if (ms > worst) worst = ms;
            if (i == 0 && launches) *launches = n;
        }
        if (total_ms) *total_ms = worst;
        return CHUNKY_OK;
    }
    LOCK_RENDER(r);
    if (int rc = collect_timing(r)) return rc;
    if (total_ms) *total_ms = r->timed_ms;
    if (launches) *launches = r->timed_launches;
    r->timed_ms = 0;
    r->timed_launches = 0;
    return CHUNKY_OK;
}

extern "C" int chunky_render_kernel_info(chunky_render* r, int32_t out8[8]) {
    if (r && !r->parts.empty()) return chunky_render_kernel_info(r->parts[0], out8);
    LOCK_RENDER(r);
    if (!out8) return fail(CHUNKY_E_INVALID, "kernel_info: NULL output");
    memset(out8, 0, 8 * sizeof(int32_t));
    out8[0] = r->last_choice.tree;
    out8[1] = r->last_choice.group;
    out8[2] = r->last_choice.bvh;
    out8[3] = r->last_choice.blocks;
    out8[4] = r->last_choice.pool;
    out8[5] = r->last_choice.ext;
    out8[7] = r->last_choice.sorted;
    if (r->launch_cap > 0) {
        out8[6] = r->launch_cap;  // (of the kernel family that ran last)
    } else {  // before the first launch: what chunky_render_passes is going to decide for this scene and option set
        SceneView S;
        int most = kMaxPassesPerLaunch;
        if (scene_view(r->scene, &S, false) == CHUNKY_OK && pool_kernel_applies(r->kernel_variant, S, r->opts, r->work_counter.p != nullptr)) most = kMaxPoolPasses;
        out8[6] = launch_pass_cap(r, kStagingBytes, most);
    }
    return CHUNKY_OK;
}

extern "C" int chunky_render_phase_stats(chunky_render* r, uint64_t* out24, int reset) {
    if (r && !r->parts.empty()) return chunky_render_phase_stats(r->parts[0], out24, reset);
    LOCK_RENDER(r);
    if (!out24) return fail(CHUNKY_E_INVALID, "phase_stats: NULL output");
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    HIP_TRY(hipMemcpy(out24, (char*)r->work_counter.p + 8, 192, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset((char*)r->work_counter.p + 8, 0, 192));
    return CHUNKY_OK;
}

extern "C" int chunky_render_preview(chunky_render* r, int32_t* argb_out) {
    if (r && !r->parts.empty()) return chunky_render_preview(r->parts[0], argb_out);  // one first-hit pass of the whole image: member 0
    LOCK_RENDER(r);
    if (!argb_out) return fail(CHUNKY_E_INVALID, "preview: NULL output");
    if (!r->have_camera) return fail(CHUNKY_E_STATE, "preview before set_camera");
    SceneView S;
    if (int rc = scene_view(r->scene, &S)) return rc;
    S.bvh_cull = r->opts.bvh_cull;
    DevBuf out;
    size_t bytes = (size_t)r->width * r->height * 4;
    HIP_TRY(hipMalloc(&out.p, bytes));
    out.bytes = bytes;
    HIP_TRY(launch_preview(r->kernel_variant, S, r->cam, r->opts, (int*)out.p, r->ctx->stream));
    HIP_TRY(hipMemcpyAsync(argb_out, out.p, bytes, hipMemcpyDeviceToHost, r->ctx->stream));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    return CHUNKY_OK;
}

extern "C" int chunky_render_trace_records(chunky_render* r, int32_t seed, const int32_t* gids, int n,
                                           chunky_hit_record* records, int32_t* counts, float* radiance) {
    if (r && !r->parts.empty()) return chunky_render_trace_records(r->parts[0], seed, gids, n, records, counts, radiance);
    LOCK_RENDER(r);
    if (n < 0 || (n > 0 && (!gids || !records || !counts || !radiance))) return fail(CHUNKY_E_INVALID, "trace_records: bad arguments");
    if (!r->have_camera) return fail(CHUNKY_E_STATE, "trace_records before set_camera");
    if (2 * r->opts.max_depth > kMaxTraces) return fail(CHUNKY_E_STATE, "trace_records holds %d traces per sample: max depth must be <= %d", kMaxTraces, kMaxTraces / 2);
    if (n == 0) return CHUNKY_OK;
    for (int i = 0; i < n; i++)
        if (gids[i] < 0 || gids[i] >= r->width * r->height) return fail(CHUNKY_E_INVALID, "trace_records: gid %d outside the image", gids[i]);
    SceneView S;
    if (int rc = scene_view(r->scene, &S)) return rc;
    S.bvh_cull = r->opts.bvh_cull;
    DevBuf dg, dr, dc, dq;
    hipStream_t st = r->ctx->stream;
    HIP_TRY(dg.upload(gids, (size_t)n * 4, st));
    HIP_TRY(hipMalloc(&dr.p, (size_t)n * kMaxTraces * sizeof(HitRecord)));
    HIP_TRY(hipMalloc(&dc.p, (size_t)n * 4));
    HIP_TRY(hipMalloc(&dq.p, (size_t)n * 12));
    HIP_TRY(hipMemsetAsync(dr.p, 0, (size_t)n * kMaxTraces * sizeof(HitRecord), st));
    HIP_TRY(launch_trace_records(r->kernel_variant, S, r->cam, r->opts, seed, (const int*)dg.p, n, (HitRecord*)dr.p, (int*)dc.p, (float*)dq.p, st));
    HIP_TRY(hipMemcpyAsync(records, dr.p, (size_t)n * kMaxTraces * sizeof(HitRecord), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(counts, dc.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(radiance, dq.p, (size_t)n * 12, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CHUNKY_OK;
}

// ------------------------------------------------------------------------------------ host loop
namespace {
struct JavaRandom {  // java.util.Random: 48-bit LCG, nextInt() = top 32 bits
    uint64_t s;
    explicit JavaRandom(int64_t seed) : s(((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1)) {}
    int32_t next_int() {
        s = (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (int32_t)(int64_t)(s >> 16);
    }
};
}  // namespace

extern "C" int chunky_java_random_ints(int64_t seed, int32_t* out, int n) {
    if (n < 0 || (n > 0 && !out)) return fail(CHUNKY_E_INVALID, "java_random_ints: bad arguments");
    JavaRandom rnd(seed);
    for (int i = 0; i < n; i++) out[i] = rnd.next_int();
    return CHUNKY_OK;
}

extern "C" int chunky_render_run_ex(chunky_render* r, double* sample_buffer, int32_t* scene_spp, int32_t target_spp,
                                    int32_t merge_interval, const chunky_run_callbacks* callbacks) {
    if (!r || !r->ctx) return fail(CHUNKY_E_INVALID, "NULL render");
    if (!sample_buffer || !scene_spp) return fail(CHUNKY_E_INVALID, "render_run: NULL buffer");
    if (merge_interval < 1) merge_interval = 1024;  // OpenClPathTracingRenderer.java:158
    // the caller's struct may be older (shorter) than this library's: copy what it holds, the rest stays NULL
    chunky_run_callbacks cb{};
    if (callbacks) {
        const size_t have = callbacks->struct_size;
        if (have < offsetof(chunky_run_callbacks, progress) || have % sizeof(void*) != 0)
            return fail(CHUNKY_E_INVALID, "render_run_ex: callbacks->struct_size %zu (set it to sizeof(chunky_run_callbacks))", have);
        memcpy(&cb, callbacks, have < sizeof cb ? have : sizeof cb);
    }
    const int64_t n = (int64_t)r->width * r->height * 3;
    std::vector<float> pass_buffer((size_t)n);
    JavaRandom rnd(0);                 // :95
    int logical_spp = *scene_spp;      // :91
    int samp_spp = *scene_spp;         // sceneSpp[0], :92
    auto last_callback = std::chrono::steady_clock::now();
    if (int rc = chunky_render_reset(r)) return rc;  // new float[] passBuffer uploaded with the buffer, :61,71
    // The launches below grow to what fits 95 ms.  Once the climb has shown where it is heading (a launch of 8 passes or more is
    // next), the staging array is sized ONCE for the launch size it will settle at instead of being regrown at every step; a
    // heavy scene that settles at a few passes never reserves anything.  The hint is dropped when the loop ends, however it ends.
    struct Reserve {
        chunky_render* r;
        void set(int passes) {
            std::lock_guard<std::recursive_mutex> g(r->ctx->mu);
            if (r->parts.empty()) r->reserve_passes = passes;
            for (chunky_render* part : r->parts) part->reserve_passes = passes;
        }
        ~Reserve() { set(0); }
    } reserve{r};
    int launch_passes = 1;             // adapts to ~95 ms per launch (below), so postRender is polled often enough
    while (logical_spp < target_spp) { // :102
        int buffer_spp = 0;            // bufferSppReal
        int until_merge = target_spp - logical_spp < merge_interval ? target_spp - logical_spp : merge_interval;
        bool stop = false, save = false, save_poll = false;
        while (buffer_spp < until_merge && !save) {
            int m = until_merge - buffer_spp < launch_passes ? until_merge - buffer_spp : launch_passes;
            if (cb.save_event)  // a snapshot / dump due inside the next launch, or a buffer to finalize, ends it there (:150)
                for (int k = 1; k <= m; k++)
                    if (const int ev = cb.save_event(cb.user, logical_spp + buffer_spp + k)) {
                        m = k;
                        save = true;
                        save_poll = ev != 2;  // a real save event is followed by one more poll (:179-182); shouldFinalizeBuffer alone is not
                        break;
                    }
            std::vector<int32_t> seeds((size_t)m);
            for (int k = 0; k < m; k++) seeds[(size_t)k] = rnd.next_int();  // :107
            auto t0 = std::chrono::steady_clock::now();
            if (int rc = chunky_render_passes(r, seeds.data(), m, buffer_spp)) return rc;
            if (int rc = chunky_render_sync(r)) return rc;                  // clWaitForEvents, :141
            auto t1 = std::chrono::steady_clock::now();
            buffer_spp += m;
            *scene_spp += m;                                                 // :144
            if (cb.progress) cb.progress(cb.user, *scene_spp);
            if (cb.regenerate_camera) cb.regenerate_camera(cb.user);         // :146-148
            double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
            // passes per launch: as many as fit ~95 ms at the rate just measured (postRender is polled between launches, at least
            // every 100 ms where a launch allows it), at most eight times the last launch — a launch of few passes overstates
            // the time per pass (its fixed costs), so the sequence climbs 1, 8, 64, ... and settles; it comes down the same way
            {
                const double per_pass = ms / (double)m;
                int want = per_pass > 0.0 ? (int)(95.0 / per_pass) : kMaxPassesPerLaunch;  // a launch stays under the 100 ms of :153
                if (want > launch_passes * 8) want = launch_passes * 8;
                if (want > kMaxPassesPerLaunch) want = kMaxPassesPerLaunch;
                if (want < 1) want = 1;
                if (want > launch_passes || ms > 90.0) launch_passes = want;
                // where the climb is heading: the rate just measured says how many passes fit 95 ms
                // (a quarter more than the rate says: the next estimate differs by a few passes, and a launch larger than the array by ONE pass
                // regrows it — 4.4 GB and 240 ms of hipMalloc in the middle of a warm render, seen on configs[1])
                if (launch_passes >= 8) reserve.set(per_pass > 0.0 && 119.0 / per_pass < (double)merge_interval ? (int)(119.0 / per_pass) + 1 : merge_interval);
            }
            if (!save && cb.post_render && std::chrono::duration<double, std::milli>(t1 - last_callback).count() > 100.0 &&
                (!cb.poll_gate || cb.poll_gate(cb.user))) {  // :153-157; the gate is `!manager.shouldFinalize()` (:154)
                last_callback = t1;
                if (cb.post_render(cb.user)) {
                    stop = true;
                    break;
                }
            }
        }
        if (!stop && cb.post_render && cb.post_render(cb.user)) stop = true;  // :163
        if (stop && buffer_spp == 0) return fail(CHUNKY_E_ABORTED, "stopped by postRender");
        if (int rc = chunky_render_read(r, pass_buffer.data(), n)) return rc;  // :164-166
        const double sinv = 1.0 / (samp_spp + buffer_spp);                   // :169
        const double a = samp_spp, b = buffer_spp;
        {   // :172-177: the reference merges on Chunky's common worker threads; here a few host threads, each its own range
            auto merge = [&](int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; i++)                             // :173
                    sample_buffer[i] = (sample_buffer[i] * a + (double)pass_buffer[(size_t)i] * b) * sinv;
            };
            unsigned workers = std::thread::hardware_concurrency();
            workers = workers > 16 ? 16 : (workers < 1 ? 1 : workers);
            if (n < (int64_t)1 << 18) workers = 1;
            std::vector<std::thread> pool;
            const int64_t chunk = (n + workers - 1) / workers;
            for (unsigned w = 1; w < workers; w++) pool.emplace_back(merge, (int64_t)w * chunk < n ? (int64_t)w * chunk : n, (int64_t)(w + 1) * chunk < n ? (int64_t)(w + 1) * chunk : n);
            merge(0, chunk < n ? chunk : n);
            for (auto& t : pool) t.join();
        }
        samp_spp += buffer_spp;
        logical_spp += buffer_spp;                                            // :178
        if (cb.merged) cb.merged(cb.user, samp_spp);                          // :174-176
        if (stop) return fail(CHUNKY_E_ABORTED, "stopped by postRender");
        if (save_poll && cb.post_render && cb.post_render(cb.user)) return fail(CHUNKY_E_ABORTED, "stopped by postRender");  // :179-182
        // bufferSppReal = 0 (:170): the next pass runs with spp = 0, i.e. (mean*0 + c)/1 — no reset needed
    }
    return CHUNKY_OK;
}

extern "C" int chunky_render_run(chunky_render* r, double* sample_buffer, int32_t* scene_spp, int32_t target_spp,
                                 int32_t merge_interval, chunky_post_render_fn post_render, void* user) {
    const chunky_run_callbacks cb{sizeof(chunky_run_callbacks), post_render, nullptr, nullptr, nullptr, nullptr, user, nullptr};
    return chunky_render_run_ex(r, sample_buffer, scene_spp, target_spp, merge_interval, &cb);
}

// ------------------------------------------------------------------------------------ wide tree hook
extern "C" int chunky_widetree_lookup(const int32_t* tree, int64_t n_ints, int depth, const int32_t* level_bits,
                                      int n_levels, const int32_t* xyz, int n, int32_t* data_out, int32_t* level_out,
                                      int64_t* n_entries) {
    if (int rc = check_ints(tree, n_ints, "widetree_lookup")) return rc;
    if (n_ints < 1 || n < 0 || (n > 0 && (!xyz || !data_out || !level_out))) return fail(CHUNKY_E_INVALID, "widetree_lookup: bad arguments");
    int bits[kWideMaxLevels];
    int nlev;
    if (level_bits) {
        if (n_levels < 1 || n_levels > kWideMaxLevels) return fail(CHUNKY_E_INVALID, "widetree_lookup: 1..%d levels", kWideMaxLevels);
        nlev = n_levels;
        for (int i = 0; i < nlev; i++) bits[i] = level_bits[i];
    } else {
        nlev = default_wide_levels(depth, bits);
    }
    WideTree wt;
    const char* why = "";
    if (!build_wide_tree(tree, n_ints, depth, bits, nlev, &wt, &why)) return fail(CHUNKY_E_INVALID, "wide tree: %s", why);
    if (n_entries) *n_entries = (int64_t)wt.data.size();
    for (int i = 0; i < n; i++) {
        int x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (((x | y | z) >> depth) != 0) return fail(CHUNKY_E_INVALID, "widetree_lookup: cell outside the world");
        int32_t e = 0;
        for (int l = 0; l < wt.nlev && e >= 0; l++) {
            const int sh = wt.shift[l], b = wt.bits[l], m = (1 << b) - 1;
            e = (int32_t)wt.data[(size_t)e + (size_t)(((((x >> sh) & m) << b) | ((y >> sh) & m)) << b | ((z >> sh) & m))];
        }
        if (e >= 0) return fail(CHUNKY_E_INVALID, "wide tree: lookup did not end in a leaf");
        level_out[i] = (e >> kWideLevelShift) & 15;
        const uint32_t ptr = (uint32_t)e & kWidePtrMask;
        data_out[i] = ptr == kWidePtrMask ? 0x7FFFFFFE : (int32_t)ptr;
    }
    return CHUNKY_OK;
}

// ------------------------------------------------------------------------------------ tone map
// The last steps of the GAMMA and ACES curves for one channel value (post_processing_filter.cl:24-27, rgba.h:9-14) on the
// host, with the rt_pow the kernels and the checkers share: pow(c, 1/2.2) * 255 + 0.5 -> (uint), saturating -> min(255).
static unsigned gamma_byte_host(float c) {
    const float f = rt_pow(c, (float)(1.0 / 2.2)) * 255.0f + 0.5f;
    const unsigned u = !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xFFFFFFFFu : (unsigned)f);
    return u > 255u ? 255u : u;
}
// T[k] (k = 1..255) = the smallest non-negative float whose byte is >= k, by bisection over the float's bit pattern (the
// byte is a non-decreasing function of c: checked over every float by tests/test_filter.py); T[0] = 0.
static const float* gamma_thresholds() {
    static float T[256];
    static std::once_flag once;
    std::call_once(once, [] {
        T[0] = 0.0f;
        for (int k = 1; k < 256; k++) {
            uint32_t lo = 0u, hi = 0x7F800000u;  // byte(+0) = 0 < k <= byte(+inf) = 255
            while (hi - lo > 1u) {
                const uint32_t mid = lo + (hi - lo) / 2;
                float c;
                memcpy(&c, &mid, 4);
                if (gamma_byte_host(c) >= (unsigned)k) hi = mid; else lo = mid;
            }
            memcpy(&T[k], &hi, 4);
        }
    });
    return T;
}
extern "C" int chunky_filter_gamma_thresholds(float* out256) {
    if (!out256) return fail(CHUNKY_E_INVALID, "gamma_thresholds: NULL output");
    memcpy(out256, gamma_thresholds(), 256 * sizeof(float));
    return CHUNKY_OK;
}
static int device_gamma_table(chunky_ctx* ctx, const float** out) {
    if (!ctx->gamma_table) {
        HIP_TRY(hipMalloc(&ctx->gamma_table, 256 * sizeof(float)));
        HIP_TRY(hipMemcpy(ctx->gamma_table, gamma_thresholds(), 256 * sizeof(float), hipMemcpyHostToDevice));
    }
    *out = (const float*)ctx->gamma_table;
    return CHUNKY_OK;
}

extern "C" int chunky_filter_frame(chunky_ctx* ctx, int width, int height, double exposure, const double* input,
                                   int32_t* argb_out, int type) {
    if (!ctx) return fail(CHUNKY_E_INVALID, "filter_frame: NULL context");
    if (!ctx->members.empty()) ctx = ctx->members[0];  // a group: the tone map of one frame runs on its first member
    if (width < 0 || height < 0) return fail(CHUNKY_E_INVALID, "filter_frame: %dx%d", width, height);
    const long long n = (long long)width * height;
    if (n == 0) return CHUNKY_OK;
    if (!input || !argb_out) return fail(CHUNKY_E_INVALID, "filter_frame: NULL buffer");
    std::lock_guard<std::recursive_mutex> guard(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf in, out;
    HIP_TRY(in.upload(input, (size_t)n * 24, ctx->stream));
    HIP_TRY(hipMalloc(&out.p, (size_t)n * 4));
    out.bytes = (size_t)n * 4;
    const float* table = nullptr;
    if (int rc = device_gamma_table(ctx, &table)) return rc;
    HIP_TRY(launch_filter(n, (float)exposure, (const double*)in.p, (unsigned*)out.p, type, ctx->stream, table));
    HIP_TRY(hipMemcpyAsync(argb_out, out.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return CHUNKY_OK;
}

extern "C" int chunky_filter_frame_device(chunky_ctx* ctx, int64_t n_pixels, float exposure, const void* d_input,
                                          void* d_argb, int type, int repeat, float* kernel_ms) {
    if (!ctx) return fail(CHUNKY_E_INVALID, "filter_frame_device: NULL context");
    if (!ctx->members.empty()) ctx = ctx->members[0];
    if (n_pixels < 0 || repeat < 1) return fail(CHUNKY_E_INVALID, "filter_frame_device: n_pixels=%lld repeat=%d", (long long)n_pixels, repeat);
    if (n_pixels > 0 && (!d_input || !d_argb)) return fail(CHUNKY_E_INVALID, "filter_frame_device: NULL buffer");
    if ((reinterpret_cast<uintptr_t>(d_input) & 7u) || (reinterpret_cast<uintptr_t>(d_argb) & 3u))
        return fail(CHUNKY_E_INVALID, "filter_frame_device: misaligned buffer");
    std::lock_guard<std::recursive_mutex> guard(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    const float* table = nullptr;
    if (int rc = device_gamma_table(ctx, &table)) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipError_t err = hipEventRecord(e0, ctx->stream);
    for (int k = 0; k < repeat && err == hipSuccess; k++)
        err = launch_filter(n_pixels, exposure, (const double*)d_input, (unsigned*)d_argb, type, ctx->stream, table);
    if (err == hipSuccess) err = hipEventRecord(e1, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    float ms = 0;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (err != hipSuccess) return fail(CHUNKY_E_HIP, "filter_frame_device: %s", hipGetErrorString(err));
    if (kernel_ms) *kernel_ms = ms / (float)repeat;
    return CHUNKY_OK;
}

// ------------------------------------------------------------------------------------ self test
extern "C" int chunky_selftest_math(chunky_ctx* ctx, int which, int n, const float* a, const float* b, float* out) {
    if (!ctx) return fail(CHUNKY_E_INVALID, "NULL context");
    if (!ctx->members.empty()) ctx = ctx->members[0];
    if (n < 0 || (n > 0 && (!a || !b || !out))) return fail(CHUNKY_E_INVALID, "selftest_math: bad arguments");
    if (n == 0) return CHUNKY_OK;
    std::lock_guard<std::recursive_mutex> g(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf da, db, dout;
    HIP_TRY(da.upload(a, (size_t)n * 4, ctx->stream));
    HIP_TRY(db.upload(b, (size_t)n * 4, ctx->stream));
    HIP_TRY(hipMalloc(&dout.p, (size_t)n * 4));
    HIP_TRY(launch_math_selftest(which, n, (const float*)da.p, (const float*)db.p, (float*)dout.p, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return CHUNKY_OK;
}

extern "C" int chunky_selftest_helpers(chunky_scene* scene, int which, int tree, int n, const float* in, float* out, int32_t* tree_used) {
    if (scene && !scene->replicas.empty()) return chunky_selftest_helpers(scene->replicas[0], which, tree, n, in, out, tree_used);
    LOCK_SCENE(scene);
    if (n < 0 || (n > 0 && (!in || !out))) return fail(CHUNKY_E_INVALID, "selftest_helpers: bad arguments");
    if (n == 0) return CHUNKY_OK;
    SceneView S;
    if (int rc = scene_view(scene, &S)) return rc;
    DevBuf din, dout;
    hipStream_t st = scene->ctx->stream;
    HIP_TRY(din.upload(in, (size_t)n * 32 * sizeof(float), st));
    HIP_TRY(hipMalloc(&dout.p, (size_t)n * 12 * sizeof(float)));
    int used = 0;
    HIP_TRY(launch_helpers_selftest(S, which, tree, n, (const float*)din.p, (float*)dout.p, &used, st));
    HIP_TRY(hipMemcpyAsync(out, dout.p, (size_t)n * 12 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (tree_used) *tree_used = used;
    return CHUNKY_OK;
}

extern "C" int chunky_selftest_gamma_scan(chunky_ctx* ctx, int curve, uint32_t first_bits, uint64_t count, uint64_t* mismatches, float* worst_estimate) {
    if (!ctx || !mismatches) return fail(CHUNKY_E_INVALID, "selftest_gamma_scan: NULL argument");
    if (!ctx->members.empty()) ctx = ctx->members[0];
    if (count > (1ull << 32) || (curve != 0 && curve != 2)) return fail(CHUNKY_E_INVALID, "selftest_gamma_scan: curve=%d count=%llu", curve, (unsigned long long)count);
    std::lock_guard<std::recursive_mutex> g(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    const float* table = nullptr;
    if (int rc = device_gamma_table(ctx, &table)) return rc;
    DevBuf out;
    HIP_TRY(hipMalloc(&out.p, 16));
    HIP_TRY(hipMemsetAsync(out.p, 0, 16, ctx->stream));
    HIP_TRY(launch_gamma_scan(first_bits, count, curve, table, (unsigned long long*)out.p, (float*)((char*)out.p + 8), ctx->stream));
    unsigned char host[16];
    HIP_TRY(hipMemcpyAsync(host, out.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(mismatches, host, 8);
    if (worst_estimate) memcpy(worst_estimate, host + 8, 4);
    return CHUNKY_OK;
}
