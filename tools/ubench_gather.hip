// tools/ubench_gather.hip — what does a gather of 64-byte records cost on gfx950, by the number of lanes that share a record?
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_gather tools/ubench_gather.hip && gpurun_out/ubench_gather
//
// The entity-BVH walk of render_pool reads one 64-byte record per walker and step as four 16-byte loads of ONE lane
// (pool_walk.hpp); a rig showed its time to be proportional to the per-lane load instructions.  This
// program times dependent chains of such gathers (the next record index comes out of the record just read) in three shapes:
//   lane : one lane per record, four 16-byte loads            (what the walk does)
//   pair : two adjacent lanes per record, two loads each
//   quad : four adjacent lanes per record, one load each      (a quad's 4 x 16 B is one aligned 64-byte line)
// at the walk's occupancy (5 waves per SIMD), on a table that fits the L2s (8 MiB) and one that fits an L1 (16 KiB).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

template <int CTRL>
__device__ inline int dpp(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

// SHARE = lanes per record (1, 2, 4); every lane of a group ends a step knowing the next record index
template <int SHARE>
__global__ __launch_bounds__(256) void chase(const int4* __restrict__ table, unsigned mask, int steps, int* out) {
    const unsigned lane = threadIdx.x + blockIdx.x * blockDim.x;
    const unsigned walker = lane / SHARE, sub = lane % SHARE;
    unsigned idx = (walker * 2654435761u) & mask;
    int acc = 0;
    for (int s = 0; s < steps; ++s) {
        const int4* p = table + (size_t)idx * 4;
        int x;
        if (SHARE == 1) {
            const int4 a = p[0], b = p[1], c = p[2], d = p[3];
            x = a.x ^ b.y ^ c.z ^ d.w;
            acc += a.y + b.z + c.w + d.x;
        } else if (SHARE == 2) {
            const int4 a = p[sub * 2], b = p[sub * 2 + 1];
            int part = a.x ^ b.y;
            acc += a.y + b.z;
            x = part ^ dpp<0xB1>(part);  // quad_perm [1,0,3,2]
        } else {
            const int4 a = p[sub];
            int part = a.x;
            acc += a.y;
            part ^= dpp<0xB1>(part);
            x = part ^ dpp<0x4E>(part);  // quad_perm [2,3,0,1]
        }
        idx = ((unsigned)x + walker * 0x9E3779B1u + (unsigned)s * 40503u) & mask;  // salted per walker: chains never merge
    }
    out[lane] = acc + idx;
}

// 64 walkers per wave, records fetched by quads: in round k the four lanes of a quad read the four 16-byte chunks of the record
// of the quad's walker k (one aligned 64-byte line per quad), then every walker collects its own four chunks.
//   MODE 0: LDS-DMA (global_load_lds, 16 bytes per lane) into four 1 KiB planes (stride 1040: conflict-free), four ds_read_b128
//   MODE 1: loads to registers, ds_write_b128 into the same planes, four ds_read_b128
//   MODE 2: loads to registers, 4 x 4 transpose inside the quad by DPP moves and selects (no LDS)
template <int CTRL>
__device__ inline unsigned long long dpp64(unsigned long long v) {
    return (unsigned long long)(unsigned)dpp<CTRL>((int)(unsigned)v) | ((unsigned long long)(unsigned)dpp<CTRL>((int)(unsigned)(v >> 32)) << 32);
}
template <int CTRL>
__device__ inline int4 dpp4(int4 v) {
    return int4{dpp<CTRL>(v.x), dpp<CTRL>(v.y), dpp<CTRL>(v.z), dpp<CTRL>(v.w)};
}
__device__ inline int4 sel4(bool c, int4 a, int4 b) { return c ? a : b; }
constexpr int kPlane = 1040;
template <int MODE>
__global__ __launch_bounds__(256) void chase_coop(const int4* __restrict__ table, unsigned mask, int steps, int* out) {
    extern __shared__ char lds_raw[];
    const unsigned lane = threadIdx.x + blockIdx.x * blockDim.x;
    const unsigned walker = lane, l = threadIdx.x & 63u, j = l & 3u;
    char* buf = lds_raw + (threadIdx.x >> 6) * (4 * kPlane);
    unsigned idx = (walker * 2654435761u) & mask;
    int acc = 0;
    for (int s = 0; s < steps; ++s) {
        const char* base = (const char*)table;
        const unsigned off = idx * 64u + j * 16u;   // this lane's chunk of ITS walker's record; the quad's walker k's via DPP
        const unsigned o0 = (unsigned)dpp<0x00>((int)off) - 0, o1 = (unsigned)dpp<0x55>((int)off), o2 = (unsigned)dpp<0xAA>((int)off), o3 = (unsigned)dpp<0xFF>((int)off);
        // offsets of walker k's record start (lane k's off minus its own chunk offset k * 16) plus this lane's chunk
        const unsigned a0 = o0 + j * 16u, a1 = o1 - 16u + j * 16u, a2 = o2 - 32u + j * 16u, a3 = o3 - 48u + j * 16u;
        int4 c0, c1, c2, c3;
        if (MODE == 0) {
            __builtin_amdgcn_global_load_lds((const void*)(base + a0), (__attribute__((address_space(3))) void*)(buf + 0 * kPlane), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)(base + a1), (__attribute__((address_space(3))) void*)(buf + 1 * kPlane), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)(base + a2), (__attribute__((address_space(3))) void*)(buf + 2 * kPlane), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)(base + a3), (__attribute__((address_space(3))) void*)(buf + 3 * kPlane), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int4* mine = (const int4*)(buf + (l & 3u) * kPlane + (l >> 2) * 64u);
            c0 = mine[0]; c1 = mine[1]; c2 = mine[2]; c3 = mine[3];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next round's DMA lands
        } else {
            const int4 r0 = *(const int4*)(base + a0), r1 = *(const int4*)(base + a1), r2 = *(const int4*)(base + a2), r3 = *(const int4*)(base + a3);
            if (MODE == 1) {
                *(int4*)(buf + 0 * kPlane + l * 16u) = r0;
                *(int4*)(buf + 1 * kPlane + l * 16u) = r1;
                *(int4*)(buf + 2 * kPlane + l * 16u) = r2;
                *(int4*)(buf + 3 * kPlane + l * 16u) = r3;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int4* mine = (const int4*)(buf + (l & 3u) * kPlane + (l >> 2) * 64u);
                c0 = mine[0]; c1 = mine[1]; c2 = mine[2]; c3 = mine[3];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                // lane j holds r[k] = chunk j of walker k; wanted: lane k holds c[j] = lane j's r[k]: a 4 x 4 transpose in two stages
                const bool b0 = j & 1u, b1 = j & 2u;
                // stage 1 (lanes j ^ 1, registers k ^ 1)
                int4 s01 = dpp4<0xB1>(sel4(b0, r0, r1)), s23 = dpp4<0xB1>(sel4(b0, r2, r3));
                int4 t0 = sel4(b0, s01, r0), t1 = sel4(b0, r1, s01), t2 = sel4(b0, s23, r2), t3 = sel4(b0, r3, s23);
                // stage 2 (lanes j ^ 2, registers k ^ 2)
                int4 s02 = dpp4<0x4E>(sel4(b1, t0, t2)), s13 = dpp4<0x4E>(sel4(b1, t1, t3));
                c0 = sel4(b1, s02, t0); c2 = sel4(b1, t2, s02); c1 = sel4(b1, s13, t1); c3 = sel4(b1, t3, s13);
            }
        }
        const int x = c0.x ^ c1.y ^ c2.z ^ c3.w;
        acc += c0.y + c1.z + c2.w + c3.x;
        idx = ((unsigned)x + walker * 0x9E3779B1u + (unsigned)s * 40503u) & mask;
    }
    out[lane] = acc + idx;
}

template <int MODE>
static double run_coop(const int4* table, unsigned records, int steps, int* out, int blocks, int* check) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const size_t lds = 4 * 4 * kPlane;
    hipLaunchKernelGGL(chase_coop<MODE>, dim3(blocks), dim3(256), lds, 0, table, records - 1, steps / 10, out);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(chase_coop<MODE>, dim3(blocks), dim3(256), lds, 0, table, records - 1, steps, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipMemcpy(check, out, 64 * 4, hipMemcpyDeviceToHost));
    return (double)blocks * 256 * steps / (ms * 1e-3);
}

// one lane per record with LOADS loads of WIDTH dwords each (the rest of the record is not read): what a lane-read costs by width
template <int LOADS, int WIDTH>
__global__ __launch_bounds__(256) void chase_width(const int* __restrict__ table, unsigned mask, int steps, int* out) {
    const unsigned lane = threadIdx.x + blockIdx.x * blockDim.x;
    unsigned idx = (lane * 2654435761u) & mask;
    int acc = 0;
    for (int s = 0; s < steps; ++s) {
        const int* p = table + (size_t)idx * 16;
        int x = 0;
#pragma unroll
        for (int l = 0; l < LOADS; l++) {
            if (WIDTH == 4) {
                const int4 v = *(const int4*)(p + 4 * l);
                x ^= v.x;
                acc += v.y + v.z + v.w;
            } else if (WIDTH == 2) {
                const int2 v = *(const int2*)(p + 4 * l);
                x ^= v.x;
                acc += v.y;
            } else {
                x ^= p[4 * l];
            }
        }
        idx = ((unsigned)x + lane * 0x9E3779B1u + (unsigned)s * 40503u) & mask;
    }
    out[lane] = acc + idx;
}
template <int LOADS, int WIDTH>
static double run_width(const int4* table, unsigned records, int steps, int* out, int blocks) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((chase_width<LOADS, WIDTH>), dim3(blocks), dim3(256), 0, 0, (const int*)table, records - 1, steps / 10, out);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((chase_width<LOADS, WIDTH>), dim3(blocks), dim3(256), 0, 0, (const int*)table, records - 1, steps, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return (double)blocks * 256 * steps / (ms * 1e-3);
}

template <int SHARE>
static double run(const int4* table, unsigned records, int steps, int* out, int blocks) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(chase<SHARE>, dim3(blocks), dim3(256), 0, 0, table, records - 1, steps / 10, out);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(chase<SHARE>, dim3(blocks), dim3(256), 0, 0, table, records - 1, steps, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double recs = (double)blocks * 256 / SHARE * steps;
    return recs / (ms * 1e-3);
}

int main() {
    const int blocks = 256 * 5;  // 5 waves per SIMD on 256 CUs
    const int steps = 4000;
    int* out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    for (unsigned records : {1u << 8, 1u << 12, 1u << 15, 1u << 16, 1u << 17}) {
        std::vector<int> h((size_t)records * 16);
        unsigned s = 12345;
        for (auto& v : h) {
            s = s * 1664525u + 1013904223u;
            v = (int)(s >> 4);
        }
        int4* table;
        CK(hipMalloc(&table, h.size() * 4));
        CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        const double l = run<1>(table, records, steps, out, blocks);
        const double p = run<2>(table, records, steps, out, blocks);
        const double q = run<4>(table, records, steps, out, blocks);
        int ref[64], got[64];
        CK(hipMemcpy(ref, out, 0, hipMemcpyDeviceToHost));
        run<1>(table, records, steps, out, blocks);
        CK(hipMemcpy(ref, out, 64 * 4, hipMemcpyDeviceToHost));
        const double g = run_coop<0>(table, records, steps, out, blocks, got);
        bool same = true;
        for (int i = 0; i < 64; i++) same &= ref[i] == got[i];
        const double w = run_coop<1>(table, records, steps, out, blocks, got);
        for (int i = 0; i < 64; i++) same &= ref[i] == got[i];
        const double t = run_coop<2>(table, records, steps, out, blocks, got);
        for (int i = 0; i < 64; i++) same &= ref[i] == got[i];
        printf("{\"table_bytes\": %zu, \"coop_ldsdma_Grec_s\": %.2f, \"coop_ldswrite_Grec_s\": %.2f, \"coop_dpp_Grec_s\": %.2f, \"vs_lane\": [%.3f, %.3f, %.3f], \"same_results_as_lane\": %s}\n",
               h.size() * 4, g / 1e9, w / 1e9, t / 1e9, g / l, w / l, t / l, same ? "true" : "false");
        printf("{\"table_bytes\": %zu, \"lane_Grec_s\": %.2f, \"pair_Grec_s\": %.2f, \"quad_Grec_s\": %.2f, \"pair_vs_lane\": %.3f, \"quad_vs_lane\": %.3f}\n",
               h.size() * 4, l / 1e9, p / 1e9, q / 1e9, p / l, q / l);
        printf("{\"table_bytes\": %zu, \"lane_Grec_s_by_loads_x_dwords\": {\"4x4\": %.2f, \"3x4\": %.2f, \"2x4\": %.2f, \"1x4\": %.2f, \"4x2\": %.2f, \"4x1\": %.2f}}\n", h.size() * 4,
               run_width<4, 4>(table, records, steps, out, blocks) / 1e9, run_width<3, 4>(table, records, steps, out, blocks) / 1e9,
               run_width<2, 4>(table, records, steps, out, blocks) / 1e9, run_width<1, 4>(table, records, steps, out, blocks) / 1e9,
               run_width<4, 2>(table, records, steps, out, blocks) / 1e9, run_width<4, 1>(table, records, steps, out, blocks) / 1e9);
        printf("{\"table_bytes\": %zu, \"one_load_per_record_Grec_s_by_dwords\": {\"1x1\": %.2f, \"1x2\": %.2f, \"1x4\": %.2f, \"2x1\": %.2f, \"2x2\": %.2f}}\n", h.size() * 4,
               run_width<1, 1>(table, records, steps, out, blocks) / 1e9, run_width<1, 2>(table, records, steps, out, blocks) / 1e9,
               run_width<1, 4>(table, records, steps, out, blocks) / 1e9, run_width<2, 1>(table, records, steps, out, blocks) / 1e9,
               run_width<2, 2>(table, records, steps, out, blocks) / 1e9);
        CK(hipFree(table));
    }
    return 0;
}
