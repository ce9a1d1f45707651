// tests/sanitize/capi_host_fuzz.cpp — TEST-ONLY: the host-side parsers of csrc/capi.hip that walk caller-supplied ints — derive_records
// (block / material / AABB / quad palettes -> aligned records), build_quad_aux, build_bvh_records + bvh_leaves_sound (entity BVHs),
// list_emitters and model_leaf_permille (octree + palettes) — under AddressSanitizer + UBSan on the CPU.  The translation unit is capi.hip itself, compiled for
// the host only; nothing here touches a device (no HIP call is reached: the functions under test are the pure halves).
//
// Inputs: palettes and trees that are well formed, and the same with ints damaged at random (pointers outside their palettes, huge and
// negative counts, cycles).  Required: no out-of-bounds access, no undefined arithmetic, termination; for sound inputs the derived records
// have to be consistent with their sources (spot checks), and every block the derivation leaves on the packed path has to be one whose
// packed reads stay inside the palettes — that is what keeps hostile scene data from faulting the GPU.   (tests/test_sanitize.py)
#include "../../chunkyclplugin_amd/csrc/capi.hip"

#include <random>

// the kernel launchers live in the other translation units; nothing under test reaches them
namespace chunky {
hipError_t launch_render(int, const SceneView&, const CameraView&, const RenderOpts&, const ShardView&, const PassSeeds&, float*, int*, hipStream_t, KernelChoice*, float*, const int*) { return hipErrorNotSupported; }
bool pool_kernel_applies(int, const SceneView&, const RenderOpts&, bool) { return false; }
hipError_t launch_gather(bool, const ShardView&, int, int, float*, float*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_clear_foreign(const ShardView&, int, int, float*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_trace_records(int, const SceneView&, const CameraView&, const RenderOpts&, int, const int*, int, HitRecord*, int*, float*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_preview(int, const SceneView&, const CameraView&, const RenderOpts&, int*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_filter(long long, float, const double*, unsigned*, int, hipStream_t, const float*) { return hipErrorNotSupported; }
hipError_t launch_gamma_scan(unsigned, unsigned long long, int, const float*, unsigned long long*, float*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_helpers_selftest(const SceneView&, int, int, int, const float*, float*, int*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_math_selftest(int, int, const float*, const float*, float*, hipStream_t) { return hipErrorNotSupported; }
}  // namespace chunky

static std::mt19937 rng(7u);
static int32_t fbits(float f) {
    int32_t i;
    memcpy(&i, &f, 4);
    return i;
}
static float frand(float lo, float hi) { return lo + (hi - lo) * (float)(rng() % 10000) / 10000.0f; }

struct Palettes {
    std::vector<int32_t> B, M, A, Q;
};
static Palettes make_palettes(int n_mats, int n_models) {
    Palettes p;
    for (int m = 0; m < n_mats; m++) {
        const int32_t mat[6] = {(int32_t)(rng() % 8), (int32_t)rng(), (16 << 16) | 16, (int32_t)rng(), (int32_t)(rng() % 256), (int32_t)rng()};
        p.M.insert(p.M.end(), mat, mat + 6);
    }
    std::vector<int32_t> aabb_ptrs, quad_ptrs;
    for (int k = 0; k < n_models; k++) {
        aabb_ptrs.push_back((int32_t)p.A.size());
        const int count = 1 + (int)(rng() % 4);
        p.A.push_back(count);
        for (int i = 0; i < count; i++) {
            for (int w = 0; w < 6; w++) p.A.push_back(fbits(frand(0, 1)));
            p.A.push_back((int32_t)(rng() & 0xFFFFFF));
            for (int w = 0; w < 6; w++) p.A.push_back(6 * (int32_t)(rng() % n_mats));
        }
        quad_ptrs.push_back((int32_t)p.Q.size());
        const int qcount = 1 + (int)(rng() % 4);
        p.Q.push_back(qcount);
        for (int i = 0; i < qcount; i++) {
            for (int w = 0; w < 13; w++) p.Q.push_back(fbits(frand(-1, 1)));
            p.Q.push_back(6 * (int32_t)(rng() % n_mats));
            p.Q.push_back((int32_t)(rng() % 2));
        }
    }
    p.B = {0, 0};  // air
    for (int k = 0; k < 3 * n_models; k++) {
        const int type = 1 + (int)(rng() % 3);
        p.B.push_back(type);
        p.B.push_back(type == 1 ? 6 * (int32_t)(rng() % n_mats) : (type == 2 ? aabb_ptrs[rng() % aabb_ptrs.size()] : quad_ptrs[rng() % quad_ptrs.size()]));
    }
    return p;
}
static void damage(std::vector<int32_t>& v, int hits) {
    static const int32_t nasty[] = {-1, -7, 0x7FFFFFFF, (int32_t)0x80000000, 1 << 30, 255, 256, 100000, -100000};
    for (int k = 0; k < hits && !v.empty(); k++) v[rng() % v.size()] = (rng() % 3) ? nasty[rng() % 9] : (int32_t)rng();
}

// would the kernels' PACKED path (rt_device.hpp block_hit -> aabb_model_hit / quad_model_hit / cube) stay inside the palettes for this block?
static bool packed_reads_inside(const Palettes& p, int32_t type, int32_t ptr) {
    if (type == 1) return ptr >= 0 && (size_t)ptr + 5 <= p.M.size();
    if (type != 2 && type != 3) return true;  // unknown model type: never read
    const std::vector<int32_t>& P = type == 2 ? p.A : p.Q;
    const int64_t stride = type == 2 ? 13 : 15;
    if (ptr < 0 || (size_t)ptr >= P.size()) return false;
    const int64_t count = P[(size_t)ptr];
    if (count < 0 || (size_t)(ptr + 1 + stride * count) > P.size()) return false;
    for (int64_t i = 0; i < count; i++) {
        const int32_t* prim = &P[(size_t)(ptr + 1 + stride * i)];
        if (type == 2) {
            for (int w = 1; w < 6; w++)
                if (prim[7 + w] < 0 || (size_t)prim[7 + w] + 6 > p.M.size()) return false;
        } else if (prim[13] < 0 || (size_t)prim[13] + 6 > p.M.size()) {
            return false;
        }
    }
    return true;
}

static int fuzz_palettes(int rounds) {
    long long never = 0, recorded = 0, packed = 0;
    for (int r = 0; r < rounds; r++) {
        Palettes p = make_palettes(2 + (int)(rng() % 6), 1 + (int)(rng() % 5));
        const bool hostile = r % 2;
        if (hostile) {
            damage(p.B, 1 + (int)(rng() % 3));
            damage(p.A, (int)(rng() % 4));
            damage(p.Q, (int)(rng() % 4));
            damage(p.M, (int)(rng() % 2));
            if (rng() % 8 == 0) p.M.resize(p.M.size() - (rng() % 7 < p.M.size() ? rng() % 7 : 0));  // a material palette cut short
            if (rng() % 8 == 0) p.B.pop_back();
        }
        DerivedRecords d;
        derive_records(p.B, p.M, p.A, p.Q, &d);
        std::vector<float> aux;
        (void)build_quad_aux(p.B, p.Q, &aux);
        if (d.info.size() != p.B.size() / 2 * 8 || d.mat8.size() != p.M.size() / 6 * 8) return fprintf(stderr, "round %d: sizes\n", r), 1;
        for (size_t k = 0; k < p.B.size() / 2; k++) {
            const int32_t* e = &d.info[k * 8];
            const int32_t type = p.B[2 * k], ptr = p.B[2 * k + 1];
            if (e[0] == 0x7FFFFFFF) {
                never++;
                if (!hostile) return fprintf(stderr, "round %d: a sound block was switched off\n", r), 1;
                continue;
            }
            if (e[0] != type || e[1] != ptr) return fprintf(stderr, "round %d: block %zu changed\n", r, k), 1;
            if ((type == 2 || type == 3) && e[7] != 0) {
                // the record path: first record and count have to lie inside the record array
                const int64_t first = (uint32_t)e[7] >> 8, count = e[7] & 0xFF;
                const size_t words = type == 2 ? 12 : 24;
                const std::vector<int32_t>& rec = type == 2 ? d.aabb_rec : d.quad_rec;
                if ((size_t)(first + count) * words > rec.size()) return fprintf(stderr, "round %d: record range\n", r), 1;
                recorded++;
            } else {
                if (!packed_reads_inside(p, type, ptr)) return fprintf(stderr, "round %d: block %zu (type %d) left on a packed path that reads outside\n", r, k, type), 1;
                packed++;
            }
        }
    }
    printf("{\"palette_rounds\": %d, \"blocks_switched_off\": %lld, \"on_records\": %lld, \"on_packed_path\": %lld}\n", rounds, never, recorded, packed);
    return 0;
}

// a random binary BVH in the packed layout (7 ints per node: [second child | -trig pointer, 6 bounds], first child at +7)
static void grow_bvh(std::vector<int32_t>& N, std::vector<int32_t>& T, int depth, int n_mats) {
    const size_t at = N.size();
    N.resize(at + 7, 0);
    for (int w = 1; w < 7; w++) N[at + w] = fbits(frand(0, 32));
    if (depth == 0 || rng() % 4 == 0) {
        const int32_t prim = (int32_t)T.size();
        const int count = (int)(rng() % 5);
        T.push_back(count);
        for (int i = 0; i < count; i++) {
            T.push_back((int32_t)(rng() % 2) << 8);
            for (int w = 0; w < 18; w++) T.push_back(fbits(frand(-2, 2)));
            T.push_back(6 * (int32_t)(rng() % n_mats));
        }
        N[at] = -prim;
        return;
    }
    grow_bvh(N, T, depth - 1, n_mats);
    N[at] = (int32_t)N.size();
    grow_bvh(N, T, depth - 1, n_mats);
}
static bool links_sound(const std::vector<int32_t>& N) {  // chunky_scene_set_bvh's own check (links inside the array, no cycle, depth <= 63)
    std::vector<std::pair<int64_t, int>> todo{{0, 0}};
    int64_t visited = 0;
    while (!todo.empty()) {
        auto [at, d] = todo.back();
        todo.pop_back();
        if (at < 0 || at + 7 > (int64_t)N.size() || ++visited > (int64_t)N.size() || d > 63) return false;
        if (N[(size_t)at] > 0) {
            todo.emplace_back(at + 7, d + 1);
            todo.emplace_back((int64_t)N[(size_t)at], d + 1);
        }
    }
    return true;
}

// the tree the records describe, independent of where they sit: depth-first, an inner record's twelve box words then its two
// subtrees, a leaf's count and triangle words
static void canonical(const std::vector<int32_t>& nodes, const std::vector<int32_t>& tris, int32_t ref, std::vector<int32_t>* out, int depth = 0) {
    if (depth > 80) return;
    if (ref < 0) {
        const int64_t leaf = -1 - (int64_t)ref, first = leaf >> 6, count = leaf & 63;
        out->push_back((int32_t)count);
        out->insert(out->end(), tris.begin() + first * 20, tris.begin() + (first + count) * 20);
        return;
    }
    const int32_t* r = &nodes[(size_t)ref * 16];
    out->insert(out->end(), r + 4, r + 16);
    canonical(nodes, tris, r[0], out, depth + 1);
    canonical(nodes, tris, r[1], out, depth + 1);
}

static int fuzz_bvh(int rounds) {
    long long built = 0, unfit = 0, refused = 0, relaid = 0;
    for (int r = 0; r < rounds; r++) {
        chunky_scene s;
        const int n_mats = 1 + (int)(rng() % 5);
        s.host_materials.assign((size_t)n_mats * 6, 0);
        s.host_trigs = {0};  // pointer 0: an empty leaf
        grow_bvh(s.host_world_bvh, s.host_trigs, 1 + (int)(rng() % 6), n_mats);
        grow_bvh(s.host_actor_bvh, s.host_trigs, (int)(rng() % 3), n_mats);
        s.world_empty = s.actor_empty = false;
        const bool hostile = r % 2;
        if (hostile) {
            damage(s.host_world_bvh, (int)(rng() % 3));
            damage(s.host_trigs, 1 + (int)(rng() % 3));
            if (rng() % 6 == 0) s.host_materials.resize(s.host_materials.size() - 3);
        }
        if (!links_sound(s.host_world_bvh) || !links_sound(s.host_actor_bvh)) continue;  // chunky_scene_set_bvh refuses these
        const bool sound = bvh_leaves_sound(s.host_world_bvh, false, s.host_trigs, s.host_materials) &&
                           bvh_leaves_sound(s.host_actor_bvh, false, s.host_trigs, s.host_materials);
        if (!hostile && !sound) return fprintf(stderr, "round %d: a sound BVH was refused\n", r), 1;
        std::vector<int32_t> nodes, tris;
        int wr = 0, ar = 0;
        const bool ok = build_bvh_records(&s, &nodes, &tris, &wr, &ar);
        if (!sound) {
            refused++;  // scene_view fails the render call; the records (built or not) are never used
            continue;
        }
        if (!ok) {
            unfit++;  // a sound BVH that does not fit the record layout: the packed walk runs, and stays inside (sound)
            continue;
        }
        built++;
        // every reference in the records has to resolve inside them
        const int64_t n_inner = (int64_t)nodes.size() / 16, n_tri = (int64_t)tris.size() / 20;
        auto ref_ok = [&](int32_t ref) {
            if (ref >= 0) return ref < n_inner || n_inner == 0;
            const int64_t leaf = -1 - (int64_t)ref, first = leaf >> 6, count = leaf & 63;
            return first + count <= n_tri;
        };
        for (int64_t i = 0; i < n_inner; i++)
            if (!ref_ok(nodes[(size_t)i * 16]) || !ref_ok(nodes[(size_t)i * 16 + 1])) return fprintf(stderr, "round %d: dangling reference\n", r), 1;
        if (!ref_ok(wr) || !ref_ok(ar)) return fprintf(stderr, "round %d: dangling root\n", r), 1;
        for (int64_t t = 0; t < n_tri; t++)
            if (tris[(size_t)t * 20 + 7] < 0 || (size_t)tris[(size_t)t * 20 + 7] / 2 >= s.host_materials.size() / 6) return fprintf(stderr, "round %d: material index\n", r), 1;
        // the same BVHs placed as a breadth-first top over treelets (relayout_bvh_records, tiny sizes so that every branch of
        // it runs on these small trees): references resolve, and the tree they describe is the same tree
        {
            static const char* const layouts[] = {"3,4", "0,2", "1,1000", "1000,3", "2,1"};
            setenv("CHUNKY_BVH_LAYOUT", layouts[r % 5], 1);
            std::vector<int32_t> nodes2, tris2;
            int wr2 = 0, ar2 = 0;
            const bool ok2 = build_bvh_records(&s, &nodes2, &tris2, &wr2, &ar2);
            unsetenv("CHUNKY_BVH_LAYOUT");
            if (!ok2 || nodes2.size() != nodes.size() || tris2.size() != tris.size()) return fprintf(stderr, "round %d: re-layout changed the record counts\n", r), 1;
            const int64_t n_tri2 = (int64_t)tris2.size() / 20;
            for (int64_t i = 0; i < n_inner; i++)
                for (int c = 0; c < 2; c++) {
                    const int32_t ref = nodes2[(size_t)i * 16 + c];
                    if (ref >= 0 ? ref >= n_inner : ((-1 - (int64_t)ref) >> 6) + ((-1 - (int64_t)ref) & 63) > n_tri2) return fprintf(stderr, "round %d: re-layout left a dangling reference\n", r), 1;
                }
            std::vector<int32_t> a, b;
            canonical(nodes, tris, wr, &a);
            canonical(nodes, tris, ar, &a);
            canonical(nodes2, tris2, wr2, &b);
            canonical(nodes2, tris2, ar2, &b);
            if (a != b) return fprintf(stderr, "round %d: re-layout %s changed the tree\n", r, layouts[r % 5]), 1;
            relaid++;
        }
    }
    printf("{\"bvh_rounds\": %d, \"records_built\": %lld, \"sound_but_unfit\": %lld, \"refused\": %lld, \"relayouts_same_tree\": %lld}\n", rounds, built, unfit, refused, relaid);
    return 0;
}

static int fuzz_emitters(int rounds) {
    long long listed = 0;
    for (int r = 0; r < rounds; r++) {
        chunky_scene s;
        Palettes p = make_palettes(3, 2);
        s.host_blocks = p.B;
        s.host_materials = p.M;
        s.octree_depth = 1 + (int)(rng() % 5);
        // a tree whose branch values stay inside the array (what chunky_scene_set_octree checks) but may form cycles and over-deep chains
        const size_t groups = 1 + rng() % 40;
        s.host_octree.assign(1 + 8 * groups, 0);
        for (auto& v : s.host_octree) {
            if (rng() % 3 == 0)
                v = (int32_t)(1 + 8 * (rng() % groups));
            else
                v = -(int32_t)(rng() % (p.B.size() + 8));
        }
        if (r % 2) damage(s.host_blocks, 2), damage(s.host_materials, 1);
        std::vector<int32_t> out;
        list_emitters(&s, &out);
        if (out.size() % 4 != 0 || out.size() / 4 > s.host_octree.size()) return fprintf(stderr, "round %d: emitter list size\n", r), 1;
        // (the same walk of caller-supplied leaves: how common model blocks are, which picks render_pool's sorted block tests)
        const int share = model_leaf_permille(s.host_octree, s.host_blocks);
        if (share < 0 || share > 1000) return fprintf(stderr, "round %d: model share %d\n", r, share), 1;
        listed += (long long)out.size() / 4;
    }
    printf("{\"emitter_rounds\": %d, \"emitters_listed\": %lld}\n", rounds, listed);
    return 0;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    if (int rc = fuzz_palettes(rounds)) return rc;
    if (int rc = fuzz_bvh(rounds)) return rc;
    if (int rc = fuzz_emitters(rounds)) return rc;
    return 0;
}
