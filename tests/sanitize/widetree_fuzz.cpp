// tests/sanitize/widetree_fuzz.cpp — TEST-ONLY: the host-side octree re-layout (csrc/widetree.cpp: what every scene upload runs on
// caller-supplied ints) under AddressSanitizer + UBSan on the CPU, on well-formed and on hostile trees.  For a well-formed tree
// every cell's (leaf value, leaf level) out of the wide tree has to equal the reference descent of K/octree.h:81-89; a hostile
// tree (branch values outside the array, cycles, branches below level 0, pointers that do not fit) has to be refused or
// expressed — never read or written out of bounds, never run away.   (tests/test_sanitize.py builds and runs it.)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../chunkyclplugin_amd/csrc/widetree.hpp"

using namespace chunky;

// a random well-formed packed octree (ClSceneLoader layout: [0] root; value > 0 = index of an 8-child group, <= 0 = -(pointer))
static void grow(std::vector<int32_t>& t, size_t at, int level, std::mt19937& rng, int palette) {
    const bool branch = level > 0 && (rng() % 100) < (level > 2 ? 70u : 45u);
    if (!branch) {
        const unsigned r = rng() % 10;
        t[at] = r == 0 ? -0x7FFFFFFE : -(int32_t)(2 * (rng() % palette));
        if (r == 0) t[at] = (int32_t)0x80000002;  // -(0x7FFFFFFE) = ANY_TYPE
        return;
    }
    const size_t kids = t.size();
    t[at] = (int32_t)kids;
    t.resize(kids + 8, 0);
    for (int c = 0; c < 8; c++) grow(t, kids + c, level - 1, rng, palette);
}

static bool descend(const std::vector<int32_t>& t, int depth, int x, int y, int z, int32_t* data, int* level) {
    int lvl = depth;
    int32_t v = t[0];
    while (v > 0) {
        lvl--;
        if (lvl < 0) return false;
        const int64_t at = (int64_t)v + ((((x >> lvl) & 1) << 2) | (((y >> lvl) & 1) << 1) | ((z >> lvl) & 1));
        if (at < 0 || at >= (int64_t)t.size()) return false;
        v = t[(size_t)at];
    }
    *data = -v;
    *level = lvl;
    return true;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937 rng(20261002u);
    long long cells = 0, refused = 0, expressed = 0;
    for (int r = 0; r < rounds; r++) {
        const int depth = 1 + (int)(rng() % 6);
        std::vector<int32_t> t(1, 0);
        grow(t, 0, depth, rng, 50);
        const bool hostile = (r % 3) == 2;
        if (hostile && t.size() > 1) {  // damage it: wild branch values, a cycle, a huge pointer
            for (int k = 0; k < 1 + (int)(rng() % 4); k++) {
                const size_t at = rng() % t.size();
                switch (rng() % 4) {
                    case 0: t[at] = (int32_t)(rng() % (2 * t.size() + 8)); break;              // branch to anywhere (cycles included)
                    case 1: t[at] = (int32_t)(t.size() - 1 - (rng() % 8 < t.size() ? rng() % 8 : 0)); break;  // 8-group hanging off the end
                    case 2: t[at] = -(int32_t)(0x2000000 + rng() % 1000); break;               // a pointer beyond 25 bits
                    default: t[at] = (int32_t)0x7FFFFFF0; break;                               // a branch far outside
                }
            }
        }
        int bits[kWideMaxLevels];
        int nlev = default_wide_levels(depth, bits);
        if (rng() % 4 == 0) {  // another legal split of the same depth
            nlev = 0;
            for (int left = depth; left > 0 && nlev < kWideMaxLevels;) {
                const int b = 1 + (int)(rng() % 3);
                bits[nlev++] = b < left ? b : left;
                left -= bits[nlev - 1];
            }
        }
        WideTree w;
        const char* why = "";
        const bool ok = build_wide_tree(t.data(), (int64_t)t.size(), depth, bits, nlev, &w, &why);
        if (!ok) {
            if (!hostile) {
                fprintf(stderr, "round %d: a well-formed tree was refused: %s\n", r, why);
                return 1;
            }
            refused++;
            continue;
        }
        expressed++;
        std::vector<int32_t> palette(100);
        for (auto& p : palette) p = (int32_t)(rng() % 4);
        annotate_wide_tree(&w, palette.data(), (int64_t)palette.size());
        const int n = 1 << depth;
        for (int x = 0; x < n; x++)
            for (int y = 0; y < n; y++)
                for (int z = 0; z < n; z++) {
                    int32_t e = 0;
                    for (int l = 0; l < w.nlev && e >= 0; l++) {
                        const int sh = w.shift[l], b = w.bits[l], m = (1 << b) - 1;
                        const size_t at = (size_t)e + (size_t)(((((x >> sh) & m) << b) | ((y >> sh) & m)) << b | ((z >> sh) & m));
                        if (at >= w.data.size()) {
                            fprintf(stderr, "round %d: entry outside the array\n", r);
                            return 1;
                        }
                        e = (int32_t)w.data[at];
                    }
                    int32_t want_data;
                    int want_level;
                    const bool ref_ok = descend(t, depth, x, y, z, &want_data, &want_level);
                    if (hostile && !ref_ok) continue;  // the reference descent itself leaves the array here: nothing to compare
                    if (e >= 0) {
                        if (hostile) continue;
                        fprintf(stderr, "round %d: lookup did not end in a leaf\n", r);
                        return 1;
                    }
                    const uint32_t ptr = (uint32_t)e & kWidePtrMask;  // (entry layout: widetree.hpp)
                    const int32_t data = ptr == kWidePtrMask ? 0x7FFFFFFE : (int32_t)ptr;
                    const int level = (e >> kWideLevelShift) & 15;
                    if (!hostile && (data != want_data || level != want_level)) {
                        fprintf(stderr, "round %d: cell (%d,%d,%d): wide (%d, level %d), reference (%d, level %d)\n", r, x, y, z, data, level, want_data, want_level);
                        return 1;
                    }
                    cells++;
                }
    }
    printf("{\"rounds\": %d, \"expressed\": %lld, \"refused\": %lld, \"cells_compared\": %lld}\n", rounds, expressed, refused, cells);
    return 0;
}
