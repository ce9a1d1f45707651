// widetree.cpp — see widetree.hpp
#include "widetree.hpp"

#include <cstdlib>
#include <deque>

namespace chunky {

int default_wide_levels(int depth, int* level_bits) {
    // one dense top node of 4..7 bits per axis over levels of 3 bits: the top octree-node fetches of a lookup become one array
    // read, and the kernels keep compile-time shifts for the levels below.  The top holds at most 64^3 entries (1 MiB) — except
    // that a 4-bit top over two or more levels gives one of them up and becomes a 7-bit top (128^3 entries, 8 MiB): a depth-10
    // world is then two dependent reads per lookup instead of three (the reference's benchmark city: 5153 -> 5424 Msamples/s;
    // a depth-7 world as ONE 128^3 node measured 8 % slower than 4 + 3 and keeps its level)
    if (depth < 0) depth = 0;
#ifdef CHUNKY_TUNING  // tuning builds only (tools/variants.sh): the shipping library reads no tuning variable
    if (const char* e = getenv("CHUNKY_WIDE_LEVELS")) {  // tuning runs: an explicit split "top,l1,l2,..." (runs the generic tree form)
        int b[kWideMaxLevels], n = 0, sum = 0;
        for (const char* p = e; *p && n < kWideMaxLevels;) {
            b[n] = atoi(p);
            sum += b[n++];
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        bool sane = n >= 1 && sum == depth;
        for (int i = 0; i < n; i++) sane = sane && b[i] >= 1 && b[i] <= 7;  // a level of 7 bits is 128^3 entries (8 MiB); anything else is ignored
        if (sane) {
            for (int i = 0; i < n; i++) level_bits[i] = b[i];
            return n;
        }
    }
#endif
    int most_top = 6;
    bool fixed_top = false;
#ifdef CHUNKY_TUNING
    if (const char* e = getenv("CHUNKY_WIDE_TOP_BITS")) {  // tuning runs: a larger dense top (7: 128^3 entries = 8 MiB) for one level less
        const int v = atoi(e);
        if (v >= 4 && v <= 7) most_top = v;
        fixed_top = true;
    }
#endif
    int n3 = depth <= most_top ? 0 : (depth - most_top + 2) / 3;
    if (n3 > kWideMaxLevels - 1) n3 = kWideMaxLevels - 1;
    if (!fixed_top && n3 >= 2 && depth - 3 * n3 == 4) n3 -= 1;
    level_bits[0] = depth - 3 * n3;
    for (int i = 1; i <= n3; i++) level_bits[i] = 3;
    return n3 + 1;
}

namespace {
struct Job {
    int64_t base;  // offset of the node to fill
    int level;     // level index
    int32_t root;  // octree value covering the node's region
};
}  // namespace

bool build_wide_tree(const int32_t* oct, int64_t n, int depth, const int* level_bits, int nlev, WideTree* out,
                     const char** why) {
    static const char* kDepth = "octree depth outside 0..15";
    static const char* kBits = "level bits do not add up to the octree depth";
    static const char* kPtr = "block pointer does not fit 25 bits";
    static const char* kTree = "octree node outside the array";
    static const char* kSize = "wide tree would exceed 2^30 entries";
    if (depth < 0 || depth > 15) return *why = kDepth, false;
    int sum = 0;
    for (int i = 0; i < nlev; i++) sum += level_bits[i];
    // the top level may address up to 2 more bits than the world has (padding: those cells are
    // outside [0, 2^depth) and are never looked up)
    const int pad = sum - depth;
    if (nlev < 1 || nlev > kWideMaxLevels || pad < 0 || pad > level_bits[0]) return *why = kBits, false;
    out->nlev = nlev;
    int s = sum;
    for (int i = 0; i < nlev; i++) {
        s -= level_bits[i];
        out->bits[i] = level_bits[i];
        out->shift[i] = s;
    }
    std::vector<uint32_t>& d = out->data;
    d.clear();
    d.resize((size_t)1 << (3 * level_bits[0]));
    std::deque<Job> queue;
    queue.push_back(Job{0, 0, oct[0]});
    while (!queue.empty()) {  // breadth-first: nodes of one level are contiguous, upper levels first
        Job j = queue.front();
        queue.pop_front();
        const int b = out->bits[j.level], sh = out->shift[j.level];
        const int side = 1 << b;
        const int real = j.level == 0 ? b - pad : b;  // address bits of this level that exist in the world
        for (int ex = 0; ex < side; ex++)
            for (int ey = 0; ey < side; ey++)
                for (int ez = 0; ez < side; ez++) {
                    int32_t val = j.root;
                    int lvl = sh + real;
                    if (((ex | ey | ez) >> real) != 0) val = 0;  // padding cell: air, never read
                    for (int k = real - 1; k >= 0 && val > 0; k--) {
                        lvl--;
                        int64_t at = (int64_t)val + ((((ex >> k) & 1) << 2) | (((ey >> k) & 1) << 1) | ((ez >> k) & 1));
                        if (at < 0 || at >= n) return *why = kTree, false;
                        val = oct[at];
                    }
                    const int64_t slot = j.base + (((int64_t)ex << (2 * b)) | ((int64_t)ey << b) | ez);
                    if (val <= 0) {
                        uint32_t code = (uint32_t)(-(int64_t)val);
                        if (code == 0x7FFFFFFEu)
                            code = kWideAny;
                        else if (code >= kWidePtrMask)
                            return *why = kPtr, false;
                        d[(size_t)slot] = kWideLeaf | ((uint32_t)lvl << kWideLevelShift) | code;
                    } else {
                        if (j.level + 1 >= nlev) return *why = kTree, false;  // a branch below level 0
                        const int64_t child = (int64_t)d.size();
                        const int64_t csize = (int64_t)1 << (3 * out->bits[j.level + 1]);
                        if (child + csize >= ((int64_t)1 << 30)) return *why = kSize, false;
                        d.resize((size_t)(child + csize));
                        d[(size_t)slot] = (uint32_t)child;
                        queue.push_back(Job{child, j.level + 1, val});
                    }
                }
    }
    return true;
}

void annotate_wide_tree(WideTree* t, const int32_t* blocks, int64_t n) {
    for (uint32_t& e : t->data) {
        if (!(e & kWideLeaf)) continue;
        const uint32_t ptr = e & kWidePtrMask;
        if (ptr == kWidePtrMask) continue;  // ANY_TYPE: marked when the tree was built
        uint32_t kind = 2;
        if (ptr != 0 && (int64_t)ptr + 1 < n) {
            const int32_t type = blocks[ptr];
            kind = type == 1 ? 0u : ((type == 2 || type == 3) ? 1u : 2u);
        }
        e = (e & ~(kWideNoHit | kWideKindLow)) | ((kind & 2u) ? kWideNoHit : 0u) | ((kind & 1u) ? kWideKindLow : 0u);
    }
}

}  // namespace chunky
