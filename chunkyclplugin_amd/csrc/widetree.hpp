// widetree.hpp — re-layout of the packed octree into a shallow, wide tree (host side, at upload).
//
// The reference looks a block up by descending the octree one bit per level from the root on every
// march step (K/octree.h:81-89: up to `depth` dependent 4-byte loads).  The lookup is a pure
// function cell -> (leaf value, leaf level), so any structure that returns the same pair for every
// cell is parity-safe.  The wide tree consumes `bits[i]` address bits per
// axis at level i, i.e. a node at level i is a dense (2^bits[i])^3 grid of 32-bit entries:
//
//   entry >= 0 : int offset of the child node (level i+1) inside the array
//   entry <  0 : leaf — bit 30 = the block cannot be hit (filled in by annotate_wide_tree once the block palette is
//                known: air, model type 0 / unknown, K/block.h:44-47, a pointer outside the palette; ANY_TYPE,
//                K/block.h:32, from the start) — so "leaf that can be hit" is ONE unsigned compare, entry < 0xC0000000;
//                bits 29..26 = level of the octree leaf that contains the cell (its cube has edge 2^level; needed
//                for the leaf-exit box of K/octree.h:103-106); bit 25 = the low bit of the block's kind (0 full
//                cube K/block.h:48 / 1 AABB or quad model :66,:92; of a block that cannot be hit: 0 / 1 = ANY_TYPE);
//                bits 24..0 = block-palette pointer (K/octree.h:88); all ones = ANY_TYPE
//
// With bits = {3,3,3} a depth-9 world needs at most 3 dependent loads per lookup instead of 9.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace chunky {

constexpr int kWideMaxLevels = 6;
constexpr uint32_t kWideLeaf = 0x80000000u;
constexpr uint32_t kWideNoHit = 0x40000000u;   // with kWideLeaf: (entry as unsigned) >= kWideLeaf | kWideNoHit <=> the march walks on
constexpr int kWideLevelShift = 26;
constexpr uint32_t kWideKindLow = 0x2000000u;
constexpr uint32_t kWidePtrMask = 0x1FFFFFFu;   // as a pointer value: ANY_TYPE
constexpr uint32_t kWideAny = kWideNoHit | kWideKindLow | kWidePtrMask;

struct WideTree {
    std::vector<uint32_t> data;
    int nlev = 0;
    int shift[kWideMaxLevels] = {0};
    int bits[kWideMaxLevels] = {0};
};

// Returns false (with *why set) when the octree cannot be expressed: depth > 15, a block pointer
// that does not fit 27 bits, or a malformed tree.
bool build_wide_tree(const int32_t* oct, int64_t n_ints, int depth, const int* level_bits, int nlev, WideTree* out,
                     const char** why);

// Sets the kind bits of every leaf from the block palette (2 ints per block: modelType, pointer).
void annotate_wide_tree(WideTree* t, const int32_t* block_palette, int64_t n_ints);

// Default split: ceil(depth/3) levels of 3 bits each (top level padded), see widetree.cpp.
int default_wide_levels(int depth, int* level_bits);

}  // namespace chunky
