// Host-side builder of the linear BVH that replaces the reference's Box hierarchy
// (src/box.h:8-97, Flyscene::divideBox_KD src/flyscene.cpp:1110-1185).
#pragma once
#include <cstdint>
#include <vector>

namespace rtx {

// 32-byte node.  count == 0: internal, children at leftFirst and leftFirst+1 (adjacent,
// so one 64-byte fetch brings both child boxes).  count > 0: leaf, triangles
// [leftFirst, leftFirst+count) of the leaf-ordered triangle arrays.
struct BvhNode {
  float lo[3];
  uint32_t leftFirst;
  float hi[3];
  uint32_t count;
};
static_assert(sizeof(BvhNode) == 32, "node must be 32 bytes");

// Marks an unused slot of a 4-slot group (see collapse_to_quads).
constexpr uint32_t kEmptySlot = 0xFFFFFFFFu;
// ... and in the packed encoding: a leaf without triangles.  The kernels never mean to visit it (its planes are
// inverted, so no finite ray enters it), but a ray with NaN or overflowing coordinates fails every comparison
// and may: the entry must then be harmless to follow (the first fuzzing run faulted the GPU on 0xFFFFFFFF).
constexpr uint32_t kEmptyPacked = 0x80000000u;

struct BvhBuildResult {
  std::vector<BvhNode> quads;      // traversal layout: groups of 4 sibling slots, group 0 = children of the root
  uint32_t quad_depth = 0;         // levels of groups (root group = 1)
  uint32_t quad_slots = 0;         // non-empty slots
  uint32_t top_groups[4] = {0, 0, 0, 0};   // [l]: groups in levels 1..l (l <= kTopLevels); they are groups 0 .. top_groups[l]-1
  std::vector<BvhNode> nodes;      // binary tree the groups are collapsed from; nodes[0] = root
  std::vector<uint32_t> order;     // leaf order -> input triangle id
  uint32_t nleaves = 0;
  uint32_t max_depth = 0;          // root = depth 0
};

constexpr int kTraversalStackDepth = 48;   // per-lane traversal stack entries of the kernels (rtx_kernels.h: kStackDepth); <= 3 per group level
constexpr int kTopLevels = 3;      // levels of groups numbered breadth first (<= 1 + 4 + 16 = 21 groups): the kernels stage them in LDS
constexpr int kMaxLeafTris = 4;    // fits the 3-bit count of the traversal stack entry (<= 7)
constexpr int kMaxDepth = 30;      // binary depth; the collapsed tree has <= 15 levels of groups

// tri_lo/tri_hi: per-triangle (already padded) bounds, 3 floats each.
// Binned-SAH top-down build; deterministic for a given input regardless of thread count.
void build_bvh(const float* tri_lo, const float* tri_hi, uint32_t ntris, int threads, BvhBuildResult* out);

// Collapses the binary tree into the layout the kernels walk: every internal node's two
// children are replaced by their own children where those are internal, giving groups of up
// to 4 sibling 32-byte nodes stored contiguously (one fetch per traversal step instead of two
// dependent ones; the kernels load the 64-byte PackedGroup form below).  A slot is {lo, ref, hi, count}:
// count > 0 leaf over triangles [ref, ref+count); count == 0 internal, ref = index of its
// child group; ref == kEmptySlot unused.  Fills out->quads / quad_depth / quad_slots.
void collapse_to_quads(BvhBuildResult* out);

// 64-byte device encoding of one group (what the kernels actually load; half the bytes of the
// fp32 layout, and the L1 return path is what bounds the traversal kernels): child boxes are
// 8-bit fixed point relative to the group's own bounds, rounded outwards.
//   origin[3]   group lower corner (fp32)
//   step_x/y/z  power-of-two step of each axis (fp32, 2^e with e chosen so that 255 steps span the group)
//   entry[4]    traversal entries: leaf = 0x80000000 | count << 28 | first triangle,
//               internal = child group index, unused = 0x80000000 (kEmptyPacked)
//   lox..hiz    byte k = child k's plane in steps from origin (lower planes rounded down,
//               upper planes rounded up); an unused slot has lower planes 255 and upper planes 0
struct PackedGroup {
  float origin[3];
  float step_x;
  uint32_t entry[4];
  uint32_t lox, loy, loz, hix;
  uint32_t hiy, hiz;
  float step_y, step_z;
};
static_assert(sizeof(PackedGroup) == 64, "packed group must be 64 bytes");
void pack_groups(const std::vector<BvhNode>& quads, std::vector<PackedGroup>* out);

}  // namespace rtx
