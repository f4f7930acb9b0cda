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

struct BvhBuildResult {
  std::vector<BvhNode> nodes;      // nodes[0] = root
  std::vector<uint32_t> order;     // leaf order -> input triangle id
  uint32_t nleaves = 0;
  uint32_t max_depth = 0;          // root = depth 0
};

constexpr int kMaxLeafTris = 4;    // fits the 3-bit count of the traversal stack entry (<= 7)
constexpr int kMaxDepth = 30;      // traversal stack has 32 entries per lane

// tri_lo/tri_hi: per-triangle (already padded) bounds, 3 floats each.
// Binned-SAH top-down build; deterministic for a given input regardless of thread count.
void build_bvh(const float* tri_lo, const float* tri_hi, uint32_t ntris, int threads, BvhBuildResult* out);

}  // namespace rtx
