// The host BVH builder on 8 threads against the same build on 1 thread (leaf order and collapsed groups must be identical),
// compiled with -fsanitize=thread by tests/test_bvh_threads.py: the helper pool, the parallel partition and the threaded collapse
// plan of bvh_build.cpp under a race detector.  No GPU.
#include "../../raytracing_project_amd/csrc/bvh_build.h"
#include <cstring>
#include <cstdio>
#include <random>
#include <vector>
int main() {
  const uint32_t n = 150000;
  std::mt19937 g(3); std::uniform_real_distribution<float> u(-1.f, 1.f), e(0.001f, 0.03f);
  std::vector<float> lo(3 * n), hi(3 * n);
  for (uint32_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { float c = u(g), h = e(g); lo[3 * i + k] = c - h; hi[3 * i + k] = c + h; }
  rtx::BvhBuildResult a, b;
  rtx::build_bvh(lo.data(), hi.data(), n, 8, &a);
  rtx::collapse_to_quads(&a);
  rtx::build_bvh(lo.data(), hi.data(), n, 1, &b);
  rtx::collapse_to_quads(&b);
  bool same = a.nodes.size() == b.nodes.size() && a.order == b.order && a.quads.size() == b.quads.size();
  for (size_t i = 0; same && i < a.quads.size(); ++i) same = std::memcmp(&a.quads[i], &b.quads[i], sizeof(rtx::BvhNode)) == 0;
  std::printf("8 threads vs 1 thread: %s (%zu nodes, %zu quads)\n", same ? "IDENTICAL" : "DIFFERENT", a.nodes.size(), a.quads.size());
  return same ? 0 : 1;
}
