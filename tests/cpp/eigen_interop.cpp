// Compiled only where Eigen is on the include path (tests/test_host_io.py adds the reference's vendored copy when it is
// there): reference call sites hand Eigen::Vector3f to the facade and take it back.
#include <cstdio>
#include <cstring>
#include "../../raytracing_project_amd/host/flyscene.hpp"
#ifndef RTXHOST_HAVE_EIGEN
#error "Eigen not found on the include path"
#endif
using namespace rtxhost;
int main() {
  Eigen::Vector3f o(0.25f, -1.5f, 3.0f), d(1.0f, 2.0f, -2.0f), n(0.0f, 0.0f, 1.0f);
  Ray r(o, d);                                   // Eigen in
  Eigen::Vector3f dir = r.getDirection();        // Eigen out
  Eigen::Vector3f want = d.normalized();
  Eigen::Vector3f p = r.getPointOnRay(2.0f);
  Ray refl = r.reflectRay(n, p);
  Eigen::Vector3f rd = refl.getDirection();
  Eigen::Vector3f wrd = (dir - 2.0f * dir.dot(n) * n).normalized();
  const float got[9] = {dir[0], dir[1], dir[2], p[0], p[1], p[2], rd[0], rd[1], rd[2]};
  const Eigen::Vector3f wp = o + 2.0f * want;
  const float exp[9] = {want[0], want[1], want[2], wp[0], wp[1], wp[2], wrd[0], wrd[1], wrd[2]};
  for (int i = 0; i < 9; ++i) if (std::memcmp(&got[i], &exp[i], 4) != 0) { std::printf("mismatch at %d: %.9g vs %.9g\n", i, got[i], exp[i]); return 1; }
  Box b; b.tmin = Eigen::Vector3f(-1, -1, -1); b.tmax = Eigen::Vector3f(1, 1, 1);
  Eigen::Vector3f lo = b.tmin;
  if (lo[0] != -1.0f) return 2;
  std::printf("eigen interop ok\n");
  return 0;
}
