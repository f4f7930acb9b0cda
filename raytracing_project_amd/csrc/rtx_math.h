// fp32 vector arithmetic in the evaluation order the reference gets from Eigen 3.3.7
// (non-vectorised Vector3f; reductions as c0 + (c1 + c2), Eigen/src/Core/Redux.h:91-103;
// normalized() = v / sqrt(v.v) when v.v > 0, Eigen/src/Core/Dot.h:124-134).
// Shared by the host flattening code and the HIP kernels; both are compiled with
// -ffp-contract=off so that no a*b+c is fused (the reference is an SSE2 build without FMA).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RTX_HD __host__ __device__ __forceinline__
#else
#define RTX_HD inline
#endif

namespace rtx {

struct V3 { float x, y, z; };

RTX_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RTX_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RTX_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RTX_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RTX_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
RTX_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RTX_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
RTX_HD V3 cmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RTX_HD float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
RTX_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RTX_HD float norm(V3 a) { return sqrtf(dot(a, a)); }
RTX_HD V3 normalized(V3 a) {
  float z = dot(a, a);
  if (z > 0.0f) return a / sqrtf(z);
  return a;
}
// Tucano::Mesh::createFaces normal (dependencies/tucano/tucano/mesh.hpp:461-463)
RTX_HD V3 face_normal(V3 a, V3 b, V3 c) {
  V3 v1 = normalized(c - a);
  V3 v0 = normalized(b - a);
  return normalized(cross(v0, v1));
}
// Ray::getPointOnRay (src/Ray.h:28-30)
RTX_HD V3 point_on_ray(V3 o, V3 d, float t) { return v3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z); }
// Ray::reflectRay direction before the Ray ctor normalises it (src/Ray.h:33);
// Shader::reflect has the same shape (src/ShaderLib.h:7-9).
RTX_HD V3 reflect_dir(V3 d, V3 n) {
  float s = 2.0f * dot(d, n);
  return d - s * n;
}
// Eigen 3x3 cofactor inverse (Eigen/src/LU/InverseImpl.h:124-170), row-major in/out.
RTX_HD void mat3_inverse(const float* m, float* out) {
#define RTX_M(i, j) m[3 * (i) + (j)]
#define RTX_COF(i, j) (RTX_M(((i) + 1) % 3, ((j) + 1) % 3) * RTX_M(((i) + 2) % 3, ((j) + 2) % 3) - RTX_M(((i) + 1) % 3, ((j) + 2) % 3) * RTX_M(((i) + 2) % 3, ((j) + 1) % 3))
  float c0 = RTX_COF(0, 0), c1 = RTX_COF(1, 0), c2 = RTX_COF(2, 0);
  float det = c0 * RTX_M(0, 0) + (c1 * RTX_M(1, 0) + c2 * RTX_M(2, 0));
  float invdet = 1.0f / det;
  out[0] = c0 * invdet; out[1] = c1 * invdet; out[2] = c2 * invdet;
  out[3] = RTX_COF(0, 1) * invdet; out[4] = RTX_COF(1, 1) * invdet; out[5] = RTX_COF(2, 1) * invdet;
  out[6] = RTX_COF(0, 2) * invdet; out[7] = RTX_COF(1, 2) * invdet; out[8] = RTX_COF(2, 2) * invdet;
#undef RTX_COF
#undef RTX_M
}
RTX_HD V3 mat3_mul(const float* m, V3 v) {
  return v3(m[0] * v.x + (m[1] * v.y + m[2] * v.z), m[3] * v.x + (m[4] * v.y + m[5] * v.z),
            m[6] * v.x + (m[7] * v.y + m[8] * v.z));
}

}  // namespace rtx
