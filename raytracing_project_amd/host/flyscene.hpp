// Host-side mirror of the reference's ray-tracing surface over the C ABI (include/rtx.h).
//
// Same class and method names, argument meaning and miss/error behaviour as
//   Flyscene  (src/flyscene.hpp:30-221)      Scene   (src/Scene.h:18-78)
//   Ray       (src/Ray.h:7-58)               HitInfo (src/hitInfo.h:4-16)
//   Box       (src/box.h:8-97)               Tucano::Face / Material::Mtl / Flycamera (the parts the tracer reads)
// so that callers of the reference (src/main.cpp:63-77) compile against this header
// instead.  Every query is forwarded to the GPU library; nothing here intersects or
// shades on the CPU.  GL preview, debug-ray drawing and GLFW input are out of scope.
//
// Vector types: a small Vector3f with the Eigen spelling the callers use (x(), y(), z(),
// operator[], + - *, dot, norm, normalized).  Eigen itself is not a dependency.
#pragma once
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <chrono>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/rtx.h"

// Call sites of the reference pass Eigen::Vector3f (src/flyscene.hpp:12-13): where Eigen is on the include path the facade's
// vector converts from and to it implicitly, so those call sites compile unchanged; without Eigen nothing of it is needed.
#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && !defined(RTXHOST_NO_EIGEN)
#include <Eigen/Dense>
#define RTXHOST_HAVE_EIGEN 1
#endif
#endif

namespace rtxhost {

struct Vector3f {
  float v[3];
  Vector3f() : v{0, 0, 0} {}
  Vector3f(float x, float y, float z) : v{x, y, z} {}
#ifdef RTXHOST_HAVE_EIGEN
  Vector3f(const Eigen::Vector3f& e) : v{e[0], e[1], e[2]} {}
  operator Eigen::Vector3f() const { return Eigen::Vector3f(v[0], v[1], v[2]); }
#endif
  float x() const { return v[0]; }
  float y() const { return v[1]; }
  float z() const { return v[2]; }
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
  Vector3f operator+(const Vector3f& o) const { return Vector3f(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  Vector3f operator-(const Vector3f& o) const { return Vector3f(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
  Vector3f operator-() const { return Vector3f(-v[0], -v[1], -v[2]); }
  Vector3f operator*(float s) const { return Vector3f(v[0] * s, v[1] * s, v[2] * s); }
  Vector3f operator/(float s) const { return Vector3f(v[0] / s, v[1] / s, v[2] / s); }
  Vector3f& operator+=(const Vector3f& o) { *this = *this + o; return *this; }
  // Eigen evaluation order (Redux.h:91-103, Dot.h:124-134)
  float dot(const Vector3f& o) const { return v[0] * o.v[0] + (v[1] * o.v[1] + v[2] * o.v[2]); }
  float squaredNorm() const { return dot(*this); }
  float norm() const { return std::sqrt(squaredNorm()); }
  Vector3f normalized() const { float z = squaredNorm(); return z > 0.0f ? *this / std::sqrt(z) : *this; }
  Vector3f cross(const Vector3f& o) const {
    return Vector3f(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
  }
  Vector3f cwiseProduct(const Vector3f& o) const { return Vector3f(v[0] * o.v[0], v[1] * o.v[1], v[2] * o.v[2]); }
};
inline Vector3f operator*(float s, const Vector3f& a) { return Vector3f(s * a.v[0], s * a.v[1], s * a.v[2]); }
struct Vector2f { float v[2]; Vector2f(float a = 0, float b = 0) : v{a, b} {} float operator[](int i) const { return v[i]; } };
struct Vector2i { int v[2]; Vector2i(int a = 0, int b = 0) : v{a, b} {} int& operator[](int i) { return v[i]; } int operator[](int i) const { return v[i]; } };
struct Vector4f { float v[4]; Vector4f(float a = 0, float b = 0, float c = 0, float d = 0) : v{a, b, c, d} {}
  float w() const { return v[3]; } Vector3f head3() const { return Vector3f(v[0], v[1], v[2]); } };

struct RtxError : std::runtime_error { int code; RtxError(int c, const std::string& m) : std::runtime_error(m), code(c) {} };
inline int rtx_check(int rc) { if (rc < 0) throw RtxError(rc, std::string("rtx: ") + rtx_last_error()); return rc; }

namespace Tucano {
// tucano/mesh.hpp:238-247
class Face { public: std::vector<unsigned> vertex_ids; int material_id = -1; Vector3f normal; };
namespace Material {
// tucano/materials/mtl.hpp:16-116
class Mtl {
 public:
  Vector3f ka{0.3f, 0.3f, 0.3f}, kd{0.5f, 0.5f, 0.5f}, ks{1.0f, 1.0f, 1.0f};
  float shininess = 10, optical_density = 0.0f, dissolve_factor = 1.0f;
  int illumination_model = 0;
  std::string name;
  void setAmbient(Vector3f v) { ka = v; } void setDiffuse(Vector3f v) { kd = v; } void setSpecular(Vector3f v) { ks = v; }
  void setShininess(float v) { shininess = v; } void setOpticalDensity(float v) { optical_density = v; }
  void setDissolveFactor(float v) { dissolve_factor = v; } void setIlluminationModel(int v) { illumination_model = v; }
  void setName(std::string n) { name = n; } std::string getName() { return name; }
  Vector3f getDiffuse() const { return kd; } Vector3f getAmbient() const { return ka; } Vector3f getSpecular() const { return ks; }
  float getShininess() const { return shininess; } float getOpticalDensity() const { return optical_density; }
  float getDissolveFactor() const { return dissolve_factor; } float getIlluminationModel() const { return (float)illumination_model; }
  rtx_material raw() const {
    rtx_material m;
    for (int k = 0; k < 3; ++k) { m.ka[k] = ka[k]; m.kd[k] = kd[k]; m.ks[k] = ks[k]; }
    m.ns = shininess; m.ni = optical_density; m.d = dissolve_factor; m.illum = illumination_model;
    return m;
  }
};
}  // namespace Material

// The state of Tucano::Flycamera the tracer reads (tucano/utils/flycamera.hpp:76-86,166-191,
// tucano/camera.hpp:42-78).  The view matrix is kept as [R|t]; yaw/pitch composition follows
// updateViewMatrix (rows = rotated X, Y, Z axes, then translate(default) and translate(user)).
class Flycamera {
 public:
  float view[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, -2};   // identity rotation, default translation (0,0,-2)
  float viewport[4] = {0, 0, 1, 1};
  float fovy = 60.0f, aspect_ratio = 1.0f;
  void setPerspectiveMatrix(float fy, float aspect, float /*near*/, float /*far*/) { fovy = fy; aspect_ratio = aspect; }
  void setViewport(const Vector2f& vp) { viewport[0] = 0; viewport[1] = 0; viewport[2] = vp[0]; viewport[3] = vp[1]; }
  Vector2i getViewportSize() const { return Vector2i((int)viewport[2], (int)viewport[3]); }
  void setViewMatrix(const float m34[12]) { for (int i = 0; i < 12; ++i) view[i] = m34[i]; }
  rtx_camera raw() const {
    rtx_camera c;
    for (int i = 0; i < 12; ++i) c.view[i] = view[i];
    for (int i = 0; i < 4; ++i) c.viewport[i] = viewport[i];
    c.fovy = fovy; c.aspect = aspect_ratio; c.subpixel[0] = c.subpixel[1] = 0.0f;
    return c;
  }
};
}  // namespace Tucano

// src/hitInfo.h:4-16
struct HitInfo { float t; int faceId; Vector3f normal; Vector3f point; };

// src/Ray.h:7-35 (construction normalises; reflectRay as the tracer uses it)
class Ray {
  Vector3f origin, dir;
 public:
  Ray(Vector3f _origin, Vector3f _dir) { origin = _origin; dir = _dir.normalized(); }
  Vector3f getOrigin() { return origin; }
  Vector3f getDirection() { return dir; }
  Vector3f getPointOnRay(float t) { return Vector3f(origin[0] + t * dir[0], origin[1] + t * dir[1], origin[2] + t * dir[2]); }
  Ray reflectRay(Vector3f normal, Vector3f point) {
    Vector3f newDir = dir - (2.0f * dir.dot(normal)) * normal;
    return Ray(point, newDir);
  }
  // src/Ray.h:36-42.  pow(float, 2) and sqrt evaluate in double under libstdc++ (the build the oracle is pinned to);
  // a negative radicand (total internal reflection) gives a NaN direction, as in the reference.
  Ray refractRay(Vector3f normal, Vector3f point, Vector3f viewDirection, float firstIndex, float secondIndex) {
    float dot = viewDirection.dot(normal);
    float numerator = (float)(((double)firstIndex * (double)firstIndex) * (1.0 - (double)dot * (double)dot));
    float root = (float)std::sqrt(1.0 - ((double)numerator / ((double)secondIndex * (double)secondIndex)));
    Vector3f t = (firstIndex / secondIndex) * (viewDirection - dot * normal) - normal * root;
    return Ray(point, t);
  }
  // src/Ray.h:43-57: the reflected direction rotated n times by 2 pi / n about the normal (Eigen::AngleAxisf's
  // rotation matrix, Eigen/src/Geometry/AngleAxis.h:218-243).
  std::vector<Ray> resendRay(Vector3f normal, Vector3f point, int n) {
    std::vector<Ray> res;
    float theta = (float)(M_PI * 2 / n);
    const float sn = std::sin(theta), c = std::cos(theta);
    const Vector3f sin_axis = sn * normal, cos1_axis = (1.0f - c) * normal;
    float m[9], tmp;
    tmp = cos1_axis.x() * normal.y(); m[1] = tmp - sin_axis.z(); m[3] = tmp + sin_axis.z();
    tmp = cos1_axis.x() * normal.z(); m[2] = tmp + sin_axis.y(); m[6] = tmp - sin_axis.y();
    tmp = cos1_axis.y() * normal.z(); m[5] = tmp - sin_axis.x(); m[7] = tmp + sin_axis.x();
    m[0] = cos1_axis.x() * normal.x() + c; m[4] = cos1_axis.y() * normal.y() + c; m[8] = cos1_axis.z() * normal.z() + c;
    Vector3f direction = reflectRay(normal, point).getDirection();
    for (int i = 0; i < n; i++) {
      direction = Vector3f(m[0] * direction.x() + (m[1] * direction.y() + m[2] * direction.z()),
                           m[3] * direction.x() + (m[4] * direction.y() + m[5] * direction.z()),
                           m[6] * direction.x() + (m[7] * direction.y() + m[8] * direction.z()));
      res.push_back(Ray(point, direction));
    }
    return res;
  }
};

// src/box.h:8-97 (value type; the GPU hierarchy lives inside librtx)
class Box {
 public:
  Vector3f tmax, tmin;
  std::vector<Tucano::Face> triangles;
  std::vector<Box> children;
  Box() {}
  Box(Vector3f tmin_, Vector3f tmax_) { tmin = tmin_; tmax = tmax_; }
  bool isInBox(Vector3f o) {
    return (o.x() >= tmin.x() && o.x() <= tmax.x()) && (o.y() >= tmin.y() && o.y() <= tmax.y()) && (o.z() >= tmin.z() && o.z() <= tmax.z());
  }
};

// src/Intersect.h:8-89: the primitive classes with their virtual intersects(Ray&), forwarded to the GPU library
// (rtx_intersect_primitive; nothing is intersected on the host).  `device` picks the GPU (default 0).
namespace Intersect {
class Face {
 protected:
  std::vector<Vector3f> vertices;
  Vector3f normal;
  const float precision = 0.001f;
  int material_id = 0;
  HitInfo run(int kind, const float* prim, Ray& ray) {
    HitInfo h{INFINITY, -1, Vector3f(), Vector3f()};
    Vector3f o = ray.getOrigin(), d = ray.getDirection();
    rtx_check(rtx_intersect_primitive(device, kind, prim, o.v, d.v, 1, &h.t, &h.faceId, h.normal.v, h.point.v));
    return h;
  }
 public:
  int device = 0;
  virtual HitInfo intersects(Ray& ray) = 0;
  Face(std::vector<Vector3f> _vertices, Vector3f _normal) { vertices = _vertices; normal = _normal; }
  Face() {}
  Face(Vector3f center) { vertices = std::vector<Vector3f>{center}; }
  void setMaterialID(int index) { material_id = index; }
  int getMaterialID() { return material_id; }
  virtual ~Face() {}
};
class Triangle : public Face {
 public:
  Triangle(std::vector<Vector3f> vertices, Vector3f normal) : Face(vertices, normal) {}
  HitInfo intersects(Ray& ray) override {
    if (vertices.size() < 3) throw RtxError(RTX_EINVAL, "Intersect::Triangle needs three vertices");
    float prim[12];
    for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) prim[3 * k + a] = vertices[k][a];
    for (int a = 0; a < 3; ++a) prim[9 + a] = normal[a];
    return run(RTX_PRIM_TRIANGLE, prim, ray);
  }
};
class Plane : public Face {
 public:
  Plane(std::vector<Vector3f> vertices, Vector3f normal) : Face(vertices, normal) {}
  HitInfo intersects(Ray& ray) override {
    if (vertices.empty()) throw RtxError(RTX_EINVAL, "Intersect::Plane needs a vertex");
    float prim[6] = {vertices[0][0], vertices[0][1], vertices[0][2], normal[0], normal[1], normal[2]};
    return run(RTX_PRIM_PLANE, prim, ray);
  }
};
class Sphere : public Face {
  float radius;
 public:
  Sphere(Vector3f center, float _radius) : Face(center) { radius = _radius; }
  HitInfo intersects(Ray& ray) override {
    float prim[4] = {vertices[0][0], vertices[0][1], vertices[0][2], radius};
    return run(RTX_PRIM_SPHERE, prim, ray);
  }
};
class BoxObject : public Face {
  Box& box;
 public:
  BoxObject(Box& _box) : Face(), box(_box) {}
  HitInfo intersects(Ray& ray) override {
    float prim[6] = {box.tmin[0], box.tmin[1], box.tmin[2], box.tmax[0], box.tmax[1], box.tmax[2]};
    return run(RTX_PRIM_BOX, prim, ray);
  }
};
}  // namespace Intersect

// src/Scene.h:18-78: the analytic sphere list
class Scene {
 public:
  struct SphereRec { Vector3f center; float radius; int material_id; };
  std::vector<SphereRec> spheres;
  static constexpr float RADIUS_CORRECTION = 3.0f;      // src/Scene.h:11
  Tucano::Material::Mtl addSphere(Vector3f center, float radius, Vector3f color) {
    spheres.push_back({center, radius / RADIUS_CORRECTION, -1});
    Tucano::Material::Mtl m;
    m.setAmbient(color / 3.0f); m.setDiffuse(color); m.setSpecular(Vector3f(0.5f, 0.5f, 0.5f));
    m.setIlluminationModel(2); m.setShininess(10); m.setDissolveFactor(0); m.setOpticalDensity(1);
    return m;
  }
  int getNumberOfSpheres() { return (int)spheres.size(); }
  SphereRec& getSphere(int i) { return spheres[i]; }
};

namespace detail {
// Three white-space separated plain decimal numbers ([+-]digits[.digits][e[+-]digits]) read with strtof, which is what
// `stream >> float` ends in for such text (libstdc++ collects the characters and calls strtof in the C locale) without
// the stream: a std::istringstream per line made the OBJ import 1.5 us per BYTE.  Fewer than three leave zeros, like failed
// extractions; false = the line holds something else and the caller lets a stream read it.
inline bool three_plain_floats(const char* p, float out[3]) {
  for (int k = 0; k < 3; ++k) {
    while (*p && std::isspace((unsigned char)*p)) ++p;
    if (!*p) return true;
    const char* q = p;
    if (*q == '+' || *q == '-') ++q;
    int digits = 0;
    while (*q >= '0' && *q <= '9') { ++q; ++digits; }
    if (*q == '.') { ++q; while (*q >= '0' && *q <= '9') { ++q; ++digits; } }
    if (!digits) return false;
    if (*q == 'e' || *q == 'E') {
      ++q;
      if (*q == '+' || *q == '-') ++q;
      if (!(*q >= '0' && *q <= '9')) return false;
      while (*q >= '0' && *q <= '9') ++q;
    }
    if (*q && !std::isspace((unsigned char)*q)) return false;
    char* end = nullptr;
    out[k] = std::strtof(p, &end);
    if (end != q || std::isinf(out[k])) return false;     // (beyond the float range the stream stores the largest float and fails)
    p = q;
  }
  return true;
}
}  // namespace detail

// ---- OBJ + MTL import with Tucano's face order (tucano/utils/objimporter.hpp:83-284,
// tucano/utils/mtlIO.hpp:45-125, tucano/mesh.hpp:441-468) and the P3 writer (tucano/utils/ppmIO.hpp:130-151)
struct MeshData {
  std::vector<float> verts;
  std::vector<uint32_t> idx;
  std::vector<int32_t> tri_mat;
};
inline bool loadMTL(std::vector<Tucano::Material::Mtl>& materials, const std::string& filename) {
  std::ifstream in(filename.c_str());
  if (!in) { std::cerr << "Cannot open " << filename << std::endl; return false; }
  for (std::string line; std::getline(in, line);) {
    if (line.empty()) continue;
    std::stringstream ss(line);
    std::vector<std::string> tok;
    for (std::string s; std::getline(ss, s, ' ');) tok.push_back(s);
    if (tok.empty()) continue;
    auto T = [&](size_t i) { return i < tok.size() ? tok[i].c_str() : ""; };
    if (tok[0] == "#") continue;
    if (tok[0] == "newmtl") { materials.emplace_back(); materials.back().setName(T(1)); }
    else if (materials.empty()) continue;
    else if (tok[0] == "Ns") materials.back().setShininess((float)atof(T(1)));
    else if (tok[0] == "Ka") materials.back().setAmbient(Vector3f((float)atof(T(1)), (float)atof(T(2)), (float)atof(T(3))));
    else if (tok[0] == "Kd") materials.back().setDiffuse(Vector3f((float)atof(T(1)), (float)atof(T(2)), (float)atof(T(3))));
    else if (tok[0] == "Ks") materials.back().setSpecular(Vector3f((float)atof(T(1)), (float)atof(T(2)), (float)atof(T(3))));
    else if (tok[0] == "Ni") materials.back().setOpticalDensity((float)atof(T(1)));
    else if (tok[0] == "d") materials.back().setDissolveFactor((float)atof(T(1)));
    else if (tok[0] == "illum") materials.back().setIlluminationModel(atoi(T(1)));
  }
  if (materials.empty()) materials.emplace_back();
  return true;
}
inline bool loadObjFile(MeshData& mesh, std::vector<Tucano::Material::Mtl>& mtls, const std::string& filename) {
  std::ifstream in(filename.c_str());
  if (!in) { std::cerr << "Cannot open " << filename << std::endl; return false; }   // the reference exit(1)s here
  std::string path;
  size_t slash = filename.find_last_of('/');
  if (slash != std::string::npos) path = filename.substr(0, slash + 1);
  std::vector<std::vector<uint32_t>> groups(1);
  std::vector<int> group_mat(1, -1);
  int current_mat = -1;
  for (std::string line; std::getline(in, line);) {
    if (line.compare(0, 6, "mtllib") == 0) {
      std::string fn = path + (line.size() > 7 ? line.substr(7) : "");
      std::string clean;
      for (char c : fn) if (c != '\n' && c != '\r') clean.push_back(c);
      loadMTL(mtls, clean);
    } else if (line.compare(0, 6, "usemtl") == 0) {
      if (!groups.back().empty()) { groups.emplace_back(); group_mat.push_back(-1); }
      std::string name = line.size() > 7 ? line.substr(7) : "";
      for (size_t i = 0; i < mtls.size(); ++i) if (mtls[i].getName() == name) current_mat = (int)i;
      group_mat.back() = current_mat;
    } else if (line.compare(0, 2, "v ") == 0) {
      float xyz[3] = {0, 0, 0};
      if (!detail::three_plain_floats(line.c_str() + 2, xyz)) {       // anything but plain decimal numbers: the stream's own reading
        std::istringstream s(line.substr(2));
        xyz[0] = xyz[1] = xyz[2] = 0;
        s >> xyz[0]; s >> xyz[1]; s >> xyz[2];
      }
      mesh.verts.push_back(xyz[0]); mesh.verts.push_back(xyz[1]); mesh.verts.push_back(xyz[2]);
    } else if (line.compare(0, 2, "f ") == 0) {
      // Tucano's importer pushes EVERY item of a face record onto the group's index stream and cuts the stream into
      // triples afterwards (objimporter.hpp:191-222, mesh.hpp:446-452): a record with four vertices is not a quad but
      // shifts every later triple of its group by one, one with two shifts them back.  Kept as it is -- face ids and
      // face order are the parity contract -- and pinned against the oracle's loader on such a file; feed
      // triangulated meshes.
      // (items are what `stream >> string` would cut at white space, each read with atoi: "12/3/4" is 12)
      for (const char* q = line.c_str() + 2; *q;) {
        while (*q && std::isspace((unsigned char)*q)) ++q;
        if (!*q) break;
        groups.back().push_back((uint32_t)(atoi(q) - 1));
        while (*q && !std::isspace((unsigned char)*q)) ++q;
      }
    }
  }
  for (size_t g = 0; g < groups.size(); ++g)
    for (size_t i = 0; i + 2 < groups[g].size(); i += 3) {
      mesh.idx.push_back(groups[g][i]); mesh.idx.push_back(groups[g][i + 1]); mesh.idx.push_back(groups[g][i + 2]);
      mesh.tri_mat.push_back(group_mat[g]);
    }
  return true;
}
// PLY reader (ascii and binary_little_endian 1.0; vertex x/y/z, any other vertex property skipped;
// face index lists of any integer type, polygons fan-triangulated in file order).  The reference's
// src/ never loads a PLY, but BASELINE config 2 is its bundled resources/models/bunny.ply
// (SURVEY.md 8(f) rank 1); faces come out in file order, which is what Tucano's face order is
// for an OBJ without groups.  All faces get material -1 (the default Mtl).
inline bool loadPlyFile(MeshData& mesh, const std::string& filename) {
  std::ifstream in(filename.c_str(), std::ios::binary);
  if (!in) { std::cerr << "Cannot open " << filename << std::endl; return false; }
  struct Prop { std::string name, type, count_type, item_type; bool list = false; };
  struct Elem { std::string name; size_t count = 0; std::vector<Prop> props; };
  std::vector<Elem> elems;
  bool binary = false, header_ok = false;
  std::string line;
  if (!std::getline(in, line) || line.compare(0, 3, "ply") != 0) { std::cerr << filename << ": not a PLY file" << std::endl; return false; }
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream ls(line);
    std::string key; ls >> key;
    if (key == "format") { std::string f; ls >> f; if (f == "binary_little_endian") binary = true; else if (f != "ascii") { std::cerr << filename << ": unsupported PLY format " << f << std::endl; return false; } }
    else if (key == "element") { Elem e; ls >> e.name >> e.count; elems.push_back(e); }
    else if (key == "property" && !elems.empty()) {
      Prop p; std::string t; ls >> t;
      if (t == "list") { p.list = true; ls >> p.count_type >> p.item_type >> p.name; } else { p.type = t; ls >> p.name; }
      elems.back().props.push_back(p);
    } else if (key == "end_header") { header_ok = true; break; }
  }
  if (!header_ok) { std::cerr << filename << ": truncated PLY header" << std::endl; return false; }
  auto size_of = [](const std::string& t) -> int {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
  };
  auto read_number = [&](const std::string& t, double* out) -> bool {
    if (!binary) { return (bool)(in >> *out); }
    unsigned char b[8];
    const int n = size_of(t);
    if (n == 0 || !in.read((char*)b, n)) return false;
    if (t == "float" || t == "float32") { float f; std::memcpy(&f, b, 4); *out = f; }
    else if (t == "double" || t == "float64") { double d; std::memcpy(&d, b, 8); *out = d; }
    else if (t == "char" || t == "int8") *out = (signed char)b[0];
    else if (t == "uchar" || t == "uint8") *out = b[0];
    else if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, b, 2); *out = v; }
    else if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, b, 2); *out = v; }
    else if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, b, 4); *out = v; }
    else { uint32_t v; std::memcpy(&v, b, 4); *out = v; }
    return true;
  };
  const size_t vert0 = mesh.verts.size() / 3;
  for (const Elem& e : elems) if (e.count > ((size_t)1 << 31)) { std::cerr << filename << ": implausible element count" << std::endl; return false; }
  for (const Elem& e : elems) {
    for (size_t k = 0; k < e.count; ++k) {
      float xyz[3] = {0, 0, 0};
      for (const Prop& p : e.props) {
        if (!p.list) {
          double v;
          if (!read_number(p.type, &v)) { std::cerr << filename << ": truncated PLY body" << std::endl; return false; }
          if (e.name == "vertex") { if (p.name == "x") xyz[0] = (float)v; else if (p.name == "y") xyz[1] = (float)v; else if (p.name == "z") xyz[2] = (float)v; }
        } else {
          double n;
          if (!read_number(p.count_type, &n)) { std::cerr << filename << ": truncated PLY body" << std::endl; return false; }
          if (!(n >= 0.0 && n <= 65536.0)) { std::cerr << filename << ": truncated PLY body" << std::endl; return false; }   // (a count no polygon has)
          std::vector<uint32_t> ids((size_t)n);
          for (size_t i = 0; i < ids.size(); ++i) {
            double v;
            if (!read_number(p.item_type, &v)) { std::cerr << filename << ": truncated PLY body" << std::endl; return false; }
            ids[i] = (uint32_t)((int64_t)v + (int64_t)vert0);
          }
          if (e.name == "face" && (p.name == "vertex_indices" || p.name == "vertex_index"))
            for (size_t i = 1; i + 1 < ids.size(); ++i) {
              mesh.idx.push_back(ids[0]); mesh.idx.push_back(ids[i]); mesh.idx.push_back(ids[i + 1]);
              mesh.tri_mat.push_back(-1);
            }
        }
      }
      if (e.name == "vertex") { mesh.verts.push_back(xyz[0]); mesh.verts.push_back(xyz[1]); mesh.verts.push_back(xyz[2]); }
    }
  }
  return true;
}
// tucano/utils/ppmIO.hpp:130-151: "P3", the size, 255, then per row the numbers min(255, (int)(255 * c)) of every pixel, each followed
// by a blank, and a newline.  The reference streams 25 million integers through operator<< for a 4K frame (2.6 s here, a thousand
// times the frame's render); this writes the SAME BYTES from a table of the 256 usual numbers, rows formatted by up to 16 threads
// into buffers of their own and written in order (4K: ~0.1 s).  `at(i, j)` returns the three floats of column i, row j.
namespace detail {
struct P3Table {
  char text[256][4]; unsigned char len[256];
  P3Table() {
    for (int v = 0; v < 256; ++v) { len[v] = (unsigned char)std::snprintf(text[v], 4, "%d", v); text[v][len[v]] = ' '; ++len[v]; }   // "N " (no terminator kept)
  }
};
inline char* p3_put(char* o, const P3Table& T, float c) {
  const int v = std::min(255, (int)(255 * c));           // the reference's expression (ppmIO.hpp:141-143)
  if (v >= 0) { std::memcpy(o, T.text[v], 4); return o + T.len[v]; }
  return o + std::sprintf(o, "%d ", v);                  // (a channel below zero prints as a negative number there too)
}
template <class At>
inline void write_p3(const std::string& filename, int width, int height, At at) {
  static const P3Table T;
  std::ofstream out(filename.c_str());
  out << "P3\n" << width << " " << height << "\n255\n";
  if (width <= 0 || height <= 0) return;
  const size_t row_max = (size_t)width * 3 * 12 + 1;     // "-2147483648 " is the longest entry
  const int rows_per_job = std::max(1, (int)((size_t)(8 << 20) / row_max));
  unsigned nthreads = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  // bands of rows, `nthreads` at a time: formatted side by side, written in order
  for (int j0 = 0; j0 < height; j0 += rows_per_job * (int)nthreads) {
    const int njobs = std::min<int>((int)nthreads, (height - j0 + rows_per_job - 1) / rows_per_job);
    std::vector<std::vector<char>> buf((size_t)njobs);
    std::vector<size_t> used((size_t)njobs, 0);
    auto job = [&](int k) {
      const int ja = j0 + k * rows_per_job, jb = std::min(height, ja + rows_per_job);
      buf[(size_t)k].resize((size_t)(jb - ja) * row_max);
      char* o = buf[(size_t)k].data();
      for (int j = ja; j < jb; ++j) {
        for (int i = 0; i < width; ++i) { const float* c = at(i, j); o = p3_put(o, T, c[0]); o = p3_put(o, T, c[1]); o = p3_put(o, T, c[2]); }
        *o++ = '\n';
      }
      used[(size_t)k] = (size_t)(o - buf[(size_t)k].data());
    };
    if (njobs == 1) job(0);
    else {
      std::vector<std::thread> th;
      for (int k = 0; k < njobs; ++k) th.emplace_back(job, k);
      for (std::thread& t : th) t.join();
    }
    for (int k = 0; k < njobs; ++k) out.write(buf[(size_t)k].data(), (std::streamsize)used[(size_t)k]);
  }
}
}  // namespace detail
// The writer with the frame's size given: data is indexed [x][y], width x height values are written row by row.
// Byte-identical to the reference's writer wherever that one is defined.
inline void writePPMImageXY(const std::string& filename, const std::vector<std::vector<Vector3f>>& data, int width, int height) {
  if ((int)data.size() < width) throw RtxError(RTX_EINVAL, "writePPMImage: the frame buffer has fewer than `width` columns");
  for (int i = 0; i < width; ++i)
    if ((int)data[i].size() < height) throw RtxError(RTX_EINVAL, "writePPMImage: a frame buffer column has fewer than `height` entries");
  detail::write_p3(filename, width, height, [&](int i, int j) { return data[(size_t)i][(size_t)j].v; });
}
// The same file from a row-major float frame (3 floats per pixel, row j at 3 * j * width): what raytraceScene holds after a render.
inline void writePPMImageRowMajor(const std::string& filename, const float* rgb, int width, int height) {
  detail::write_p3(filename, width, height, [&](int i, int j) { return rgb + 3 * ((size_t)j * (size_t)width + (size_t)i); });
}
// The reference's signature: width = data[0].size(), height = data.size(), and data[i][j] with i < width, j < height
// -- inside the buffer only when it is square.  Anything else is refused instead of read out of bounds.
inline void writePPMImage(const std::string& filename, const std::vector<std::vector<Vector3f>>& data) {
  if (data.empty()) throw RtxError(RTX_EINVAL, "writePPMImage: empty frame buffer");
  const int width = (int)data[0].size(), height = (int)data.size();
  if (width != height)
    throw RtxError(RTX_EINVAL, "writePPMImage(data): the reference indexes data[x][y] in a [height][width] buffer, which is only inside it for "
                               "square frames; use writePPMImageXY(filename, data, width, height) with a buffer indexable [x][y]");
  writePPMImageXY(filename, data, width, height);
}

// ---- Flyscene (src/flyscene.hpp:30-221), ray-tracing half
//
// Threading.  The reference calls its queries from 16 std::threads at once (raytraceScene spawns them over
// updating_pixels, src/flyscene.cpp:417-430) and mutates the scene only between renders (src/main.cpp:63-77).  The C
// ABI underneath wants the calls on one rtx_scene serialised (include/rtx.h), and this object caches a frame, so every
// public method takes the object's mutex (recursive: Shader calls getIntersections, the debug ray calls canSeeLight):
// sixteen threads running the reference's own updating_pixels(pixel_data, origin, size, 16, tid) render ONE frame --
// the first to arrive launches it, the others find it cached -- and each copies its own rows.  Per-ray queries from
// many threads are correct but serialised, one GPU round trip each; the vector overloads (getIntersections(rays),
// traceRay(bounce, rays, inside), canSeeLight(light, positions)) answer a whole batch in one launch and are what a
// caller with more than a handful of rays should use.
class Flyscene {
 public:
  std::string model_path = "resources/models/planeobj.obj";     // src/flyscene.cpp:22-23
  int device = 0;
  // the reference's compile-time switches (src/flyscene.cpp:11,704): {MAX_BOUNCES 2, reflect, refract, 8 GI, shadows}
  rtx_options options{2, 1, 1, 8, 1, 0};
  Box root_box;                                                  // src/flyscene.hpp:219

  Flyscene() {}
  ~Flyscene() { drop_replicas(); if (tri_scene) rtx_scene_destroy(tri_scene); if (scene) rtx_scene_destroy(scene); }
  Flyscene(const Flyscene&) = delete;
  Flyscene& operator=(const Flyscene&) = delete;

  // src/flyscene.cpp:13-120 without the GL parts
  void initialize(int width, int height) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    flycamera.setPerspectiveMatrix(60.0f, width / (float)height, 0.1f, 100.0f);
    flycamera.setViewport(Vector2f((float)width, (float)height));
    mesh = MeshData(); materials.clear();
    if (!loadObjFile(mesh, materials, model_path)) throw RtxError(RTX_EINVAL, "cannot open " + model_path);
    mesh_materials = materials.size();
    lights.clear(); sphericalLights.clear();
    lights.push_back(Vector3f(-1.0f, 1.0f, 1.0f));             // :48
    spheres_added = false; currentScene = Scene();
    rebuild();
  }
  Tucano::Flycamera* getCamera() { return &flycamera; }
  void addLight() { std::lock_guard<std::recursive_mutex> guard(mu); lights.push_back(getCenter()); dirty = true; }                       // src/flyscene.hpp:61
  void addSphericalLight() { std::lock_guard<std::recursive_mutex> guard(mu); sphericalLights.push_back({getCenter(), 0.01f}); dirty = true; }   // :62
  int getNumberOfFaces() const { return (int)mesh.tri_mat.size(); }

  // src/flyscene.cpp:395-439.  The frame buffer is indexed [x][y] as the reference does (:465); the
  // reference allocates it [height][width], which holds that indexing only for square frames and is out of bounds
  // otherwise.  Here it is allocated [width][height]: the same object for square frames (result.ppm byte-identical),
  // and a defined result for every other size.
  void raytraceScene(int width = 0, int height = 0) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    std::cout << "ray tracing ..." << std::endl;
    auto t0 = std::chrono::steady_clock::now();
    Vector2i image_size(width, height);
    if (width == 0 || height == 0) image_size = flycamera.getViewportSize();
    if (image_size[0] <= 0 || image_size[1] <= 0) throw RtxError(RTX_EINVAL, "raytraceScene: bad image size");
    // One GPU frame instead of 16 racing threads.  The reference's pixel_data is a local that only feeds the
    // writer (src/flyscene.cpp:410-432): the file is written straight from the rendered frame, same bytes, without the
    // 100 MB [x][y] copy of a 4K frame.
    render_frame(image_size[0], image_size[1]);
    writePPMImageRowMajor("result.ppm", frame.data(), image_size[0], image_size[1]);
    auto ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "Time elapsed: " << (long)ms << " ms" << std::endl << "ray tracing done! " << std::endl;
  }

  // A mesh that moves.  No counterpart in the reference, which loads its mesh once (src/flyscene.cpp:22-23) and builds
  // Box(mesh) / divideBox_KD once (:77-82): new positions for the loaded vertices (same count, same faces); the
  // hierarchy is refitted on the GPU (rtx_scene_update_vertices; rebuilt when `rebuild` is set or a face degenerates),
  // the leaf list of the GI probes is split again from the new vertices as initialize would.
  const std::vector<float>& getVertices() const { return mesh.verts; }
  void updateVertices(const std::vector<float>& xyz, bool rebuild = false) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    if (xyz.size() != mesh.verts.size()) throw RtxError(RTX_EINVAL, "updateVertices: the mesh has another number of vertices");
    sync();
    drop_replicas();
    mesh.verts = xyz;
    rtx_check(rtx_scene_update_vertices(scene, mesh.verts.data(), (uint32_t)(mesh.verts.size() / 3), rebuild ? RTX_UPDATE_REBUILD : RTX_UPDATE_REFIT));
    boxes.clear();
    if (options.gi_rays > 0) {
      if (rtx_scene_build_reference_leafboxes(scene, 0, 0) < 0) { std::cerr << "warning: " << rtx_last_error() << "; global illumination probes disabled" << std::endl; options.gi_rays = 0; }
      else fetch_boxes();
    }
    dirty = true;
    sync();
  }

  // How many scene handles render a frame (rtx_render_multi: part i of the tile grid on handle i, all at once):
  // 0 = one per visible GPU, the multi-GPU form of the reference's 16 threads (src/flyscene.cpp:417-430); 1 = a
  // single rtx_render.  Handles beyond the number of GPUs share devices (device = index modulo GPUs).
  int render_parts = 0;
  int tile_size = 32;

  // src/flyscene.cpp:441-468.  Rows thread_id, thread_id + number_threads, ... of the frame are
  // written to pixel_data[i][j] (the reference's [x][y] indexing): the buffer must be
  // indexable that way -- at least image_size[0] columns of at least image_size[1] entries --
  // or the call is refused (the reference writes out of bounds instead).  The spheres are
  // added once, not once per thread.
  void updating_pixels(std::vector<std::vector<Vector3f>>& pixel_data, Vector3f& /*origin*/, Vector2i& image_size,
                       int number_threads, int thread_id) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    const int W = image_size[0], H = image_size[1];
    if (W <= 0 || H <= 0 || number_threads <= 0 || thread_id < 0) throw RtxError(RTX_EINVAL, "updating_pixels: bad image size or thread index");
    if ((int)pixel_data.size() < W) throw RtxError(RTX_EINVAL, "updating_pixels: pixel_data[x][y] needs image_size[0] columns (SURVEY F8)");
    for (int i = 0; i < W; ++i)
      if ((int)pixel_data[i].size() < H) throw RtxError(RTX_EINVAL, "updating_pixels: pixel_data[x][y] needs image_size[1] entries per column (SURVEY F8)");
    render_frame(W, H);
    for (int j = thread_id; j < H; j += number_threads)
      for (int i = 0; i < W; ++i) {
        const float* p = &frame[3 * ((size_t)j * W + i)];
        pixel_data[i][j] = Vector3f(p[0], p[1], p[2]);
      }
  }
  const float* frame_data() const { return frame.data(); }
  // The frame of the current camera at W x H in `frame` (row-major floats), rendered unless it is there already.
  void render_frame(int W, int H) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    if (!spheres_added) add_reference_spheres();
    sync();
    rtx_camera cam = flycamera.raw();
    // (the camera is reachable through getCamera(): a frame cached for another view is not this call's frame)
    if (frame_w != W || frame_h != H || frame_dirty || std::memcmp(&cam, &frame_cam, sizeof cam) != 0) {
      frame.resize((size_t)W * H * 3);
      rtx_rect r{0, 0, W, H};
      int ndev = rtx_device_count();
      int parts = render_parts > 0 ? render_parts : ndev;
      if (parts > 1 && ndev >= 1) {
        ensure_replicas(parts, ndev);
        std::vector<rtx_scene*> hs(1, scene);
        hs.insert(hs.end(), replicas.begin(), replicas.end());
        rtx_check(rtx_render_multi(hs.data(), parts, &cam, &options, r, tile_size, frame.data(), nullptr, nullptr));
      } else {
        rtx_check(rtx_render(scene, &cam, &options, r, frame.data(), nullptr, nullptr));
      }
      frame_w = W; frame_h = H; frame_dirty = false; frame_cam = cam; ++frames_rendered;
    }
  }

  // src/flyscene.cpp:587-601
  Vector3f traceRay(int bounce, Ray ray, bool insideObject) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    sync();
    Vector3f o = ray.getOrigin(), d = ray.getDirection(), c;
    rtx_check(rtx_trace_shade(scene, &options, o.v, d.v, 1, bounce, insideObject ? 1 : 0, c.v));
    return c;
  }
  // src/flyscene.cpp:603-629: the colour of an already found hit, summed over the lights.  The reference only ever calls it
  // with the (hit, face) that getIntersections(ray) has just returned (:588-594), and for that pair it is traceRay's value:
  // this forwards to traceRay, and refuses a hit that is not the ray's own closest hit (calculateColor, its per-light term,
  // takes intermediate state of that sum and has no entry point of its own: call traceRay / Shader).
  Vector3f Shader(int bounce, Tucano::Face /*face*/, HitInfo hit, Ray ray, bool insideObject) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    const std::pair<HitInfo, Tucano::Face> own = getIntersections(ray);
    if (own.first.faceId != hit.faceId || std::memcmp(&own.first.t, &hit.t, 4) != 0)
      throw RtxError(RTX_EINVAL, "rtx: Shader() takes the hit getIntersections() returns for the same ray");
    return traceRay(bounce, ray, insideObject);
  }
  // src/flyscene.cpp:517-585
  std::pair<HitInfo, Tucano::Face> getIntersections(Ray ray) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    sync();
    Vector3f o = ray.getOrigin(), d = ray.getDirection();
    HitInfo h; int mat = -1;
    rtx_check(rtx_trace_closest(scene, o.v, d.v, 1, &h.t, &h.faceId, h.normal.v, h.point.v, &mat));
    Tucano::Face f;
    if (h.faceId >= 0) f = make_face(h.faceId);
    f.material_id = mat;
    return std::make_pair(h, f);
  }
  // src/flyscene.cpp:819-879: closest triangle of a face list; faceId indexes that list.
  // The scene built for a face list is kept until another list (or other vertices) comes: the reference's callers ask
  // ray after ray against the same leaf (src/flyscene.cpp:553-577).
  HitInfo intersectTriangle(std::vector<Tucano::Face>& faces, Vector3f origin, Vector3f dir) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    std::vector<uint32_t> idx; idx.reserve(faces.size() * 3);
    for (auto& f : faces) for (int k = 0; k < 3; ++k) idx.push_back(f.vertex_ids[k]);
    if (!tri_scene || idx != tri_idx || mesh.verts != tri_verts) {
      if (tri_scene) { rtx_scene_destroy(tri_scene); tri_scene = nullptr; }
      rtx_scene* tmp = nullptr;
      rtx_check(rtx_scene_create(mesh.verts.data(), (uint32_t)(mesh.verts.size() / 3), idx.data(), nullptr, (uint32_t)faces.size(),
                                 nullptr, 0, &tmp));
      int rc = rtx_scene_set_flags(tmp, RTX_FLAG_NO_ROOT_GATE);
      if (rc >= 0) rc = rtx_scene_commit(tmp, device);
      if (rc < 0) { std::string err = rtx_last_error(); rtx_scene_destroy(tmp); throw RtxError(rc, "rtx: " + err); }
      tri_scene = tmp; tri_idx = idx; tri_verts = mesh.verts;
    }
    HitInfo h{INFINITY, -1, Vector3f(), Vector3f()};
    rtx_check(rtx_trace_closest(tri_scene, origin.v, dir.v, 1, &h.t, &h.faceId, h.normal.v, h.point.v, nullptr));
    return h;
  }
  // src/flyscene.cpp:906-972
  HitInfo intersectBox(Box& box, Vector3f origin, Vector3f dest) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    sync();
    HitInfo h{INFINITY, -1, Vector3f(), Vector3f()};
    rtx_check(rtx_intersect_box(scene, box.tmin.v, box.tmax.v, origin.v, dest.v, 1, &h.t));
    return h;
  }
  // src/flyscene.cpp:747-782
  std::pair<bool, Tucano::Material::Mtl> canSeeLight(Vector3f lightPos, Vector3f position) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    sync();
    uint8_t vis = 0; int bm = -1;
    rtx_check(rtx_trace_visible(scene, lightPos.v, position.v, 1, &vis, &bm));
    Tucano::Material::Mtl m;
    if (bm >= 0 && bm < (int)materials.size()) m = materials[bm];
    return std::make_pair(vis != 0, m);
  }
  // ---- the same queries for many rays in ONE launch (no counterpart in the reference, whose callers loop; results are
  // element for element those of the single-ray methods above)
  std::vector<std::pair<HitInfo, Tucano::Face>> getIntersections(const std::vector<Ray>& rays) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    sync();
    const size_t n = rays.size();
    std::vector<std::pair<HitInfo, Tucano::Face>> out(n);
    if (n == 0) return out;
    std::vector<float> o(3 * n), d(3 * n), t(n), nrm(3 * n), pt(3 * n);
    std::vector<int32_t> face(n), mat(n);
    flatten(rays, o, d);
    rtx_check(rtx_trace_closest(scene, o.data(), d.data(), (uint32_t)n, t.data(), face.data(), nrm.data(), pt.data(), mat.data()));
    for (size_t i = 0; i < n; ++i) {
      HitInfo h{t[i], face[i], Vector3f(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), Vector3f(pt[3 * i], pt[3 * i + 1], pt[3 * i + 2])};
      Tucano::Face f;
      if (face[i] >= 0) f = make_face(face[i]);
      f.material_id = mat[i];
      out[i] = std::make_pair(h, f);
    }
    return out;
  }
  std::vector<Vector3f> traceRay(int bounce, const std::vector<Ray>& rays, bool insideObject) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    sync();
    const size_t n = rays.size();
    std::vector<Vector3f> out(n);
    if (n == 0) return out;
    std::vector<float> o(3 * n), d(3 * n), c(3 * n);
    flatten(rays, o, d);
    rtx_check(rtx_trace_shade(scene, &options, o.data(), d.data(), (uint32_t)n, bounce, insideObject ? 1 : 0, c.data()));
    for (size_t i = 0; i < n; ++i) out[i] = Vector3f(c[3 * i], c[3 * i + 1], c[3 * i + 2]);
    return out;
  }
  std::vector<std::pair<bool, Tucano::Material::Mtl>> canSeeLight(Vector3f lightPos, const std::vector<Vector3f>& positions) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    sync();
    const size_t n = positions.size();
    std::vector<std::pair<bool, Tucano::Material::Mtl>> out(n);
    if (n == 0) return out;
    std::vector<float> L(3 * n), P(3 * n);
    std::vector<uint8_t> vis(n);
    std::vector<int32_t> bm(n);
    for (size_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { L[3 * i + k] = lightPos[k]; P[3 * i + k] = positions[i][k]; }
    rtx_check(rtx_trace_visible(scene, L.data(), P.data(), (uint32_t)n, vis.data(), bm.data()));
    for (size_t i = 0; i < n; ++i) {
      Tucano::Material::Mtl m;
      if (bm[i] >= 0 && bm[i] < (int)materials.size()) m = materials[bm[i]];
      out[i] = std::make_pair(vis[i] != 0, m);
    }
    return out;
  }
  // src/flyscene.cpp:631-669: (Kd, t) of the closest hit over the spheres and the leaf list `boxes`; zeros on a miss
  Vector4f trace_global_illum(Ray& ray) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    sync();
    Vector3f o = ray.getOrigin(), d = ray.getDirection();
    Vector4f r;
    rtx_check(rtx_trace_gi(scene, o.v, d.v, 1, r.v));
    return r;
  }
  // src/flyscene.cpp:1110-1185: the flat leaf list `boxes` (the hierarchy itself is never attached to root_box in the
  // reference either).  Throws where the reference's split would never terminate.
  std::vector<Box> boxes;
  void divideBox_KD(int max_numberFaces) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    int leaves = rtx_scene_build_reference_leafboxes(scene, max_numberFaces, 0);
    if (leaves < 0) throw RtxError(leaves, std::string("rtx: ") + rtx_last_error());
    dirty = true;
    fetch_boxes();
    std::cout << "Number of leafs: " << boxes.size() << std::endl;            // :1184
  }
  Vector3f getCenter() {
    std::lock_guard<std::recursive_mutex> guard(mu);
    sync();
    rtx_camera cam = flycamera.raw();
    Vector3f c;
    rtx_check(rtx_camera_points(scene, &cam, nullptr, 0, nullptr, c.v));
    return c;
  }
  Vector3f screenToWorld(const Vector2f& p) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    sync();
    rtx_camera cam = flycamera.raw();
    Vector3f w;
    rtx_check(rtx_camera_points(scene, &cam, p.v, 1, w.v, nullptr));
    return w;
  }
  // ---- debug ray (src/flyscene.cpp:288-393), as data instead of Tucano cylinders: `rays` = the
  // polyline (white segments end on a surface, the last one is 10 long when it left the scene),
  // `ray_hitbox` = indices of the leaf boxes the ray entered from outside, `light_rays` = segments
  // from hit points to the lights that see them.  `jumps` = number of reflections followed
  // (src/flyscene.hpp:124, keys in src/flyscene.cpp:238-249).
  struct DebugSegment { Vector3f origin, dir; float length; bool hit; int faceId; };
  std::vector<DebugSegment> rays, light_rays;
  std::vector<int> ray_hitbox;
  int jumps = 0;
  void createDebugRay(const Vector2f& mouse_pos) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    rays.clear(); ray_hitbox.clear(); light_rays.clear();
    Vector3f screen_pos = screenToWorld(mouse_pos);
    Vector3f dir = (screen_pos - getCenter()).normalized();
    ReflectDebugRay(getCenter(), dir, 0, jumps);
  }
  void ReflectDebugRay(Vector3f origin, Vector3f dir, int bounce, int jumps_) {
    std::lock_guard<std::recursive_mutex> guard(mu);
    sync();
    uint32_t nl = 0, nf = 0;
    rtx_check(rtx_scene_get_leafboxes(scene, nullptr, nullptr, nullptr, nullptr, nullptr, &nl, &nf));
    std::vector<uint8_t> flags(nl ? nl : 1);
    for (; bounce <= jumps_; ++bounce) {
      Ray r(origin, dir);
      HitInfo h;
      std::fill(flags.begin(), flags.end(), 0);
      rtx_check(rtx_debug_hit(scene, origin.v, dir.v, 1, &h.t, &h.faceId, h.normal.v, h.point.v, flags.data()));
      for (uint32_t i = 0; i < nl; ++i)
        if (flags[i]) ray_hitbox.push_back((int)i);
      if (h.t == INFINITY) { rays.push_back({origin, dir, 10.0f, false, -1}); return; }
      rays.push_back({origin, dir, h.t, true, h.faceId});
      for (size_t i = 0; i < lights.size(); ++i)
        if (canSeeLight(lights[i], h.point).first)
          light_rays.push_back({h.point, (lights[i] - h.point).normalized(), (lights[i] - h.point).norm(), true, -1});
      Ray nr = r.reflectRay(h.normal, h.point);
      origin = nr.getOrigin(); dir = nr.getDirection();
    }
  }
  rtx_stats lastStats() { std::lock_guard<std::recursive_mutex> guard(mu); rtx_get_stats(scene, &stats); return stats; }
  void markCameraChanged() { std::lock_guard<std::recursive_mutex> guard(mu); frame_dirty = true; }
  int framesRendered() { std::lock_guard<std::recursive_mutex> guard(mu); return frames_rendered; }   // GPU frames launched by updating_pixels so far

  std::vector<Vector3f> lights;
  std::vector<std::pair<Vector3f, float>> sphericalLights;
  std::vector<Tucano::Material::Mtl> materials;
  Scene currentScene;

  // src/flyscene.hpp:177-206
  std::vector<Vector4f> spherePositions{
      {-0.5f * 0.5f, 0.25f * 0.5f, 0.8f * 0.5f, 0.75f * 0.5f}, {0.75f * 0.5f, 0.25f * 0.5f, 0.7f * 0.5f, 0.75f * 0.5f},
      {0.6f * 0.5f, 0.25f * 0.5f, -0.4f * 0.5f, 0.75f * 0.5f}, {-0.45f * 0.5f, 0.25f * 0.5f, -0.4f * 0.5f, 0.75f * 0.5f},
      {0.0f * 0.5f, 0.167f * 0.5f, 1.2f * 0.5f, 0.5f * 0.5f}, {1.0f * 0.5f, 0.167f * 0.5f, 1.2f * 0.5f, 0.5f * 0.5f},
      {-1.0f * 0.5f, 0.25f * 0.5f, 1.2f * 0.5f, 0.75f * 0.5f}, {0.0f * 0.5f, 0.33f * 0.5f, -2.2f * 0.5f, 1.0f * 0.5f},
      {0.0f * 0.5f, 0.33f * 0.5f, 2.2f * 0.5f, 1.0f * 0.5f}, {-3.0f * 0.5f, 0.167f * 0.5f, -2.2f * 0.5f, 0.5f * 0.5f},
      {3.0f * 0.5f, 0.25f * 0.5f, -2.2f * 0.5f, 0.75f * 0.5f}, {3.5f * 0.5f, 0.33f * 0.5f, 0.0f * 0.5f, 1.0f * 0.5f},
      {-3.5f * 0.5f, 0.167f * 0.5f, 0.0f * 0.5f, 0.5f * 0.5f}};
  std::vector<Vector3f> sphereColors{
      {0.8f, 0.2401f, 0.0f}, {0.5f, 0.5f, 0}, {(float)0.98824, (float)0.72941, (float)0.01176}, {0.0f, 0.5f, 0.5f},
      {1.0f, 0.0f, 1.0f}, {1.0f, 0.75f, 1.0f}, {0.5f, 1.0f, 0.75f}, {0.4f, 0.3f, 1.0f}, {0.8f, 0.667f, 0.0f},
      {1.0f, 0.75f, 1.0f}, {0.0f, 0.7f, 1.0f}, {1.0f, 0.5f, 0.3f}, {0.0f, 0.3f, 0.123f}};

 private:
  std::recursive_mutex mu;                   // every public method holds it (see "Threading" above)
  int frames_rendered = 0;
  static void flatten(const std::vector<Ray>& rays, std::vector<float>& o, std::vector<float>& d) {
    for (size_t i = 0; i < rays.size(); ++i) {
      Ray r = rays[i];
      const Vector3f ro = r.getOrigin(), rd = r.getDirection();
      for (int k = 0; k < 3; ++k) { o[3 * i + k] = ro[k]; d[3 * i + k] = rd[k]; }
    }
  }
  Tucano::Flycamera flycamera;
  MeshData mesh;
  size_t mesh_materials = 0;
  rtx_scene* scene = nullptr;
  rtx_scene* tri_scene = nullptr;            // intersectTriangle's face list, committed (kept between calls)
  std::vector<uint32_t> tri_idx;
  std::vector<float> tri_verts;
  bool dirty = true, spheres_added = false, frame_dirty = true;
  // The rendered frame in page-locked memory (rtx_host_alloc): rtx_render's read-back then runs at the PCIe rate (a 4K frame
  // in ~3 ms instead of ~10 through the runtime's staging copies).  Plain memory where the library has none to give.
  struct FrameBuffer {
    float* p = nullptr; size_t cap = 0, n = 0; bool pinned = false;
    FrameBuffer() {}
    FrameBuffer(const FrameBuffer&) = delete;
    FrameBuffer& operator=(const FrameBuffer&) = delete;
    ~FrameBuffer() { release(); }
    void release() { if (p) { if (pinned) rtx_host_free(p); else std::free(p); } p = nullptr; cap = n = 0; }
    void resize(size_t count) {
      if (count > cap) {
        release();
        void* q = nullptr;
        if (rtx_host_alloc((uint64_t)count * sizeof(float), &q) == RTX_OK && q) { p = (float*)q; pinned = true; }
        else { p = (float*)std::malloc(std::max<size_t>(count, 1) * sizeof(float)); pinned = false; if (!p) throw std::bad_alloc(); }
        cap = count;
      }
      n = count;
    }
    float* data() { return p; }
    const float* data() const { return p; }
    float& operator[](size_t i) { return p[i]; }
    const float& operator[](size_t i) const { return p[i]; }
  };
  FrameBuffer frame;
  int frame_w = 0, frame_h = 0;
  rtx_camera frame_cam{};
  rtx_stats stats{};

  std::vector<rtx_scene*> replicas;        // handles 1 .. parts-1 of rtx_render_multi (clones of `scene`, committed on their devices)
  void drop_replicas() { for (rtx_scene* r : replicas) rtx_scene_destroy(r); replicas.clear(); }
  void ensure_replicas(int parts, int ndev) {
    if ((int)replicas.size() == parts - 1) return;
    drop_replicas();
    for (int i = 1; i < parts; ++i) {
      rtx_scene* r = nullptr;
      rtx_check(rtx_scene_clone(scene, &r));
      replicas.push_back(r);
      rtx_check(rtx_scene_commit(r, (device + i) % ndev));
    }
  }
  void fetch_boxes() {
    boxes.clear();
    uint32_t nl = 0, nf = 0;
    if (rtx_scene_get_leafboxes(scene, nullptr, nullptr, nullptr, nullptr, nullptr, &nl, &nf) < 0) return;
    std::vector<float> lo(3 * (size_t)nl + 1), hi(3 * (size_t)nl + 1);
    std::vector<uint32_t> st(nl + 1), ct(nl + 1);
    std::vector<int32_t> fa(nf + 1);
    rtx_check(rtx_scene_get_leafboxes(scene, lo.data(), hi.data(), st.data(), ct.data(), fa.data(), nullptr, nullptr));
    for (uint32_t i = 0; i < nl; ++i) {
      Box b(Vector3f(lo[3 * i], lo[3 * i + 1], lo[3 * i + 2]), Vector3f(hi[3 * i], hi[3 * i + 1], hi[3 * i + 2]));
      for (uint32_t k = 0; k < ct[i]; ++k) b.triangles.push_back(make_face(fa[st[i] + k]));
      boxes.push_back(b);
    }
  }
  Tucano::Face make_face(int f) {
    Tucano::Face face;
    for (int k = 0; k < 3; ++k) face.vertex_ids.push_back(mesh.idx[3 * (size_t)f + k]);
    face.material_id = mesh.tri_mat[f];
    return face;
  }
  void rebuild() {
    drop_replicas();
    if (scene) { rtx_scene_destroy(scene); scene = nullptr; }
    std::vector<rtx_material> raw;
    for (size_t i = 0; i < mesh_materials; ++i) raw.push_back(materials[i].raw());
    rtx_check(rtx_scene_create(mesh.verts.data(), (uint32_t)(mesh.verts.size() / 3), mesh.idx.data(), mesh.tri_mat.data(),
                               (uint32_t)mesh.tri_mat.size(), raw.data(), (uint32_t)raw.size(), &scene));
    pushed_materials = mesh_materials;
    // Box(mesh) + divideBox_KD(sqrt(F + 100)) (src/flyscene.cpp:77-82): the leaf list the GI probes walk
    int leaves = rtx_scene_build_reference_leafboxes(scene, 0, 0);
    if (leaves < 0) { std::cerr << "warning: " << rtx_last_error() << "; global illumination probes disabled" << std::endl; options.gi_rays = 0; }
    else fetch_boxes();
    root_box.triangles.clear();
    for (int f = 0; f < (int)mesh.tri_mat.size(); ++f) root_box.triangles.push_back(make_face(f));
    dirty = true;
    sync();
    rtx_scene_info info;
    rtx_check(rtx_scene_get_info(scene, &info));
    std::cout << "Number of leafs: " << leaves << std::endl;
  }
  // src/flyscene.cpp:452-458, once
  void add_reference_spheres() {
    if (spherePositions.size() != sphereColors.size()) abort();
    for (size_t i = 0; i < spherePositions.size(); i++) {
      auto tempMat = currentScene.addSphere(spherePositions[i].head3(), spherePositions[i].w(), sphereColors[i]);
      int mat_id = (int)materials.size();
      materials.push_back(tempMat);
      currentScene.getSphere((int)i).material_id = mat_id;
    }
    spheres_added = true; dirty = true;
  }
  // push lights / spheres / appended materials and (re)commit when something changed
  void sync() {
    if (!dirty) return;
    drop_replicas();                       // clones of the scene as it was
    rtx_scene_info info{};
    std::vector<float> L, SL, S;
    std::vector<int32_t> SM;
    for (auto& l : lights) { L.push_back(l[0]); L.push_back(l[1]); L.push_back(l[2]); }
    for (auto& s : sphericalLights) { SL.push_back(s.first[0]); SL.push_back(s.first[1]); SL.push_back(s.first[2]); SL.push_back(s.second); }
    for (auto& s : currentScene.spheres) { S.push_back(s.center[0]); S.push_back(s.center[1]); S.push_back(s.center[2]); S.push_back(s.radius); SM.push_back(s.material_id); }
    rtx_check(rtx_scene_set_lights(scene, L.data(), (uint32_t)lights.size(), SL.data(), (uint32_t)sphericalLights.size()));
    rtx_check(rtx_scene_set_spheres(scene, S.data(), SM.data(), (uint32_t)SM.size()));
    rtx_check(rtx_scene_build(scene));
    rtx_check(rtx_scene_get_info(scene, &info));
    // materials appended after the MTL ones (sphere materials, src/flyscene.cpp:455-456)
    while (pushed_materials < materials.size()) {
      rtx_material m = materials[pushed_materials].raw();
      rtx_check(rtx_scene_add_material(scene, &m));
      ++pushed_materials;
    }
    rtx_check(rtx_scene_commit(scene, device));
    dirty = false; frame_dirty = true;
  }
  size_t pushed_materials = 0;
};

}  // namespace rtxhost
