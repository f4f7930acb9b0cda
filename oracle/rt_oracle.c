/* oracle/rt_oracle.c -- TEST INFRASTRUCTURE ONLY (see rt_oracle.h for the contract).
 *
 * Plain-C restatement of the reference hot path.  Every function cites the reference
 * lines it follows (paths relative to /root/reference).  All arithmetic is IEEE fp32 in
 * the evaluation order Eigen 3.3.7 produces for Vector3f (non-vectorised; reductions in
 * tree order a0 + (a1 + a2), Eigen/src/Core/Redux.h:91-103), doubles only where the C++
 * source promotes (literals such as -0.5, pow(float,int) under libstdc++, M_PI).
 * Build with -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef ora_v3 v3;

/* ---------------------------------------------------------------- Eigen-order vector math */
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float* p) { return V3(p[0], p[1], p[2]); }
static inline void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline v3 scale3(float s, v3 a) { return V3(s * a.x, s * a.y, s * a.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 div3(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
/* Eigen dot = (a.cwiseProduct(b)).sum(), redux tree for 3 = c0 + (c1 + c2) */
static inline float dot3(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
/* Eigen/src/Core/Dot.h:124-134 */
static inline v3 normalized3(v3 a) {
  float z = dot3(a, a);
  if (z > 0.0f) return div3(a, sqrtf(z));
  return a;
}
static inline float norm3(v3 a) { return sqrtf(dot3(a, a)); }
/* Eigen/src/Geometry/OrthoMethods.h cross3 */
static inline v3 cross3(v3 a, v3 b) {
  return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* Matrix3f * Vector3f, coefficient based: row.cwiseProduct(v).sum() */
static inline v3 mat3_mul(const float m[9], v3 v) { /* m row-major */
  return V3(m[0] * v.x + (m[1] * v.y + m[2] * v.z), m[3] * v.x + (m[4] * v.y + m[5] * v.z),
            m[6] * v.x + (m[7] * v.y + m[8] * v.z));
}
/* Eigen/src/Geometry/AngleAxis.h:218-243 toRotationMatrix (row-major out) */
static void angle_axis_matrix(float angle, v3 axis, float m[9]) {
  v3 sin_axis = scale3(sinf(angle), axis);
  float c = cosf(angle);
  v3 cos1_axis = scale3(1.0f - c, axis);
  float tmp;
  tmp = cos1_axis.x * axis.y;
  m[1] = tmp - sin_axis.z;
  m[3] = tmp + sin_axis.z;
  tmp = cos1_axis.x * axis.z;
  m[2] = tmp + sin_axis.y;
  m[6] = tmp - sin_axis.y;
  tmp = cos1_axis.y * axis.z;
  m[5] = tmp - sin_axis.x;
  m[7] = tmp + sin_axis.x;
  m[0] = cos1_axis.x * axis.x + c;
  m[4] = cos1_axis.y * axis.y + c;
  m[8] = cos1_axis.z * axis.z + c;
}

/* ---------------------------------------------------------------- Ray (src/Ray.h) */
typedef struct { v3 o, d; } ray_t;
/* src/Ray.h:15-18 */
static inline ray_t make_ray(v3 o, v3 d) { ray_t r; r.o = o; r.d = normalized3(d); return r; }
/* src/Ray.h:28-30 */
static inline v3 point_on_ray(const ray_t* r, float t) {
  return V3(r->o.x + t * r->d.x, r->o.y + t * r->d.y, r->o.z + t * r->d.z);
}
/* src/Ray.h:32-35 */
static inline ray_t reflect_ray(const ray_t* r, v3 n, v3 p) {
  float s = 2.0f * dot3(r->d, n);
  return make_ray(p, sub3(r->d, scale3(s, n)));
}
/* src/Ray.h:36-42 (pow(float,int) and sqrt evaluate in double under libstdc++) */
static inline ray_t refract_ray(v3 n, v3 p, v3 view, float n1, float n2) {
  float dot = dot3(view, n);
  float numerator = (float)(((double)n1 * (double)n1) * (1.0 - (double)dot * (double)dot));
  float root = (float)sqrt(1.0 - ((double)numerator / ((double)n2 * (double)n2)));
  float s = n1 / n2;
  v3 a = scale3(s, sub3(view, scale3(dot, n)));
  v3 b = V3(n.x * root, n.y * root, n.z * root);
  return make_ray(p, sub3(a, b));
}
/* src/Ray.h:43-57 */
static void resend_dirs(const ray_t* r, v3 n, v3 p, int count, v3* out) {
  float theta = (float)(M_PI * 2 / count);
  float m[9];
  angle_axis_matrix(theta, n, m);
  v3 dir = reflect_ray(r, n, p).d;
  for (int i = 0; i < count; ++i) {
    dir = mat3_mul(m, dir);
    out[i] = normalized3(dir);
  }
}
/* src/ShaderLib.h:7-9 */
static inline v3 shader_reflect(v3 light, v3 normal) {
  float s = 2.0f * dot3(normal, light);
  return sub3(light, scale3(s, normal));
}
/* src/ShaderLib.h:11-31 */
static int points_on_circle(v3 center, float radius, v3 normal, int n, v3* out) {
  v3 not_normal = V3(1.0f, 0.0f, 0.0f);
  v3 mn = neg3(normal);
  if ((not_normal.x == normal.x && not_normal.y == normal.y && not_normal.z == normal.z) ||
      (not_normal.x == mn.x && not_normal.y == mn.y && not_normal.z == mn.z))
    not_normal = normalized3(V3(1.0f, 1.0f, 0.0f));
  v3 rv = normalized3(cross3(normal, not_normal));
  rv = V3(rv.x * radius, rv.y * radius, rv.z * radius);
  float theta = (float)(M_PI * 2 / n);
  float m[9];
  angle_axis_matrix(theta, normal, m);
  int k = 0;
  out[k++] = add3(rv, center);
  for (int i = 0; i < n - 1; ++i) {
    rv = mat3_mul(m, rv);
    out[k++] = add3(rv, center);
  }
  return k;
}

/* ---------------------------------------------------------------- primitives */
typedef struct { float t; v3 n; v3 p; } phit;
static const float PRECISION = 0.001f; /* src/Intersect.h:14 */
static inline phit phit_miss(void) { phit h; h.t = INFINITY; h.n = V3(0, 0, 0); h.p = V3(0, 0, 0); return h; }

/* src/Intersect.cpp:24-46 */
static inline int in_triangle(v3 hit, v3 v0, v3 v1, v3 v2) {
  v3 u = sub3(v1, v0), v = sub3(v2, v0), w = sub3(hit, v0);
  float uu = dot3(u, u), uv = dot3(u, v), vv = dot3(v, v);
  float wu = dot3(w, u), wv = dot3(w, v);
  float denom = uu * vv - uv * uv;
  float s1 = vv * wu - uv * wv;
  float t1 = uu * wv - uv * wu;
  float s = s1 / denom;
  float t = t1 / denom;
  if (s < 0 || t < 0 || s + t > 1) return 0;
  return 1;
}
/* src/Intersect.cpp:6-23 */
static inline phit triangle_hit(const ray_t* r, v3 v0, v3 v1, v3 v2, v3 n) {
  phit res = phit_miss();
  float D = dot3(n, v0);
  float den = dot3(n, r->d);
  if (den != 0.0f) {
    float t = (D - dot3(r->o, n)) / den;
    v3 P = point_on_ray(r, t);
    if (t >= PRECISION && in_triangle(P, v0, v1, v2)) { res.t = t; res.n = n; res.p = P; }
  }
  return res;
}
/* src/Intersect.cpp:48-64 */
static inline phit plane_hit(const ray_t* r, v3 v0, v3 n) {
  phit res = phit_miss();
  float D = dot3(n, v0);
  float den = dot3(n, r->d);
  if (den != 0.0f) {
    float t = (D - dot3(r->o, n)) / den;
    v3 P = point_on_ray(r, t);
    if (t >= PRECISION) { res.t = t; res.n = n; res.p = P; }
  }
  return res;
}
/* src/Intersect.cpp:67-145 */
static inline float box_hit(const ray_t* r, v3 bmin, v3 bmax) {
  v3 dir = r->d, o = r->o;
  float ix = 1 / dir.x, iy = 1 / dir.y, iz = 1 / dir.z;
  float txmin, txmax, tymin, tymax, tzmin, tzmax;
  if (ix >= 0) { txmin = (bmin.x - o.x) * ix; txmax = (bmax.x - o.x) * ix; }
  else { txmin = (bmax.x - o.x) * ix; txmax = (bmin.x - o.x) * ix; }
  if (iy >= 0) { tymin = (bmin.y - o.y) * iy; tymax = (bmax.y - o.y) * iy; }
  else { tymin = (bmax.y - o.y) * iy; tymax = (bmin.y - o.y) * iy; }
  if ((txmin > tymax) || (tymin > txmax)) return INFINITY;
  if (tymin > txmin) txmin = tymin;
  if (tymax < txmax) txmax = tymax;
  if (iz >= 0) { tzmin = (bmin.z - o.z) * iz; tzmax = (bmax.z - o.z) * iz; }
  else { tzmin = (bmax.z - o.z) * iz; tzmax = (bmin.z - o.z) * iz; }
  if ((txmin > tzmax) || (tzmin > txmax)) return INFINITY;
  if (tzmin > txmin) txmin = tzmin;
  if (tzmax < txmax) txmax = tzmax;
  return txmin;
}
/* src/Intersect.cpp:178-193 */
static inline int solve_quadratic(float a, float b, float c, float* x0, float* x1) {
  float discr = b * b - 4.0f * a * c;
  if (discr < 0) return 0;
  else if (discr == 0) *x0 = *x1 = (float)(-0.5 * (double)b / (double)a);
  else {
    float q = (b > 0) ? (float)(-0.5 * (double)(b + sqrtf(discr))) : (float)(-0.5 * (double)(b - sqrtf(discr)));
    *x0 = q / a;
    *x1 = c / q;
  }
  if (*x0 > *x1) { float tmp = *x0; *x0 = *x1; *x1 = tmp; }
  return 1;
}
/* src/Intersect.cpp:147-176 */
static inline phit sphere_hit(const ray_t* r, v3 c, float radius) {
  phit res = phit_miss();
  float t0, t1;
  float radius2 = radius * radius;
  v3 L = sub3(r->o, c);
  float a = dot3(r->d, r->d);
  float b = 2.0f * dot3(r->d, L);
  float cc = dot3(L, L) - radius2;
  if (solve_quadratic(a, b, cc, &t0, &t1)) {
    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 < 0) t0 = t1;
    if (t0 > 0 && t0 >= PRECISION) {
      res.t = t0;
      res.p = point_on_ray(r, t0);
      res.n = normalized3(sub3(res.p, c));
    }
  }
  return res;
}

/* ---------------------------------------------------------------- camera */
/* Eigen/src/LU/InverseImpl.h:124-170 (3x3 cofactor inverse), m and out row-major */
static void mat3_inverse(const float m[9], float out[9]) {
#define M(i, j) m[3 * (i) + (j)]
#define COF(i, j) (M(((i) + 1) % 3, ((j) + 1) % 3) * M(((i) + 2) % 3, ((j) + 2) % 3) - M(((i) + 1) % 3, ((j) + 2) % 3) * M(((i) + 2) % 3, ((j) + 1) % 3))
  float c0 = COF(0, 0), c1 = COF(1, 0), c2 = COF(2, 0);
  float det = c0 * M(0, 0) + (c1 * M(1, 0) + c2 * M(2, 0));
  float invdet = 1.0f / det;
  out[0] = c0 * invdet; out[1] = c1 * invdet; out[2] = c2 * invdet;
  out[3] = COF(0, 1) * invdet; out[4] = COF(1, 1) * invdet; out[5] = COF(2, 1) * invdet;
  out[6] = COF(0, 2) * invdet; out[7] = COF(1, 2) * invdet; out[8] = COF(2, 2) * invdet;
#undef COF
#undef M
}
/* tucano/camera.hpp:115-118 */
static v3 camera_center(const ora_camera* cam) {
  float lin[9], inv[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) lin[3 * r + c] = cam->view[4 * r + c];
  mat3_inverse(lin, inv);
  v3 mt = V3(-cam->view[3], -cam->view[7], -cam->view[11]);
  return mat3_mul(inv, mt);
}
/* Eigen/src/Geometry/Transform.h:1202-1229: inverse of the Affine3f view, as 3x4 row-major */
static void view_inverse(const ora_camera* cam, float inv34[12]) {
  float lin[9], inv[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) lin[3 * r + c] = cam->view[4 * r + c];
  mat3_inverse(lin, inv);
  v3 t = V3(cam->view[3], cam->view[7], cam->view[11]);
  /* translation = (-inv) * t, coefficient based */
  float ninv[9];
  for (int k = 0; k < 9; ++k) ninv[k] = -inv[k];
  v3 it = mat3_mul(ninv, t);
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) inv34[4 * r + c] = inv[3 * r + c]; }
  inv34[3] = it.x; inv34[7] = it.y; inv34[11] = it.z;
}
/* tucano/camera.hpp:155-173, :263-266.  The 4x4 * (v,1) product is Eigen's packet path:
 * ((c0*v0 + c1*v1) + c2*v2) + c3*1 per row (Transform.h:1384-1392). */
static v3 screen_to_world(const ora_camera* cam, const float inv34[12], float rx, float ry) {
  float nx = (float)(2.0 * (double)(rx - cam->vp[0]) / (double)cam->vp[2] - 1.0);
  float ny = (float)(1.0 - 2.0 * (double)(ry - cam->vp[1]) / (double)cam->vp[3]);
  float nz = -1.0f;
  float pscale = (float)((double)1.0f / tan((double)(cam->fovy / 2.0f) * (M_PI / (double)180.0f)));
  float scale = (float)(1.0 / (double)pscale);
  nx *= cam->aspect * scale;
  ny *= scale;
  v3 w;
  w.x = ((inv34[0] * nx + inv34[1] * ny) + inv34[2] * nz) + inv34[3] * 1.0f;
  w.y = ((inv34[4] * nx + inv34[5] * ny) + inv34[6] * nz) + inv34[7] * 1.0f;
  w.z = ((inv34[8] * nx + inv34[9] * ny) + inv34[10] * nz) + inv34[11] * 1.0f;
  return w;
}

/* ---------------------------------------------------------------- scene */
typedef struct { v3 bmin, bmax; int start, count; } leaf_t;

struct ora_scene {
  int nverts, ntris, nmats_mesh, nmats;
  float* verts;
  uint32_t* idx;
  int32_t* tri_mat;
  v3 *V0, *V1, *V2; /* world-space vertices per face (identity model matrix) */
  v3* nface;        /* Tucano::Mesh::createFaces normal (tucano/mesh.hpp:461-463) */
  v3* nquery;       /* re-normalised per query (src/flyscene.cpp:839) */
  ora_mtl* mats;
  int nspheres; float* sph; int32_t* sph_mat;
  int nlights; float* lights;
  int nsl; float* slights;
  v3 root_min, root_max; /* Box(Mesh) (src/box.h:41-85) */
  int nleaves; leaf_t* leaves; int32_t* leaf_faces; int leaf_faces_n;
  ora_options opt;
  uint64_t counts[5];
  pthread_mutex_t mu;
};

static const ora_mtl DEFAULT_MTL = {{0.3f, 0.3f, 0.3f}, {0.5f, 0.5f, 0.5f}, {1.0f, 1.0f, 1.0f}, 10.0f, 0.0f, 1.0f, 0};

static v3 face_normal(v3 a, v3 b, v3 c) {
  v3 v1 = normalized3(sub3(c, a));
  v3 v0 = normalized3(sub3(b, a));
  return normalized3(cross3(v0, v1));
}

ora_scene* ora_scene_create(const float* verts, int nverts, const uint32_t* idx, const int32_t* tri_mat,
                            int ntris, const ora_mtl* mats, int nmats) {
  ora_scene* s = (ora_scene*)calloc(1, sizeof(ora_scene));
  s->nverts = nverts; s->ntris = ntris; s->nmats_mesh = nmats; s->nmats = nmats;
  s->verts = (float*)malloc(sizeof(float) * 3 * (nverts > 0 ? nverts : 1));
  memcpy(s->verts, verts, sizeof(float) * 3 * nverts);
  s->idx = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (ntris > 0 ? ntris : 1));
  memcpy(s->idx, idx, sizeof(uint32_t) * 3 * ntris);
  s->tri_mat = (int32_t*)malloc(sizeof(int32_t) * (ntris > 0 ? ntris : 1));
  for (int i = 0; i < ntris; ++i) s->tri_mat[i] = tri_mat ? tri_mat[i] : -1;
  s->mats = (ora_mtl*)malloc(sizeof(ora_mtl) * (nmats > 0 ? nmats : 1));
  if (nmats) memcpy(s->mats, mats, sizeof(ora_mtl) * nmats);
  size_t nb = sizeof(v3) * (ntris > 0 ? ntris : 1);
  s->V0 = (v3*)malloc(nb); s->V1 = (v3*)malloc(nb); s->V2 = (v3*)malloc(nb);
  s->nface = (v3*)malloc(nb); s->nquery = (v3*)malloc(nb);
  for (int f = 0; f < ntris; ++f) {
    v3 a = ld3(verts + 3 * idx[3 * f]), b = ld3(verts + 3 * idx[3 * f + 1]), c = ld3(verts + 3 * idx[3 * f + 2]);
    s->V0[f] = a; s->V1[f] = b; s->V2[f] = c;
    s->nface[f] = face_normal(a, b, c);
    s->nquery[f] = normalized3(s->nface[f]);
  }
  /* Box(Tucano::Mesh) (src/box.h:41-85): max seeded with numeric_limits<float>::min() */
  float mx = 1.17549435e-38f, my = mx, mz = mx;
  float nx = 3.40282347e+38f, ny = nx, nz = nx;
  for (int i = 0; i < nverts; ++i) {
    float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
    if (x > mx) mx = x;
    if (x < nx) nx = x;
    if (y > my) my = y;
    if (y < ny) ny = y;
    if (z > mz) mz = z;
    if (z < nz) nz = z;
  }
  s->root_max = V3(mx, my, mz); s->root_min = V3(nx, ny, nz);
  ora_options o = {2, 1, 1, 8, 1, 0};
  s->opt = o;
  pthread_mutex_init(&s->mu, NULL);
  return s;
}
void ora_scene_destroy(ora_scene* s) {
  if (!s) return;
  free(s->verts); free(s->idx); free(s->tri_mat); free(s->mats); free(s->V0); free(s->V1); free(s->V2);
  free(s->nface); free(s->nquery); free(s->sph); free(s->sph_mat); free(s->lights); free(s->slights);
  free(s->leaves); free(s->leaf_faces);
  pthread_mutex_destroy(&s->mu);
  free(s);
}
void ora_scene_set_spheres(ora_scene* s, const float* cr4, const int32_t* mat, int n) {
  free(s->sph); free(s->sph_mat);
  s->nspheres = n;
  s->sph = (float*)malloc(sizeof(float) * 4 * (n > 0 ? n : 1));
  s->sph_mat = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
  if (n) { memcpy(s->sph, cr4, sizeof(float) * 4 * n); memcpy(s->sph_mat, mat, sizeof(int32_t) * n); }
}
void ora_scene_set_lights(ora_scene* s, const float* xyz, int n, const float* sxyzr, int m) {
  free(s->lights); free(s->slights);
  s->nlights = n; s->nsl = m;
  s->lights = (float*)malloc(sizeof(float) * 3 * (n > 0 ? n : 1));
  s->slights = (float*)malloc(sizeof(float) * 4 * (m > 0 ? m : 1));
  if (n) memcpy(s->lights, xyz, sizeof(float) * 3 * n);
  if (m) memcpy(s->slights, sxyzr, sizeof(float) * 4 * m);
}
void ora_scene_set_options(ora_scene* s, const ora_options* o) { s->opt = *o; }
void ora_scene_root_box(const ora_scene* s, float* bmin, float* bmax) { st3(bmin, s->root_min); st3(bmax, s->root_max); }
void ora_scene_face_normal(const ora_scene* s, int f, float* nf, float* nq) { st3(nf, s->nface[f]); st3(nq, s->nquery[f]); }
/* materials may be appended after creation (sphere materials, src/flyscene.cpp:454-457) */
int ora_scene_add_material(ora_scene* s, const ora_mtl* m) {
  s->mats = (ora_mtl*)realloc(s->mats, sizeof(ora_mtl) * (s->nmats + 1));
  s->mats[s->nmats] = *m;
  return s->nmats++;
}

/* ------------------------------------------------- leaf boxes: Flyscene::divideBox_KD */
typedef struct { v3 bmin, bmax; int32_t* faces; int n; } qbox;

/* src/flyscene.cpp:1070-1082 */
static int axis_to_divide(v3 tmax, v3 tmin) {
  float max = 1.17549435e-38f;
  int result = -1;
  float d[3] = {tmax.x - tmin.x, tmax.y - tmin.y, tmax.z - tmin.z};
  for (int i = 0; i < 3; ++i) if (max <= d[i]) { max = d[i]; result = i; }
  return result;
}
/* src/flyscene.cpp:1084-1108 */
static int face_in_box(const ora_scene* s, v3 tmin, v3 tmax, int f) {
  v3 vs[3] = {s->V0[f], s->V1[f], s->V2[f]};
  for (int i = 0; i < 3; ++i) {
    v3 v = vs[i];
    if (tmax.x >= v.x && tmin.x <= v.x && tmax.y >= v.y && tmin.y <= v.y && tmax.z >= v.z && tmin.z <= v.z) return 1;
  }
  return 0;
}
int ora_scene_build_leafboxes(ora_scene* s, int max_faces, long iter_cap) {
  free(s->leaves); free(s->leaf_faces);
  s->leaves = NULL; s->leaf_faces = NULL; s->nleaves = 0; s->leaf_faces_n = 0;
  size_t qcap = 1024, qhead = 0, qtail = 0;
  qbox* q = (qbox*)malloc(sizeof(qbox) * qcap);
  qbox root; root.bmin = s->root_min; root.bmax = s->root_max; root.n = s->ntris;
  root.faces = (int32_t*)malloc(sizeof(int32_t) * (s->ntris > 0 ? s->ntris : 1));
  for (int i = 0; i < s->ntris; ++i) root.faces[i] = i;
  q[qtail++] = root;
  int lcap = 64, fcap = 1024;
  s->leaves = (leaf_t*)malloc(sizeof(leaf_t) * lcap);
  s->leaf_faces = (int32_t*)malloc(sizeof(int32_t) * fcap);
  long iters = 0;
  int failed = 0;
  while (qhead < qtail) {
    if (++iters > iter_cap) { failed = 1; break; }
    qbox box = q[qhead++];
    if (box.n <= max_faces && box.n > 0) { /* src/flyscene.cpp:1119-1125 */
      if (s->nleaves == lcap) { lcap *= 2; s->leaves = (leaf_t*)realloc(s->leaves, sizeof(leaf_t) * lcap); }
      while (s->leaf_faces_n + box.n > fcap) { fcap *= 2; s->leaf_faces = (int32_t*)realloc(s->leaf_faces, sizeof(int32_t) * fcap); }
      leaf_t L; L.bmin = box.bmin; L.bmax = box.bmax; L.start = s->leaf_faces_n; L.count = box.n;
      memcpy(s->leaf_faces + s->leaf_faces_n, box.faces, sizeof(int32_t) * box.n);
      s->leaf_faces_n += box.n;
      s->leaves[s->nleaves++] = L;
      free(box.faces);
      continue;
    }
    int axis = axis_to_divide(box.bmax, box.bmin);
    /* averagePoint (src/flyscene.cpp:974-993) */
    float sx = 0, sy = 0, sz = 0;
    for (int i = 0; i < box.n; ++i) {
      int f = box.faces[i];
      v3 vs[3] = {s->V0[f], s->V1[f], s->V2[f]};
      for (int k = 0; k < 3; ++k) { sx += vs[k].x; sy += vs[k].y; sz += vs[k].z; }
    }
    float size = (float)((size_t)box.n * 3);
    v3 avg = V3(sx / size, sy / size, sz / size);
    v3 midMax, midMin;
    if (axis == 0) { midMax = V3(avg.x, box.bmax.y, box.bmax.z); midMin = V3(avg.x, box.bmin.y, box.bmin.z); }
    else if (axis == 1) { midMax = V3(box.bmax.x, avg.y, box.bmax.z); midMin = V3(box.bmin.x, avg.y, box.bmin.z); }
    else { midMax = V3(box.bmax.x, box.bmax.y, avg.z); midMin = V3(box.bmin.x, box.bmin.y, avg.z); }
    qbox b1, b2;
    b1.bmin = box.bmin; b1.bmax = midMax; b2.bmin = midMin; b2.bmax = box.bmax;
    b1.faces = (int32_t*)malloc(sizeof(int32_t) * (box.n > 0 ? box.n : 1));
    b2.faces = (int32_t*)malloc(sizeof(int32_t) * (box.n > 0 ? box.n : 1));
    b1.n = b2.n = 0;
    for (int i = 0; i < box.n; ++i) {
      int f = box.faces[i];
      if (face_in_box(s, b1.bmin, b1.bmax, f)) b1.faces[b1.n++] = f;
      if (face_in_box(s, b2.bmin, b2.bmax, f)) b2.faces[b2.n++] = f;
    }
    free(box.faces);
    if (qtail + 2 > qcap) {
      if (qhead > 0) { memmove(q, q + qhead, sizeof(qbox) * (qtail - qhead)); qtail -= qhead; qhead = 0; }
      if (qtail + 2 > qcap) { qcap *= 2; q = (qbox*)realloc(q, sizeof(qbox) * qcap); }
    }
    q[qtail++] = b1;
    q[qtail++] = b2;
  }
  for (size_t i = qhead; i < qtail; ++i) free(q[i].faces);
  free(q);
  if (failed) { s->nleaves = 0; s->leaf_faces_n = 0; return -1; }
  return s->nleaves;
}
int ora_scene_num_leaves(const ora_scene* s) { return s->nleaves; }
int ora_scene_leaf(const ora_scene* s, int i, float* bmin, float* bmax, int32_t* faces, int cap) {
  const leaf_t* L = &s->leaves[i];
  st3(bmin, L->bmin); st3(bmax, L->bmax);
  for (int k = 0; k < L->count && k < cap; ++k) faces[k] = s->leaf_faces[L->start + k];
  return L->count;
}

/* ---------------------------------------------------------------- queries */
typedef struct { uint64_t c[5]; } counters; /* primary, shadow, reflection, refraction, gi */
enum { C_PRIMARY = 0, C_SHADOW = 1, C_REFLECT = 2, C_REFRACT = 3, C_GI = 4 };

/* Flyscene::intersectTriangle over a face list (src/flyscene.cpp:819-879).  `faces`==NULL
 * means the root list 0..ntris-1.  Returns the index into the list (or -1). */
static int intersect_triangles(const ora_scene* s, const int32_t* faces, int n, v3 origin, v3 dir, phit* out) {
  phit small = phit_miss();
  int small_i = -1;
  ray_t r = make_ray(origin, dir); /* src/flyscene.cpp:845 (re-normalises) */
  for (int i = 0; i < n; ++i) {
    int f = faces ? faces[i] : i;
    phit h = triangle_hit(&r, s->V0[f], s->V1[f], s->V2[f], s->nquery[f]);
    if (h.t < small.t) { small = h; small_i = i; }
  }
  *out = small;
  return small_i;
}
/* Flyscene::intersectBox (src/flyscene.cpp:906-972) */
static float intersect_box(v3 bmin, v3 bmax, v3 origin, v3 dir) {
  ray_t r = make_ray(origin, dir);
  return box_hit(&r, bmin, bmax);
}
static inline ora_hit hit_miss(void) {
  ora_hit h; h.t = INFINITY; h.face = -1; h.n[0] = h.n[1] = h.n[2] = 0; h.p[0] = h.p[1] = h.p[2] = 0; h.mat = -1;
  return h;
}
/* Flyscene::getIntersections, KD branch with the child-less root (src/flyscene.cpp:517-585) */
static ora_hit get_intersections(const ora_scene* s, const ray_t* ray) {
  ora_hit best = hit_miss();
  float smallestT = INFINITY;
  for (int i = 0; i < s->nspheres; ++i) {
    phit h = sphere_hit(ray, ld3(s->sph + 4 * i), s->sph[4 * i + 3]);
    if (h.t != INFINITY && h.t < smallestT) {
      smallestT = h.t;
      best.t = h.t; best.face = -1; st3(best.n, h.n); st3(best.p, h.p); best.mat = s->sph_mat[i];
    }
  }
  float bt = intersect_box(s->root_min, s->root_max, ray->o, ray->d);
  if (bt != INFINITY) {
    phit h;
    int i = intersect_triangles(s, NULL, s->ntris, ray->o, ray->d, &h);
    if (h.t != INFINITY && smallestT > h.t) {
      smallestT = h.t;
      best.t = h.t; best.face = i; st3(best.n, h.n); st3(best.p, h.p); best.mat = s->tri_mat[i];
    }
  }
  return best;
}
static inline const ora_mtl* material(const ora_scene* s, int id) {
  /* materials[-1] is undefined behaviour in the reference; the default Mtl is used. */
  if (id < 0 || id >= s->nmats) return &DEFAULT_MTL;
  return &s->mats[id];
}
/* Flyscene::canSeeLight (src/flyscene.cpp:747-782) */
static int can_see_light(const ora_scene* s, v3 light, v3 pos, int* blocker_mat, ora_hit* blocker, counters* cn) {
  v3 toLight = sub3(light, pos);
  ray_t ray = make_ray(pos, normalized3(toLight));
  float len = norm3(toLight);
  int behind = 0, sphereBlock = 0;
  for (int i = 0; i < s->nspheres; ++i) {
    phit h = sphere_hit(&ray, ld3(s->sph + 4 * i), s->sph[4 * i + 3]);
    if (h.t != INFINITY && h.t > len) behind = 1;
    else if (h.t != INFINITY) sphereBlock = 1;
  }
  ora_hit th = get_intersections(s, &ray);
  if (cn) cn->c[C_SHADOW]++;
  *blocker_mat = -1;
  if (th.t != INFINITY && th.t > len) behind = 1;
  if (th.t != INFINITY && th.t < len) *blocker_mat = th.mat < 0 ? -2 : th.mat; /* -2: materials[-1] UB */
  if (blocker) *blocker = th;
  return ((!sphereBlock && th.t == INFINITY) || behind) ? 1 : 0;
}
/* Flyscene::trace_global_illum (src/flyscene.cpp:631-669): rgb = Kd, w = t (0 = nothing hit) */
static void trace_gi(const ora_scene* s, const ray_t* ray, float out[4]) {
  float smallestT = INFINITY;
  int mat = -1;
  for (int i = 0; i < s->nspheres; ++i) {
    phit h = sphere_hit(ray, ld3(s->sph + 4 * i), s->sph[4 * i + 3]);
    if (h.t != INFINITY && h.t < smallestT) { smallestT = h.t; mat = s->sph_mat[i]; }
  }
  for (int b = 0; b < s->nleaves; ++b) {
    const leaf_t* L = &s->leaves[b];
    float bt = intersect_box(L->bmin, L->bmax, ray->o, ray->d);
    if (bt != INFINITY) {
      phit h;
      int i = intersect_triangles(s, s->leaf_faces + L->start, L->count, ray->o, ray->d, &h);
      if (h.t != INFINITY && smallestT > h.t) { smallestT = h.t; mat = s->tri_mat[s->leaf_faces[L->start + i]]; }
    }
  }
  if (smallestT != INFINITY) {
    const ora_mtl* m = material(s, mat);
    out[0] = m->kd[0]; out[1] = m->kd[1]; out[2] = m->kd[2]; out[3] = smallestT;
  } else { out[0] = out[1] = out[2] = out[3] = 0.0f; }
}

static v3 trace_ray(const ora_scene* s, int bounce, const ray_t* ray, int inside, counters* cn);

/* Flyscene::calculateColor (src/flyscene.cpp:671-745) */
static v3 calculate_color(const ora_scene* s, int bounce, v3 lightPos, const ora_hit* hit, v3 normalN,
                          const ray_t* ray, const ora_mtl* mat, int inside, counters* cn) {
  int numLights = s->nlights + s->nsl;
  float I = 0.8f / (float)numLights;
  v3 P = ld3(hit->p), HN = ld3(hit->n);
  v3 lightDir = normalized3(sub3(lightPos, P));
  v3 reflectedLight = shader_reflect(neg3(lightDir), normalN);
  v3 eyeDir = normalized3(sub3(ray->o, P));
  float dotted = dot3(eyeDir, normalized3(reflectedLight));
  v3 reflected = V3(0, 0, 0), refracted = V3(0, 0, 0);
  float wr = 1.0f - mat->d, wt = mat->d;
  if (bounce < s->opt.max_bounces) {
    if (s->opt.reflection && !(s->opt.skip_zero_weight && wr == 0.0f)) {
      ray_t rr = reflect_ray(ray, HN, P);
      cn->c[C_REFLECT]++;
      reflected = trace_ray(s, bounce + 1, &rr, inside, cn);
    }
    if (s->opt.refraction && !(s->opt.skip_zero_weight && wt == 0.0f)) {
      ray_t rf;
      if (inside) rf = refract_ray(neg3(HN), P, neg3(eyeDir), mat->ni, 1.0f);
      else rf = refract_ray(HN, P, neg3(eyeDir), 1.0f, mat->ni);
      cn->c[C_REFRACT]++;
      refracted = trace_ray(s, bounce + 1, &rf, !inside, cn);
    }
  }
  v3 gi = V3(0, 0, 0);
  int G = s->opt.gi_rays;
  if (G > 0) {
    v3 dirs[64];
    if (G > 64) G = 64;
    resend_dirs(ray, HN, P, G, dirs);
    float cols[64][4];
    for (int i = 0; i < G; ++i) {
      ray_t gr; gr.o = P; gr.d = dirs[i];
      cn->c[C_GI]++;
      trace_gi(s, &gr, cols[i]);
    }
    for (int i = 0; i < G; ++i) {
      if (cols[i][3] != 0.0f) {
        float distance = cols[i][3];
        /* min(1 / pow(50*distance, 2), 1.0f): double under libstdc++ */
        float fd = 50.0f * distance;
        double w = 1.0 / ((double)fd * (double)fd);
        float wf = (w < 1.0) ? (float)w : 1.0f;
        gi = add3(gi, V3(cols[i][0] * wf, cols[i][1] * wf, cols[i][2] * wf));
      }
    }
    gi = div3(gi, (float)G);
  }
  int visible = 1, bmat = -1;
  if (s->opt.shadows) visible = can_see_light(s, lightPos, P, &bmat, NULL, cn);
  v3 Iv = V3(I, I, I);
  v3 ka = ld3(mat->ka), kd = ld3(mat->kd), ks = ld3(mat->ks);
  v3 ambient = mul3(ka, Iv);
  float ndl = dot3(normalN, lightDir);
  float md = (ndl < 0.0f) ? 0.0f : ndl; /* std::max(a, b) = (a < b) ? b : a */
  v3 diffuse = scale3(md, mul3(kd, Iv));
  float mdot = (dotted < 0.0f) ? 0.0f : dotted;
  float sp = (float)pow((double)mdot, 33.0);
  v3 isk = mul3(Iv, ks);
  v3 specular = V3(isk.x * sp, isk.y * sp, isk.z * sp);
  v3 color = add3(add3(ambient, diffuse), specular);
  v3 rterm = scale3(wr, reflected), tterm = scale3(wt, refracted);
  if (!visible) {
    const ora_mtl* bm = (bmat == -1) ? &DEFAULT_MTL : material(s, bmat);
    v3 ambientS = mul3(ld3(bm->ka), Iv);
    v3 diffuseS = scale3(md, mul3(ld3(bm->kd), Iv));
    float spS = powf(mdot, bm->ns);
    v3 iskS = mul3(Iv, ld3(bm->ks));
    v3 specularS = V3(iskS.x * spS, iskS.y * spS, iskS.z * spS);
    v3 colorS = add3(add3(ambientS, diffuseS), specularS);
    v3 base = add3(add3(mul3(ka, Iv), rterm), tterm);
    return add3(base, V3(colorS.x * bm->d, colorS.y * bm->d, colorS.z * bm->d));
  }
  return add3(add3(add3(color, gi), rterm), tterm);
}
/* Flyscene::Shader (src/flyscene.cpp:603-629) */
static v3 shade(const ora_scene* s, int bounce, const ora_hit* hit, const ray_t* ray, int inside, counters* cn) {
  const ora_mtl* mat = material(s, hit->mat);
  v3 normalN = ld3(hit->n);
  v3 total = V3(0, 0, 0);
  for (int l = 0; l < s->nlights; ++l)
    total = add3(total, calculate_color(s, bounce, ld3(s->lights + 3 * l), hit, normalN, ray, mat, inside, cn));
  for (int l = 0; l < s->nsl; ++l) {
    v3 c = ld3(s->slights + 4 * l);
    v3 normal = sub3(ld3(hit->p), c);
    v3 pts[8];
    int n = points_on_circle(c, s->slights[4 * l + 3], normalized3(normal), 4, pts);
    pts[n++] = c;
    for (int k = 0; k < n; ++k)
      total = add3(total, div3(calculate_color(s, bounce, pts[k], hit, normalN, ray, mat, inside, cn), (float)n));
  }
  return total;
}
/* Flyscene::traceRay (src/flyscene.cpp:587-601) */
static v3 trace_ray(const ora_scene* s, int bounce, const ray_t* ray, int inside, counters* cn) {
  ora_hit hit = get_intersections(s, ray);
  if (hit.t != INFINITY) return shade(s, bounce, &hit, ray, inside, cn);
  if (bounce == 0) return V3((float)0.69019607843, 1.0f, 1.0f);
  return V3(0, 0, 0);
}

static void merge_counts(ora_scene* s, const counters* cn) {
  pthread_mutex_lock(&s->mu);
  for (int i = 0; i < 5; ++i) s->counts[i] += cn->c[i];
  pthread_mutex_unlock(&s->mu);
}
/* The hit search of Flyscene::ReflectDebugRay (src/flyscene.cpp:324-357) for one (origin, dir):
 * leaf list in order; a leaf is scanned when Box::isInBox(origin) (src/box.h:87-93) or when
 * intersectBox gives t != INFINITY && t > 0 (then the box is recorded, ray_hitbox); strict <. */
/* Box::isInBox (src/box.h:87-93): closed on every side. */
static int is_in_box(v3 bmin, v3 bmax, v3 p) {
  return (p.x >= bmin.x && p.x <= bmax.x) && (p.y >= bmin.y && p.y <= bmax.y) && (p.z >= bmin.z && p.z <= bmax.z);
}
int ora_is_in_box(const float* bmin, const float* bmax, const float* p) { return is_in_box(ld3(bmin), ld3(bmax), ld3(p)); }
static ora_hit debug_hit(const ora_scene* s, v3 origin, v3 dir, uint8_t* box_flags) {
  ora_hit best = hit_miss();
  float smallestT = INFINITY;
  for (int b = 0; b < s->nleaves; ++b) {
    const leaf_t* L = &s->leaves[b];
    int scan = is_in_box(L->bmin, L->bmax, origin);
    if (!scan) {
      float bt = intersect_box(L->bmin, L->bmax, origin, dir);
      if (bt != INFINITY && bt > 0) { scan = 1; if (box_flags) box_flags[b] = 1; }
    }
    if (scan) {
      phit h;
      int i = intersect_triangles(s, s->leaf_faces + L->start, L->count, origin, dir, &h);
      if (h.t != INFINITY && smallestT > h.t) {
        smallestT = h.t;
        best.t = h.t; best.face = s->leaf_faces[L->start + i]; st3(best.n, h.n); st3(best.p, h.p); best.mat = s->tri_mat[best.face];
      }
    }
  }
  return best;
}
void ora_debug_hit(const ora_scene* s, const float* o, const float* d, int n, ora_hit* out, uint8_t* box_flags) {
  for (int i = 0; i < n; ++i) out[i] = debug_hit(s, ld3(o + 3 * i), ld3(d + 3 * i), box_flags ? box_flags + (size_t)i * s->nleaves : NULL);
}
/* Flyscene::ReflectDebugRay (src/flyscene.cpp:311-393) as data.  segs: 8 floats per segment
 * {origin, dir as passed, length, hit flag (1 = ends on a surface; a miss is drawn 10 long)};
 * seg_face: face id per segment; box_flags: OR over the bounces; light_segs: 7 floats per
 * visible point light {hit point, normalised direction, distance}.  Returns the segment count. */
int ora_debug_ray(const ora_scene* s, const float* origin, const float* dir, int jumps, float* segs, int32_t* seg_face, int max_segs,
                  uint8_t* box_flags, float* light_segs, int max_light_segs, int* n_light_segs) {
  v3 o = ld3(origin), d = ld3(dir);
  int ns = 0, nl = 0;
  for (int bounce = 0; bounce <= jumps && ns < max_segs; ++bounce) {
    ray_t r = make_ray(o, d);
    ora_hit h = debug_hit(s, o, d, box_flags);
    float* sg = segs + 8 * ns;
    st3(sg, o); st3(sg + 3, d);
    if (h.t == INFINITY) { sg[6] = 10.0f; sg[7] = 0.0f; seg_face[ns++] = -1; break; }
    sg[6] = h.t; sg[7] = 1.0f; seg_face[ns++] = h.face;
    v3 P = ld3(h.p);
    for (int i = 0; i < s->nlights; ++i) {
      v3 light = ld3(s->lights + 3 * i);
      int bm;
      if (can_see_light(s, light, P, &bm, NULL, NULL) && nl < max_light_segs) {
        float* lg = light_segs + 7 * nl++;
        st3(lg, P); st3(lg + 3, normalized3(sub3(light, P))); lg[6] = norm3(sub3(light, P));
      }
    }
    ray_t nr = reflect_ray(&r, ld3(h.n), P);
    o = nr.o; d = nr.d;
  }
  if (n_light_segs) *n_light_segs = nl;
  return ns;
}

void ora_ray_counts(const ora_scene* s, uint64_t out[5]) { for (int i = 0; i < 5; ++i) out[i] = s->counts[i]; }

/* Flyscene::trace_global_illum(Ray(o, d)) for n rays: {Kd.x, Kd.y, Kd.z, t}, zeros when nothing is hit. */
void ora_trace_gi(const ora_scene* s, const float* o, const float* d, int n, float* kd_t4) {
  for (int i = 0; i < n; ++i) { ray_t r = make_ray(ld3(o + 3 * i), ld3(d + 3 * i)); trace_gi(s, &r, kd_t4 + 4 * i); }
}
void ora_closest(const ora_scene* s, const float* o, const float* d, int n, ora_hit* out) {
  for (int i = 0; i < n; ++i) {
    ray_t r = make_ray(ld3(o + 3 * i), ld3(d + 3 * i));
    out[i] = get_intersections(s, &r);
  }
}
void ora_visible(const ora_scene* s, const float* light, const float* pos, int n, uint8_t* visible,
                 int32_t* blocker_mat, float* blocker_t, int32_t* blocker_face) {
  for (int i = 0; i < n; ++i) {
    int bm; ora_hit bh;
    visible[i] = (uint8_t)can_see_light(s, ld3(light + 3 * i), ld3(pos + 3 * i), &bm, &bh, NULL);
    if (blocker_mat) blocker_mat[i] = bm;
    if (blocker_t) blocker_t[i] = bh.t;
    if (blocker_face) blocker_face[i] = bh.face;
  }
}
void ora_trace(const ora_scene* cs, const float* o, const float* d, int n, float* rgb) {
  ora_scene* s = (ora_scene*)cs;
  counters cn; memset(&cn, 0, sizeof cn);
  memset(s->counts, 0, sizeof s->counts);
  for (int i = 0; i < n; ++i) {
    ray_t r = make_ray(ld3(o + 3 * i), ld3(d + 3 * i));
    cn.c[C_PRIMARY]++;
    st3(rgb + 3 * i, trace_ray(s, 0, &r, 0, &cn));
  }
  merge_counts(s, &cn);
}

typedef struct {
  ora_scene* s; const ora_camera* cam; int x0, y0, x1, y1, tid, nthreads;
  float* rgb; float* t; int32_t* face; float inv34[12]; v3 origin;
} render_job;
/* Flyscene::updating_pixels hot loop (src/flyscene.cpp:460-467) */
static void* render_worker(void* arg) {
  render_job* j = (render_job*)arg;
  counters cn; memset(&cn, 0, sizeof cn);
  int W = j->x1 - j->x0;
  for (int y = j->y0 + j->tid; y < j->y1; y += j->nthreads) {
    for (int x = j->x0; x < j->x1; ++x) {
      v3 p = screen_to_world(j->cam, j->inv34, (float)x + j->cam->sub[0], (float)y + j->cam->sub[1]);
      ray_t r = make_ray(j->origin, sub3(p, j->origin));
      size_t k = (size_t)(y - j->y0) * W + (x - j->x0);
      cn.c[C_PRIMARY]++;
      if (j->rgb) st3(j->rgb + 3 * k, trace_ray(j->s, 0, &r, 0, &cn));
      if (j->t || j->face) {
        ora_hit h = get_intersections(j->s, &r);
        if (j->t) j->t[k] = h.t;
        if (j->face) j->face[k] = h.face;
      }
    }
  }
  merge_counts(j->s, &cn);
  return NULL;
}
void ora_render(const ora_scene* cs, const ora_camera* cam, int x0, int y0, int x1, int y1, int threads,
                float* rgb, float* t, int32_t* face) {
  ora_scene* s = (ora_scene*)cs;
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  memset(s->counts, 0, sizeof s->counts);
  render_job jobs[256];
  pthread_t th[256];
  for (int i = 0; i < threads; ++i) {
    render_job* j = &jobs[i];
    j->s = s; j->cam = cam; j->x0 = x0; j->y0 = y0; j->x1 = x1; j->y1 = y1; j->tid = i; j->nthreads = threads;
    j->rgb = rgb; j->t = t; j->face = face;
    view_inverse(cam, j->inv34);
    j->origin = camera_center(cam);
  }
  if (threads == 1) { render_worker(&jobs[0]); return; }
  for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, render_worker, &jobs[i]);
  for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
}

typedef struct { const ora_scene* s; const float* o; const float* d; int n, tid, nthreads; ora_hit* out; } closest_job;
static void* closest_worker(void* arg) {
  closest_job* j = (closest_job*)arg;
  for (int i = j->tid; i < j->n; i += j->nthreads) {
    ray_t r = make_ray(ld3(j->o + 3 * i), ld3(j->d + 3 * i));
    j->out[i] = get_intersections(j->s, &r);
  }
  return NULL;
}
void ora_closest_mt(const ora_scene* s, const float* o, const float* d, int n, int threads, ora_hit* out) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  closest_job jobs[256];
  pthread_t th[256];
  for (int i = 0; i < threads; ++i) {
    jobs[i].s = s; jobs[i].o = o; jobs[i].d = d; jobs[i].n = n; jobs[i].tid = i; jobs[i].nthreads = threads; jobs[i].out = out;
  }
  if (threads == 1) { closest_worker(&jobs[0]); return; }
  for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, closest_worker, &jobs[i]);
  for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
}

/* ---------------------------------------------------------------- primitive doors (pinned vs _ref) */
static void st_hit7(float* h, phit p) { h[0] = p.t; st3(h + 1, p.n); st3(h + 4, p.p); }
void ora_ray(const float* o, const float* d, float* oo, float* od) { ray_t r = make_ray(ld3(o), ld3(d)); st3(oo, r.o); st3(od, r.d); }
void ora_triangle(const float* o, const float* d, const float* v0, const float* v1, const float* v2, const float* n, float* hit7) {
  ray_t r = make_ray(ld3(o), ld3(d));
  st_hit7(hit7, triangle_hit(&r, ld3(v0), ld3(v1), ld3(v2), ld3(n)));
}
void ora_plane(const float* o, const float* d, const float* v0, const float* n, float* hit7) {
  ray_t r = make_ray(ld3(o), ld3(d));
  st_hit7(hit7, plane_hit(&r, ld3(v0), ld3(n)));
}
float ora_box(const float* o, const float* d, const float* bmin, const float* bmax) {
  ray_t r = make_ray(ld3(o), ld3(d));
  return box_hit(&r, ld3(bmin), ld3(bmax));
}
void ora_sphere(const float* o, const float* d, const float* c, float radius, float* hit7) {
  ray_t r = make_ray(ld3(o), ld3(d));
  st_hit7(hit7, sphere_hit(&r, ld3(c), radius));
}
void ora_reflect(const float* o, const float* d, const float* n, const float* p, float* oo, float* od) {
  ray_t r = make_ray(ld3(o), ld3(d)); ray_t q = reflect_ray(&r, ld3(n), ld3(p)); st3(oo, q.o); st3(od, q.d);
}
void ora_refract(const float* o, const float* d, const float* n, const float* p, const float* view, float n1, float n2,
                 float* oo, float* od) {
  (void)o; (void)d;
  ray_t q = refract_ray(ld3(n), ld3(p), ld3(view), n1, n2); st3(oo, q.o); st3(od, q.d);
}
void ora_resend(const float* o, const float* d, const float* n, const float* p, int count, float* od) {
  ray_t r = make_ray(ld3(o), ld3(d));
  v3 dirs[64];
  if (count > 64) count = 64;
  resend_dirs(&r, ld3(n), ld3(p), count, dirs);
  for (int i = 0; i < count; ++i) st3(od + 3 * i, dirs[i]);
}
void ora_face_normal(const float* a, const float* b, const float* c, float* out) { st3(out, face_normal(ld3(a), ld3(b), ld3(c))); }
void ora_renormal(const float* n, float* out) { st3(out, normalized3(ld3(n))); }
float ora_dot(const float* a, const float* b) { return dot3(ld3(a), ld3(b)); }
float ora_norm(const float* a) { return norm3(ld3(a)); }
void ora_normalized(const float* a, float* out) { st3(out, normalized3(ld3(a))); }
void ora_camera_point(const ora_camera* cam, float i, float j, float* world, float* center) {
  float inv34[12];
  view_inverse(cam, inv34);
  st3(world, screen_to_world(cam, inv34, i, j));
  st3(center, camera_center(cam));
}
void ora_shader_reflect(const float* light, const float* normal, float* out) { st3(out, shader_reflect(ld3(light), ld3(normal))); }
int ora_points_on_circle(const float* center, float radius, const float* normal, int n, float* out) {
  v3 pts[64];
  if (n > 64) n = 64;
  int k = points_on_circle(ld3(center), radius, ld3(normal), n, pts);
  for (int i = 0; i < k; ++i) st3(out + 3 * i, pts[i]);
  return k;
}
