/* oracle/rt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the Raytracing-Project hot path:
 * camera ray -> closest hit -> shadow / reflection / refraction / GI rays -> shade.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product (librtx.so) never links, includes or calls anything in oracle/.
 *
 * Parity pin: the primitive arithmetic here (Eigen 3.3.7 evaluation order, Ray
 * re-normalisations, triangle / sphere / slab tests, camera ray generation, reflect /
 * refract / resend) is checked bit-for-bit against oracle/_ref/libref.so, which is the
 * reference's own src/Intersect.cpp + src/Ray.h + tucano/camera.hpp compiled in place.
 * The drivers above those primitives (getIntersections, canSeeLight, traceRay, Shader,
 * calculateColor, trace_global_illum, divideBox_KD) live in src/flyscene.cpp, which cannot
 * be linked here without GL/GLEW/GLFW libraries; they are restated from the source text
 * and pinned only by the reference outputs recorded in SURVEY.md section 8(c).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } ora_v3;

/* Tucano::Material::Mtl (dependencies/tucano/tucano/materials/mtl.hpp:16-116) */
typedef struct {
  float ka[3], kd[3], ks[3];
  float ns;     /* shininess */
  float ni;     /* optical density */
  float d;      /* dissolve factor */
  int illum;
} ora_mtl;

/* HitInfo (src/hitInfo.h:4-16) + the material id carried by the Tucano::Face copy that
 * Flyscene::getIntersections returns next to it (src/flyscene.cpp:517-585). */
typedef struct {
  float t;
  int face;       /* global mesh face index, -1 for spheres / miss */
  float n[3];
  float p[3];
  int mat;        /* material id of the hit face / sphere; -1 when none */
} ora_hit;

/* Tucano::Camera state used by screenToWorld/getCenter (tucano/camera.hpp:42-78). */
typedef struct {
  float view[12];   /* row-major 3x4 [R|t] of the Affine3f view matrix */
  float vp[4];      /* viewport minX, minY, width, height */
  float fovy;       /* degrees */
  float aspect;
  float sub[2];     /* sub-pixel offset added to (i,j); (0,0) = reference's top-left sampling */
} ora_camera;

/* Feature switches.  The reference itself is {max_bounces=2, reflection=1, refraction=1,
 * gi_rays=8, shadows=1, skip_zero_weight=0} (src/flyscene.cpp:11,686-721). */
typedef struct {
  int max_bounces;
  int reflection;
  int refraction;
  int gi_rays;
  int shadows;
  int skip_zero_weight;  /* do not trace a child ray whose blend weight is exactly 0 */
} ora_options;

typedef struct ora_scene ora_scene;

ora_scene* ora_scene_create(const float* verts_xyz, int nverts, const uint32_t* tri_idx,
                            const int32_t* tri_mat, int ntris, const ora_mtl* mats, int nmats);
void ora_scene_destroy(ora_scene*);
/* center_radius4: already-divided radius as stored by Scene::addSphere (src/Scene.h:37). */
void ora_scene_set_spheres(ora_scene*, const float* center_radius4, const int32_t* mat, int n);
void ora_scene_set_lights(ora_scene*, const float* xyz, int n, const float* sphere_xyzr, int m);
void ora_scene_set_options(ora_scene*, const ora_options*);
/* append a material (sphere materials are appended after the MTL ones, src/flyscene.cpp:454-457) */
int ora_scene_add_material(ora_scene*, const ora_mtl*);
/* Flyscene::divideBox_KD (src/flyscene.cpp:1110-1185). Returns #leaves, or -1 when the
 * reference would never terminate (detected by an iteration cap). */
int ora_scene_build_leafboxes(ora_scene*, int max_faces, long iter_cap);
/* Flyscene::ReflectDebugRay (src/flyscene.cpp:311-393): its hit search per ray, and the whole walk as data. */
void ora_debug_hit(const ora_scene*, const float* o, const float* d, int n, ora_hit* out, uint8_t* box_flags);
int ora_debug_ray(const ora_scene*, const float* origin, const float* dir, int jumps, float* segs, int32_t* seg_face, int max_segs,
                  uint8_t* box_flags, float* light_segs, int max_light_segs, int* n_light_segs);
int ora_scene_num_leaves(const ora_scene*);
int ora_scene_leaf(const ora_scene*, int i, float* bmin, float* bmax, int32_t* faces, int cap);
void ora_scene_root_box(const ora_scene*, float* bmin, float* bmax);
void ora_scene_face_normal(const ora_scene*, int f, float* n_face, float* n_query);

/* primitives (bit-pinned against oracle/_ref) */
void ora_ray(const float* o, const float* d, float* out_o, float* out_d);
void ora_triangle(const float* o, const float* d, const float* v0, const float* v1,
                  const float* v2, const float* n, float* hit7);
void ora_plane(const float* o, const float* d, const float* v0, const float* n, float* hit7);
float ora_box(const float* o, const float* d, const float* bmin, const float* bmax);
void ora_sphere(const float* o, const float* d, const float* c, float radius, float* hit7);
void ora_reflect(const float* o, const float* d, const float* n, const float* p, float* oo, float* od);
void ora_refract(const float* o, const float* d, const float* n, const float* p, const float* view,
                 float n1, float n2, float* oo, float* od);
void ora_resend(const float* o, const float* d, const float* n, const float* p, int count, float* od);
void ora_face_normal(const float* a, const float* b, const float* c, float* out);
void ora_renormal(const float* n, float* out);
float ora_dot(const float* a, const float* b);
float ora_norm(const float* a);
void ora_normalized(const float* a, float* out);
void ora_camera_point(const ora_camera* cam, float i, float j, float* world, float* center);
void ora_shader_reflect(const float* light, const float* normal, float* out);
int ora_points_on_circle(const float* center, float radius, const float* normal, int n, float* out);

/* drivers */
/* Flyscene::getIntersections for n rays given as Ray-ctor inputs (origin, un-normalised dir). */
void ora_closest(const ora_scene*, const float* o, const float* d, int n, ora_hit* out);
/* Flyscene::canSeeLight(lightPos, position): visible flag + blocker material id
 * (-1 = the default-constructed Mtl, src/flyscene.cpp:769). */
void ora_visible(const ora_scene*, const float* light, const float* pos, int n,
                 uint8_t* visible, int32_t* blocker_mat, float* blocker_t, int32_t* blocker_face);
/* Flyscene::traceRay(0, Ray(o,d), false) */
void ora_trace(const ora_scene*, const float* o, const float* d, int n, float* rgb);
/* Flyscene::updating_pixels over the pixel rectangle [x0,x1)x[y0,y1), row-major [y][x] outputs
 * (any of rgb / t / face may be NULL).  threads >= 1 (rows strided like the reference). */
void ora_render(const ora_scene*, const ora_camera*, int x0, int y0, int x1, int y1, int threads,
                float* rgb, float* t, int32_t* face);
/* ray counters of the last ora_trace / ora_render call: primary, shadow, reflection, refraction, gi */
void ora_ray_counts(const ora_scene*, uint64_t out[5]);

/* Brute-force closest hit only (spheres + root gate + all triangles), the O(F) loop that
 * the cpu_baseline times.  Same as ora_closest but multi-threaded over rays. */
void ora_closest_mt(const ora_scene*, const float* o, const float* d, int n, int threads, ora_hit* out);

/* --- BVH traversal mirror (rt_oracle_bvh.c): checker for the exported BVH + visit counters */
void ora_bvh_trace(const uint32_t* nodes8, const float* tri16, const int32_t* tri_face, uint32_t ntris, uint32_t nwild,
                   const float* root_lo, const float* root_hi, const float* o, const float* d1, int n, int threads,
                   float* t, int32_t* face, uint32_t* nodes_visited, uint32_t* tris_tested);
void ora_camera_rays(const ora_camera* cam, int x0, int y0, int x1, int y1, float* o, float* d1);

/* --- loaders / writers, in rt_oracle_io.c ------------------- */
typedef struct {
  int nverts, ntris, nmats;
  float* verts;        /* nverts*3 */
  uint32_t* idx;       /* ntris*3 */
  int32_t* tri_mat;    /* ntris */
  ora_mtl* mats;       /* nmats */
  char (*mat_names)[64];
} ora_mesh;
/* Tucano::MeshImporter::loadObjFile + MaterialImporter::loadMTL
 * (tucano/utils/objimporter.hpp:83-284, tucano/utils/mtlIO.hpp:45-125). */
int ora_obj_load(const char* path, ora_mesh* out);
void ora_mesh_free(ora_mesh*);
/* Tucano::ImageImporter::writePPMImage(filename, vector<vector<Vector3f>>)
 * (tucano/utils/ppmIO.hpp:130-151) for a row-major [y][x] rgb buffer; `reference_indexing`
 * reproduces the reference's data[i][j] transposition (square images only). */
int ora_ppm_write(const char* path, const float* rgb, int w, int h, int reference_indexing);

#ifdef __cplusplus
}
#endif
#endif
