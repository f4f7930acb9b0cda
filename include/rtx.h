/* include/rtx.h -- C ABI of librtx.so, the MI355X (gfx950) ray-scene intersection engine.
 *
 * This is the drop-in boundary for the Raytracing-Project hot path: every entry point
 * names the reference interface it replaces (paths relative to the reference repo).
 * Plain pointers and sizes only; no C++/torch types.  All functions return 0 on success
 * or a negative RTX_E* code; rtx_last_error() gives the message for the calling thread.
 *
 * Pointer residency: entry points ending in `_dev` take DEVICE pointers (hipMalloc'd or
 * torch.cuda tensors on the scene's device) and run asynchronously on the scene's stream
 * followed by a stream synchronise; all others take HOST pointers.
 *
 * Threading: calls on one rtx_scene are serialised by the caller; different scenes are
 * independent (HIP streams of their own).  The reference mutates its scene only between
 * renders (src/main.cpp:63-77); so does this API (mutators need a new rtx_scene_commit).
 *
 * Limits (RTX_ELIMIT beyond them): fewer than 2^28 triangles and 2^26 BVH node groups (~1.4e8 triangles); fewer than 2^31 pixels per call
 * and fewer than 2^31 rays per query call and per ray generation of a render (pixels x light
 * samples, or x 2^bounces with reflection and refraction both on); max_bounces <= 16 (8 for
 * reflection + refraction trees); gi_rays <= 64.
 */
#ifndef RTX_H
#define RTX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_OK 0
#define RTX_EINVAL (-1)    /* bad argument */
#define RTX_ESTATE (-2)    /* scene not committed / already committed */
#define RTX_EHIP (-3)      /* HIP runtime error (no device, out of memory, launch failure) */
#define RTX_ELIMIT (-4)    /* size beyond a documented limit */

typedef struct rtx_scene rtx_scene;

/* Tucano::Material::Mtl (dependencies/tucano/tucano/materials/mtl.hpp:16-116). */
typedef struct {
  float ka[3], kd[3], ks[3];
  float ns;        /* shininess */
  float ni;        /* optical density */
  float d;         /* dissolve factor */
  int32_t illum;
} rtx_material;

/* The state Tucano::Camera::screenToWorld / getCenter read
 * (dependencies/tucano/tucano/camera.hpp:42-78,115-118,155-173,263-266). */
typedef struct {
  float view[12];      /* row-major 3x4 [R|t] of the Affine3f view matrix */
  float viewport[4];   /* minX, minY, width, height */
  float fovy;          /* degrees */
  float aspect;        /* width / height as set by setPerspectiveMatrix */
  float subpixel[2];   /* added to the raster (i,j); (0,0) = the reference's top-left sample */
} rtx_camera;

/* Compile-time switches of the reference made run-time (src/flyscene.cpp:11,704 and the
 * unconditional reflect/refract/shadow calls at :686-723).  The reference itself is
 * {2, 1, 1, 8, 1, 0}.  gi_rays > 0 needs the reference leaf list
 * (rtx_scene_build_reference_leafboxes), because trace_global_illum walks that list. */
typedef struct {
  int32_t max_bounces;
  int32_t reflection;
  int32_t refraction;
  int32_t gi_rays;
  int32_t shadows;
  int32_t skip_zero_weight;  /* do not trace a child ray whose blend weight is exactly 0 */
} rtx_options;

typedef struct { int32_t x0, y0, x1, y1; } rtx_rect;   /* pixel rectangle [x0,x1) x [y0,y1) */

/* Counters of the last trace/render call on a scene. */
typedef struct {
  uint64_t rays_primary, rays_shadow, rays_reflection, rays_refraction, rays_gi;
  float ms_trace;      /* HIP-event time of the traversal kernels, summed over launches (a render with bounces runs the
                          bounce launch of a generation beside its shadow launch: both spans count in full) */
  float ms_total;      /* HIP-event time of every kernel of the call */
  uint32_t launches_trace;
  float ms_trace_camera;   /* the primary-ray traversal launch alone (0 when the call had none) */
  uint32_t rays_out_of_domain;   /* caller / camera rays whose origin lies beyond the validity range of the culling boxes
                                    (~30 x the largest |coordinate| of the mesh): answered by a brute-force scan of every
                                    triangle, as the reference answers all rays, so that parity holds by construction */
  uint32_t verify_checked;       /* RTX_FLAG_VERIFY: rays of this call re-checked by that brute-force scan ... */
  uint32_t verify_mismatches;    /* ... and how many of them differed from the hierarchy's answer (expected: 0) */
} rtx_stats;

typedef struct {
  uint32_t ntris, nverts, nnodes, nleaves, max_depth, nspheres, nlights, nmaterials;
  uint64_t device_bytes;
  float build_ms;
  uint32_t nwild;     /* ill-conditioned triangles kept out of the BVH (tested by every ray) */
  uint32_t nslots;    /* non-empty node slots; nnodes counts all 32-byte slots (4 per group) */
  uint32_t ntop_groups;   /* groups 0 .. ntop_groups-1 are the top three levels of the hierarchy, numbered breadth first
                             (the group-by-group traversal kernels keep them in LDS); deeper groups follow depth first */
  uint32_t build_kind;    /* 0: the host's binned-SAH build; 1: built on the GPU (RTX_FLAG_DEVICE_BUILD) */
  float device_build_ms;  /* wall time of the last build on the GPU (upload of the mesh included), 0 if none */
  float refit_ms;         /* wall time of the last refit (rtx_scene_update_vertices), 0 if none */
  uint32_t refits;        /* refits since the hierarchy's topology was built */
} rtx_scene_info;

const char* rtx_last_error(void);
int rtx_device_count(void);

/* Page-locked host memory for the host-pointer entry points (rtx_render, rtx_trace_closest, ...): what they return lands in it at
 * the PCIe rate, where a pageable destination goes through the runtime's staging copies (a 4K float frame: ~10 ms of read-back
 * against 0.4-2 ms of rendering).  No counterpart in the reference, whose frame never leaves host memory
 * (src/flyscene.cpp:410-415); for callers that do not link HIP themselves (the C++ facade keeps its frame in it).
 * rtx_host_alloc: *out = `bytes` bytes (at least one), RTX_EHIP without a GPU; rtx_host_free(NULL) is a no-op. */
int rtx_host_alloc(uint64_t bytes, void** out);
int rtx_host_free(void* p);

/* Replaces the scene state Flyscene::initialize builds from a Tucano::Mesh
 * (src/flyscene.cpp:13-120; vertices/faces: dependencies/tucano/tucano/mesh.hpp:238-247,
 * 441-468).  Triangles keep the caller's order: face ids returned by queries are indices
 * into tri_idx, exactly like HitInfo::faceId over root_box.triangles.
 * tri_material may be NULL (all -1). */
int rtx_scene_create(const float* verts_xyz, uint32_t nverts, const uint32_t* tri_idx,
                     const int32_t* tri_material, uint32_t ntris, const rtx_material* mats,
                     uint32_t nmats, rtx_scene** out);
int rtx_scene_destroy(rtx_scene*);

/* materials.push_back (src/flyscene.cpp:455-456); returns the new id (>= 0). */
int rtx_scene_add_material(rtx_scene*, const rtx_material*);
/* Scene::addSphere (src/Scene.h:36-49): center + the radius as stored (already / 3). */
int rtx_scene_set_spheres(rtx_scene*, const float* center_radius4, const int32_t* material, uint32_t n);
/* Flyscene::lights / sphericalLights (src/flyscene.hpp:140-141, addLight :61-62). */
int rtx_scene_set_lights(rtx_scene*, const float* xyz, uint32_t n, const float* sphere_xyzr, uint32_t m);

/* RTX_FLAG_NO_ROOT_GATE: closest-hit queries skip the root-AABB test that getIntersections
 * performs before its triangle scan, which gives Flyscene::intersectTriangle(faces, o, d)
 * semantics (src/flyscene.cpp:819-879) for a scene made of exactly that face list. */
#define RTX_FLAG_NO_ROOT_GATE 1u
/* RTX_FLAG_SCHED_REFILL / RTX_FLAG_SCHED_GROUPS: force one of the wave schedulers of the
 * traversal kernels for every launch of this scene (default: chosen per launch kind by timing
 * them on the scene's first seven render calls; results are identical either way).
 * RTX_FLAG_SCHED_SPLIT (only together with RTX_FLAG_SCHED_REFILL): the refilling scheduler's
 * leaf phases stop when few lanes have triangles left and keep the rest of a leaf pending.
 * No counterpart in the reference; for tests and tuning. */
#define RTX_FLAG_SCHED_REFILL 2u
#define RTX_FLAG_SCHED_GROUPS 4u
#define RTX_FLAG_SCHED_SPLIT 8u
/* RTX_FLAG_VERIFY: every closest-hit launch of caller or camera rays re-checks a sample of its rays (every k-th, about
 * 2,000 per launch) by brute force over all triangles and reports the outcome in rtx_stats.verify_*: a standing
 * check of "the BVH only skips triangles the ray cannot hit" on the caller's own scenes, at ~ntris x 2,000 triangle
 * tests per launch.  No counterpart in the reference (which always scans every triangle). */
#define RTX_FLAG_VERIFY 16u
/* RTX_FLAG_FAST_TRIANGLES (takes effect at the next rtx_scene_commit): the traversal kernels test triangles with a
 * watertight Moeller-Trumbore functor (48-byte {v0, e1, e2, face} records, edge functions compared before the one
 * reciprocal, edges inclusive with a 1e-6 overlap so that neighbouring triangles leave no gap) instead of the
 * reference's plane + barycentric arithmetic (src/Intersect.cpp:6-46).  Acceptance rules stay the reference's
 * (t >= 0.001f, lowest face id on equal t, spheres first), but t differs by ulps and a hit within ~1e-6 of an edge may
 * go to the neighbouring face: hit ids are then NOT bit-exact (DESIGN.md 3 has the measured match rates per scene).
 * Off by default: the default path is the reference's arithmetic.  With RTX_FLAG_VERIFY the sampled re-check (always
 * in the reference's arithmetic) counts those differences in rtx_stats.verify_mismatches. */
#define RTX_FLAG_FAST_TRIANGLES 32u
/* RTX_FLAG_PARENT_LIST: a diagnostic switch without effect on results.  A camera frame whose nodes spawn nothing but shadow
 * rays (no reflection / refraction generations, no GI probes) is shaded straight from its hit records: the shadow launch walks
 * the output slots, no compacted parent list is made.  With this flag such frames take the general path (parent list, one
 * hit-point record per parent), as frames with child rays always do; the two give the same pixels bit for bit (tested). */
#define RTX_FLAG_PARENT_LIST 64u
/* RTX_FLAG_DEVICE_BUILD (takes effect at the next rtx_scene_commit of a scene not built yet): the acceleration structure
 * that replaces Box(mesh) + divideBox_KD (src/box.h:41-85, src/flyscene.cpp:77-82,1110-1185) is built ON THE GPU instead
 * of on the host -- per-triangle records and padded boxes in the reference's arithmetic, Morton order, a binary radix
 * tree, 4-wide groups chosen by the same expected-cost programme as the host's collapse (a few milliseconds per million
 * triangles against 0.3 s for the host's binned-SAH build; DESIGN.md 5 has the traversal speed of either tree).  Results
 * do not depend on the hierarchy: hits are a (t, face id) minimum over everything a ray can reach, whichever builder made
 * the boxes (tested against the oracle both ways).  Meshes of fewer than 8 regular triangles, and trees too deep for the
 * traversal stack, take the host builder all the same (rtx_scene_info.build_kind says which ran).
 * RTX_BUILD=device|host in the environment overrides the flag (A/B runs). */
#define RTX_FLAG_DEVICE_BUILD 128u
int rtx_scene_set_flags(rtx_scene*, uint32_t flags);

/* Flyscene::divideBox_KD + averagePoint / axisToDivide / isInBox and Box(mesh)
 * (src/flyscene.cpp:974-993,1070-1185, src/box.h:41-85): the flat leaf list `boxes` that
 * trace_global_illum (and the debug ray) walk; triangles are assigned by "any vertex inside"
 * exactly as the reference does.  max_faces <= 0 means (int)sqrt(F + 100)
 * (src/flyscene.cpp:79).  Returns the leaf count, or RTX_ELIMIT when the reference's split
 * never terminates on this mesh (iter_cap <= 0 picks 64 F + 4096 queue pops).
 * rtx_scene_set_leafboxes installs a caller-built list instead. */
int rtx_scene_build_reference_leafboxes(rtx_scene*, int32_t max_faces, int64_t iter_cap);
int rtx_scene_set_leafboxes(rtx_scene*, const float* bmin_xyz, const float* bmax_xyz, const uint32_t* start,
                            const uint32_t* count, uint32_t nleaves, const int32_t* faces, uint32_t nfaces);
/* Any output may be NULL; nleaves / nfaces receive the sizes. */
int rtx_scene_get_leafboxes(const rtx_scene*, float* bmin_xyz, float* bmax_xyz, uint32_t* start, uint32_t* count,
                            int32_t* faces, uint32_t* nleaves, uint32_t* nfaces);

/* Replaces Box(mesh) + divideBox_KD as the acceleration structure of closest-hit queries
 * (src/box.h:41-85, src/flyscene.cpp:77-82,517-585).
 * rtx_scene_build flattens the mesh into per-triangle records (reference arithmetic) and
 * builds the linear BVH on the host; it needs no GPU.  rtx_scene_commit builds if needed
 * and uploads everything to `device`. */
int rtx_scene_build(rtx_scene*);
int rtx_scene_commit(rtx_scene*, int device);
/* New vertex positions for a committed scene (same vertex count, same triangles): an animated mesh.  The reference has
 * no counterpart -- its mesh is loaded once and Box(mesh) / divideBox_KD run once in Flyscene::initialize
 * (src/flyscene.cpp:22-23,77-82) -- but everything it derives from the vertices is derived again here, in its
 * arithmetic: face normals, the per-triangle constants of Triangle::intersects, the root box of the root gate.
 *   RTX_UPDATE_REFIT    keeps the hierarchy's topology and refits its boxes on the GPU, level by level from the leaves
 *                       (tens of microseconds per level; the right choice while the mesh deforms smoothly);
 *   RTX_UPDATE_REBUILD  builds the hierarchy again: on the GPU when the scene has RTX_FLAG_DEVICE_BUILD, else on the host.
 * A refit that finds a triangle of the hierarchy degenerated into one whose accepted set no box can bound (DESIGN.md 3,
 * "wild" triangles) rebuilds instead.  Hits after either are what a fresh scene with the new vertices gives, bit for bit.
 * The reference leaf list of the GI probes (rtx_scene_build_reference_leafboxes) is a function of the vertices and is
 * dropped: build it again if gi_rays > 0 is wanted.  Finishes the scene's enqueued renders first.
 * rtx_scene_update_vertices takes a host pointer, rtx_scene_update_vertices_dev a device pointer on the scene's device. */
#define RTX_UPDATE_REFIT 0
#define RTX_UPDATE_REBUILD 1
int rtx_scene_update_vertices(rtx_scene*, const float* verts_xyz, uint32_t nverts, int32_t how);
int rtx_scene_update_vertices_dev(rtx_scene*, const float* verts_xyz, uint32_t nverts, int32_t how);
int rtx_scene_get_info(const rtx_scene*, rtx_scene_info*);
/* Host copy of the flattened acceleration structure (tests / roofline accounting).
 * nodes: nnodes x 8 words {lo.xyz, ref, hi.xyz, count}, in groups of 4 siblings (128 bytes); group 0
 * holds the root's children.  count > 0: leaf over triangles [ref, ref+count); count == 0: internal,
 * ref = index of its child group; ref == 0xFFFFFFFF: unused slot.  max_depth counts group levels.
 * tri_face: ntris ids in leaf order,
 * the last rtx_scene_info.nwild of them being the triangles outside the hierarchy.
 * (On the device the same groups are held in a 64-byte quantised encoding with outward-rounded
 * boxes, and the triangle record in the order {n,D}{v0,v.v}{u,u.u}{v,u.v}; these exports are the
 * fp32 source of both.) */
int rtx_scene_get_bvh(const rtx_scene*, uint32_t* nodes8, uint32_t* tri_face);
/* The same groups in the 64-byte encoding the kernels load: nnodes / 4 x 16 words
 * {origin.xyz, step_x (floats), entry[4], lox, loy, loz, hix, hiy, hiz, step_y, step_z (floats)}; plane =
 * origin + byte * step per axis (byte k of a plane word = child k; steps are powers of two), rounded
 * outwards; entry = 0x80000000 | count << 28 | first triangle (leaf), child group (internal),
 * 0x80000000 (unused: a leaf of no triangles, harmless to visit).  Inspection only (tests); no counterpart
 * in the reference. */
int rtx_scene_get_packed_bvh(const rtx_scene*, uint32_t* groups16);
/* Host copy of the leaf-ordered 64-byte triangle records: ntris x 16 floats
 * {v0.xyz, D}{u.xyz, u.u}{v.xyz, u.v}{n.xyz, v.v} (src/Intersect.cpp:10,26-32 hoisted per triangle). */
int rtx_scene_get_tris(const rtx_scene*, float* tri16);

/* Flyscene::getIntersections(Ray(o,d)) for n rays (src/flyscene.cpp:517-585): closest hit
 * over spheres, then the root-box gate and every triangle; face_id = mesh face index
 * (-1 for spheres and misses), t = INFINITY on a miss.  normal/point/material may be NULL. */
int rtx_trace_closest(rtx_scene*, const float* origin_xyz, const float* dir_xyz, uint32_t n,
                      float* t, int32_t* face_id, float* normal_xyz, float* point_xyz,
                      int32_t* material);
int rtx_trace_closest_dev(rtx_scene*, const float* origin_xyz, const float* dir_xyz, uint32_t n,
                          float* t, int32_t* face_id);

/* rtx_trace_closest_dev for callers that cannot allocate device memory (tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Same kernels and results, but the
 * ray records, hit records and traversal-stack overflow that rtx_trace_closest_dev keeps in the scene's
 * growable workspace live in a fixed device array inside the library (~11.7 MB per GPU, held until the
 * process ends), and the call makes no hipMalloc / hipFree / event creation.  At most RTX_NOALLOC_MAX_RAYS
 * rays per call (RTX_ELIMIT beyond).  All calls on one GPU share that array, whichever scene they name: calls
 * that overlap on a GPU race; serialise them.  Returns when the results are in t / face_id (device pointers,
 * like the inputs): it launches on a stream of the library's own and waits for it, so the call itself cannot
 * be captured into a graph.  Unlike the other synchronous entry points it leaves enqueued renders in flight
 * and rtx_get_stats untouched (no events, no counters read back). */
#define RTX_NOALLOC_MAX_RAYS 65536u
int rtx_trace_closest_noalloc(rtx_scene*, const float* origin_xyz, const float* dir_xyz, uint32_t n,
                              float* t, int32_t* face_id);

/* Flyscene::canSeeLight(lightPos, position) for n pairs (src/flyscene.cpp:747-782).
 * blocker_material: material id the reference would return, -1 for the default Mtl. */
int rtx_trace_visible(rtx_scene*, const float* light_xyz, const float* position_xyz, uint32_t n,
                      uint8_t* visible, int32_t* blocker_material);

/* Tucano::Camera::screenToWorld for n raster positions (i,j pairs) and Camera::getCenter
 * (dependencies/tucano/tucano/camera.hpp:115-118,155-173).  center may be NULL. */
int rtx_camera_points(rtx_scene*, const rtx_camera*, const float* ij, uint32_t n, float* world_xyz, float* center_xyz);

/* Flyscene::intersectBox(box, origin, dest) for n rays against one box (src/flyscene.cpp:906-972,
 * src/Intersect.cpp:67-145): the slab-test t (may be negative), INFINITY on a miss. */
int rtx_intersect_box(rtx_scene*, const float* bmin_xyz, const float* bmax_xyz, const float* origin_xyz,
                      const float* dir_xyz, uint32_t n, float* t);
/* Flyscene::trace_global_illum(Ray(o, d)) for n rays (src/flyscene.cpp:631-669): closest hit over the spheres and the
 * reference's flat leaf list (not the BVH: the leaf lists are lossy and the answer depends on it);
 * kd_t4 = {Kd.x, Kd.y, Kd.z, t} per ray, zeros when nothing is hit.  Needs the reference leaf list before commit. */
int rtx_trace_gi(rtx_scene*, const float* origin_xyz, const float* dir_xyz, uint32_t n, float* kd_t4);

/* Intersect::Triangle / Plane / BoxObject / Sphere ::intersects(Ray&) (src/Intersect.h:8-89, src/Intersect.cpp:6-193)
 * for n Ray objects against one primitive; needs no scene, only a device.  origin / dir are the Ray's members as
 * they stand (its constructor has normalised dir; nothing is normalised again).  prim: RTX_PRIM_TRIANGLE {v0, v1,
 * v2, normal} (12 floats), RTX_PRIM_PLANE {v0, normal} (6), RTX_PRIM_BOX {tmin, tmax} (6), RTX_PRIM_SPHERE {centre,
 * radius} (4).  Outputs are the HitInfo members the reference returns: t (INFINITY on a miss; for the box the slab
 * distance, which may be negative), face_id (0 for a triangle / plane hit, else -1), normal, point (zeros where the
 * reference leaves them unset); face_id / normal / point may be NULL. */
#define RTX_PRIM_TRIANGLE 0
#define RTX_PRIM_PLANE 1
#define RTX_PRIM_BOX 2
#define RTX_PRIM_SPHERE 3
int rtx_intersect_primitive(int device, int32_t kind, const float* prim, const float* origin_xyz, const float* dir_xyz, uint32_t n,
                            float* t, int32_t* face_id, float* normal_xyz, float* point_xyz);

/* The hit search of Flyscene::ReflectDebugRay (src/flyscene.cpp:324-357) for n rays (o, d = its
 * origin / dir arguments): closest triangle over the reference's leaf list, a leaf being scanned
 * when the origin is inside its box (Box::isInBox, src/box.h:87-93) or the slab test gives t > 0;
 * no spheres, no root gate.  face = mesh face id or -1, t = INFINITY on a miss, normal / point as
 * HitInfo holds them.  box_flags (optional, n x number of leaves): 1 where the slab test passed
 * from outside -- the boxes the reference highlights (ray_hitbox).  Needs the reference leaf
 * list (rtx_scene_build_reference_leafboxes) before commit. */
int rtx_debug_hit(rtx_scene*, const float* o, const float* d, uint32_t n, float* t, int32_t* face, float* normal,
                  float* point, uint8_t* box_flags);

/* Flyscene::traceRay(bounce, Ray(o,d), insideObject) for n rays (src/flyscene.cpp:587-601):
 * the shaded colour, 3 floats per ray. */
int rtx_trace_shade(rtx_scene*, const rtx_options*, const float* origin_xyz, const float* dir_xyz, uint32_t n,
                    int32_t bounce, int32_t inside_object, float* rgb);

/* Flyscene::updating_pixels / raytraceScene over a pixel rectangle
 * (src/flyscene.cpp:395-468): camera ray -> traceRay(0, ray, false) per pixel.
 * Outputs are row-major [y - y0][x - x0]; any of rgb (3 floats/pixel), t, face_id may be NULL. */
int rtx_render(rtx_scene*, const rtx_camera*, const rtx_options*, rtx_rect rect,
               float* rgb, float* t, int32_t* face_id);
int rtx_render_dev(rtx_scene*, const rtx_camera*, const rtx_options*, rtx_rect rect,
                   float* rgb, float* t, int32_t* face_id);
/* rtx_render_dev in two halves, for sequences of frames: rtx_render_dev_enqueue queues the call's
 * kernels and returns; rtx_render_finish waits for the oldest call enqueued and not yet handed out and
 * reports its statistics (stats may be NULL; RTX_ESTATE when there is none).  Up to four calls may be
 * in flight, each on HIP streams and a workspace of its own: they run SIDE BY SIDE on the GPU (frame
 * k+1's launches take the CUs that frame k's traversal kernels leave free while their last waves
 * drain: +29 % rays/s on the 4K soup, +70 % on 1080p frames), so their outputs must not overlap, and
 * the per-kernel times in their rtx_stats include the time shared with the other calls.  The outputs
 * of a call are complete once rtx_render_finish has returned for it; a fifth enqueue first finishes
 * the oldest call (its statistics are then lost), and any synchronous entry point on the scene
 * finishes all of them.  While the library is still timing its wave schedulers on a scene (its first
 * seven render calls) enqueued calls run one at a time.  RTX_OVERLAP_FRAMES=0 in the environment
 * keeps them one at a time always.
 * No counterpart in the reference (raytraceScene returns when the frame is done,
 * src/flyscene.cpp:395-468). */
int rtx_render_dev_enqueue(rtx_scene*, const rtx_camera*, const rtx_options*, rtx_rect rect,
                           float* rgb, float* t, int32_t* face_id);
int rtx_render_finish(rtx_scene*, rtx_stats* stats);

/* The row-strided thread decomposition of raytraceScene (src/flyscene.cpp:418-430,460) as an
 * image-tile partition for several GPUs: `rect` is cut into tile_size x tile_size tiles
 * (row-major tile order), part p of nparts renders tiles p, p + nparts, ... and stores its
 * k-th tile at output pixels [k*tile_size^2, (k+1)*tile_size^2), row-major inside the tile
 * (slots of partial edge tiles that fall outside `rect` are left untouched).
 * rtx_tiles_count returns how many tiles that part owns (>= 0), or a negative error. */
int rtx_tiles_count(rtx_rect rect, int32_t tile_size, int32_t part, int32_t nparts);
int rtx_render_tiles_dev(rtx_scene*, const rtx_camera*, const rtx_options*, rtx_rect rect,
                         int32_t tile_size, int32_t part, int32_t nparts,
                         float* rgb, float* t, int32_t* face_id);
/* ... and its enqueue form, finished by rtx_render_finish like rtx_render_dev_enqueue. */
int rtx_render_tiles_dev_enqueue(rtx_scene*, const rtx_camera*, const rtx_options*, rtx_rect rect,
                                 int32_t tile_size, int32_t part, int32_t nparts,
                                 float* rgb, float* t, int32_t* face_id);

/* A batch of frames in one call: nframes (1 .. RTX_MAX_BATCH = 64) cameras over the same rectangle and tile partition,
 * traced and shaded by ONE set of kernel launches; frame f's pixels land at rgb + 3 f P, t + f P, face_id + f P with
 * P = rtx_tiles_count(rect, tile_size, 0, nparts) * tile_size^2 -- the size of the LARGEST part (part 0), the same for
 * every part, so that the parts of all ranks have one layout [frame][local tile] and gather into one buffer whatever
 * their own tile counts (slots beyond a part's last tile are left untouched) -- or, with tile_size 0 (part 0 of 1), the
 * whole rectangle row-major and P = its pixels.  Every frame's values are those of rtx_render_tiles_dev (rtx_render_dev)
 * with that camera, bit for bit.  For the ranks of an N-GPU run: a rank's share of one frame gives its persistent
 * traversal kernels too few rays to run at full rate (1/8 of the 4K soup frame: 0.37 ms instead of 0.23 ms), its
 * share of N frames of an animation gives them as many as one whole frame.  The enqueue form is finished by
 * rtx_render_finish like the others; rtx_stats counts the rays of all frames of the batch.  Workspace: the call's queues are
 * sized for all frames at once -- about 8 + 4 + 40 L bytes per pixel and frame (hit records, parents, L light samples' shadow rays
 * and hits), more with bounces -- per call slot in use: 3.4 GB for 8 frames of 3840x2160 with one light.
 * No counterpart in the reference (raytraceScene renders one camera, src/flyscene.cpp:395-468). */
#define RTX_MAX_BATCH 64
int rtx_render_tiles_batch_dev(rtx_scene*, const rtx_camera* cams, int32_t nframes, const rtx_options*, rtx_rect rect,
                               int32_t tile_size, int32_t part, int32_t nparts,
                               float* rgb, float* t, int32_t* face_id);
int rtx_render_tiles_batch_dev_enqueue(rtx_scene*, const rtx_camera* cams, int32_t nframes, const rtx_options*, rtx_rect rect,
                                       int32_t tile_size, int32_t part, int32_t nparts,
                                       float* rgb, float* t, int32_t* face_id);

/* The inverse of that layout, for the rank that collects the parts: `parts` holds the compact buffers of
 * parts 0 .. nparts-1 one after the other, part_stride_pixels apart (>= rtx_tiles_count(rect, tile_size, 0,
 * nparts) * tile_size^2), words_per_pixel 32-bit words per pixel (3 for rgb, 1 for t or face_id);
 * out receives the row-major rectangle [y - y0][x - x0].  Device pointers; runs on `stream` (a
 * hipStream_t as void*, NULL = default stream) and returns without waiting for it.  No scene needed. */
int rtx_tiles_assemble_dev(const void* parts, int32_t nparts, uint64_t part_stride_pixels, rtx_rect rect,
                           int32_t tile_size, int32_t words_per_pixel, void* out, void* stream);
/* ... for pixels of any size in bytes (1 .. 64; 3 for the P3 bytes below): the same call, sizes in bytes. */
int rtx_tiles_assemble_bytes_dev(const void* parts, int32_t nparts, uint64_t part_stride_pixels, rtx_rect rect,
                                 int32_t tile_size, int32_t bytes_per_pixel, void* out, void* stream);

/* The quantisation of Tucano::ImageImporter::writePPMImage (dependencies/tucano/tucano/utils/ppmIO.hpp:130-151, called by
 * raytraceScene, src/flyscene.cpp:433) on the device: out[3 i + c] = min(255, (int)(255 * rgb[3 i + c])) as a byte, for
 * npixels pixels in whatever order they lie (row-major frames and compact tile buffers alike) -- the numbers result.ppm
 * holds, bit-equal to quantising the fp32 frame on the host.  What a rank of a multi-GPU run hands over when the frame is
 * wanted as the reference writes it: 3 bytes per pixel instead of 12.  A channel whose P3 number no byte can hold (the
 * reference prints a negative number for a negative channel and for NaN / infinities) becomes 0 (255 for +infinity).
 * Device pointers; runs on `stream` (a hipStream_t as void*, NULL = default stream) without waiting for it.  No scene needed. */
int rtx_rgb_to_ppm8_dev(const float* rgb, uint64_t npixels, uint8_t* out, void* stream);

/* Several GPUs in one process: Flyscene::raytraceScene's thread decomposition (src/flyscene.cpp:417-430,460:
 * 16 threads striding the rows of one frame) as nscenes parts of the tile grid, part i rendered by scenes[i]
 * on the device it was committed on -- all parts at once -- then moved to scenes[0]'s device over xGMI (grouped
 * ncclSend / ncclRecv: RCCL is opened with dlopen the first time two handles of a call sit on different devices, so
 * the library has no link-time dependency on it; hipMemcpyPeerAsync where RCCL cannot be had or with
 * RTX_MULTI_GATHER=peer in the environment) and put back into row-major order there.  Every entry of `scenes` is a handle of its own holding the same
 * scene (rtx_scene_clone copies a scene's host description and its built BVH for the next device; two handles
 * may share a device).  Outputs as rtx_render: row-major [y - y0][x - x0], any may be NULL; host pointers for
 * rtx_render_multi, device pointers on scenes[0]'s device for rtx_render_multi_dev.  The frame is bit-identical
 * to rtx_render's on one scene.  Ray counts and times stay per scene (rtx_get_stats of each handle). */
int rtx_scene_clone(const rtx_scene* src, rtx_scene** out);
int rtx_render_multi(rtx_scene* const* scenes, int32_t nscenes, const rtx_camera*, const rtx_options*, rtx_rect rect,
                     int32_t tile_size, float* rgb, float* t, int32_t* face_id);
int rtx_render_multi_dev(rtx_scene* const* scenes, int32_t nscenes, const rtx_camera*, const rtx_options*, rtx_rect rect,
                         int32_t tile_size, float* rgb, float* t, int32_t* face_id);

int rtx_get_stats(const rtx_scene*, rtx_stats*);

#ifdef __cplusplus
}
#endif
#endif
