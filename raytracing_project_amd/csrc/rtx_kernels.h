// HIP kernels of librtx (gfx950 / CDNA4, wave64).  Included only by rtx.hip.
//
// Arithmetic contract: everything that decides a hit (ray normalisation, the plane +
// barycentric triangle test, the sphere quadratic, the root-box gate, t >= 0.001f, strict-<
// reduction with the lowest face id on ties) is the reference's fp32 arithmetic in the
// reference's order (src/Intersect.cpp, src/flyscene.cpp:517-585,819-879), compiled with
// -ffp-contract=off.  The BVH only *skips* triangles whose padded boxes the ray misses.
#pragma once
#include <hip/hip_runtime.h>

#include "rtx_math.h"

#ifndef RTX_BLOCKS_REFILL
#define RTX_BLOCKS_REFILL 7   // resident blocks per CU the refilling kernels are compiled for: 72 VGPRs, no
                              // spills (at 8 the camera kernel spilled 15 VGPRs and wrote 0.4 GB of scratch per
                              // launch for the same speed)
#endif
#ifndef RTX_REFILL_MIN
#define RTX_REFILL_MIN 8    // idle lanes of a wave before parked rays are handed out (16 -> 8 with three node steps per iteration: kernels -3 %)
#endif
#ifndef RTX_DEFER_MIN
#define RTX_DEFER_MIN 24    // lanes at leaves before the triangle tests run (while node work exists)
#endif
#ifndef RTX_SORT_CX
#define RTX_SORT_CX 4       // compare-exchanges of the child sort: 5 = full order; 4 leaves the two middle children unordered
#endif                    // (+2 % node steps on the soup, +0.6 % on the hall by the CPU replay; 5 VALU instructions less per step)
#ifndef RTX_SPEC
#define RTX_SPEC 1         // 1: lanes go on past a pending leaf; 0: they wait at it (A/B switch); 2: on, except for any-blocker shadow rays
#endif
#ifndef RTX_NODE_UNROLL
#define RTX_NODE_UNROLL 3    // node steps per iteration of the refilling scheduler's loop (1 -> 3: soup kernels -5 %, hall -6 %)
#endif
#ifndef RTX_BLOCK_MIN
#define RTX_BLOCK_MIN 8     // ... or lanes that hold a leaf and have no node work
#endif

namespace rtx {

constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kStackDepth = 48;      // per-lane traversal stack entries in total (<= 3 per level of 4-node groups, <= 15 levels);
                                     // rtx.hip asserts it equals bvh_build.h's kTraversalStackDepth, which the builder's depth fallback uses
constexpr int kLdsStackGroups = 16;  // same for the group-by-group scheduler, which parks nothing
// Node groups of the top levels of the hierarchy kept in LDS per block (64 B each; 21 = three levels, 5 = two, 0 = none),
// per scheduler.  Measured: the group-by-group kernels, whose 64 rays start at the root together, issue 10 %
// fewer vector memory instructions with three levels staged (bunny shadow launch +3 %, camera launch -1 %); the refilling
// kernels, whose lanes are at all depths at once, gain nothing (a wave's load costs the L1 path the same however few
// lanes take part: -0.6 % TA wavefronts) and would have to give a stack level for the room (-1 %): 0 there.
#ifndef RTX_TOP_GROUPS
#define RTX_TOP_GROUPS 21
#endif
#ifndef RTX_TOP_GROUPS_REFILL
#define RTX_TOP_GROUPS_REFILL 0
#endif
#ifndef RTX_TOP_UNIFORM
#define RTX_TOP_UNIFORM 1            // 1: the LDS copy is read only when every stepping lane of the wave is in the top levels
#endif
constexpr int kTopGroups = RTX_TOP_GROUPS;
constexpr int kTopGroupsRefill = RTX_TOP_GROUPS_REFILL;
// The first entries of a lane's traversal stack live in LDS, deeper ones (rare) in a per-thread global column.  Per block of the
// refilling kernels: 4 B x 256 x kLdsStack of stack + 9 KB of parked rays + 64 B per staged group -> 7 blocks per CU of 160 KB
// with 13 levels and up to 5 groups, or 12 levels and up to 21 groups.
constexpr int kLdsStack = kTopGroupsRefill > 5 ? 12 : 13;
                                     // deeper ones (rare) spill to a per-thread global column
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr float kPrecision = 0.001f; // Intersect::Face::precision (src/Intersect.h:14)
constexpr float kPruneSlack = 1.0001f;

struct DevMat { float ka[3], kd[3], ks[3]; float ns, ni, d; int illum; int pad[3]; };
static_assert(sizeof(DevMat) == 64, "material record is 64 bytes");

struct DevScene {
  const uint4* nodes;      // 4 x uint4 per 64-byte packed group of 4 sibling nodes (PackedGroup, bvh_build.h)
  const float4* tris;      // 4 x float4 per leaf-ordered triangle: {n,D} {v0,vv} {u,uu} {v,uv}
  const float4* tris_mt;   // RTX_FLAG_FAST_TRIANGLES only: 3 x float4 per leaf-ordered triangle, {v0, e1.x} {e1.yz, e2.xy} {e2.z, face id, -, -}
  const int* tri_face;     // leaf order -> mesh face id
  const float4* face_nm;   // per mesh face: {query normal xyz, material id bits}
  const float4* spheres;   // {center xyz, radius}
  const int* sph_mat;
  const DevMat* mats;
  const float4* lights;    // point lights {xyz, 0}
  const float4* slights;   // spherical lights {center, radius}
  const float4* leafboxes; // reference leaf list (GI only): {bmin, start bits}, {bmax, count bits}
  const int* leaf_faces;   // mesh face ids of the leaf lists
  const int* face_pos;     // mesh face id -> position in tris
  uint32_t ntris, nspheres, nlights, nmats;
  uint32_t nsl, nleafboxes;
  uint32_t flags, ntop;           // ntop: node groups 0 .. ntop-1 are the top levels of the hierarchy (breadth first), staged in LDS
                                  // bit 0: skip the reference's root-box gate (intersectTriangle semantics); bit 1: any blocker
                                  // shades alike; bits 2-5: reflection / refraction weights zero for all or for no material (rtx.hip)
                                  // bit 1: every material shades a blocked light alike (set at commit)
  float root_lo[3], root_hi[3];   // Box(mesh) of the reference, used as its root gate
  uint32_t root_entry;            // traversal entry of the BVH root
  uint32_t nwild;                 // ill-conditioned triangles [ntris - nwild, ntris): tested by every ray
  float sl_sin, sl_cos;           // sinf / cosf of (float)(2 pi / 4): the ring of Shader::getNPointsOnCircle (evaluated on the host, glibc)
};

struct DevCamera {
  float inv34[12];     // inverse view, row-major 3x4
  float origin[3];     // Camera::getCenter
  float vp[4];
  float aspect_scale;  // aspect_ratio * scale
  float scale;         // 1 / getPerspectiveScale
  float sub[2];
  int x0, y0, w, h;    // pixel rectangle
  // Tiled-compact layout (multi-GPU image partition): the rectangle is cut into ts x ts
  // tiles, this scene renders tiles part, part + nparts, ... and stores local tile k at
  // compact pixels [k*ts*ts, (k+1)*ts*ts), row-major inside the tile.  ts == 0: plain
  // row-major rectangle.
  int ts, part, nparts, tiles_x;
  uint32_t ntiles_local;
  // A batch of frames in one call (rtx_render_tiles_batch_dev): nframes cameras over the same rectangle / tile
  // partition; frame f's output slots are [f * frame_slots, (f + 1) * frame_slots), its 16x16 pixel blocks
  // [f * frame_blocks, ...), its tables at tab + f * (w + h), its inverse view and centre in frames[16 f ..]
  // (12 + 3 floats + pad).  nframes == 1: the fields above, nothing read from `frames`.
  uint32_t nframes, frame_slots, frame_blocks;
  const float* frames;
  const float* tab;    // nx of the w columns, then ny of the h rows (k_camera_tables)
};


// ---------------------------------------------------------------- reference primitives
// Intersect::BoxObject::intersects (src/Intersect.cpp:67-145); used only for the root gate.
__device__ __forceinline__ float ref_box_t(V3 o, V3 dir, const float* bmin, const float* bmax) {
  float ix = 1 / dir.x, iy = 1 / dir.y, iz = 1 / dir.z;
  float txmin, txmax, tymin, tymax, tzmin, tzmax;
  if (ix >= 0) { txmin = (bmin[0] - o.x) * ix; txmax = (bmax[0] - o.x) * ix; }
  else { txmin = (bmax[0] - o.x) * ix; txmax = (bmin[0] - o.x) * ix; }
  if (iy >= 0) { tymin = (bmin[1] - o.y) * iy; tymax = (bmax[1] - o.y) * iy; }
  else { tymin = (bmax[1] - o.y) * iy; tymax = (bmin[1] - o.y) * iy; }
  if ((txmin > tymax) || (tymin > txmax)) return INFINITY;
  if (tymin > txmin) txmin = tymin;
  if (tymax < txmax) txmax = tymax;
  if (iz >= 0) { tzmin = (bmin[2] - o.z) * iz; tzmax = (bmax[2] - o.z) * iz; }
  else { tzmin = (bmax[2] - o.z) * iz; tzmax = (bmin[2] - o.z) * iz; }
  if ((txmin > tzmax) || (tzmin > txmax)) return INFINITY;
  if (tzmin > txmin) txmin = tzmin;
  if (tzmax < txmax) txmax = tzmax;
  return txmin;
}

// Intersect::Sphere::intersects + solveQuadratic (src/Intersect.cpp:147-193): t or INFINITY.
__device__ __forceinline__ float ref_sphere_t(V3 o, V3 d, float4 s) {
  float radius2 = s.w * s.w;
  V3 L = o - v3(s.x, s.y, s.z);
  float a = dot(d, d);
  float b = 2.0f * dot(d, L);
  float c = dot(L, L) - radius2;
  float discr = b * b - 4.0f * a * c;
  if (discr < 0) return INFINITY;
  float x0, x1;
  if (discr == 0) { x0 = x1 = (float)(-0.5 * (double)b / (double)a); }
  else {
    float sq = sqrtf(discr);
    float q = (b > 0) ? -0.5f * (b + sq) : -0.5f * (b - sq);
    x0 = q / a;
    x1 = c / q;
  }
  if (x0 > x1) { float tmp = x0; x0 = x1; x1 = tmp; }
  if (x0 < 0) x0 = x1;
  if (x0 > 0 && x0 >= kPrecision) return x0;
  return INFINITY;
}

struct Hit { float t; int id; };   // id >= 0 mesh face, -1 miss, <= -2 sphere (-2 - index)

__device__ __forceinline__ float cull_inv(float d) {
  const float iv = 1.0f / d;
  return (fabsf(iv) <= 1e18f) ? iv : copysignf(1e18f, d);
}

// A ray with a NaN or an infinity among its six numbers hits nothing in the reference (every comparison of
// Triangle::intersects fails, or t is infinite and loses against any hit or the miss alike), so the
// hierarchy need not be walked for it -- and should not be: its slab tests all "pass" and it would visit
// every node.  x - x is 0 for a finite x and NaN otherwise.
__device__ __forceinline__ bool ray_is_finite(V3 o, V3 d) {
  return ((o.x - o.x) + (o.y - o.y) + (o.z - o.z)) + ((d.x - d.x) + (d.y - d.y) + (d.z - d.z)) == 0.0f;
}

// Intersect::Triangle::intersects + isInTriangle (src/Intersect.cpp:6-46) on leaf-ordered
// triangle k; u, v, u.u, u.v, v.v, n.v0 are ray independent and were hoisted into the record.
// Updates `best` with the reference's reduction: strict < over faces in mesh order, i.e. the
// lexicographic min of (t, face); a sphere at the same t keeps winning (src/flyscene.cpp:569).
__device__ __forceinline__ void test_triangle(const DevScene& S, uint32_t k, V3 o, V3 d, Hit* best) {
  const float4 qn = S.tris[4 * k];                 // n, D
  const V3 n = v3(qn.x, qn.y, qn.z);
  const float den = dot(n, d);
  const float num = qn.w - dot(o, n);
  // (A reciprocal-based pre-rejection of t was measured here: 87 % of the triangles a ray reaches
  // pass the plane stage, so it cost more than the divisions it saved.)
  // den == 0 (the reference returns early) is folded into this test rather than branched on, so
  // that {n, D} stays one 16-byte load: with a branch in between the compiler fetches D separately,
  // a second round trip on the critical path.
  const float t = num / den;
  if (!(den != 0.0f && t >= kPrecision && t <= best->t)) return;
  // (fetching these three before the test above, for one round trip instead of two, measured equal)
  const float4 q0 = S.tris[4 * k + 1];             // v0, vv
  const float4 q1 = S.tris[4 * k + 2];             // u, uu
  const float4 q2 = S.tris[4 * k + 3];             // v, uv
  const V3 P = point_on_ray(o, d, t);
  const V3 w = P - v3(q0.x, q0.y, q0.z);
  const float wu = dot(w, v3(q1.x, q1.y, q1.z)), wv = dot(w, v3(q2.x, q2.y, q2.z));
  const float uu = q1.w, uv = q2.w, vv = q0.w;
  const float denom = uu * vv - uv * uv;
  const float s1 = vv * wu - uv * wv;
  const float t1 = uu * wv - uv * wu;
  // Same idea for the barycentrics s = s1 / denom, tt = t1 / denom of the reference: for
  // denom > 0 (every triangle outside the wild list), a numerator below -1e-30 * denom gives a
  // quotient that is negative and not flushed to -0, and s1 + t1 > 1.00001 * denom gives
  // s + tt > 1 after all three roundings; both are rejections of the exact test below.
  if (denom > 0.0f && (s1 < -1e-30f * denom || t1 < -1e-30f * denom || s1 + t1 > 1.00001f * denom)) return;
  const float s = s1 / denom;
  const float tt = t1 / denom;
  if (!(s < 0 || tt < 0 || s + tt > 1)) {
    const int face = S.tri_face[k];
    if (t < best->t || (t == best->t && face < best->id)) { best->t = t; best->id = face; }
  }
}

// The triangle test of the traversal kernels is a functor, so that the default path stays exactly the code above.
struct RefTri {
  static __device__ __forceinline__ void test(const DevScene& S, uint32_t k, V3 o, V3 d, Hit* best) { test_triangle(S, k, o, d, best); }
};
// Opt-in (RTX_FLAG_FAST_TRIANGLES): Moeller-Trumbore over a 48-byte {v0, e1, e2, face} record, made watertight by
// overlap -- the edge functions U, V, det - U - V are compared before any division, edges inclusive, with a tolerance
// of 1e-6 |det|, so that neighbouring triangles overlap by a millionth of their size rather than leave a gap (a double
// hit is harmless to a closest-hit search).  One reciprocal instead of the reference's three divisions, fused
// multiply-adds, three loads of which none depends on a comparison, the face id in the record.  Same acceptance rules
// as the reference: t >= 0.001f, (t, face) lexicographic minimum, a NaN anywhere rejects.  NOT the reference's
// arithmetic: t differs by ulps and hits within ~1e-6 of an edge may go to the neighbour (DESIGN.md 3 has the measured
// id-match rates).
struct FastTri {
  static __device__ __forceinline__ void test(const DevScene& S, uint32_t k, V3 o, V3 d, Hit* best) {
    const float4 a = S.tris_mt[3 * k], b = S.tris_mt[3 * k + 1], c = S.tris_mt[3 * k + 2];
    const V3 e1 = v3(a.w, b.x, b.y), e2 = v3(b.z, b.w, c.x);
    const V3 tv = v3(o.x - a.x, o.y - a.y, o.z - a.z);
    // p = d x e2, q = tv x e1
    const float px = __builtin_fmaf(d.y, e2.z, -(d.z * e2.y)), py = __builtin_fmaf(d.z, e2.x, -(d.x * e2.z)), pz = __builtin_fmaf(d.x, e2.y, -(d.y * e2.x));
    const float det = __builtin_fmaf(e1.x, px, __builtin_fmaf(e1.y, py, e1.z * pz));
    const float U = __builtin_fmaf(tv.x, px, __builtin_fmaf(tv.y, py, tv.z * pz));
    const float qx = __builtin_fmaf(tv.y, e1.z, -(tv.z * e1.y)), qy = __builtin_fmaf(tv.z, e1.x, -(tv.x * e1.z)), qz = __builtin_fmaf(tv.x, e1.y, -(tv.y * e1.x));
    const float V = __builtin_fmaf(d.x, qx, __builtin_fmaf(d.y, qy, d.z * qz));
    const float T = __builtin_fmaf(e2.x, qx, __builtin_fmaf(e2.y, qy, e2.z * qz));
    // everything times sign(det): the tests below are those of a front-facing triangle
    const uint32_t sg = __float_as_uint(det) & 0x80000000u;
    const float ad = fabsf(det);
    const float Us = __uint_as_float(__float_as_uint(U) ^ sg), Vs = __uint_as_float(__float_as_uint(V) ^ sg), Ts = __uint_as_float(__float_as_uint(T) ^ sg);
    const float tol = 1e-6f * ad;
    if (!(ad > 0.0f && Us >= -tol && Vs >= -tol && Us + Vs <= ad + tol && Ts > 0.0f)) return;
    const float t = Ts * __builtin_amdgcn_rcpf(ad);
    if (!(t >= kPrecision && t <= best->t)) return;
    const int face = __float_as_int(c.y);
    if (t < best->t || (t == best->t && face < best->id)) { best->t = t; best->id = face; }
  }
};

// Per-lane traversal stack: entries [0, kLdsStack) in LDS (column of this thread, stride
// kBlock, so lane <-> bank and no conflicts), entries beyond in global memory (column of this
// thread, stride = threads in the grid, so a wave's accesses to one level coalesce).
typedef __attribute__((address_space(3))) uint32_t lds_u32;   // LDS word: 32-bit addresses, ds_* instructions
// rank of this lane among the set bits of a wave mask (bits below the lane)
__device__ __forceinline__ uint32_t lane_rank(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
template <int kLds>
struct Stack {
  static constexpr int kLdsEntries = kLds;
  lds_u32* lds;         // &lds_stack[threadIdx.x]
  uint32_t* ovf;        // overflow base (uniform)
  uint32_t ovf_col;     // global thread id = this thread's column
  uint32_t ovf_stride;  // threads in the grid
  __device__ __forceinline__ void push(int sp, uint32_t v) const {
    if (sp < kLds) lds[sp * kBlock] = v;
    else ovf[(uint32_t)(sp - kLds) * ovf_stride + ovf_col] = v;
  }
  __device__ __forceinline__ uint32_t pop(int sp) const {
    return sp < kLds ? lds[sp * kBlock] : ovf[(uint32_t)(sp - kLds) * ovf_stride + ovf_col];
  }
};

// ---------------------------------------------------------------- camera rays
// Camera::screenToWorld + Ray ctor (tucano/camera.hpp:155-173, src/flyscene.cpp:463-465)
// The two screen coordinates of Camera::screenToWorld for raster position (rx, ry): the
// reference evaluates them in double and rounds to float before the aspect / fov scaling.
__device__ __forceinline__ float camera_nx(const DevCamera& C, float rx) {
  return (float)(2.0 * (double)(rx - C.vp[0]) / (double)C.vp[2] - 1.0) * C.aspect_scale;
}
__device__ __forceinline__ float camera_ny(const DevCamera& C, float ry) {
  return (float)(1.0 - 2.0 * (double)(ry - C.vp[1]) / (double)C.vp[3]) * C.scale;
}
// nx depends on the column only and ny on the row only: k_camera_tables evaluates them once
// per render call (w + h values), which keeps the fp64 divisions and their registers out of
// the traversal kernels.
__global__ void k_camera_tables(DevCamera C, float* tab) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < C.w) tab[i] = camera_nx(C, (float)(C.x0 + i) + C.sub[0]);
  else if (i < C.w + C.h) tab[i] = camera_ny(C, (float)(C.y0 + (i - C.w)) + C.sub[1]);
}
// The per-frame records of a batch travel as kernel arguments (no host buffer to keep alive behind an async copy).
constexpr int kMaxBatch = 64;
constexpr int kFramesPerStore = 16;       // records per k_store_frames launch (1 KB of kernel arguments)
struct FrameBatch { float rec[kFramesPerStore][16]; };
__global__ void k_store_frames(FrameBatch B, uint32_t n, float* dst) {
  const uint32_t i = threadIdx.x;
  if (i < 16u * n) dst[i] = B.rec[i >> 4][i & 15];
}
// Ray of pixel (lx, ly) of the rectangle (relative coordinates).
__device__ __forceinline__ void camera_ray_of(const float* M, const float* c, float nx, float ny, V3* o, V3* d1) {
  const float nz = -1.0f;
  V3 w;
  w.x = ((M[0] * nx + M[1] * ny) + M[2] * nz) + M[3] * 1.0f;
  w.y = ((M[4] * nx + M[5] * ny) + M[6] * nz) + M[7] * 1.0f;
  w.z = ((M[8] * nx + M[9] * ny) + M[10] * nz) + M[11] * 1.0f;
  const V3 org = v3(c[0], c[1], c[2]);
  *o = org;
  *d1 = normalized(w - org);
}
__device__ __forceinline__ void camera_ray(const DevCamera& C, uint32_t f, int lx, int ly, V3* o, V3* d1) {
  if (C.nframes > 1) {          // frame f of a batch: its own view and tables
    const float* tab = C.tab + (size_t)f * (uint32_t)(C.w + C.h);
    const float* fr = C.frames + 16u * f;
    camera_ray_of(fr, fr + 12, tab[lx], tab[C.w + ly], o, d1);
    return;
  }
  camera_ray_of(C.inv34, C.origin, C.tab[lx], C.tab[C.w + ly], o, d1);
}
// Output slot p -> pixel (x, y) relative to the rectangle; false for the dead slots of
// partial edge tiles in the tiled-compact layout.
__device__ __forceinline__ bool slot_pixel(const DevCamera& C, uint32_t p, uint32_t* frame, int* px, int* py) {
  *frame = 0;
  if (C.nframes > 1) { *frame = p / C.frame_slots; p -= *frame * C.frame_slots; }
  if (C.ts == 0) { *px = (int)(p % (uint32_t)C.w); *py = (int)(p / (uint32_t)C.w); return true; }
  const uint32_t tt = (uint32_t)(C.ts * C.ts);
  const uint32_t k = p / tt, in = p % tt;
  const uint32_t g = k * (uint32_t)C.nparts + (uint32_t)C.part;
  const int x = (int)(g % (uint32_t)C.tiles_x) * C.ts + (int)(in % (uint32_t)C.ts);
  const int y = (int)(g / (uint32_t)C.tiles_x) * C.ts + (int)(in / (uint32_t)C.ts);
  *px = x; *py = y;
  return x < C.w && y < C.h;
}
// (block, thread) -> output slot: a block covers 16x16 pixels, a wave an 8x8 sub-tile, so
// that the 64 rays of a wave stay coherent.  Returns false where there is no pixel.
__device__ __forceinline__ bool block_slot(const DevCamera& C, uint32_t block, uint32_t tid, uint32_t* frame, uint32_t* slot, int* px, int* py) {
  const uint32_t wave = tid >> 6, lane = tid & 63;
  uint32_t f = 0;
  if (C.nframes > 1) { f = block / C.frame_blocks; block -= f * C.frame_blocks; }
  *frame = f;
  const uint32_t slot0 = f * C.frame_slots;
  const int ix = (int)((wave & 1) * 8 + (lane & 7)), iy = (int)((wave >> 1) * 8 + (lane >> 3));
  if (C.ts == 0) {
    const uint32_t bxn = (uint32_t)(C.w + 15) >> 4;
    const int lx = (int)(block % bxn) * 16 + ix, ly = (int)(block / bxn) * 16 + iy;
    *px = lx; *py = ly; *slot = slot0 + (uint32_t)ly * (uint32_t)C.w + (uint32_t)lx;
    return lx < C.w && ly < C.h;
  }
  const uint32_t sub_n = (uint32_t)(C.ts >> 4);                 // 16x16 blocks per tile side
  const uint32_t k = block / (sub_n * sub_n), sb = block % (sub_n * sub_n);
  const int tx = (int)(sb % sub_n) * 16 + ix, ty = (int)(sb / sub_n) * 16 + iy;
  const uint32_t g = k * (uint32_t)C.nparts + (uint32_t)C.part;
  const int x = (int)(g % (uint32_t)C.tiles_x) * C.ts + tx, y = (int)(g / (uint32_t)C.tiles_x) * C.ts + ty;
  *px = x; *py = y; *slot = slot0 + k * (uint32_t)(C.ts * C.ts) + (uint32_t)(ty * C.ts + tx);
  return x < C.w && y < C.h;
}

// ---------------------------------------------------------------- kernels
// Persistent waves: the grid is sized to what is resident (7 or 8 blocks of 4 waves per CU).  A
// block owns a fixed, image-wide scattered set of 64-ray units and its four waves (one per SIMD
// of the CU) draw from it through a counter in LDS (struct Work): no global atomics.
template <int kLds>
__device__ __forceinline__ Stack<kLds> make_stack(uint32_t* lds, uint32_t* ovf) {
  Stack<kLds> st;
  st.lds = (lds_u32*)lds + threadIdx.x;
  st.ovf = ovf;
  st.ovf_col = blockIdx.x * kBlock + threadIdx.x;
  st.ovf_stride = gridDim.x * kBlock;
  return st;
}
// Work distribution: block b owns the unit quads b, b + B, b + 2B, ... (B = blocks in the grid; a
// quad = the four 8x8 tiles of a 16x16 pixel block, or 256 consecutive queue rays) and its waves
// take the next unit of that list with an LDS atomic, so the four SIMDs of a CU finish together
// whatever the single units cost.  Measured alternatives: fully static lists per
// wave (1.5-3% slower: the longest wave of a block runs on alone), one global cursor, 64
// per-stripe global counters and a global pool for the last quarter (agent-scope atomics run at
// ~90 per us on this part whatever the address, and serialise).
struct Work {
  uint32_t n, grab;
  lds_u32* ctr;       // the block's unit counter (LDS)
  __device__ __forceinline__ bool next(uint32_t* base) {
    uint32_t i = 0;
    if ((threadIdx.x & 63u) == 0) i = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
    // the block's units: quads of consecutive units (the four 8x8 tiles of a 16x16 pixel block), a grid apart
    const uint32_t unit = ((i >> 2) * gridDim.x + blockIdx.x) * 4u + (i & 3u);
    *base = unit * grab;
    return unit < (n + grab - 1u) / grab;
  }
};
__device__ __forceinline__ Work make_work(uint32_t n, uint32_t grab, uint32_t* ctr) {
  Work w; w.n = n; w.grab = grab; w.ctr = (lds_u32*)ctr;
  if (threadIdx.x == 0) *ctr = 0;
  __syncthreads();
  return w;
}

// One traversal step over the packed group of 4 sibling nodes *cur: slab tests against the
// padded (and outward-quantised) boxes, near-to-far order, the other hits pushed far-to-near.  Returns false when no
// child is hit (the caller pops); else *cur = the nearest child.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u4;
// The top levels of the hierarchy in LDS: every ray walks them, and a lane that reads them there does not
// wait for (or occupy) the L1 path.  Loaded once per block.
template <int kTop>
struct TopLevels { static constexpr int kMax = kTop; const lds_u4* lds; uint32_t n; };
template <int kTop>
__device__ __forceinline__ TopLevels<kTop> load_top_levels(const DevScene& S, uint4* lds_top) {
  TopLevels<kTop> T; T.lds = (const lds_u4*)lds_top; T.n = 0;
  if constexpr (kTop > 0) {
    T.n = S.ntop < (uint32_t)kTop ? S.ntop : (uint32_t)kTop;
    for (uint32_t i = threadIdx.x; i < 4u * T.n; i += kBlock) lds_top[i] = S.nodes[i];
    __syncthreads();
  }
  return T;
}
// kOrdered = false (rays whose first hit decides: the shadow rays of a scene where any blocker shades alike): the children a ray
// enters are taken in slot order, without the sort -- which blocker such a ray finds does not show in the result, and a ray
// that finds none walks the same boxes in any order.
template <bool kOrdered, class StackT, class TopT>
__device__ __forceinline__ bool node_step(const DevScene& S, const TopT& top, const StackT& stk, float lim, float ivx, float ivy, float ivz, float oix,
                                          float oiy, float oiz, uint32_t* cur, int* sp_io) {
  int sp = *sp_io;
  // one 64-byte packed group of 4 sibling nodes per step: from LDS for the top levels, else from memory (uniform
  // base + 32-bit byte offset)
  uint4 h0, h1, h2, h3;
#if RTX_TOP_UNIFORM
  // (only when every lane of the wave that steps is in the top levels: one branch for the wave, and the loads
  //  from memory -- which cost the L1 path the same however few lanes take part -- are then really skipped)
  if (TopT::kMax > 0 && __ballot(*cur >= top.n) == 0) {
#else
  if (TopT::kMax > 0 && *cur < top.n) {
#endif
    const lds_u4* q = top.lds + (*cur << 2);
    const u32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    h0 = make_uint4(a.x, a.y, a.z, a.w); h1 = make_uint4(b.x, b.y, b.z, b.w);
    h2 = make_uint4(c.x, c.y, c.z, c.w); h3 = make_uint4(d.x, d.y, d.z, d.w);
  } else {
    const uint4* q = (const uint4*)((const char*)S.nodes + (*cur << 6));
    h0 = q[0]; h1 = q[1]; h2 = q[2]; h3 = q[3];
  }
  // planes are origin + byte * step per axis; fold origin, step and the ray into one fma per plane
  const float ax = __uint_as_float(h0.w) * ivx, ay = __uint_as_float(h3.z) * ivy, az = __uint_as_float(h3.w) * ivz;
  const float bx = __builtin_fmaf(__uint_as_float(h0.x), ivx, -oix);
  const float by = __builtin_fmaf(__uint_as_float(h0.y), ivy, -oiy);
  const float bz = __builtin_fmaf(__uint_as_float(h0.z), ivz, -oiz);
  const uint32_t ents[4] = {h1.x, h1.y, h1.z, h1.w};
  // Which plane of each axis the ray enters through depends on the sign of the direction only:
  // pick the words once per step instead of a min and a max per child and axis.  An unused slot
  // is encoded inverted (lower planes 255, upper planes 0), so that its entry lies beyond its
  // exit whatever the signs -- no separate "empty" test.  (That holds for finite arithmetic; a ray with NaN or
  // overflowing coordinates fails every comparison and can "hit" such a slot, so its entry word is a leaf of
  // no triangles, harmless to follow.)
  const bool nx = ivx < 0.0f, ny = ivy < 0.0f, nz = ivz < 0.0f;
  const uint32_t wnx = nx ? h2.w : h2.x, wfx = nx ? h2.x : h2.w;
  const uint32_t wny = ny ? h3.x : h2.y, wfy = ny ? h2.y : h3.x;
  const uint32_t wnz = nz ? h3.y : h2.z, wfz = nz ? h2.z : h3.y;
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az}, bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
  float tn[4]; uint32_t en[4]; bool hit[4];
#pragma unroll
  for (int k = 0; k < 4; k += 2) {
    const int s0 = 8 * k, s1 = 8 * k + 8;
    // two children per packed fma
    const f2 qnx = {(float)((wnx >> s0) & 0xFFu), (float)((wnx >> s1) & 0xFFu)}, qfx = {(float)((wfx >> s0) & 0xFFu), (float)((wfx >> s1) & 0xFFu)};
    const f2 qny = {(float)((wny >> s0) & 0xFFu), (float)((wny >> s1) & 0xFFu)}, qfy = {(float)((wfy >> s0) & 0xFFu), (float)((wfy >> s1) & 0xFFu)};
    const f2 qnz = {(float)((wnz >> s0) & 0xFFu), (float)((wnz >> s1) & 0xFFu)}, qfz = {(float)((wfz >> s0) & 0xFFu), (float)((wfz >> s1) & 0xFFu)};
    const f2 tnx = __builtin_elementwise_fma(qnx, ax2, bx2), tfx = __builtin_elementwise_fma(qfx, ax2, bx2);
    const f2 tny = __builtin_elementwise_fma(qny, ay2, by2), tfy = __builtin_elementwise_fma(qfy, ay2, by2);
    const f2 tnz = __builtin_elementwise_fma(qnz, az2, bz2), tfz = __builtin_elementwise_fma(qfz, az2, bz2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // entry clamped to the ray start, exit to the pruning limit: hit <=> entry <= exit
      const float n_ = fmaxf(fmaxf(tnx[j], tny[j]), fmaxf(tnz[j], 0.0f));
      const float f_ = fminf(fminf(tfx[j], tfy[j]), fminf(tfz[j], lim));
      en[k + j] = ents[k + j];
      hit[k + j] = n_ <= f_;
      tn[k + j] = hit[k + j] ? n_ : INFINITY;
    }
  }
  if constexpr (!kOrdered) {
    // (the comparison itself, not "entry distance is finite": one select per child less)
    const bool h0 = hit[0], h1 = hit[1], h2 = hit[2], h3 = hit[3];
    if (sp <= StackT::kLdsEntries - 4) {
      // four unconditional LDS stores, the stack pointer advancing over the real ones; the last of them is taken at once
      stk.lds[sp * kBlock] = en[0]; sp += h0 ? 1 : 0;
      stk.lds[sp * kBlock] = en[1]; sp += h1 ? 1 : 0;
      stk.lds[sp * kBlock] = en[2]; sp += h2 ? 1 : 0;
      stk.lds[sp * kBlock] = en[3]; sp += h3 ? 1 : 0;
      if (!(h0 || h1 || h2 || h3)) { *sp_io = sp; return false; }
      --sp;
    } else {
      // (near the end of the LDS part: push all but the last hit, so that no more than three entries are added per level)
      const int last = h3 ? 3 : h2 ? 2 : h1 ? 1 : h0 ? 0 : -1;
      if (last < 0) { *sp_io = sp; return false; }
      if (h0 && last != 0) { stk.push(sp, en[0]); ++sp; }
      if (h1 && last != 1) { stk.push(sp, en[1]); ++sp; }
      if (h2 && last != 2) { stk.push(sp, en[2]); ++sp; }
    }
    *sp_io = sp;
    *cur = h3 ? en[3] : h2 ? en[2] : h1 ? en[1] : en[0];
    return true;
  }
  // sort the 4 candidates near-to-far; misses carry tn = +inf.  The nearest and the farthest are exact
  // after four compare-exchanges; the order of the two in between only matters for efficiency (RTX_SORT_CX).
#define RTX_CX(i, j) { const bool sw = tn[j] < tn[i]; const float tt_ = sw ? tn[j] : tn[i]; tn[j] = sw ? tn[i] : tn[j]; tn[i] = tt_; \
                   const uint32_t ee_ = sw ? en[j] : en[i]; en[j] = sw ? en[i] : en[j]; en[i] = ee_; }
  RTX_CX(0, 1) RTX_CX(2, 3) RTX_CX(0, 2)
#if RTX_SORT_CX > 3
  RTX_CX(1, 3)
#endif
#if RTX_SORT_CX > 4
  RTX_CX(1, 2)
#endif
#undef RTX_CX
  // far-to-near pushes of the other hits.  Fast path: three unconditional LDS stores, the
  // stack pointer only advances over the real ones (a miss is overwritten by the next store
  // or left above the top).
  if (sp <= StackT::kLdsEntries - 3) {
    stk.lds[sp * kBlock] = en[3]; sp += (tn[3] != INFINITY) ? 1 : 0;
    stk.lds[sp * kBlock] = en[2]; sp += (tn[2] != INFINITY) ? 1 : 0;
    stk.lds[sp * kBlock] = en[1]; sp += (tn[1] != INFINITY) ? 1 : 0;
  } else {
    if (tn[3] != INFINITY) { stk.push(sp, en[3]); ++sp; }
    if (tn[2] != INFINITY) { stk.push(sp, en[2]); ++sp; }
    if (tn[1] != INFINITY) { stk.push(sp, en[1]); ++sp; }
  }
  *sp_io = sp;
  if (tn[0] == INFINITY) return false;
  *cur = en[0];
  return true;
}

// Eigen::AngleAxisf(theta, axis).toRotationMatrix() (Eigen/src/Geometry/AngleAxis.h:218-243), row-major
__device__ __forceinline__ void angle_axis_matrix(float s, float c, V3 axis, float* m) {
  const V3 sin_axis = s * axis;
  const V3 cos1_axis = (1.0f - c) * axis;
  float tmp;
  tmp = cos1_axis.x * axis.y; m[1] = tmp - sin_axis.z; m[3] = tmp + sin_axis.z;
  tmp = cos1_axis.x * axis.z; m[2] = tmp + sin_axis.y; m[6] = tmp - sin_axis.y;
  tmp = cos1_axis.y * axis.z; m[5] = tmp - sin_axis.x; m[7] = tmp + sin_axis.x;
  m[0] = cos1_axis.x * axis.x + c; m[4] = cos1_axis.y * axis.y + c; m[8] = cos1_axis.z * axis.z + c;
}
// Position of light sample ls seen from hit point P: the point lights, then per spherical
// light the 4 ring points of Shader::getNPointsOnCircle + its centre (src/flyscene.cpp:616-626,
// src/ShaderLib.h:11-31).  *fifth: the sample's colour is divided by 5 by the caller.
__device__ __forceinline__ V3 light_sample_s(const DevScene& S, V3 P, uint32_t ls, bool* fifth) {
  if (ls < S.nlights) { const float4 L = S.lights[ls]; *fifth = false; return v3(L.x, L.y, L.z); }
  *fifth = true;
  const uint32_t j = (ls - S.nlights) / 5u, k = (ls - S.nlights) % 5u;
  const float4 sl = S.slights[j];
  const V3 center = v3(sl.x, sl.y, sl.z);
  if (k == 4u) return center;
  const V3 normal = normalized(P - center);
  V3 not_normal = v3(1.0f, 0.0f, 0.0f);
  const bool eq = (normal.x == 1.0f && normal.y == 0.0f && normal.z == 0.0f) || (-normal.x == 1.0f && -normal.y == 0.0f && -normal.z == 0.0f);
  if (eq) not_normal = normalized(v3(1.0f, 1.0f, 0.0f));
  V3 rv = normalized(cross(normal, not_normal)) * sl.w;
  float m[9];
  angle_axis_matrix(S.sl_sin, S.sl_cos, normal, m);
  for (uint32_t i = 0; i < k; ++i) rv = mat3_mul(m, rv);
  return rv + center;
}

// Where a wave's rays come from.  get(j): ray j of the launch -> origin, Ray direction d1 and
// the output slot of its hit record; false where index j holds no ray.
// Traversal bound of a shadow ray: canSeeLight compares hit distances with |light - P|
// (src/flyscene.cpp:751-778); anything farther than that (with slack for the ulp-level spread
// between slab and plane distances, as in kPruneSlack) cannot change its answer.
__device__ __forceinline__ float shadow_tmax(V3 light, V3 P) { return norm(light - P) * 1.001f; }
// *tmax: hits beyond it do not matter to the caller (shadow rays: the light's distance with
// slack; canSeeLight gives the same answer for "closest hit beyond the light" and "no hit"),
// so traversal starts pruned at tmax; a ray that finds nothing reports t = INFINITY as always.
struct CameraSource {
  DevCamera C;
  __device__ __forceinline__ bool get(uint32_t j, V3* o, V3* d1, uint32_t* slot, float* tmax) const {
    int lx, ly; uint32_t f;
    *tmax = INFINITY;
    if (!block_slot(C, j / kBlock, j % kBlock, &f, slot, &lx, &ly)) return false;
    camera_ray(C, f, lx, ly, o, d1);
    return true;
  }
  __device__ __forceinline__ bool first_hit_decides() const { return false; }
  __device__ __forceinline__ bool decides(uint32_t, V3, float) const { return false; }
};
// Shadow rays of a node with several light samples: ray (parent p, sample ls) lives at shadow_index(p, ls, NLS) --
// light-major within groups of 64 parents, so that the 64 rays of a wave unit go from 64 neighbouring hit points to
// ONE light (parent-major, a unit would be 4 hit points x 16 lights: 16 directions in one wave).  The queue then
// spans roundup64(parents) * NLS entries, of which those of parents beyond the count hold nothing.
__device__ __forceinline__ size_t shadow_index(uint32_t parent, uint32_t ls, uint32_t nls) {
  return (size_t)(parent >> 6) * 64u * nls + (size_t)ls * 64u + (parent & 63u);
}
// Index arithmetic of a queue in shadow_index order: (parent, light sample) of entry j.
struct ShadowLayout {
  uint32_t tile_stride = 0;   // != 0: shadow_index layout with this many samples per parent, nparents parents
  uint32_t nparents = 0;
  uint32_t magic = 0;         // ceil(2^32 / tile_stride) where (j >> 6) / tile_stride == mulhi(j >> 6, magic); 0: divide
  __device__ __forceinline__ void set_layout(uint32_t stride, uint32_t parents) {
    tile_stride = stride; nparents = parents;
    // exact while (j >> 6) * tile_stride < 2^32: queues are below 2^32 entries, so for up to 64 samples per parent
    magic = (stride > 1u && stride <= 64u) ? 0xFFFFFFFFu / stride + 1u : 0u;
  }
  __device__ __forceinline__ uint32_t unit_parent(uint32_t unit) const {      // (unit = j >> 6) / tile_stride
    return tile_stride == 1u ? unit : magic ? __umulhi(unit, magic) : unit / tile_stride;
  }
  __device__ __forceinline__ uint32_t parent_of(uint32_t j) const { return unit_parent(j >> 6) * 64u + (j & 63u); }
  __device__ __forceinline__ uint32_t sample_of(uint32_t j) const { return tile_stride == 1u ? 0u : (j >> 6) - unit_parent(j >> 6) * tile_stride; }
};
// Does a hit at distance t end the search of a shadow ray whose first hit decides (t clearly before the light: tmax = 1.001 |light - P|,
// and a hit below 0.998 tmax is before the light whatever the rounding)?  o: the ray's origin, which IS the hit point P the ray was
// formed from.  For a point light the test is made on squares, t^2 < 0.995 |light - o|^2, i.e. t < 0.9975 |light - o| -- a little
// stricter, no square root; which hits end a search early does not show in the result (any blocker shades alike).
__device__ __forceinline__ bool shadow_hit_decides(const DevScene& S, uint32_t ls, V3 o, float t) {
  if (ls < S.nlights) {
    const float4 L = S.lights[ls];
    const V3 v = v3(L.x, L.y, L.z) - o;
    return t * t < 0.995f * dot(v, v);
  }
  bool fifth;
  return t < 0.998f * shadow_tmax(light_sample_s(S, o, ls, &fifth), o);
}
// The shadow ray from hit point P towards light sample ls, as canSeeLight forms it: Ray(position, toLight.normalized()) normalises
// twice (src/flyscene.cpp:749-750).
__device__ __forceinline__ void shadow_ray_of(const DevScene& S, V3 P, uint32_t ls, V3* o, V3* d1, float* tmax) {
  bool fifth;
  const V3 L = light_sample_s(S, P, ls, &fifth);
  *o = P; *d1 = normalized(normalized(L - P)); *tmax = shadow_tmax(L, P);
}
struct QueueSource : ShadowLayout {
  const float4* rays;
  uint32_t bounded;    // rd.w of the records is tmax (shadow queue) rather than a throughput
  uint32_t anyhit;     // shadow queue of a scene whose materials all shade a blocked light alike: any
                       // hit clearly before the light gives canSeeLight's answer *and* colour, so the
                       // ray stops there instead of looking for the closest one
  // The pipeline's shadow queue (round 3): `rays` holds ONE 16-byte record per parent, its hit point; the ray of index j
  // (shadow_index layout: 64 parents x one light sample per wave unit) is formed here from that point and light sample
  // (j / 64) % tile_stride -- the expression k_shade used to evaluate and store as 64 bytes per ray.
  const DevScene* scene = nullptr;      // != nullptr: compact records
  __device__ __forceinline__ bool first_hit_decides() const { return anyhit != 0; }
  __device__ __forceinline__ bool decides(uint32_t slot, V3 o, float t) const {
    if (scene) return shadow_hit_decides(*scene, sample_of(slot), o, t);     // (compact records hold no tmax)
    return t < 0.998f * rays[2 * (size_t)slot + 1].w;
  }
  __device__ __forceinline__ bool get(uint32_t j, V3* o, V3* d1, uint32_t* slot, float* tmax) const {
    if (tile_stride != 0 && parent_of(j) >= nparents) return false;
    if (scene) {
      const float4 p = rays[parent_of(j)];
      shadow_ray_of(*scene, v3(p.x, p.y, p.z), sample_of(j), o, d1, tmax);       // (the light is the same for the 64 rays of a unit)
      *slot = j;
      return true;
    }
    const float4 ro = rays[2 * (size_t)j], rd = rays[2 * (size_t)j + 1];
    *o = v3(ro.x, ro.y, ro.z); *d1 = v3(rd.x, rd.y, rd.z); *slot = j;
    *tmax = bounded ? rd.w : INFINITY;
    return true;
  }
};

// The shadow rays of a camera frame WITHOUT a parent list (frames whose nodes spawn no children: no reflection / refraction / GI):
// queue index j is (output slot p, light sample ls) in shadow_index order over ALL slots of the call; a slot whose camera ray hit
// something gives a ray -- its hit point is formed again from the camera ray and the hit distance, the expressions of surface_of --
// and the others give none.  The refilling scheduler's parked-ray buffer does the compaction k_shade's two passes did.
struct DirectSource : ShadowLayout {
  const DevCamera* C;       // device copy: read in the set-up of a chunk only
  const uint2* hits0;       // closest hits of the camera rays, per slot (dead slots of the tiled layout hold anything)
  const DevScene* scene;
  uint32_t anyhit;
  __device__ __forceinline__ bool first_hit_decides() const { return anyhit != 0; }
  __device__ __forceinline__ bool decides(uint32_t slot, V3 o, float t) const { return shadow_hit_decides(*scene, sample_of(slot), o, t); }
  __device__ __forceinline__ bool get(uint32_t j, V3* o, V3* d1, uint32_t* slot, float* tmax) const {
    const uint32_t p = parent_of(j);
    if (p >= nparents) return false;
    const uint2 h = hits0[p];
    const int id = (int)h.y;
    uint32_t f; int x, y;
    if (!slot_pixel(*C, p, &f, &x, &y) || id == -1) return false;
    V3 co, cd1;
    camera_ray(*C, f, x, y, &co, &cd1);
    const V3 P = point_on_ray(co, id >= 0 ? normalized(cd1) : cd1, __uint_as_float(h.x));      // surface_of's hit point
    shadow_ray_of(*scene, P, sample_of(j), o, d1, tmax);
    *slot = j;
    return true;
  }
};

// ---------------------------------------------------------------- closest hit, one wave
// Flyscene::getIntersections (src/flyscene.cpp:517-585) for a stream of rays, scheduled for
// a 64-wide wave:
//  * setup (spheres, the Ray re-normalisation, the reference's root-box gate, the "wild"
//    triangles) runs for 64 rays at a time with all lanes busy; rays that still need the BVH
//    are parked in a small per-wave LDS buffer (wbuf, 9 words x 64);
//  * lanes whose ray has finished take a parked ray (refill) once kRefillMin lanes are idle,
//    instead of idling until the slowest ray of a 64-ray group is done;
//  * triangle tests run in leaf phases, one triangle per lane and pass, when kDeferMin lanes hold a
//    leaf (or kBlockMin hold one and have nothing else to do, or no lane has node work); a lane
//    that reaches a leaf in between keeps it in a one-entry pending slot and goes on with its
//    next stack entry, so that the expensive tests run with more lanes active and the node steps
//    with fewer lanes waiting.
// None of this changes a ray's result: every triangle whose box the ray enters before its final hit
// distance is still tested, only later (closest hits are order independent: (t, faceId) minimum);
// scripts/wave_replay.py replays real traversals under these policies on the CPU.
constexpr uint32_t kDone = 0x7FFFFFFFu;   // lane holds no ray (never a valid entry: group indices stay below 2^26)
constexpr int kRefillMin = RTX_REFILL_MIN;
constexpr int kDeferMin = RTX_DEFER_MIN;
constexpr int kBlockMin = RTX_BLOCK_MIN;
constexpr int kWbufWords = 9;             // o.xyz, d.xyz, best.t, best.id, slot

template <class Tri, bool kOrdered, class Source>
__device__ __forceinline__ void trace_wave(const DevScene& S, const TopLevels<kTopGroupsRefill>& top, const Source& src, Work work, const Stack<kLdsStack>& stk, lds_u32* wbuf,
                                           uint32_t tri_min, uint2* hits) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t nbuf = 0;                       // parked rays (wave-uniform)
  uint32_t chunk = 0, chunk_end = 0;       // rest of the current grab (wave-uniform)
  bool more = true;
  V3 o = v3(0, 0, 0), d = v3(0, 0, 0);
  float ivx = 0, ivy = 0, ivz = 0, oix = 0, oiy = 0, oiz = 0;
  Hit best; best.t = INFINITY; best.id = -1;
  uint32_t cur = kDone, pend = kDone, slot = 0;   // pend: a leaf reached outside a leaf phase, tested at the next one
  int sp = 0;

  for (;;) {
    // entry kinds: leaf = top bit set, node group = below kDone, none = kDone; a lane is free when it holds neither
    uint64_t act = __ballot(cur != kDone || pend != kDone);
    const uint32_t nidle = 64u - (uint32_t)__popcll(act);
    if ((nidle >= (uint32_t)kRefillMin || act == 0) && (nbuf != 0 || more)) {
      while (nbuf == 0 && more) {
        if (chunk >= chunk_end) {
          uint32_t base;
          more = work.next(&base);
          if (!more) break;
          chunk = base;
          chunk_end = base + work.grab < work.n ? base + work.grab : work.n;
        }
        const uint32_t j = chunk + lane;
        chunk += 64;
        // ---- setup of ray j on every lane
        V3 so, sd1; uint32_t sslot = 0;
        bool park = false;
        Hit sb; sb.t = INFINITY; sb.id = -1;
        V3 sd = v3(0, 0, 0);
        if (j < chunk_end && src.get(j, &so, &sd1, &sslot, &sb.t)) {
          const float tmax0 = sb.t;
          // spheres first, strict < (src/flyscene.cpp:523-532)
          for (uint32_t i = 0; i < S.nspheres; ++i) {
            const float t = ref_sphere_t(so, sd1, S.spheres[i]);
            if (t != INFINITY && t < sb.t) { sb.t = t; sb.id = -2 - (int)i; }
          }
          if (S.ntris != 0) {
            // intersectBox / intersectTriangle rebuild the Ray, re-normalising d1 (src/flyscene.cpp:845,965)
            sd = normalized(sd1);
            if ((S.flags & 1u) || ref_box_t(so, sd, S.root_lo, S.root_hi) != INFINITY) {
              for (uint32_t k = S.ntris - S.nwild; k < S.ntris; ++k) Tri::test(S, k, so, sd, &sb);
              park = S.nwild != S.ntris && ray_is_finite(so, sd);
            }
          }
          if (src.first_hit_decides() && sb.id != -1 && sb.t < 0.998f * tmax0) park = false;
          if (!park) hits[sslot] = make_uint2(__float_as_uint(sb.id == -1 ? INFINITY : sb.t), (uint32_t)sb.id);
        }
        const uint64_t pm = __ballot(park);
        if (park) {
          lds_u32* w = wbuf + lane_rank(pm);
          w[0] = __float_as_uint(so.x); w[64] = __float_as_uint(so.y); w[128] = __float_as_uint(so.z);
          w[192] = __float_as_uint(sd.x); w[256] = __float_as_uint(sd.y); w[320] = __float_as_uint(sd.z);
          w[384] = __float_as_uint(sb.t); w[448] = (uint32_t)sb.id; w[512] = sslot;
        }
        nbuf = (uint32_t)__popcll(pm);
        __builtin_amdgcn_wave_barrier();
      }
      if (nbuf != 0) {
        const uint32_t r = lane_rank(~act);      // rank among the free lanes
        if (!((act >> lane) & 1ull) && r < nbuf) {
          const lds_u32* w = wbuf + (nbuf - 1u - r);
          o = v3(__uint_as_float(w[0]), __uint_as_float(w[64]), __uint_as_float(w[128]));
          d = v3(__uint_as_float(w[192]), __uint_as_float(w[256]), __uint_as_float(w[320]));
          best.t = __uint_as_float(w[384]); best.id = (int)w[448]; slot = w[512];
          // slab constants for the padded BVH boxes (culling only; not part of the parity
          // contract).  |1/d| is clamped so that an axis-parallel ray gives huge finite slab
          // distances of the right sign instead of inf - inf = NaN in the fused form below.
          ivx = cull_inv(d.x); ivy = cull_inv(d.y); ivz = cull_inv(d.z);
          oix = o.x * ivx; oiy = o.y * ivy; oiz = o.z * ivz;
          cur = S.root_entry; sp = 0;
        }
        nbuf -= nidle < nbuf ? nidle : nbuf;
        __builtin_amdgcn_wave_barrier();
      }
      act = __ballot(cur != kDone || pend != kDone);
    }
    if (act == 0) break;                   // no ray in flight, nothing parked, nothing left to set up

    // Node phase: every lane at a node group steps.  Leaf phase: lanes holding a leaf test it --
    // run only when enough lanes hold one (kDeferMin), enough lanes can do nothing else
    // (kBlockMin), or no lane has node work.  In between a lane that reaches a leaf keeps it in
    // `pend` and goes on with its next stack entry: the pending leaf's hit would have pruned a few
    // of those steps (+2 % node steps, CPU replay in scripts/wave_replay.py), but the lane would
    // have idled through them anyway, and closest hits do not depend on the order of the tests.
    const bool alive = cur != kDone || pend != kDone;
#if RTX_NODE_UNROLL > 1
    // Extra node steps per iteration: the bookkeeping below (ballots, leaf-phase decision, refill check) is then paid once per
    // RTX_NODE_UNROLL steps; a lane that reaches a leaf on the way parks it in the pending slot and goes on, as always.
#pragma unroll
    for (int rep = 1; rep < RTX_NODE_UNROLL; ++rep) {
      if (cur < kDone) {
        bool p1 = !node_step<kOrdered>(S, top, stk, best.t * kPruneSlack, ivx, ivy, ivz, oix, oiy, oiz, &cur, &sp);
        if (RTX_SPEC != 0 && (RTX_SPEC == 1 || !src.first_hit_decides()) && (int)cur < 0 && pend == kDone) { pend = cur; p1 = true; }
        if (p1) { if (sp == 0) cur = kDone; else { --sp; cur = stk.pop(sp); } }
      }
    }
#endif
    const bool at_node = cur < kDone;
    const bool has_pend = pend != kDone;
    const bool want_leaf = has_pend || (int)cur < 0;
    const uint64_t leaf_mask = __ballot(want_leaf);
    const uint64_t node_mask = __ballot(at_node);
    bool pop = false;
    if (at_node) {
      pop = !node_step<kOrdered>(S, top, stk, best.t * kPruneSlack, ivx, ivy, ivz, oix, oiy, oiz, &cur, &sp);
    }
    if (leaf_mask != 0 && (node_mask == 0 || __popcll(leaf_mask) >= kDeferMin || __popcll(leaf_mask & ~node_mask) >= kBlockMin)) {
      // One triangle per lane and pass; passes go on while at least tri_min lanes still have
      // triangles of their leaf left (1: whole leaves).  What is left after that stays pending and
      // the lane goes on with node work meanwhile: on a triangle soup, where few tests hit, the wave
      // is not held for the biggest leaf (shadow kernel -5 %); on closed surfaces the late tests
      // cost more pruning than they save (-2 to -7 %), so the host picks per scene and launch kind.
      const uint32_t leaf = has_pend ? pend : cur;
      uint32_t first = leaf & 0x0FFFFFFFu, cnt = want_leaf ? (leaf >> 28) & 7u : 0u;
      do {
        if (cnt != 0) { Tri::test(S, first, o, d, &best); ++first; --cnt; }
      } while ((uint32_t)__popcll(__ballot(cnt != 0)) >= tri_min);
      if (want_leaf) {
        const uint32_t rest = cnt != 0 ? (0x80000000u | (cnt << 28) | first) : kDone;
        if (has_pend) pend = rest; else if (cnt != 0) cur = rest; else pop = true;
        // tmax = 1.001 |light - P|: a hit below 0.998 tmax is before the light whatever the rounding
        if (src.first_hit_decides() && best.id != -1 && src.decides(slot, o, best.t)) { sp = 0; cur = kDone; pend = kDone; pop = false; }
      }
    }
    if (RTX_SPEC != 0 && (RTX_SPEC == 1 || !src.first_hit_decides()) && (int)cur < 0 && pend == kDone) { pend = cur; pop = true; }
    if (pop) {
      if (sp == 0) cur = kDone; else { --sp; cur = stk.pop(sp); }
    }
    if (alive && cur == kDone && pend == kDone) hits[slot] = make_uint2(__float_as_uint(best.id == -1 ? INFINITY : best.t), (uint32_t)best.id);
  }
}

// The same per-ray work scheduled group by group: a wave sets 64 rays up, walks them until the
// slowest is done, then takes the next 64.  No parking, no refill, no deferral -- less
// bookkeeping per step, which wins where the rays of a group stay in step (coherent rays on
// closed surfaces); the refilling scheduler above wins on incoherent work (triangle soups,
// diffuse bounces).
template <class Tri, bool kOrdered, class Source>
__device__ __forceinline__ void trace_wave_groups(const DevScene& S, const TopLevels<kTopGroups>& top, const Source& src, Work work, const Stack<kLdsStackGroups>& stk,
                                                  uint2* hits) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t base;
  while (work.next(&base)) {
    const uint32_t end = base + work.grab < work.n ? base + work.grab : work.n;
    for (uint32_t j = base + lane; j < end; j += 64) {
      V3 o, d1; uint32_t slot;
      Hit best; best.id = -1;
      if (!src.get(j, &o, &d1, &slot, &best.t)) continue;
      const float tmax0 = best.t;
      for (uint32_t i = 0; i < S.nspheres; ++i) {
        const float t = ref_sphere_t(o, d1, S.spheres[i]);
        if (t != INFINITY && t < best.t) { best.t = t; best.id = -2 - (int)i; }
      }
      bool walk = false;
      V3 d = v3(0, 0, 0);
      if (S.ntris != 0) {
        d = normalized(d1);
        if ((S.flags & 1u) || ref_box_t(o, d, S.root_lo, S.root_hi) != INFINITY) {
          for (uint32_t k = S.ntris - S.nwild; k < S.ntris; ++k) Tri::test(S, k, o, d, &best);
          walk = S.nwild != S.ntris && ray_is_finite(o, d);
        }
      }
      if (src.first_hit_decides() && best.id != -1 && best.t < 0.998f * tmax0) walk = false;
      if (walk) {
        const float ivx = cull_inv(d.x), ivy = cull_inv(d.y), ivz = cull_inv(d.z);
        const float oix = o.x * ivx, oiy = o.y * ivy, oiz = o.z * ivz;
        uint32_t cur = S.root_entry;
        int sp = 0;
        for (;;) {
          if ((int)cur >= 0) {
            if (node_step<kOrdered>(S, top, stk, best.t * kPruneSlack, ivx, ivy, ivz, oix, oiy, oiz, &cur, &sp)) continue;
          } else {
            const uint32_t first = cur & 0x0FFFFFFFu, cnt = (cur >> 28) & 7u;
            for (uint32_t k = first; k < first + cnt; ++k) Tri::test(S, k, o, d, &best);
            if (src.first_hit_decides() && best.id != -1 && best.t < 0.998f * tmax0) break;
          }
          if (sp == 0) break;
          --sp;
          cur = stk.pop(sp);
        }
      }
      hits[slot] = make_uint2(__float_as_uint(best.id == -1 ? INFINITY : best.t), (uint32_t)best.id);
    }
  }
}


// ---------------------------------------------------------------- brute force (domain guard, verification)
// The hierarchy's boxes are padded for the rounding of o + t d up to origins ~30 scene extents away (rtx.hip,
// rtx_scene_build).  A ray from beyond that is answered the way the reference answers every ray -- all triangles,
// no boxes -- by a whole block: 256 threads share the triangles of one ray and reduce to the (t, face) minimum.
// The same routine re-checks sampled rays of any launch when the scene asks for verification.
__device__ __forceinline__ bool out_of_domain(V3 o, float domain) {
  return fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) > domain;       // (false for NaN: such a ray hits nothing anyway)
}
__device__ __forceinline__ bool hit_before(const Hit& a, const Hit& b) {   // does a win over b in getIntersections' reduction?
  if (a.id == -1) return false;
  if (b.id == -1) return true;
  if (a.t != b.t) return a.t < b.t;
  if ((a.id < 0) != (b.id < 0)) return a.id < 0;                            // a sphere found first keeps an equal t (strict <)
  return a.id >= 0 ? a.id < b.id : a.id > b.id;                             // lowest face id / first sphere
}
// All threads of the block call this with the same ray; returns the closest hit over spheres + gate + every triangle.
__device__ __forceinline__ Hit brute_closest(const DevScene& S, V3 o, V3 d1, float tmax, Hit* red) {
  Hit best; best.t = tmax; best.id = -1;
  for (uint32_t i = 0; i < S.nspheres; ++i) {
    const float t = ref_sphere_t(o, d1, S.spheres[i]);
    if (t != INFINITY && t < best.t) { best.t = t; best.id = -2 - (int)i; }
  }
  if (S.ntris != 0) {
    const V3 d = normalized(d1);
    if ((S.flags & 1u) || ref_box_t(o, d, S.root_lo, S.root_hi) != INFINITY)
      for (uint32_t k = threadIdx.x; k < S.ntris; k += kBlock) test_triangle(S, k, o, d, &best);
  }
  red[threadIdx.x] = best;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off && hit_before(red[threadIdx.x + off], red[threadIdx.x])) red[threadIdx.x] = red[threadIdx.x + off];
    __syncthreads();
  }
  const Hit r = red[0];
  __syncthreads();
  return r;
}
// ctr: [0] rays answered by brute force (out of the domain), [1] rays re-checked, [2] of which differed
template <class Source>
__device__ __forceinline__ void guard_and_verify(const DevScene& S, const Source& src, uint32_t n, float domain, uint32_t verify_stride,
                                                 uint2* hits, uint32_t* ctr) {
  __shared__ Hit red[kBlock];
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t beg = blockIdx.x * per < n ? blockIdx.x * per : n, end = beg + per < n ? beg + per : n;
  for (uint32_t j = beg; j < end; ++j) {
    V3 o, d1; uint32_t slot; float tmax;
    if (!src.get(j, &o, &d1, &slot, &tmax)) continue;
    const bool far = out_of_domain(o, domain);
    const bool check = verify_stride != 0 && j % verify_stride == 0;
    if (!far && !check) continue;
    const Hit h = brute_closest(S, o, d1, tmax, red);
    if (threadIdx.x == 0) {
      const uint2 want = make_uint2(__float_as_uint(h.id == -1 ? INFINITY : h.t), (uint32_t)h.id);
      if (check && !far) {
        const uint2 got = hits[slot];
        atomicAdd(&ctr[1], 1u);
        if (got.x != want.x || got.y != want.y) atomicAdd(&ctr[2], 1u);
      }
      if (far) { hits[slot] = want; atomicAdd(&ctr[0], 1u); }
    }
  }
}
struct PlainCameraSource {          // the pixels of the rectangle in slot order (no 16x16 block mapping)
  DevCamera C;
  __device__ __forceinline__ bool get(uint32_t j, V3* o, V3* d1, uint32_t* slot, float* tmax) const {
    int x, y; uint32_t f;
    *tmax = INFINITY; *slot = j;
    if (!slot_pixel(C, j, &f, &x, &y)) return false;
    camera_ray(C, f, x, y, o, d1);
    return true;
  }
};
__global__ __launch_bounds__(kBlock) void k_guard_camera(DevScene S, DevCamera C, uint32_t nslots, float domain, uint32_t verify_stride, uint2* hits,
                                                       uint32_t* ctr) {
  PlainCameraSource src; src.C = C;
  guard_and_verify(S, src, nslots, domain, verify_stride, hits, ctr);
}
__global__ __launch_bounds__(kBlock) void k_guard_queue(DevScene S, const float4* rays, const uint32_t* count, uint32_t stride, uint32_t bounded,
                                                      float domain, uint32_t verify_stride, uint2* hits, uint32_t* ctr) {
  QueueSource src; src.rays = rays; src.bounded = bounded; src.anyhit = 0u;
  guard_and_verify(S, src, *count * stride, domain, verify_stride, hits, ctr);
}

// Camera rays of a pixel rectangle.  Ray index j = (16x16 pixel block, thread): consecutive
// indices walk 8x8 pixel sub-tiles, so the 64 rays of a chunk are coherent.  Writes {t, id}.
template <bool kRefill, class Tri = RefTri>
__global__ __launch_bounds__(kBlock, kRefill ? RTX_BLOCKS_REFILL : 8) void k_trace_camera(DevScene S, DevCamera C, uint32_t nblocks, uint32_t grab, uint32_t tri_min, uint32_t* ovf,
                                                          uint2* hits) {
  __shared__ uint32_t lds_stack[(kRefill ? kLdsStack : kLdsStackGroups) * kBlock];
  CameraSource src; src.C = C;
  constexpr int kTop = kRefill ? kTopGroupsRefill : kTopGroups;
  __shared__ uint4 lds_top[kTop > 0 ? 4 * kTop : 1];
  const TopLevels<kTop> top = load_top_levels<kTop>(S, lds_top);
  __shared__ uint32_t lds_ctr;
  const Work work = make_work(nblocks * (uint32_t)kBlock, grab, &lds_ctr);
  if constexpr (kRefill) {
    __shared__ uint32_t lds_wbuf[(kBlock / 64) * kWbufWords * 64];
    trace_wave<Tri, true>(S, top, src, work, make_stack<kLdsStack>(lds_stack, ovf), (lds_u32*)lds_wbuf + (threadIdx.x >> 6) * (kWbufWords * 64), tri_min, hits);
  } else {
    trace_wave_groups<Tri, true>(S, top, src, work, make_stack<kLdsStackGroups>(lds_stack, ovf), hits);
  }
}

// Queued rays: rays[2i] = {o.xyz, aux}, rays[2i+1] = {d1.xyz, aux}; *count * stride rays.
// kAnyOrder: the launch's rays all stop at their first hit (the host says so: a shadow queue of a scene where any blocker shades alike);
// their node steps then skip the child sort.
template <bool kRefill, class Tri = RefTri, bool kAnyOrder = false>
__global__ __launch_bounds__(kBlock, kRefill ? RTX_BLOCKS_REFILL : 8) void k_trace_queue(DevScene S, const float4* rays, const uint32_t* count, uint32_t stride,
                                                         uint32_t bounded, uint32_t grab, uint32_t tri_min, uint32_t* ovf, uint2* hits) {
  __shared__ uint32_t lds_stack[(kRefill ? kLdsStack : kLdsStackGroups) * kBlock];
  QueueSource src; src.rays = rays; src.bounded = bounded; src.anyhit = ((bounded & 2u) && (S.flags & 2u)) ? 1u : 0u;
  const uint32_t cnt = *count;
  uint32_t n = cnt * stride;
  if (bounded & 4u) { src.set_layout(stride, cnt); n = ((cnt + 63u) & ~63u) * stride; }   // shadow_index layout
  if (bounded & 8u) src.scene = &S;                                                                         // ... over one hit point per parent
  constexpr int kTop = kRefill ? kTopGroupsRefill : kTopGroups;
  __shared__ uint4 lds_top[kTop > 0 ? 4 * kTop : 1];
  const TopLevels<kTop> top = load_top_levels<kTop>(S, lds_top);
  __shared__ uint32_t lds_ctr;
  const Work work = make_work(n, grab, &lds_ctr);
  if constexpr (kRefill) {
    __shared__ uint32_t lds_wbuf[(kBlock / 64) * kWbufWords * 64];
    trace_wave<Tri, !kAnyOrder>(S, top, src, work, make_stack<kLdsStack>(lds_stack, ovf), (lds_u32*)lds_wbuf + (threadIdx.x >> 6) * (kWbufWords * 64), tri_min, hits);
  } else {
    trace_wave_groups<Tri, !kAnyOrder>(S, top, src, work, make_stack<kLdsStackGroups>(lds_stack, ovf), hits);
  }
}

// Shadow rays of a camera frame straight from its hit records (DirectSource): nslots output slots x nls light samples.
template <bool kRefill, class Tri = RefTri, bool kAnyOrder = false>
__global__ __launch_bounds__(kBlock, kRefill ? RTX_BLOCKS_REFILL : 8) void k_trace_shadow_direct(DevScene S, const DevCamera* C, const uint2* hits0, uint32_t nslots,
                                                         uint32_t nls, uint32_t any_blocker, uint32_t grab, uint32_t tri_min, uint32_t* ovf, uint2* hits) {
  __shared__ uint32_t lds_stack[(kRefill ? kLdsStack : kLdsStackGroups) * kBlock];
  DirectSource src; src.C = C; src.hits0 = hits0; src.scene = &S; src.anyhit = (any_blocker && (S.flags & 2u)) ? 1u : 0u;
  src.set_layout(nls, nslots);
  const uint32_t n = ((nslots + 63u) & ~63u) * nls;
  constexpr int kTop = kRefill ? kTopGroupsRefill : kTopGroups;
  __shared__ uint4 lds_top[kTop > 0 ? 4 * kTop : 1];
  const TopLevels<kTop> top = load_top_levels<kTop>(S, lds_top);
  __shared__ uint32_t lds_ctr;
  const Work work = make_work(n, grab, &lds_ctr);
  if constexpr (kRefill) {
    __shared__ uint32_t lds_wbuf[(kBlock / 64) * kWbufWords * 64];
    trace_wave<Tri, !kAnyOrder>(S, top, src, work, make_stack<kLdsStack>(lds_stack, ovf), (lds_u32*)lds_wbuf + (threadIdx.x >> 6) * (kWbufWords * 64), tri_min, hits);
  } else {
    trace_wave_groups<Tri, !kAnyOrder>(S, top, src, work, make_stack<kLdsStackGroups>(lds_stack, ovf), hits);
  }
}
__global__ void k_store_camera(DevCamera C, DevCamera* dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = C;
}

// The inverse of the tiled-compact layout: parts[p][k * ts * ts + (y % ts) * ts + (x % ts)][c] of every
// part p -> out[y][x][c]; tile g = (y / ts) * tiles_x + (x / ts) is local tile g / nparts of part g % nparts.
// One element (a 32-bit word, or a byte for pixel sizes that are not whole words: the 3-byte P3 pixels) per thread:
// consecutive threads read and write consecutive elements within a tile row.
template <class T>
__global__ void k_assemble_tiles(const T* parts, uint32_t nparts, uint64_t part_stride_elems, uint32_t w, uint32_t h,
                                 uint32_t ts, uint32_t elems, T* out) {
  const uint64_t n = (uint64_t)w * h * elems;
  const uint32_t tiles_x = (w + ts - 1) / ts;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t pix = (uint32_t)(i / elems), c = (uint32_t)(i - (uint64_t)pix * elems);
    const uint32_t y = pix / w, x = pix - y * w;
    const uint32_t g = (y / ts) * tiles_x + (x / ts);
    const uint32_t p = g % nparts, k = g / nparts;
    const uint64_t src = (uint64_t)p * part_stride_elems + ((uint64_t)k * ts * ts + (uint64_t)(y % ts) * ts + (x % ts)) * elems + c;
    out[i] = parts[src];
  }
}

// The same, 16 bytes per thread, where every image row and every tile row is a whole number of 16-byte pieces
// (w * bytes-per-pixel and ts * bytes-per-pixel multiples of 16: any pixel size at widths that are multiples of 16,
// since tile sizes are) and both buffers are 16-byte aligned.  Thread -> (row y, piece q of that row).
__global__ void k_assemble_tiles_v4(const uint4* parts, uint32_t nparts, uint64_t part_stride_v4, uint32_t w, uint32_t h,
                                    uint32_t ts, uint32_t bpp, uint4* out) {
  const uint32_t row_v4 = w * bpp / 16u, tile_v4 = ts * bpp / 16u;      // pieces per image row / per full tile row
  const uint32_t tiles_x = (w + ts - 1) / ts;
  const uint64_t n = (uint64_t)row_v4 * h;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t y = (uint32_t)(i / row_v4), q = (uint32_t)(i - (uint64_t)y * row_v4);
    const uint32_t tx = q / tile_v4, r = q - tx * tile_v4;               // tile column, piece within the tile's row
    const uint32_t g = (y / ts) * tiles_x + tx;
    const uint32_t p = g % nparts, k = g / nparts;
    const uint64_t src = (uint64_t)p * part_stride_v4 + ((uint64_t)k * ts + (y % ts)) * tile_v4 + r;
    out[i] = parts[src];
  }
}

// The channel values Tucano::ImageImporter::writePPMImage prints (tucano/utils/ppmIO.hpp:145): min(255, (int)(255 * c)),
// the product in fp32, the conversion truncating -- as a byte.  Where the reference's text holds a number no byte can
// (a negative channel prints negative; NaN and infinities convert to INT_MIN on x86) the byte is 0, or 255 for +inf.
__device__ __forceinline__ uint32_t ppm8(float c) {
  const float v = 255.0f * c;
  if (!(v >= 1.0f)) return 0u;              // [0, 1) truncates to 0; negatives and NaN have no byte value
  return v >= 255.0f ? 255u : (uint32_t)(int)v;
}
// n channel values (3 per pixel) -> bytes; four per thread where the count allows (float4 in, one word out).
__global__ void k_rgb_to_ppm8(const float* rgb, uint64_t n, uint8_t* out) {
  const uint64_t n4 = n / 4;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
    const float4 v = ((const float4*)rgb)[i];
    ((uint32_t*)out)[i] = ppm8(v.x) | (ppm8(v.y) << 8) | (ppm8(v.z) << 16) | (ppm8(v.w) << 24);
  }
  if (blockIdx.x == 0 && threadIdx.x < (uint32_t)(n - 4 * n4)) out[4 * n4 + threadIdx.x] = (uint8_t)ppm8(rgb[4 * n4 + threadIdx.x]);
}
__global__ void k_rgb_to_ppm8_unaligned(const float* rgb, uint64_t n, uint8_t* out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (uint8_t)ppm8(rgb[i]);
}

// Ray ctor over caller rays (src/Ray.h:15-18): xyz arrays -> queue records.
__global__ void k_make_rays(const float* o, const float* d, uint32_t n, float4* rays) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 d1 = normalized(v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
  rays[2 * (size_t)i] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], 0.0f);
  rays[2 * (size_t)i + 1] = make_float4(d1.x, d1.y, d1.z, 0.0f);
}

// Ray ctor over caller rays for Flyscene::traceRay: queue records with pixel = ray index.
__global__ void k_make_nodes(const float* o, const float* d, uint32_t n, int inside, float4* rays) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 d1 = normalized(v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
  rays[2 * (size_t)i] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], __uint_as_float(i | (inside ? 0x80000000u : 0u)));
  rays[2 * (size_t)i + 1] = make_float4(d1.x, d1.y, d1.z, 1.0f);
}
// Flyscene::intersectBox: Ray ctor + Intersect::BoxObject::intersects for one box.
__global__ void k_ref_box(const float* o, const float* d, uint32_t n, float lx, float ly, float lz, float hx, float hy, float hz,
                          float* t) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 d1 = normalized(v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
  const float lo[3] = {lx, ly, lz}, hi[3] = {hx, hy, hz};
  t[i] = ref_box_t(v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), d1, lo, hi);
}

// Intersect::{Triangle, Plane, BoxObject, Sphere}::intersects(Ray&) (src/Intersect.cpp:6-193) for n Ray objects against
// one primitive: o / d are the Ray's members as they stand (the ctor has normalised d already; nothing is
// normalised again here).  prim: triangle {v0, v1, v2, normal}, plane {v0, normal}, box {tmin, tmax}, sphere
// {centre, radius}.  Outputs as the HitInfo the reference returns: t (INFINITY = miss), faceId (0 for a triangle
// or plane hit, else -1), normal, point (zeros where the reference leaves them unset).
__global__ void k_primitive(int kind, const float* prim, const float* o, const float* d, uint32_t n, float* t, int* face,
                            float* normal, float* point) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 org = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dir = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
  float ht = INFINITY; int id = -1;
  V3 N = v3(0, 0, 0), P = v3(0, 0, 0);
  if (kind == 0 || kind == 1) {
    const V3 v0 = v3(prim[0], prim[1], prim[2]);
    const V3 nrm = kind == 0 ? v3(prim[9], prim[10], prim[11]) : v3(prim[3], prim[4], prim[5]);
    const float D = dot(nrm, v0);
    const float den = dot(nrm, dir);
    if (den != 0.0f) {
      const float tt = (D - dot(org, nrm)) / den;
      const V3 hp = point_on_ray(org, dir, tt);
      bool ok = tt >= kPrecision;
      if (ok && kind == 0) {
        const V3 u = v3(prim[3], prim[4], prim[5]) - v0, v = v3(prim[6], prim[7], prim[8]) - v0, w = hp - v0;
        const float uu = dot(u, u), uv = dot(u, v), vv = dot(v, v), wu = dot(w, u), wv = dot(w, v);
        const float denom = uu * vv - uv * uv;
        const float s1 = vv * wu - uv * wv, t1 = uu * wv - uv * wu;
        const float sb = s1 / denom, tb = t1 / denom;
        if (sb < 0 || tb < 0 || sb + tb > 1) ok = false;
      }
      if (ok) { ht = tt; id = 0; N = nrm; P = hp; }
    }
  } else if (kind == 2) {
    ht = ref_box_t(org, dir, prim, prim + 3);
  } else {
    ht = ref_sphere_t(org, dir, make_float4(prim[0], prim[1], prim[2], prim[3]));
    if (ht != INFINITY) { P = point_on_ray(org, dir, ht); N = normalized(P - v3(prim[0], prim[1], prim[2])); }
  }
  t[i] = ht; face[i] = id;
  normal[3 * i] = N.x; normal[3 * i + 1] = N.y; normal[3 * i + 2] = N.z;
  point[3 * i] = P.x; point[3 * i + 1] = P.y; point[3 * i + 2] = P.z;
}

// Camera::screenToWorld for caller pixel coordinates (tucano/camera.hpp:155-173): the world
// point on the image plane (before the Ray ctor normalises the direction).
__global__ void k_camera_points(DevCamera C, const float* ij, uint32_t n, float* world) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float nx = camera_nx(C, ij[2 * i]), ny = camera_ny(C, ij[2 * i + 1]);
  const float nz = -1.0f;
  world[3 * i] = ((C.inv34[0] * nx + C.inv34[1] * ny) + C.inv34[2] * nz) + C.inv34[3] * 1.0f;
  world[3 * i + 1] = ((C.inv34[4] * nx + C.inv34[5] * ny) + C.inv34[6] * nz) + C.inv34[7] * 1.0f;
  world[3 * i + 2] = ((C.inv34[8] * nx + C.inv34[9] * ny) + C.inv34[10] * nz) + C.inv34[11] * 1.0f;
}

// Unpack hit records of caller rays into the HitInfo fields (src/hitInfo.h:4-16).
__global__ void k_unpack_hits(DevScene S, const float4* rays, const uint2* hits, uint32_t n, float* t, int* face,
                              float* normal, float* point, int* mat) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint2 h = hits[i];
  const float ht = __uint_as_float(h.x);
  const int id = (int)h.y;
  if (t) t[i] = ht;
  if (face) face[i] = id >= 0 ? id : -1;
  if (!normal && !point && !mat) return;
  V3 N = v3(0, 0, 0), P = v3(0, 0, 0);
  int m = -1;
  if (id != -1) {
    const float4 ro = rays[2 * (size_t)i], rd = rays[2 * (size_t)i + 1];
    const V3 o = v3(ro.x, ro.y, ro.z), d1 = v3(rd.x, rd.y, rd.z);
    if (id >= 0) {
      const float4 nm = S.face_nm[id];
      N = v3(nm.x, nm.y, nm.z); m = __float_as_int(nm.w);
      P = point_on_ray(o, normalized(d1), ht);
    } else {
      const int si = -2 - id;
      const float4 s = S.spheres[si];
      P = point_on_ray(o, d1, ht);
      N = normalized(P - v3(s.x, s.y, s.z));
      m = S.sph_mat[si];
    }
  }
  if (normal) { normal[3 * i] = N.x; normal[3 * i + 1] = N.y; normal[3 * i + 2] = N.z; }
  if (point) { point[3 * i] = P.x; point[3 * i + 1] = P.y; point[3 * i + 2] = P.z; }
  if (mat) mat[i] = m;
}

// ---------------------------------------------------------------- shading pipeline
// One "node" = one traceRay call of the reference: a ray at bounce b for pixel p with the
// product of blend weights on the way down (throughput) and the insideObject flag.
// Generation g holds all nodes of bounce b0 + g; each generation is
//   trace -> k_shade (spawn shadow / GI / child rays, compacted) -> trace queues -> k_resolve.
struct PipeArgs {
  DevScene S;
  DevCamera C;
  int bounce;            // bounce index of the nodes of this generation
  int from_camera;       // nodes are the pixels of C (implicit rays); else queue records
  int first_gen;         // first generation of the call: initialises the output pixels
  int max_bounces, reflection, refraction, gi_rays, shadows, skip_zero_weight;
  int atomic_accum;      // several nodes of one generation may share a pixel (refraction tree)
  uint32_t n0;           // node count when from_camera / first generation from a caller queue
  const uint32_t* nrays; // device-side node count otherwise
  const float4* rays;    // {o, pixel | inside<<31}, {d1, throughput}
  const uint2* hits;
  float* rgb; float* out_t; int* out_face;
  uint32_t* parents;     // compacted list of nodes that hit something
  float4* shadow_rays;   // parents x light samples
  const uint2* shadow_hits;
  float4* gi_q;          // parents x gi_rays
  const uint2* gi_hits;  // {t bits (0 = nothing), material id}
  float4* next_rays;     // nodes of the next generation
  uint32_t* counters;    // [0] = #parents, [1] = #next nodes
  long long* acc;        // atomic_accum: per pixel 3 x 40.24 fixed-point sums (integer addition is associative: the
  uint32_t* acc_flag;    // frame does not depend on the order the nodes arrive in) + flags for non-finite terms
  float gi_sin, gi_cos;  // sinf/cosf of (float)(2*pi / gi_rays), evaluated on the host (glibc, like the reference)
  float sl_sin, sl_cos;  // same for the 4-point ring of spherical lights
};

struct RayRec { V3 o, d1; uint32_t pixel; float thr; bool inside; bool valid; };

// Order-independent accumulation for ray trees (reflection AND refraction: several nodes of a generation share a
// pixel and arrive in whatever order the queue compaction gave them).  A float atomic sum would depend on that
// order ((p + a) + b != (p + b) + a in fp32); a 64-bit integer sum does not.  2^-40 resolution (1e-12, against an
// RGB tolerance of 1e-4), range +-8e6.  Terms that are not finite (a material with a negative shininess can give
// pow(0, Ns) = inf) set per-channel flags instead: 1 = +inf, 2 = -inf, 4 = NaN, three bits per channel.
constexpr double kAccScale = 1099511627776.0;     // 2^40
__device__ __forceinline__ long long acc_fixed(float x) { return __double2ll_rn((double)x * kAccScale); }
__device__ __forceinline__ void acc_add(const PipeArgs& A, uint32_t pixel, int c, float x) {
  if (x - x == 0.0f && fabsf(x) < 4.0e6f) {
    atomicAdd((unsigned long long*)(A.acc + 3 * (size_t)pixel + c), (unsigned long long)acc_fixed(x));
  } else {
    const uint32_t bit = (x != x) ? 4u : (x > 0.0f ? 1u : 2u);     // (a finite term beyond the range counts as an infinity)
    atomicOr(A.acc_flag + pixel, bit << (3 * c));
  }
}
__device__ __forceinline__ float acc_value(long long v, uint32_t flags) {
  if (flags & 4u) return __uint_as_float(0x7FC00000u);
  if ((flags & 3u) == 3u) return __uint_as_float(0x7FC00000u);
  if (flags & 1u) return INFINITY;
  if (flags & 2u) return -INFINITY;
  return (float)((double)v / kAccScale);
}

__device__ __forceinline__ RayRec load_node(const PipeArgs& A, uint32_t i) {
  RayRec r;
  if (A.from_camera) {
    int x, y; uint32_t f;
    r.valid = slot_pixel(A.C, i, &f, &x, &y);
    camera_ray(A.C, f, x, y, &r.o, &r.d1);
    r.pixel = i; r.thr = 1.0f; r.inside = false;
    return r;
  }
  const float4 ro = A.rays[2 * (size_t)i], rd = A.rays[2 * (size_t)i + 1];
  r.o = v3(ro.x, ro.y, ro.z); r.d1 = v3(rd.x, rd.y, rd.z);
  const uint32_t pw = __float_as_uint(ro.w);
  r.pixel = pw & 0x7FFFFFFFu; r.inside = (pw >> 31) != 0; r.thr = rd.w;
  r.valid = true;
  return r;
}

struct Surface { V3 P, N; int mat; };
// HitInfo point / normal / material of a hit record (src/flyscene.cpp:849-854, Intersect.cpp:167-173)
__device__ __forceinline__ Surface surface_of(const DevScene& S, V3 o, V3 d1, float t, int id) {
  Surface s;
  if (id >= 0) {
    const float4 nm = S.face_nm[id];
    s.N = v3(nm.x, nm.y, nm.z); s.mat = __float_as_int(nm.w);
    s.P = point_on_ray(o, normalized(d1), t);
  } else {
    const int si = -2 - id;
    const float4 sp = S.spheres[si];
    s.P = point_on_ray(o, d1, t);
    s.N = normalized(s.P - v3(sp.x, sp.y, sp.z));
    s.mat = S.sph_mat[si];
  }
  return s;
}
__device__ __forceinline__ DevMat default_mtl() {   // Tucano::Material::Mtl() (materials/mtl.hpp:21-39)
  DevMat m;
  for (int k = 0; k < 3; ++k) { m.ka[k] = 0.3f; m.kd[k] = 0.5f; m.ks[k] = 1.0f; }
  m.ns = 10.0f; m.ni = 0.0f; m.d = 1.0f; m.illum = 0;
  return m;
}
__device__ __forceinline__ DevMat material_of(const DevScene& S, int id) {
  if (id < 0 || id >= (int)S.nmats) return default_mtl();   // materials[-1] is UB in the reference
  return S.mats[id];
}

__device__ __forceinline__ V3 light_sample(const PipeArgs& A, V3 P, uint32_t ls, bool* fifth) { return light_sample_s(A.S, P, ls, fifth); }
__device__ __forceinline__ uint32_t light_samples(const DevScene& S) { return S.nlights + 5u * S.nsl; }

// Ray::refractRay (src/Ray.h:36-42); pow(float,int) and sqrt run in double under libstdc++
__device__ __forceinline__ V3 refract_dir(V3 n, V3 view, float n1, float n2) {
  const float dt = dot(view, n);
  const float numerator = (float)(((double)n1 * (double)n1) * (1.0 - (double)dt * (double)dt));
  const float root = (float)sqrt(1.0 - ((double)numerator / ((double)n2 * (double)n2)));
  const float s = n1 / n2;
  const V3 a = s * (view - dt * n);
  const V3 b = v3(n.x * root, n.y * root, n.z * root);
  return normalized(a - b);
}

// wave64 compaction inside a block: returns this lane's offset in the block's share of a
// list (LDS running counter), or ~0u when !want.
__device__ __forceinline__ uint32_t wave_alloc_lds(bool want, uint32_t* lds_counter) {
  const unsigned long long m = __ballot(want);
  if (m == 0) return ~0u;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
  uint32_t base = 0;
  const int leader = __ffsll((long long)m) - 1;
  if ((int)lane == leader) base = atomicAdd(lds_counter, (uint32_t)__popcll(m));
  base = __shfl(base, leader, 64);
  return want ? base + rank : ~0u;
}

// What a node spawns, from its hit record alone (shared by the counting and the writing pass).
struct Spawn { bool valid, hit, refl, refr; float ht; int id; };
__device__ __forceinline__ Spawn classify(const PipeArgs& A, uint32_t i) {
  Spawn sp; sp.valid = true; sp.hit = sp.refl = sp.refr = false; sp.ht = INFINITY; sp.id = -1;
  if (A.from_camera) { int x, y; uint32_t f; sp.valid = slot_pixel(A.C, i, &f, &x, &y); }
  if (!sp.valid) return sp;
  const uint2 h = A.hits[i];
  sp.ht = __uint_as_float(h.x); sp.id = (int)h.y;
  sp.hit = sp.id != -1;
  if (sp.hit && A.bounce < A.max_bounces) {
    sp.refl = A.reflection != 0; sp.refr = A.refraction != 0;
    if (A.skip_zero_weight && (sp.refl || sp.refr)) {
      // where all materials in use answer alike (DevScene.flags bits 2-5, set at commit) no lookup is needed
      const uint32_t f = A.S.flags;
      if (f & 4u) sp.refl = false;
      if (f & 16u) sp.refr = false;
      const bool ask = (sp.refl && !(f & 8u)) || (sp.refr && !(f & 32u));
      if (ask) {
        const int m = sp.id >= 0 ? __float_as_int(A.S.face_nm[sp.id].w) : A.S.sph_mat[-2 - sp.id];
        const float d = (m < 0 || m >= (int)A.S.nmats) ? 1.0f : A.S.mats[m].d;      // default Mtl: d = 1
        if (1.0f - d == 0.0f) sp.refl = false;
        if (d == 0.0f) sp.refr = false;
      }
    }
  }
  return sp;
}

// Flyscene::traceRay miss branch + the ray-spawning half of Shader/calculateColor
// (src/flyscene.cpp:587-601, 686-702, 704-712, 723, 747-750).
// Each block owns a contiguous range of nodes.  Pass 1 counts what the range spawns, one thread
// reserves the block's share of the three output lists with ONE global atomic each, pass 2
// writes the records using LDS counters.  (A global atomic per wave was the bottleneck of the
// first version: ~90k atomics on three addresses per 4K frame, ~0.6 ms.)
__global__ __launch_bounds__(kBlock) void k_shade(PipeArgs A) {
  __shared__ uint32_t s_cnt[4], s_base[4], s_run[4];
  const uint32_t n = A.nrays ? *A.nrays : A.n0;
  const uint32_t per = (((n + gridDim.x - 1) / gridDim.x) + kBlock - 1) / kBlock * kBlock;
  const uint32_t beg = blockIdx.x * per < n ? blockIdx.x * per : n;
  const uint32_t end = beg + per < n ? beg + per : n;
  if (threadIdx.x < 4) { s_cnt[threadIdx.x] = 0; s_run[threadIdx.x] = 0; s_base[threadIdx.x] = 0; }
  __syncthreads();
  if (A.rgb) {
    uint32_t c_par = 0, c_next = 0, c_refr = 0;
    for (uint32_t i = beg + threadIdx.x; i < end; i += kBlock) {
      const Spawn sp = classify(A, i);
      c_par += sp.hit; c_next += (uint32_t)sp.refl + (uint32_t)sp.refr; c_refr += sp.refr;
    }
    for (int off = 32; off > 0; off >>= 1) {
      c_par += __shfl_down(c_par, off, 64); c_next += __shfl_down(c_next, off, 64); c_refr += __shfl_down(c_refr, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cnt[0], c_par); atomicAdd(&s_cnt[1], c_next); atomicAdd(&s_cnt[2], c_refr); }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (s_cnt[0]) s_base[0] = atomicAdd(&A.counters[0], s_cnt[0]);
      if (s_cnt[1]) s_base[1] = atomicAdd(&A.counters[1], s_cnt[1]);
      if (s_cnt[2]) atomicAdd(&A.counters[2], s_cnt[2]);        // refraction tally for the statistics
    }
    __syncthreads();
  }
  for (uint32_t i0 = beg; i0 < end; i0 += kBlock) {
    const uint32_t i = i0 + threadIdx.x;
    const bool live = i < end;
    Spawn sp; sp.valid = false; sp.hit = sp.refl = sp.refr = false; sp.ht = INFINITY; sp.id = -1;
    if (live) sp = classify(A, i);
    if (live && sp.valid && A.first_gen) {
      if (A.out_t) A.out_t[i] = sp.ht;
      if (A.out_face) A.out_face[i] = sp.id >= 0 ? sp.id : -1;
    }
    if (!A.rgb) continue;   // hits only
    RayRec r; r.valid = false; r.pixel = 0; r.thr = 0; r.inside = false; r.o = v3(0, 0, 0); r.d1 = v3(0, 0, 0);
    if (live && sp.valid && (sp.hit || A.first_gen)) r = load_node(A, i);
    if (live && sp.valid && A.first_gen) {
      // traceRay's miss colour at bounce 0 (src/flyscene.cpp:596-600); hit pixels start at 0 and
      // receive their nodes' colours from k_resolve
      const bool sky = !sp.hit && A.bounce == 0;
      const float c0 = sky ? (float)0.69019607843 : 0.0f, c1 = sky ? 1.0f : 0.0f;
      if (A.atomic_accum) {
        long long* ax = A.acc + 3 * (size_t)r.pixel;
        ax[0] = acc_fixed(c0); ax[1] = acc_fixed(c1); ax[2] = acc_fixed(c1);
        A.acc_flag[r.pixel] = 0u;
      } else if (!sp.hit) {               // (a hit pixel receives its first value from k_resolve: 0 + colour)
        float* px = A.rgb + 3 * (size_t)r.pixel;
        px[0] = c0; px[1] = c1; px[2] = c1;
      }
    }
    const uint32_t off_par = wave_alloc_lds(sp.hit, &s_run[0]);
    const uint32_t off_r = wave_alloc_lds(sp.refl, &s_run[1]);
    const uint32_t off_t = wave_alloc_lds(sp.refr, &s_run[1]);
    if (!sp.hit) continue;
    const uint32_t slot = s_base[0] + off_par;
    // the hit point alone where that is all a parent needs (its shadow record): the normal and the material are a gather from the
    // per-face table, wanted by GI probes and child rays only
    Surface sf;
    if (A.gi_rays > 0 || sp.refl || sp.refr) sf = surface_of(A.S, r.o, r.d1, sp.ht, sp.id);
    else { sf.P = point_on_ray(r.o, sp.id >= 0 ? normalized(r.d1) : r.d1, sp.ht); sf.N = v3(0, 0, 0); sf.mat = -1; }
    A.parents[slot] = i;
    // the shadow rays of this parent: one record, its hit point (QueueSource forms the rays of every light sample from it)
    if (A.shadows) A.shadow_rays[slot] = make_float4(sf.P.x, sf.P.y, sf.P.z, 0.0f);
    if (A.gi_rays > 0) {
      // Ray::resendRay (src/Ray.h:43-57): the reflected direction rotated about the normal
      float m[9];
      angle_axis_matrix(A.gi_sin, A.gi_cos, sf.N, m);
      V3 dir = normalized(reflect_dir(r.d1, sf.N));
      for (int k = 0; k < A.gi_rays; ++k) {
        dir = mat3_mul(m, dir);
        const V3 dn = normalized(dir);
        const size_t q = (size_t)slot * (uint32_t)A.gi_rays + (uint32_t)k;
        A.gi_q[2 * q] = make_float4(sf.P.x, sf.P.y, sf.P.z, 0.0f);
        A.gi_q[2 * q + 1] = make_float4(dn.x, dn.y, dn.z, 0.0f);
      }
    }
    if (sp.refl || sp.refr) {
      const DevMat mat = material_of(A.S, sf.mat);
      // every light's calculateColor adds (1-d)*reflected + d*refracted once (src/flyscene.cpp:611-626,744)
      const float nl_tot = (float)(A.S.nlights + A.S.nsl);
      if (sp.refl) {
        const V3 nd = normalized(reflect_dir(r.d1, sf.N));            // Ray::reflectRay (src/Ray.h:32-35)
        const uint32_t pw = r.pixel | (r.inside ? 0x80000000u : 0u);
        const size_t q = s_base[1] + off_r;
        A.next_rays[2 * q] = make_float4(sf.P.x, sf.P.y, sf.P.z, __uint_as_float(pw));
        A.next_rays[2 * q + 1] = make_float4(nd.x, nd.y, nd.z, r.thr * ((1.0f - mat.d) * nl_tot));
      }
      if (sp.refr) {
        const V3 eye = normalized(r.o - sf.P);
        const V3 nd = r.inside ? refract_dir(-sf.N, -eye, mat.ni, 1.0f) : refract_dir(sf.N, -eye, 1.0f, mat.ni);   // :691-701
        const uint32_t pw = r.pixel | (r.inside ? 0u : 0x80000000u);
        const size_t q = s_base[1] + off_t;
        A.next_rays[2 * q] = make_float4(sf.P.x, sf.P.y, sf.P.z, __uint_as_float(pw));
        A.next_rays[2 * q + 1] = make_float4(nd.x, nd.y, nd.z, r.thr * (mat.d * nl_tot));
      }
    }
  }
}

// Flyscene::canSeeLight decision from the closest hit of the shadow ray (src/flyscene.cpp:752-781).
// Returns visible; *bmat = blocker material id (-1: default Mtl).
__device__ __forceinline__ bool can_see_light(const DevScene& S, V3 light, V3 P, V3 d1s, float ht, int hid, int* bmat) {
  const float len = norm(light - P);
  bool behind = false, sphere_block = false;
  for (uint32_t i = 0; i < S.nspheres; ++i) {
    const float t = ref_sphere_t(P, d1s, S.spheres[i]);
    if (t != INFINITY && t > len) behind = true;
    else if (t != INFINITY) sphere_block = true;
  }
  *bmat = -1;
  if (ht != INFINITY && ht > len) behind = true;
  if (ht != INFINITY && ht < len) {
    int m = hid >= 0 ? __float_as_int(S.face_nm[hid].w) : S.sph_mat[-2 - hid];
    *bmat = m < 0 ? -2 : m;
  }
  return (!sphere_block && ht == INFINITY) || behind;
}

// The local (non-recursive) part of Flyscene::calculateColor (src/flyscene.cpp:671-745) summed
// over the light samples as Flyscene::Shader does (:611-626), times the node's throughput.
// (h: the node's index in the lists that hang off it -- shadow hits, GI probes; i: the node)
__device__ __forceinline__ void resolve_node(const PipeArgs& A, uint32_t h, uint32_t i, uint32_t NLS) {
  const RayRec r = load_node(A, i);
  const uint2 hh = A.hits[i];
  const Surface sf = surface_of(A.S, r.o, r.d1, __uint_as_float(hh.x), (int)hh.y);
  const DevMat mat = material_of(A.S, sf.mat);
  const float I = 0.8f / (float)(A.S.nlights + A.S.nsl);
  const V3 Iv = v3(I, I, I);
  const V3 ka = v3(mat.ka[0], mat.ka[1], mat.ka[2]);
  // global illumination term (src/flyscene.cpp:704-721), the same for every light
  V3 gi = v3(0, 0, 0);
  if (A.gi_rays > 0) {
    for (int k = 0; k < A.gi_rays; ++k) {
      const uint2 g = A.gi_hits[(size_t)h * (uint32_t)A.gi_rays + (uint32_t)k];
      const float dist = __uint_as_float(g.x);
      if (dist != 0.0f) {
        const DevMat gm = material_of(A.S, (int)g.y);
        const float fd = 50.0f * dist;
        const double w = 1.0 / ((double)fd * (double)fd);
        const float wf = (w < 1.0) ? (float)w : 1.0f;
        gi = gi + v3(gm.kd[0] * wf, gm.kd[1] * wf, gm.kd[2] * wf);
      }
    }
    gi = gi / (float)A.gi_rays;
  }
  V3 total = v3(0, 0, 0);
  const V3 eye = normalized(r.o - sf.P);
  for (uint32_t ls = 0; ls < NLS; ++ls) {
    bool fifth;
    const V3 light = light_sample(A, sf.P, ls, &fifth);
    const V3 lightDir = normalized(light - sf.P);
    const V3 reflectedLight = reflect_dir(-lightDir, sf.N);
    const float dotted = dot(eye, normalized(reflectedLight));
    bool visible = true; int bmat = -1;
    if (A.shadows) {
      // (the shadow ray's direction: the expression QueueSource formed it with, from the same light and point)
      const uint2 sh = A.shadow_hits[shadow_index(h, ls, NLS)];
      visible = can_see_light(A.S, light, sf.P, normalized(normalized(light - sf.P)), __uint_as_float(sh.x), (int)sh.y, &bmat);
    }
    const float ndl = dot(sf.N, lightDir);
    const float md = (ndl < 0.0f) ? 0.0f : ndl;          // std::max(a, b) = (a < b) ? b : a
    const float mdot = (dotted < 0.0f) ? 0.0f : dotted;
    V3 c;
    if (visible) {
      const V3 ambient = cmul(ka, Iv);
      const V3 diffuse = md * cmul(v3(mat.kd[0], mat.kd[1], mat.kd[2]), Iv);
      // std::pow(float, 33) (src/flyscene.cpp:729): evaluated in double like libstdc++ does, by
      // repeated squaring (x^32 * x) instead of the generic pow -- a few double ulps apart at
      // most, far inside the 1e-4 RGB tolerance, and ~20x fewer instructions
      const double xd = (double)mdot;
      const double x2 = xd * xd, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
      const float sp = (float)((x16 * x16) * xd);
      const V3 specular = cmul(Iv, v3(mat.ks[0], mat.ks[1], mat.ks[2])) * sp;
      c = ((ambient + diffuse) + specular) + gi;
    } else {
      const DevMat bm = (bmat == -1) ? default_mtl() : material_of(A.S, bmat);
      const V3 ambientS = cmul(v3(bm.ka[0], bm.ka[1], bm.ka[2]), Iv);
      const V3 diffuseS = md * cmul(v3(bm.kd[0], bm.kd[1], bm.kd[2]), Iv);
      const float spS = powf(mdot, bm.ns);
      const V3 specularS = cmul(Iv, v3(bm.ks[0], bm.ks[1], bm.ks[2])) * spS;
      const V3 colorS = (ambientS + diffuseS) + specularS;
      c = cmul(ka, Iv) + colorS * bm.d;
    }
    total = total + (fifth ? c / 5.0f : c);
  }
  float* px = A.rgb + 3 * (size_t)r.pixel;
  const V3 add = v3(r.thr * total.x, r.thr * total.y, r.thr * total.z);
  if (A.atomic_accum) { acc_add(A, r.pixel, 0, add.x); acc_add(A, r.pixel, 1, add.y); acc_add(A, r.pixel, 2, add.z); }
  else if (A.first_gen) { px[0] = 0.0f + add.x; px[1] = 0.0f + add.y; px[2] = 0.0f + add.z; }      // the pixel's first node: traceRay's sum starts at zero
  else { px[0] += add.x; px[1] += add.y; px[2] += add.z; }
}
__global__ __launch_bounds__(kBlock) void k_resolve(PipeArgs A) {
  const uint32_t n = A.counters[0];
  const uint32_t NLS = light_samples(A.S);
  for (uint32_t h = blockIdx.x * kBlock + threadIdx.x; h < n; h += gridDim.x * kBlock) resolve_node(A, h, A.parents[h], NLS);
}
// The same for a camera frame without a parent list (run_pipeline's direct mode): every output slot of the call -- dead slots
// skipped, a miss gets traceRay's miss colour (src/flyscene.cpp:596-600), a hit is its own parent (its shadow hits lie at
// shadow_index(slot, ls)); out_t / out_face as k_shade writes them; counters[0] = the number of hits (ray statistics).
__global__ __launch_bounds__(kBlock) void k_resolve_direct(PipeArgs A) {
  __shared__ uint32_t s_hits;
  if (threadIdx.x == 0) s_hits = 0;
  __syncthreads();
  const uint32_t NLS = light_samples(A.S);
  uint32_t mine = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < A.n0; i += gridDim.x * kBlock) {
    uint32_t f; int x, y;
    if (!slot_pixel(A.C, i, &f, &x, &y)) continue;
    const uint2 hh = A.hits[i];
    const int id = (int)hh.y;
    if (A.out_t) A.out_t[i] = __uint_as_float(hh.x);
    if (A.out_face) A.out_face[i] = id >= 0 ? id : -1;
    if (id == -1) {
      const bool sky = A.bounce == 0;
      float* px = A.rgb + 3 * (size_t)i;
      px[0] = sky ? (float)0.69019607843 : 0.0f; px[1] = sky ? 1.0f : 0.0f; px[2] = sky ? 1.0f : 0.0f;
      continue;
    }
    ++mine;
    resolve_node(A, i, i, NLS);
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_hits, mine);
  __syncthreads();
  if (threadIdx.x == 0 && s_hits) atomicAdd(&A.counters[0], s_hits);
}

// The fixed-point sums of a ray-tree call -> the caller's float pixels (slots of partial edge tiles stay untouched).
__global__ __launch_bounds__(kBlock) void k_acc_to_rgb(PipeArgs A) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < A.n0; i += gridDim.x * kBlock) {
    if (A.from_camera) { int x, y; uint32_t f; if (!slot_pixel(A.C, i, &f, &x, &y)) continue; }
    const uint32_t f = A.acc_flag[i];
    for (int c = 0; c < 3; ++c) A.rgb[3 * (size_t)i + c] = acc_value(A.acc[3 * (size_t)i + c], (f >> (3 * c)) & 7u);
  }
}

// Flyscene::trace_global_illum (src/flyscene.cpp:631-669): spheres, then the reference's flat
// leaf-box list (slab gate per leaf, that leaf's own triangle list, strict < in list order).
// Deliberately NOT the BVH: the leaf lists are lossy and the colour depends on it.
__global__ __launch_bounds__(kBlock) void k_trace_gi(DevScene S, const float4* rays, const uint32_t* count, uint32_t stride,
                                                   uint2* out) {
  const uint32_t n = *count * stride;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 ro = rays[2 * (size_t)i], rd = rays[2 * (size_t)i + 1];
    const V3 o = v3(ro.x, ro.y, ro.z), d1 = v3(rd.x, rd.y, rd.z);
    float smallest = INFINITY; int mat = -1;
    for (uint32_t s = 0; s < S.nspheres; ++s) {
      const float t = ref_sphere_t(o, d1, S.spheres[s]);
      if (t != INFINITY && t < smallest) { smallest = t; mat = S.sph_mat[s]; }
    }
    const V3 d = normalized(d1);
    for (uint32_t b = 0; b < S.nleafboxes; ++b) {
      const float4 lo = S.leafboxes[2 * b], hi = S.leafboxes[2 * b + 1];
      const float blo[3] = {lo.x, lo.y, lo.z}, bhi[3] = {hi.x, hi.y, hi.z};
      if (ref_box_t(o, d, blo, bhi) != INFINITY) {
        Hit best; best.t = INFINITY; best.id = -1;
        const uint32_t start = __float_as_uint(lo.w), cnt = __float_as_uint(hi.w);
        for (uint32_t k = 0; k < cnt; ++k) {
          Hit one; one.t = INFINITY; one.id = -1;
          test_triangle(S, (uint32_t)S.face_pos[S.leaf_faces[start + k]], o, d, &one);
          if (one.t < best.t) best = one;             // intersectTriangle: strict < in list order
        }
        if (best.t != INFINITY && smallest > best.t) { smallest = best.t; mat = __float_as_int(S.face_nm[best.id].w); }
      }
    }
    out[i] = (smallest != INFINITY) ? make_uint2(__float_as_uint(smallest), (uint32_t)mat) : make_uint2(0u, 0xFFFFFFFFu);
  }
}

// The closest-hit search of Flyscene::ReflectDebugRay (src/flyscene.cpp:324-357): over the
// reference's flat leaf list, a leaf is scanned when the origin lies inside its box
// (Box::isInBox, src/box.h:87-93) or when the slab test gives a positive distance (those boxes
// are the ones the reference highlights: flags); strict < over leaves in list order, no
// spheres, no root gate.  d1 = the direction after the Ray ctor's single normalisation, which
// is what intersectTriangle / intersectBox see here (the debug path does not go through
// getIntersections, so there is no second one).
__global__ void k_debug_hit(DevScene S, const float4* rays, uint32_t n, float* t, int* face, float* normal, float* point,
                            unsigned char* flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 ro = rays[2 * (size_t)i], rd = rays[2 * (size_t)i + 1];
  const V3 o = v3(ro.x, ro.y, ro.z), d = v3(rd.x, rd.y, rd.z);
  float smallest = INFINITY; int best_face = -1;
  for (uint32_t b = 0; b < S.nleafboxes; ++b) {
    const float4 lo = S.leafboxes[2 * b], hi = S.leafboxes[2 * b + 1];
    const float blo[3] = {lo.x, lo.y, lo.z}, bhi[3] = {hi.x, hi.y, hi.z};
    bool scan = (o.x >= lo.x && o.x <= hi.x) && (o.y >= lo.y && o.y <= hi.y) && (o.z >= lo.z && o.z <= hi.z);
    if (!scan) {
      const float bt = ref_box_t(o, d, blo, bhi);
      if (bt != INFINITY && bt > 0) { scan = true; if (flags) flags[(size_t)i * S.nleafboxes + b] = 1; }
    }
    if (scan) {
      Hit best; best.t = INFINITY; best.id = -1;
      const uint32_t start = __float_as_uint(lo.w), cnt = __float_as_uint(hi.w);
      for (uint32_t k = 0; k < cnt; ++k) {
        Hit one; one.t = INFINITY; one.id = -1;
        test_triangle(S, (uint32_t)S.face_pos[S.leaf_faces[start + k]], o, d, &one);
        if (one.t < best.t) best = one;               // intersectTriangle: strict < in list order
      }
      if (best.t != INFINITY && smallest > best.t) { smallest = best.t; best_face = best.id; }
    }
  }
  t[i] = smallest; face[i] = best_face;
  V3 N = v3(0, 0, 0), P = v3(0, 0, 0);
  if (best_face >= 0) { const float4 nm = S.face_nm[best_face]; N = v3(nm.x, nm.y, nm.z); P = point_on_ray(o, d, smallest); }
  normal[3 * i] = N.x; normal[3 * i + 1] = N.y; normal[3 * i + 2] = N.z;
  point[3 * i] = P.x; point[3 * i + 1] = P.y; point[3 * i + 2] = P.z;
}

// canSeeLight over caller (light, position) pairs: ray construction and decision.
__global__ void k_make_shadow_rays(const float* light, const float* pos, uint32_t n, float4* rays) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 P = v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
  const V3 L = v3(light[3 * i], light[3 * i + 1], light[3 * i + 2]);
  const V3 d1s = normalized(normalized(L - P));
  rays[2 * (size_t)i] = make_float4(P.x, P.y, P.z, 0.0f);
  rays[2 * (size_t)i + 1] = make_float4(d1s.x, d1s.y, d1s.z, shadow_tmax(L, P));
}
__global__ void k_visible(DevScene S, const float* light, const float4* rays, const uint2* hits, uint32_t n,
                          unsigned char* visible, int* bmat) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 ro = rays[2 * (size_t)i], rd = rays[2 * (size_t)i + 1];
  const uint2 h = hits[i];
  int bm;
  const bool v = can_see_light(S, v3(light[3 * i], light[3 * i + 1], light[3 * i + 2]), v3(ro.x, ro.y, ro.z),
                               v3(rd.x, rd.y, rd.z), __uint_as_float(h.x), (int)h.y, &bm);
  visible[i] = v ? 1 : 0;
  if (bmat) bmat[i] = bm;
}

}  // namespace rtx
