// The hierarchy built and refitted ON THE GPU (RTX_FLAG_DEVICE_BUILD, rtx_scene_update_vertices).  Included only by rtx.hip.
//
// Replaces the same thing as bvh_build.cpp -- Box(mesh) + Flyscene::divideBox_KD (src/box.h:41-85,
// src/flyscene.cpp:77-82,1110-1185) -- for meshes that change between frames, where the host's binned-SAH build
// (0.2-0.3 s per million triangles) is a wall.  Any conservative hierarchy gives the reference's hits, because hits are
// reduced with a (t, faceId) lexicographic minimum: what is built here differs from the host's tree in
// quality only, never in results (tests/test_gpu_device_build.py renders both against the oracle).
//
// Build: per-triangle records and padded boxes in the reference's arithmetic (the host's expressions, shared through
// tri_setup below) -> 63-bit extended Morton keys of the boxes (centre and size) -> radix sort (rocPRIM through hipcub; not a hot-path kernel)
// -> Karras' binary radix tree, one thread per inner node -> boxes, triangle counts and the collapse plan bottom-up ->
// 4-wide groups top-down, written in the 64-byte packed encoding the traversal kernels load.
// No data is handed from one workgroup to another INSIDE a launch: the bottom-up pass runs in rounds (a node is
// processed in the launch after the one that completed its second child; only counters are raised by atomics within a
// launch), the top-down pass one group level per launch.  Per-XCD L2s are not coherent with each other
// (MI355X_MICROARCH.md, inter-workgroup visibility), kernel boundaries are the cheapest correct hand-off here: ~120
// launches of a few microseconds each for a build that takes milliseconds.
// Refit: records and boxes again from the new vertices (same kernel), then the groups level by level from the deepest.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>

#include "bvh_build.h"
#include "rtx_math.h"

namespace rtx {
namespace devbuild {

constexpr int kMaxLevels = kTraversalStackDepth / 3;   // levels of groups the traversal stack allows (3 entries per level)
constexpr int kMaxRounds = 100;                        // bottom-up rounds = height of the binary tree: <= 63 key bits + 28 index bits
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kLeafRef = 0x80000000u;             // child reference of the binary tree: leaf-ordered triangle k
constexpr float kCoordMax = 1e18f;                     // as in rtx_scene_build

// Counters and reductions of one build / refit (device memory, copied back once).
struct Header {
  uint32_t abs_bits;                 // largest |coordinate| <= kCoordMax (non-negative floats order as their bits)
  uint32_t root_lo[3], root_hi[3];   // Box(Tucano::Mesh) (src/box.h:41-85), ordered encoding (enc_f)
  uint32_t fin_lo[3], fin_hi[3];     // bounds of the in-range coordinates: the Morton grid
  uint32_t nwild;                    // triangles kept out of the hierarchy
  uint32_t wild_changed;             // refit: triangles of the hierarchy that have become wild (the tree cannot hold them)
  uint32_t ngroups, nslots, nleaves;
  uint32_t level_count[kMaxLevels + 2];
  uint32_t round_count[kMaxRounds + 2];
};

// floats <-> unsigned integers of the same order (atomicMin / atomicMax on floats of any sign)
__host__ __device__ __forceinline__ uint32_t enc_f(float f) {
  uint32_t b; __builtin_memcpy(&b, &f, 4);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float dec_f(uint32_t u) {
  const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  float f; __builtin_memcpy(&f, &b, 4);
  return f;
}
inline void header_init(Header* h) {
  std::memset(h, 0, sizeof *h);
  for (int k = 0; k < 3; ++k) {
    h->root_lo[k] = enc_f(3.40282347e+38f);        // src/box.h:43-45: min seeded with max(), max with numeric_limits<float>::min()
    h->root_hi[k] = enc_f(1.17549435e-38f);
    h->fin_lo[k] = enc_f(FLT_MAX); h->fin_hi[k] = enc_f(-FLT_MAX);
  }
}

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

// Largest in-range |coordinate| (the padding scale), the reference's root box, the Morton grid.
__global__ __launch_bounds__(256) void k_vert_bounds(const float* __restrict__ verts, uint32_t nverts, Header* H) {
  float amax = 0.0f;
  float mn[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f}, mx[3] = {1.17549435e-38f, 1.17549435e-38f, 1.17549435e-38f};
  float flo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, fhi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nverts; i += gridDim.x * blockDim.x)
    for (int k = 0; k < 3; ++k) {
      const float x = verts[3 * (size_t)i + k];
      if (fabsf(x) <= kCoordMax) { amax = fmaxf(amax, fabsf(x)); flo[k] = fminf(flo[k], x); fhi[k] = fmaxf(fhi[k], x); }
      if (x > mx[k]) mx[k] = x;            // (a NaN fails both, as in the reference's loop)
      if (x < mn[k]) mn[k] = x;
    }
  // 13 reductions: wave, then block through LDS, then one atomic per value and block (an atomic per WAVE on these few
  // addresses, 200 k of them at 4096 blocks, took 2.4 ms of a 4 ms build)
  amax = wave_max(amax);
  for (int k = 0; k < 3; ++k) { mn[k] = wave_min(mn[k]); mx[k] = wave_max(mx[k]); flo[k] = wave_min(flo[k]); fhi[k] = wave_max(fhi[k]); }
  __shared__ float red[4][13];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave][0] = amax;
    for (int k = 0; k < 3; ++k) { red[wave][1 + k] = mn[k]; red[wave][4 + k] = mx[k]; red[wave][7 + k] = flo[k]; red[wave][10 + k] = fhi[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 13) {
    const int j = threadIdx.x;
    const bool is_min = (j >= 1 && j <= 3) || (j >= 7 && j <= 9);
    float v = red[0][j];
    for (int w = 1; w < 4; ++w) v = is_min ? fminf(v, red[w][j]) : fmaxf(v, red[w][j]);
    if (j == 0) atomicMax(&H->abs_bits, __float_as_uint(v));
    else if (j <= 3) atomicMin(&H->root_lo[j - 1], enc_f(v));
    else if (j <= 6) atomicMax(&H->root_hi[j - 4], enc_f(v));
    else if (j <= 9) atomicMin(&H->fin_lo[j - 7], enc_f(v));
    else atomicMax(&H->fin_hi[j - 10], enc_f(v));
  }
}

// What rtx_scene_build computes per face, expression for expression (rtx.hip; both compiled with -ffp-contract=off, fp32
// and fp64 division and square root correctly rounded on either side): the 64-byte record in the reference's
// arithmetic, the conservative culling box, and whether the face has to stay out of the hierarchy.
struct TriSetup {
  float4 a_D, u_uu, v_uv, n_vv;      // the export layout of rtx_scene_get_tris
  float lo[3], hi[3];
  bool wild;
};
struct PadParams { float pad_abs, pad_scale; double wild_below; };
__device__ __forceinline__ TriSetup tri_setup(V3 a, V3 b, V3 c, const PadParams& pp) {
  TriSetup S;
  const V3 nf = face_normal(a, b, c);
  const V3 nq = normalized(nf);
  const V3 u = b - a, v = c - a;
  const float D = dot(nq, a);
  const float uu = dot(u, u), uv = dot(u, v), vv = dot(v, v);
  S.a_D = make_float4(a.x, a.y, a.z, D);
  S.u_uu = make_float4(u.x, u.y, u.z, uu);
  S.v_uv = make_float4(v.x, v.y, v.z, uv);
  S.n_vv = make_float4(nq.x, nq.y, nq.z, vv);
  const double du[3] = {(double)b.x - a.x, (double)b.y - a.y, (double)b.z - a.z};
  const double dv[3] = {(double)c.x - a.x, (double)c.y - a.y, (double)c.z - a.z};
  const double de[3] = {dv[0] - du[0], dv[1] - du[1], dv[2] - du[2]};
  const double duu = du[0] * du[0] + du[1] * du[1] + du[2] * du[2], dvv = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
  const double duv = du[0] * dv[0] + du[1] * dv[1] + du[2] * dv[2];
  const double dee = de[0] * de[0] + de[1] * de[1] + de[2] * de[2];
  const double q = (duu * dvv > 0) ? (duu * dvv - duv * duv) / (duu * dvv) : 0.0;
  bool in_range = true;
  const float co[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z};
  for (int k = 0; k < 9; ++k) in_range = in_range && fabsf(co[k]) <= kCoordMax;
  const bool finite_rec = in_range && isfinite(D) && isfinite(uu) && isfinite(vv) && isfinite(nq.x + nq.y + nq.z);
  S.wild = !(q >= pp.wild_below) || !(dee > 0) || !finite_rec;
  for (int k = 0; k < 3; ++k) { S.lo[k] = 0.0f; S.hi[k] = 0.0f; }
  if (!S.wild) {
    const double lu = sqrt(duu), lv = sqrt(dvv), le = sqrt(dee);
    const double longest = fmax(lu, fmax(lv, le));
    const double fuzz = 3e-6 / sqrt(q) * (1.0 + (lu + lv) / le);
    float pad = pp.pad_abs + (float)(longest * (2e-3 + fuzz) + fmax(lu, lv) * (1.2e-6 / q));
    pad *= pp.pad_scale;
    const float pa[3] = {a.x, a.y, a.z}, pb[3] = {b.x, b.y, b.z}, pc[3] = {c.x, c.y, c.z};
    for (int k = 0; k < 3; ++k) {
      S.lo[k] = fminf(pa[k], fminf(pb[k], pc[k])) - pad;
      S.hi[k] = fmaxf(pa[k], fmaxf(pb[k], pc[k])) + pad;
    }
  }
  return S;
}
__device__ __forceinline__ PadParams pad_params(const Header* H, float pad_scale, double wild_below) {
  PadParams pp;
  const float scene_abs = __uint_as_float(H->abs_bits);
  pp.pad_abs = fmaxf(2e-5f * scene_abs, 1e-30f);
  pp.pad_scale = pad_scale; pp.wild_below = wild_below;
  return pp;
}
__device__ __forceinline__ V3 vert_of(const float* verts, uint32_t i) { return v3(verts[3 * (size_t)i], verts[3 * (size_t)i + 1], verts[3 * (size_t)i + 2]); }

__device__ __forceinline__ uint64_t spread21(uint32_t v) {
  uint64_t x = v & 0x1FFFFFu;
  x = (x | x << 32) & 0x1F00000000FFFFull;
  x = (x | x << 16) & 0x1F0000FF0000FFull;
  x = (x | x << 8) & 0x100F00F00F00F00Full;
  x = (x | x << 4) & 0x10C30C30C30C30C3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

// Mesh order: the sort key of every face (Morton code of its padded box's centre; all ones for a wild face, so that
// the wild list is the tail of the leaf order, where the kernels expect it) and the face id as the value.
// Extended Morton code (Vinkler, Bittner, Havran 2017): the size of the box is a fourth coordinate, interleaved with the
// three of its centre -- x y z s, 16 + 16 + 16 + 15 bits -- so that a triangle much larger than its neighbours leaves
// their subtree within the first few splits instead of sitting in a leaf among them and stretching every box above it
// (the hall's twelve wall triangles among 262 k column facets: +5.5 %; soup and bunny unchanged).
__device__ __forceinline__ uint64_t extended_key(const uint32_t q[3], uint32_t qs) {
  uint64_t key = 0;
  for (int i = 15; i >= 0; --i) {
    key = key << 3 | (uint64_t)((q[0] >> i) & 1u) << 2 | (uint64_t)((q[1] >> i) & 1u) << 1 | (uint64_t)((q[2] >> i) & 1u);
    if (i > 0) key = key << 1 | (uint64_t)((qs >> (i - 1)) & 1u);
  }
  return key;
}
__global__ __launch_bounds__(256) void k_tri_keys(const float* __restrict__ verts, const uint32_t* __restrict__ idx, uint32_t ntris, Header* H,
                                                  float pad_scale, double wild_below, int extended, uint64_t* keys, uint32_t* vals) {
  const PadParams pp = pad_params(H, pad_scale, wild_below);
  float glo[3], gext[3];
  for (int k = 0; k < 3; ++k) { glo[k] = dec_f(H->fin_lo[k]); gext[k] = dec_f(H->fin_hi[k]) - glo[k]; }
  uint32_t wild_here = 0;
  for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < ntris; f += gridDim.x * blockDim.x) {
    const TriSetup S = tri_setup(vert_of(verts, idx[3 * (size_t)f]), vert_of(verts, idx[3 * (size_t)f + 1]), vert_of(verts, idx[3 * (size_t)f + 2]), pp);
    uint64_t key = ~0ull;
    if (!S.wild) {
      uint32_t q[3];
      for (int k = 0; k < 3; ++k) {
        const float c = 0.5f * S.lo[k] + 0.5f * S.hi[k];
        const double x = gext[k] > 0.0f ? ((double)c - (double)glo[k]) / (double)gext[k] * 2097152.0 : 0.0;
        q[k] = (uint32_t)fmin(fmax(x, 0.0), 2097151.0);
      }
      key = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
      if (extended) {
        const float gmax = fmaxf(gext[0], fmaxf(gext[1], gext[2]));
        const float ext = fmaxf(S.hi[0] - S.lo[0], fmaxf(S.hi[1] - S.lo[1], S.hi[2] - S.lo[2]));
        const double xs = gmax > 0.0f ? (double)ext / (double)gmax * 32768.0 : 0.0;
        const uint32_t q16[3] = {q[0] >> 5, q[1] >> 5, q[2] >> 5};
        key = extended_key(q16, (uint32_t)fmin(fmax(xs, 0.0), 32767.0));
      }
    } else {
      ++wild_here;
    }
    keys[f] = key; vals[f] = f;
  }
  if (wild_here) atomicAdd(&H->nwild, wild_here);
}

// Leaf order: everything the traversal kernels read per triangle (and the shading kernels per face), and the padded
// boxes the hierarchy is fitted to.  Runs after the sort of a build and again, on the new vertices, for a refit.
// tri_box: {lo.xyz, wild ? 1 : 0}, {hi.xyz, 0} per leaf-ordered triangle.
__global__ __launch_bounds__(256) void k_tri_setup_leaf(const float* __restrict__ verts, const uint32_t* __restrict__ idx, const int* __restrict__ tri_mat,
                                                        const int* __restrict__ tri_face, uint32_t ntris, uint32_t nreg, Header* H, float pad_scale,
                                                        double wild_below, float4* tris, float4* tris_mt, float4* face_nm, int* face_pos,
                                                        float4* tri_box) {
  const PadParams pp = pad_params(H, pad_scale, wild_below);
  uint32_t changed = 0;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < ntris; k += gridDim.x * blockDim.x) {
    const uint32_t f = (uint32_t)tri_face[k];
    const TriSetup S = tri_setup(vert_of(verts, idx[3 * (size_t)f]), vert_of(verts, idx[3 * (size_t)f + 1]), vert_of(verts, idx[3 * (size_t)f + 2]), pp);
    // device order of the record: {n, D} {v0, vv} {u, uu} {v, uv} (rtx_scene_commit)
    tris[4 * (size_t)k] = make_float4(S.n_vv.x, S.n_vv.y, S.n_vv.z, S.a_D.w);
    tris[4 * (size_t)k + 1] = make_float4(S.a_D.x, S.a_D.y, S.a_D.z, S.n_vv.w);
    tris[4 * (size_t)k + 2] = S.u_uu;
    tris[4 * (size_t)k + 3] = S.v_uv;
    if (tris_mt) {
      tris_mt[3 * (size_t)k] = make_float4(S.a_D.x, S.a_D.y, S.a_D.z, S.u_uu.x);
      tris_mt[3 * (size_t)k + 1] = make_float4(S.u_uu.y, S.u_uu.z, S.v_uv.x, S.v_uv.y);
      tris_mt[3 * (size_t)k + 2] = make_float4(S.v_uv.z, __int_as_float((int)f), 0.0f, 0.0f);
    }
    face_nm[f] = make_float4(S.n_vv.x, S.n_vv.y, S.n_vv.z, __int_as_float(tri_mat[f]));
    face_pos[f] = (int)k;
    tri_box[2 * (size_t)k] = make_float4(S.lo[0], S.lo[1], S.lo[2], S.wild ? 1.0f : 0.0f);
    tri_box[2 * (size_t)k + 1] = make_float4(S.hi[0], S.hi[1], S.hi[2], 0.0f);
    if (k < nreg && S.wild) ++changed;
  }
  if (changed) atomicAdd(&H->wild_changed, changed);
}

// ---------------------------------------------------------------- binary radix tree (Karras 2012)
// Inner node i of n - 1 covers the keys [first, last] it shares its longest common prefix with; equal keys are told
// apart by their index, so runs of identical keys become balanced subtrees.
__device__ __forceinline__ int lcp(const uint64_t* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clzll((long long)(a ^ b));
}
__global__ __launch_bounds__(256) void k_radix_tree(const uint64_t* __restrict__ keys, int n, uint2* bin_lr, uint32_t* par_int, uint32_t* par_leaf) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
    const int d = (lcp(keys, n, i, i + 1) - lcp(keys, n, i, i - 1)) < 0 ? -1 : 1;
    const int dmin = lcp(keys, n, i, i - d);
    int lmax = 2;
    while (lcp(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
      if (lcp(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lcp(keys, n, i, j);
    int s = 0, t = l;
    do {
      t = (t + 1) / 2;
      if (lcp(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    uint32_t left, right;
    if (first == gamma) { left = kLeafRef | (uint32_t)gamma; par_leaf[gamma] = (uint32_t)i; }
    else { left = (uint32_t)gamma; par_int[gamma] = (uint32_t)i; }
    if (last == gamma + 1) { right = kLeafRef | (uint32_t)(gamma + 1); par_leaf[gamma + 1] = (uint32_t)i; }
    else { right = (uint32_t)(gamma + 1); par_int[gamma + 1] = (uint32_t)i; }
    bin_lr[i] = make_uint2(left, right);
    if (i == 0) par_int[0] = kNone;
  }
}

// ---------------------------------------------------------------- bottom-up: boxes, counts, the collapse plan
// Per inner node: {lo, c1} {hi, c2} {c3, c4, plan, count | first << ... } -- c_k = the least expected cost of the subtree
// when it is given k slots of its parent's group (k = 1: it is a slot itself, either a leaf of <= 4 triangles or the
// root of a group of its own); the plan byte 0 says which (0 = leaf, else the number of slots its own group uses),
// bytes 1..3 how many of k = 2, 3, 4 slots go to the left child.  This is bvh_build.cpp's dynamic programme (expected
// group steps = summed surface area of the group roots) with one more term, because here the leaves are not given: a
// leaf slot costs its surface area x its triangles x kTriCost.
struct BinInfo { float lo[3], hi[3], c[4]; uint32_t count, first; };
__device__ __forceinline__ float half_area(const float* lo, const float* hi) {
  const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
  return ex * ey + ey * ez + ez * ex;
}
struct BinArrays { float4* a; float4* b; float4* c; uint2* lr; const float4* tri_box; };
__device__ __forceinline__ BinInfo bin_info(const BinArrays& B, uint32_t ref, float tri_cost) {
  BinInfo I;
  if (ref & kLeafRef) {
    const uint32_t k = ref & ~kLeafRef;
    const float4 l = B.tri_box[2 * (size_t)k], h = B.tri_box[2 * (size_t)k + 1];
    I.lo[0] = l.x; I.lo[1] = l.y; I.lo[2] = l.z; I.hi[0] = h.x; I.hi[1] = h.y; I.hi[2] = h.z;
    I.c[0] = half_area(I.lo, I.hi) * tri_cost; I.c[1] = I.c[2] = I.c[3] = FLT_MAX;
    I.count = 1; I.first = k;
  } else {
    const float4 a = B.a[ref], b = B.b[ref], c = B.c[ref];
    I.lo[0] = a.x; I.lo[1] = a.y; I.lo[2] = a.z; I.hi[0] = b.x; I.hi[1] = b.y; I.hi[2] = b.z;
    I.c[0] = a.w; I.c[1] = b.w; I.c[2] = c.x; I.c[3] = c.y;
    I.count = __float_as_uint(c.w) & 0x0FFFFFFFu; I.first = 0;
  }
  return I;
}
__device__ __forceinline__ uint32_t bin_plan(const BinArrays& B, uint32_t node) { return __float_as_uint(B.c[node].z); }

__global__ __launch_bounds__(256) void k_bu_leaves(uint32_t n, const uint32_t* __restrict__ par_leaf, uint32_t* arrived, uint32_t* list, Header* H) {
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const uint32_t p = par_leaf[k];
    if (atomicAdd(&arrived[p], 1u) == 1u) list[atomicAdd(&H->round_count[1], 1u)] = p;
  }
}
// Round r: the nodes whose second child was completed by round r - 1 (a launch earlier: their children's records are
// read across a kernel boundary).
__global__ __launch_bounds__(256) void k_bu_round(int r, const uint32_t* __restrict__ list, uint32_t* next, BinArrays B, const uint32_t* __restrict__ par_int,
                                                  uint32_t* arrived, Header* H, float tri_cost) {
  const uint32_t n = H->round_count[r];
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const uint32_t x = list[e];
    const uint2 lr = B.lr[x];
    const BinInfo L = bin_info(B, lr.x, tri_cost), R = bin_info(B, lr.y, tri_cost);
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = fminf(L.lo[k], R.lo[k]); hi[k] = fmaxf(L.hi[k], R.hi[k]); }
    const uint32_t count = L.count + R.count;
    float c[4] = {FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX};
    uint32_t plan = 0;
    for (int k = 2; k <= 4; ++k)
      for (int a = 1; a < k; ++a) {
        if (L.c[a - 1] == FLT_MAX || R.c[k - a - 1] == FLT_MAX) continue;
        const float v = L.c[a - 1] + R.c[k - a - 1];
        if (v < c[k - 1]) { c[k - 1] = v; plan = (plan & ~(0xFFu << (8 * (k - 1)))) | ((uint32_t)a << (8 * (k - 1))); }
      }
    const float area = half_area(lo, hi);
    int bk = 2;
    for (int k = 3; k <= 4; ++k) if (c[k - 1] < c[bk - 1]) bk = k;
    float c1 = area + c[bk - 1];
    uint32_t own = (uint32_t)bk;
    if (count <= (uint32_t)kMaxLeafTris) {
      const float as_leaf = area * (float)count * tri_cost;
      if (as_leaf <= c1) { c1 = as_leaf; own = 0; }
    }
    c[0] = c1; plan |= own;
    B.a[x] = make_float4(lo[0], lo[1], lo[2], c[0]);
    B.b[x] = make_float4(hi[0], hi[1], hi[2], c[1]);
    B.c[x] = make_float4(c[2], c[3], __uint_as_float(plan), __uint_as_float(count));
    const uint32_t p = par_int[x];
    if (p != kNone && atomicAdd(&arrived[p], 1u) == 1u) next[atomicAdd(&H->round_count[r + 1], 1u)] = p;
  }
}
// first triangle of an inner node's run: the leftmost leaf below it (runs are contiguous in leaf order)
__device__ __forceinline__ uint32_t first_tri(const BinArrays& B, uint32_t node) {
  uint32_t r = node;
  for (int guard = 0; guard < 8 && !(r & kLeafRef); ++guard) r = B.lr[r].x;
  return r & ~kLeafRef;
}

// ---------------------------------------------------------------- the 64-byte encoding (bvh_build.cpp: pack_groups)
// Same construction as the host's: the group's own bounds, a power-of-two step per axis such that 255 steps span
// them, child planes rounded outwards and checked in the fp32 arithmetic the kernels decode with.
__device__ __forceinline__ void pack_group(const float lo[4][3], const float hi[4][3], const uint32_t entry[4], const bool used[4], uint4* out) {
  float glo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, ghi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  bool any = false;
  for (int k = 0; k < 4; ++k) {
    if (!used[k]) continue;
    any = true;
    for (int a = 0; a < 3; ++a) { glo[a] = fminf(glo[a], lo[k][a]); ghi[a] = fmaxf(ghi[a], hi[k][a]); }
  }
  uint32_t planes[6] = {0, 0, 0, 0, 0, 0};
  float steps[3];
  for (int a = 0; a < 3; ++a) {
    if (!any) { glo[a] = 0.0f; ghi[a] = 0.0f; }
    int e = 27;
    const float ext = ghi[a] - glo[a];
    if (ext > 0.0f) {
      int ex; (void)frexp((double)ext / 255.0, &ex);
      e = ex + 127; e = e < 1 ? 1 : (e > 254 ? 254 : e);
      while (e < 254 && glo[a] + 255.0f * __uint_as_float((uint32_t)e << 23) < ghi[a]) ++e;
    }
    const float step = __uint_as_float((uint32_t)e << 23);
    steps[a] = step;
    for (int k = 0; k < 4; ++k) {
      uint32_t ql = 255, qh = 0;
      if (used[k]) {
        const double fl = floor(((double)lo[k][a] - (double)glo[a]) / (double)step);
        const double fh = ceil(((double)hi[k][a] - (double)glo[a]) / (double)step);
        int il = (int)fmax(0.0, fmin(255.0, fl)), ih = (int)fmax(0.0, fmin(255.0, fh));
        while (il > 0 && glo[a] + (float)il * step > lo[k][a]) --il;
        while (ih < 255 && glo[a] + (float)ih * step < hi[k][a]) ++ih;
        ql = (uint32_t)il; qh = (uint32_t)ih;
      }
      planes[a] |= ql << (8 * k);
      planes[3 + a] |= qh << (8 * k);
    }
  }
  uint32_t ent[4];
  for (int k = 0; k < 4; ++k) ent[k] = used[k] ? entry[k] : kEmptyPacked;
  out[0] = make_uint4(__float_as_uint(glo[0]), __float_as_uint(glo[1]), __float_as_uint(glo[2]), __float_as_uint(steps[0]));
  out[1] = make_uint4(ent[0], ent[1], ent[2], ent[3]);
  out[2] = make_uint4(planes[0], planes[1], planes[2], planes[3]);
  out[3] = make_uint4(planes[4], planes[5], __float_as_uint(steps[1]), __float_as_uint(steps[2]));
}
__device__ __forceinline__ uint32_t slot_entry(uint32_t ref, uint32_t count) { return count ? (0x80000000u | (count << 28) | ref) : ref; }

// ---------------------------------------------------------------- top-down: one level of groups per launch
// Item = {inner node that becomes a group, its group index}.  Child groups are numbered with an atomic counter, level
// after level, so the groups of a level are one index range (and the first three levels are groups 0 .. ntop-1, as the
// kernels' LDS staging wants them); their order inside a level depends on the run, the hits a ray finds do not.
// quads: the fp32 form {lo, ref, hi, count} x 4 per group (rtx_scene_get_bvh's layout; what a refit works on).
__global__ __launch_bounds__(256) void k_emit_level(int level, const uint2* __restrict__ qin, uint2* qout, Header* H, BinArrays B, float tri_cost, int fixed,
                                                    BvhNode* quads, uint4* nodes) {
  const uint32_t n = H->level_count[level];
  uint32_t slots_here = 0, leaves_here = 0;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const uint32_t x = qin[e].x, g = qin[e].y;
    uint32_t cand[4]; int nc = 0;
    if (!fixed) {
      uint32_t st_node[4]; int st_k[4]; int sp = 0;
      st_node[0] = x; st_k[0] = (int)(bin_plan(B, x) & 0xFFu); sp = 1;
      while (sp > 0 && nc < 4) {
        --sp;
        const uint32_t nd = st_node[sp]; const int k = st_k[sp];
        if (k <= 1 || (nd & kLeafRef)) { cand[nc++] = nd; continue; }
        const int a = (int)((bin_plan(B, nd) >> (8 * (k - 1))) & 0xFFu);
        const uint2 lr = B.lr[nd];
        st_node[sp] = lr.y; st_k[sp] = k - a; ++sp;       // right below left: the left subtree's slots come first
        st_node[sp] = lr.x; st_k[sp] = a; ++sp;
      }
    } else {
      const uint2 kids = B.lr[x];
      const uint32_t kid[2] = {kids.x, kids.y};
      for (int k = 0; k < 2; ++k) {
        const bool leaf = (kid[k] & kLeafRef) || (__float_as_uint(B.c[kid[k]].w) & 0x0FFFFFFFu) <= (uint32_t)kMaxLeafTris;
        if (leaf) cand[nc++] = kid[k];
        else { const uint2 gk = B.lr[kid[k]]; cand[nc++] = gk.x; cand[nc++] = gk.y; }
      }
    }
    float lo[4][3], hi[4][3]; uint32_t ref[4], cnt[4], entry[4]; bool used[4], inner[4];
    uint32_t n_inner = 0;
    for (int k = 0; k < 4; ++k) {
      used[k] = k < nc; inner[k] = false; ref[k] = kEmptySlot; cnt[k] = 0; entry[k] = kEmptyPacked;
      for (int a = 0; a < 3; ++a) { lo[k][a] = FLT_MAX; hi[k][a] = -FLT_MAX; }
      if (!used[k]) continue;
      const BinInfo I = bin_info(B, cand[k], tri_cost);
      for (int a = 0; a < 3; ++a) { lo[k][a] = I.lo[a]; hi[k][a] = I.hi[a]; }
      bool leaf;
      if (cand[k] & kLeafRef) leaf = true;
      else leaf = fixed ? I.count <= (uint32_t)kMaxLeafTris : (bin_plan(B, cand[k]) & 0xFFu) == 0;
      if (leaf) { ref[k] = (cand[k] & kLeafRef) ? I.first : first_tri(B, cand[k]); cnt[k] = I.count; ++leaves_here; }
      else { inner[k] = true; ++n_inner; }
      ++slots_here;
    }
    if (n_inner) {
      const uint32_t base_g = atomicAdd(&H->ngroups, n_inner);
      const uint32_t base_q = atomicAdd(&H->level_count[level + 1], n_inner);
      uint32_t j = 0;
      for (int k = 0; k < 4; ++k)
        if (inner[k]) { ref[k] = base_g + j; qout[base_q + j] = make_uint2(cand[k], base_g + j); ++j; }
    }
    for (int k = 0; k < 4; ++k) {
      BvhNode s;
      for (int a = 0; a < 3; ++a) { s.lo[a] = lo[k][a]; s.hi[a] = hi[k][a]; }
      s.leftFirst = ref[k]; s.count = cnt[k];
      quads[4 * (size_t)g + k] = s;
      entry[k] = slot_entry(ref[k], cnt[k]);
    }
    pack_group(lo, hi, entry, used, nodes + 4 * (size_t)g);
  }
  if (slots_here) { atomicAdd(&H->nslots, slots_here); atomicAdd(&H->nleaves, leaves_here); }
}
__global__ void k_emit_seed(uint2* q, Header* H) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { q[0] = make_uint2(0u, 0u); H->level_count[0] = 1; H->ngroups = 1; }
}

// ---------------------------------------------------------------- refit: one level of groups per launch, deepest first
// list: group ids of the level (null: the groups base .. base + count - 1, a device-built tree's own numbering).
// A leaf slot takes the union of its triangles' boxes, an inner slot the union of its child group's slots, which the
// launch before this one has brought up to date.
__global__ __launch_bounds__(256) void k_refit_level(const uint32_t* __restrict__ list, uint32_t base, uint32_t count, BvhNode* quads,
                                                     const float4* __restrict__ tri_box, uint4* nodes) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const uint32_t g = list ? list[base + i] : base + i;
    float lo[4][3], hi[4][3]; uint32_t entry[4]; bool used[4];
    for (int k = 0; k < 4; ++k) {
      BvhNode s = quads[4 * (size_t)g + k];
      used[k] = !(s.leftFirst == kEmptySlot && s.count == 0);
      entry[k] = kEmptyPacked;
      for (int a = 0; a < 3; ++a) { lo[k][a] = FLT_MAX; hi[k][a] = -FLT_MAX; }
      if (!used[k]) continue;
      if (s.count) {
        for (uint32_t t = s.leftFirst; t < s.leftFirst + s.count; ++t) {
          const float4 l = tri_box[2 * (size_t)t], h = tri_box[2 * (size_t)t + 1];
          lo[k][0] = fminf(lo[k][0], l.x); lo[k][1] = fminf(lo[k][1], l.y); lo[k][2] = fminf(lo[k][2], l.z);
          hi[k][0] = fmaxf(hi[k][0], h.x); hi[k][1] = fmaxf(hi[k][1], h.y); hi[k][2] = fmaxf(hi[k][2], h.z);
        }
      } else {
        for (int j = 0; j < 4; ++j) {
          const BvhNode c = quads[4 * (size_t)s.leftFirst + j];
          if (c.leftFirst == kEmptySlot && c.count == 0) continue;
          for (int a = 0; a < 3; ++a) { lo[k][a] = fminf(lo[k][a], c.lo[a]); hi[k][a] = fmaxf(hi[k][a], c.hi[a]); }
        }
      }
      for (int a = 0; a < 3; ++a) { s.lo[a] = lo[k][a]; s.hi[a] = hi[k][a]; }
      quads[4 * (size_t)g + k] = s;
      entry[k] = slot_entry(s.leftFirst, s.count);
    }
    pack_group(lo, hi, entry, used, nodes + 4 * (size_t)g);
  }
}

}  // namespace devbuild
}  // namespace rtx
