// librtx.so -- C ABI (include/rtx.h) over the HIP kernels in rtx_kernels.h.
// Host side: scene flattening with the reference's arithmetic, BVH build, uploads,
// the generation loop of the ray queue, statistics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtx.h"
#include "bvh_build.h"
#include "rtx_kernels.h"
#include "bvh_device.h"

using namespace rtx;
static_assert(kStackDepth == kTraversalStackDepth, "the builder's depth fallback and the kernels' stack must agree");

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(RTX_EHIP, "%s: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return fail(RTX_EHIP, "hipMalloc(%zu): %s", want, hipGetErrorString(e)); }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct EventPool {
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  hipEvent_t get() {
    if (used == ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; ev.push_back(e); }
    return ev[used++];
  }
  void reset() { used = 0; }
  void release() { for (auto e : ev) (void)hipEventDestroy(e); ev.clear(); used = 0; }
};

}  // namespace

// Calls that may be in flight on one scene (rtx_render_*_enqueue): each has its own streams and workspace.
#ifndef RTX_SLOTS
#define RTX_SLOTS 4
#endif
constexpr int kSlots = RTX_SLOTS;

struct rtx_scene {
  // host description
  std::vector<float> verts;
  std::vector<uint32_t> idx;
  std::vector<int32_t> tri_mat;
  std::vector<rtx_material> mats;
  std::vector<float> spheres;
  std::vector<int32_t> sph_mat;
  std::vector<float> lights, slights;
  std::vector<float> leaf_lo, leaf_hi;        // reference leaf list (divideBox_KD), 3 floats per leaf
  std::vector<uint32_t> leaf_start, leaf_count;
  std::vector<int32_t> leaf_faces;
  bool have_leaves = false;
  uint32_t flags = 0;
  // host-built acceleration data (rtx_scene_build)
  bool built = false;
  std::vector<float4> h_tri_leaf, h_face_nm;
  std::vector<int32_t> h_tri_face;
  float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
  uint32_t nwild = 0;
  float domain = INFINITY;             // ray origins beyond this |coordinate| are outside what the boxes' padding covers
  // committed state
  bool committed = false;
  int device = -1;
  BvhBuildResult bvh;
  DevScene dev{};
  DevBuf d_nodes, d_tris, d_tris_mt, d_tri_face, d_face_nm, d_spheres, d_sph_mat, d_mats, d_lights;
  DevBuf d_slights, d_leafboxes, d_leaf_faces, d_face_pos;
  // The hierarchy built / refitted on the GPU (bvh_device.h; RTX_FLAG_DEVICE_BUILD, rtx_scene_update_vertices).
  bool geom_dev = false;               // d_nodes / d_tris / d_tri_face / d_face_nm / d_face_pos hold the current mesh on `device`
  bool device_built = false;           // ... and the hierarchy among them was built on the GPU
  bool host_stale = false;             // h_tri_leaf / h_tri_face / h_face_nm / bvh.quads / bvh.order are behind the device's state
  bool verts_stale = false;            // `verts` is behind d_verts (rtx_scene_update_vertices_dev)
  bool mesh_dev = false;               // d_verts / d_idx / d_tri_mat hold the mesh
  bool refit_ready = false;            // d_quads / d_tri_box / the level lists match d_nodes
  bool keep_build_ws = false;          // keep the build's temporaries (a caller that rebuilds per frame)
  DevBuf d_verts, d_idx, d_tri_mat, d_quads, d_tri_box, d_level_groups, d_hdr;
  std::vector<uint32_t> level_start;   // groups of level l: d_level_groups[level_start[l] .. level_start[l + 1]) -- or, for a tree built
  bool level_identity = false;         // on the GPU (level_identity), that range of group indices itself
  DevBuf b_keys[2], b_vals, b_sort_tmp, b_lr, b_par_int, b_par_leaf, b_arrived, b_list[2], b_bin[3], b_queue[2], b_nodes, b_quads;
  float device_build_ms = 0.0f, refit_ms = 0.0f;
  // Workspace of one call: its streams and every buffer its kernels write.  Two of them (one per call slot,
  // below), so that two render calls can be in flight on the GPU at the same time: frame k+1's launches fill the
  // CUs that frame k's persistent traversal kernels leave free while their last waves drain.
  struct Slot {
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;     // bounce launches that overlap the shadow launch of the same generation (run_pipeline)
    DevBuf w_cam_tab;                  // camera tables (k_camera_tables)
    std::vector<DevCamera> cam_tab_of; // the cameras w_cam_tab (and w_frames) currently hold the tables of (valid: cam_tab_ok)
    bool cam_tab_ok = false;
    DevBuf w_frames;                   // per-frame inverse views and centres of a batch (rtx_render_tiles_batch_dev)
    DevBuf w_stack_ovf;                // global overflow columns of the per-lane traversal stacks
    DevBuf w_stack_ovf2;               // ... of the launches on stream2 (they run beside launches on `stream`)
    DevBuf w_rays[2], w_hits[3], w_parents, w_shadow_rays, w_shadow_hits, w_gi_rays, w_gi_hits, w_counters;
    DevBuf w_camera;                   // device copy of a call's DevCamera (run_pipeline's direct mode: k_trace_shadow_direct reads it)
    DevBuf w_guard;                    // counters of the domain guard / verification kernels
    DevBuf w_acc, w_acc_flag;          // fixed-point pixel sums of ray-tree calls (reflection + refraction)
    template <class F> void for_each_buf(F f) {
      f(w_cam_tab); f(w_frames); f(w_stack_ovf); f(w_stack_ovf2); f(w_rays[0]); f(w_rays[1]); f(w_hits[0]); f(w_hits[1]); f(w_hits[2]);
      f(w_camera); f(w_parents); f(w_shadow_rays); f(w_shadow_hits); f(w_gi_rays); f(w_gi_hits); f(w_counters); f(w_acc); f(w_acc_flag); f(w_guard);
    }
  };
  Slot slots[kSlots];
  DevBuf w_io[6];                      // staging of the host-pointer entry points (synchronous: never in use by a call in flight)
  DevBuf m_part[3];                    // rtx_render_multi: this scene's compact tile buffers (rgb, t, face id) on its device
  DevBuf m_gather[3];                  // ... and, on the first scene, the parts of all scenes one after the other
  // Per-call bookkeeping.  Two of them, so that a render call can be enqueued while the one before is
  // still running (rtx_render_dev_enqueue / rtx_render_finish); every other entry point begins and
  // finishes its call in one go.
  struct Call {
    EventPool events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_spans, all_spans;
    size_t camera_span = (size_t)-1;   // index into trace_spans of the k_trace_camera launch
    std::vector<int> span_kind;        // kind of each trace span, -1 = not a scheduler-driven launch
    int mode[3] = {-1, -1, -1};        // scheduler mode of this call per launch kind (-1: none picked yet)
    bool timed[3] = {false, false, false};   // ... and whether that pick is one of the tuner's trials
    uint32_t* hg = nullptr;            // pinned host copy of the guard / verification counters (4 words)
    bool guard = false;                // ... in use by this call
    uint32_t* hc = nullptr;            // pinned host copy of the pipeline's counters
    size_t hc_words = 0;
    bool pending = false;              // enqueued, not finished yet
    bool tunable = false;              // a render call: its launches may be tuner trials (set before the first launch)
    bool throughput = false;           // enqueued, and other calls may run beside it
    bool alone = false;                // its launches may be tuner trials: no other call runs beside it
    int bucket = 0;                    // size class of the call (primary rays), selects the tuner state
    bool unreported = false;           // enqueued by rtx_render_*_enqueue and not yet handed out by rtx_render_finish
    bool pipeline = false, shade = false, shadows = false, tune = false;   // what finish_call has to account for
    int max_gen = 0;
    uint32_t NLS = 0, G = 0;
    rtx_stats stats{};
  };
  Call calls[kSlots];
  int cur = 0;
  // Scheduler tuning per launch kind (camera / shadow queue / bounce queue) and size class of the call (its
  // primary rays, classes a factor of 4 wide: a frame, a batch of 8 frames and a 1/8 share each have a state of
  // their own, and calls of different sizes taking turns do not disturb each other): after a first untimed
  // call, the next calls of a class run the scheduler modes in turn on the real work, twice each, and
  // the fastest is kept (results are identical whichever runs, tests/test_gpu_parity.py).
  struct Tune {
    bool started = false, too_small = false;
    int choice = -1;
    int launched[3] = {0, 0, 0}, done[3] = {0, 0, 0};
    double ms_per_ray[3] = {0, 0, 0};
  };
  static constexpr int kBuckets = 16;
  Tune tune[3][kBuckets];
  rtx_stats stats{};                   // of the last finished call
  rtx_scene_info info{};
  int ncu = 256;
};

static rtx_scene::Call& cur_call(rtx_scene* s) { return s->calls[s->cur]; }
static rtx_scene::Slot& ws(rtx_scene* s) { return s->slots[s->cur]; }          // the workspace of the current call
static int finish_call(rtx_scene* s, rtx_scene::Call& c);
static int scratch_init(int device);
static int trace_grid(const rtx_scene* s) { return s->ncu * 8; }      // 8 blocks x 4 waves = 32 waves per CU

extern "C" const char* rtx_last_error(void) { return g_err.c_str(); }

extern "C" int rtx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int rtx_host_alloc(uint64_t bytes, void** out) {
  if (!out) return fail(RTX_EINVAL, "null argument");
  *out = nullptr;
  HIP_TRY(hipHostMalloc(out, (size_t)std::max<uint64_t>(bytes, 1), hipHostMallocDefault));
  return RTX_OK;
}
extern "C" int rtx_host_free(void* p) {
  if (!p) return RTX_OK;
  HIP_TRY(hipHostFree(p));
  return RTX_OK;
}

extern "C" int rtx_scene_create(const float* verts, uint32_t nverts, const uint32_t* idx, const int32_t* tri_mat,
                                uint32_t ntris, const rtx_material* mats, uint32_t nmats, rtx_scene** out) {
  if (!out) return fail(RTX_EINVAL, "out is null");
  if ((nverts && !verts) || (ntris && !idx) || (nmats && !mats)) return fail(RTX_EINVAL, "null array with non-zero count");
  if (ntris >= (1u << 28)) return fail(RTX_ELIMIT, "at most 2^28-1 triangles (got %u)", ntris);
  for (size_t i = 0; i < (size_t)ntris * 3; ++i)
    if (idx[i] >= nverts) return fail(RTX_EINVAL, "vertex index %u out of range at position %zu (nverts=%u)", idx[i], i, nverts);
  rtx_scene* s = new rtx_scene();
  s->verts.assign(verts, verts + (size_t)nverts * 3);
  s->idx.assign(idx, idx + (size_t)ntris * 3);
  if (tri_mat) s->tri_mat.assign(tri_mat, tri_mat + ntris); else s->tri_mat.assign(ntris, -1);
  s->mats.assign(mats, mats + nmats);
  *out = s;
  return RTX_OK;
}

static void release_device(rtx_scene* s) {
  if (s->device >= 0) (void)hipSetDevice(s->device);
  DevBuf* bufs[] = {&s->d_nodes, &s->d_tris, &s->d_tris_mt, &s->d_tri_face, &s->d_face_nm, &s->d_spheres, &s->d_sph_mat, &s->d_mats,
                    &s->d_lights, &s->d_slights, &s->d_leafboxes, &s->d_leaf_faces, &s->d_face_pos,
                    &s->d_verts, &s->d_idx, &s->d_tri_mat, &s->d_quads, &s->d_tri_box, &s->d_level_groups, &s->d_hdr,
                    &s->b_keys[0], &s->b_keys[1], &s->b_vals, &s->b_sort_tmp, &s->b_lr, &s->b_par_int, &s->b_par_leaf, &s->b_arrived,
                    &s->b_list[0], &s->b_list[1], &s->b_bin[0], &s->b_bin[1], &s->b_bin[2], &s->b_queue[0], &s->b_queue[1], &s->b_nodes, &s->b_quads};
  for (rtx_scene::Slot& w : s->slots) {                        // calls still in flight
    if (w.stream) (void)hipStreamSynchronize(w.stream);
    if (w.stream2) (void)hipStreamSynchronize(w.stream2);
  }
  for (DevBuf* b : bufs) b->release();
  for (DevBuf& b : s->w_io) b.release();
  for (DevBuf& b : s->m_part) b.release();
  for (DevBuf& b : s->m_gather) b.release();
  for (rtx_scene::Slot& w : s->slots) {
    w.for_each_buf([](DevBuf& b) { b.release(); });
    w.cam_tab_ok = false;
    if (w.stream) { (void)hipStreamDestroy(w.stream); w.stream = nullptr; }
    if (w.stream2) { (void)hipStreamDestroy(w.stream2); w.stream2 = nullptr; }
  }
  for (rtx_scene::Call& c : s->calls) {
    c.events.release();
    if (c.hc) (void)hipHostFree(c.hc);
    if (c.hg) (void)hipHostFree(c.hg);
    c.hg = nullptr;
    c.hc = nullptr; c.hc_words = 0; c.pending = false; c.unreported = false;
  }
  s->committed = false;
  s->geom_dev = false; s->mesh_dev = false; s->refit_ready = false;
}

extern "C" int rtx_scene_destroy(rtx_scene* s) {
  if (!s) return RTX_OK;
  release_device(s);
  delete s;
  return RTX_OK;
}

extern "C" int rtx_scene_add_material(rtx_scene* s, const rtx_material* m) {
  if (!s || !m) return fail(RTX_EINVAL, "null argument");
  s->mats.push_back(*m);
  s->committed = false;
  return (int)s->mats.size() - 1;
}

extern "C" int rtx_scene_set_spheres(rtx_scene* s, const float* cr4, const int32_t* mat, uint32_t n) {
  if (!s || (n && (!cr4 || !mat))) return fail(RTX_EINVAL, "null argument");
  s->spheres.assign(cr4, cr4 + (size_t)n * 4);
  s->sph_mat.assign(mat, mat + n);
  s->committed = false;
  return RTX_OK;
}

extern "C" int rtx_scene_set_flags(rtx_scene* s, uint32_t flags) {
  if (!s) return fail(RTX_EINVAL, "null scene");
  if (flags & ~(uint32_t)(RTX_FLAG_NO_ROOT_GATE | RTX_FLAG_SCHED_REFILL | RTX_FLAG_SCHED_GROUPS | RTX_FLAG_SCHED_SPLIT | RTX_FLAG_VERIFY | RTX_FLAG_FAST_TRIANGLES | RTX_FLAG_PARENT_LIST | RTX_FLAG_DEVICE_BUILD))
    return fail(RTX_EINVAL, "unknown flag bits 0x%x", flags);
  if ((flags & RTX_FLAG_SCHED_REFILL) && (flags & RTX_FLAG_SCHED_GROUPS)) return fail(RTX_EINVAL, "pick one scheduler");
  if ((flags & RTX_FLAG_SCHED_SPLIT) && !(flags & RTX_FLAG_SCHED_REFILL)) return fail(RTX_EINVAL, "RTX_FLAG_SCHED_SPLIT modifies RTX_FLAG_SCHED_REFILL");
  if ((flags ^ s->flags) & (RTX_FLAG_NO_ROOT_GATE | RTX_FLAG_FAST_TRIANGLES)) s->committed = false;   // the scheduler bits are host-side only
  s->flags = flags;
  return RTX_OK;
}

extern "C" int rtx_scene_set_lights(rtx_scene* s, const float* xyz, uint32_t n, const float* sxyzr, uint32_t m) {
  if (!s || (n && !xyz) || (m && !sxyzr)) return fail(RTX_EINVAL, "null argument");
  s->lights.assign(xyz, xyz + (size_t)n * 3);
  s->slights.assign(sxyzr, sxyzr + (size_t)m * 4);
  s->committed = false;
  return RTX_OK;
}

template <class T>
static int upload(DevBuf& b, const std::vector<T>& v, hipStream_t st) {
  size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
  if (int rc = b.ensure(bytes)) return rc;
  if (!v.empty()) HIP_TRY(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
  return RTX_OK;
}


// ------------------------------------------------------------------ reference leaf list
extern "C" int rtx_scene_set_leafboxes(rtx_scene* s, const float* bmin_xyz, const float* bmax_xyz, const uint32_t* start,
                                       const uint32_t* count, uint32_t nleaves, const int32_t* faces, uint32_t nfaces) {
  if (!s || (nleaves && (!bmin_xyz || !bmax_xyz || !start || !count)) || (nfaces && !faces)) return fail(RTX_EINVAL, "null argument");
  const uint32_t ntris = (uint32_t)s->tri_mat.size();
  for (uint32_t i = 0; i < nleaves; ++i)
    if ((uint64_t)start[i] + count[i] > nfaces) return fail(RTX_EINVAL, "leaf %u exceeds the face list", i);
  for (uint32_t i = 0; i < nfaces; ++i)
    if (faces[i] < 0 || (uint32_t)faces[i] >= ntris) return fail(RTX_EINVAL, "leaf face id %d out of range", faces[i]);
  s->leaf_lo.assign(bmin_xyz, bmin_xyz + (size_t)nleaves * 3);
  s->leaf_hi.assign(bmax_xyz, bmax_xyz + (size_t)nleaves * 3);
  s->leaf_start.assign(start, start + nleaves);
  s->leaf_count.assign(count, count + nleaves);
  s->leaf_faces.assign(faces, faces + nfaces);
  s->have_leaves = true;
  s->committed = false;
  return RTX_OK;
}

// Flyscene::divideBox_KD with its helpers averagePoint, axisToDivide, isInBox
// (src/flyscene.cpp:974-993,1070-1185) and Box(mesh) (src/box.h:41-85): the flat, lossy leaf
// list that trace_global_illum and the debug ray walk.  One-off host work.
// Returns the number of leaves, or RTX_ELIMIT when the reference's split would never
// terminate (detected by an iteration cap).
extern "C" int rtx_scene_build_reference_leafboxes(rtx_scene* s, int32_t max_faces, int64_t iter_cap) {
  if (!s) return fail(RTX_EINVAL, "null scene");
  const uint32_t ntris = (uint32_t)s->tri_mat.size();
  const uint32_t nverts = (uint32_t)(s->verts.size() / 3);
  if (max_faces <= 0) max_faces = (int32_t)std::sqrt((double)(ntris + 100));     // src/flyscene.cpp:79
  if (iter_cap <= 0) iter_cap = 64ll * ntris + 4096;
  struct QBox { float lo[3], hi[3]; std::vector<int32_t> faces; };
  float mx[3] = {1.17549435e-38f, 1.17549435e-38f, 1.17549435e-38f};
  float mn[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f};
  for (uint32_t i = 0; i < nverts; ++i)
    for (int k = 0; k < 3; ++k) {
      float x = s->verts[3 * (size_t)i + k];
      if (x > mx[k]) mx[k] = x;
      if (x < mn[k]) mn[k] = x;
    }
  auto vtx = [&](int32_t f, int c, int k) { return s->verts[3 * (size_t)s->idx[3 * (size_t)f + c] + k]; };
  auto in_box = [&](const float* lo, const float* hi, int32_t f) {
    for (int c = 0; c < 3; ++c) {
      const float x = vtx(f, c, 0), y = vtx(f, c, 1), z = vtx(f, c, 2);
      if (hi[0] >= x && lo[0] <= x && hi[1] >= y && lo[1] <= y && hi[2] >= z && lo[2] <= z) return true;
    }
    return false;
  };
  std::vector<float> llo, lhi; std::vector<uint32_t> lstart, lcount; std::vector<int32_t> lfaces;
  std::vector<QBox> queue(1);
  std::memcpy(queue[0].lo, mn, 12); std::memcpy(queue[0].hi, mx, 12);
  queue[0].faces.resize(ntris);
  for (uint32_t i = 0; i < ntris; ++i) queue[0].faces[i] = (int32_t)i;
  size_t head = 0;
  int64_t iters = 0;
  size_t queued_ids = ntris;                                  // face ids held by boxes still in the queue
  const size_t id_cap = (size_t)64 * ntris + 65536;
  while (head < queue.size()) {
    if (++iters > iter_cap) return fail(RTX_ELIMIT, "the reference's divideBox_KD does not terminate on this mesh (SURVEY F13)");
    QBox box = std::move(queue[head]); ++head;
    if (box.faces.size() <= (size_t)max_faces && !box.faces.empty()) {
      for (int k = 0; k < 3; ++k) { llo.push_back(box.lo[k]); lhi.push_back(box.hi[k]); }
      lstart.push_back((uint32_t)lfaces.size()); lcount.push_back((uint32_t)box.faces.size());
      lfaces.insert(lfaces.end(), box.faces.begin(), box.faces.end());
      continue;
    }
    int axis = -1; float best = 1.17549435e-38f;                       // axisToDivide: '<=' keeps the last of equals
    for (int k = 0; k < 3; ++k) { float d = box.hi[k] - box.lo[k]; if (best <= d) { best = d; axis = k; } }
    float sx = 0, sy = 0, sz = 0;                                       // averagePoint: running fp32 sums
    for (int32_t f : box.faces) for (int c = 0; c < 3; ++c) { sx += vtx(f, c, 0); sy += vtx(f, c, 1); sz += vtx(f, c, 2); }
    const float size = (float)(box.faces.size() * 3);
    const float avg[3] = {sx / size, sy / size, sz / size};
    const int ax = (axis == 0 || axis == 1) ? axis : 2;
    QBox b1, b2;
    std::memcpy(b1.lo, box.lo, 12); std::memcpy(b1.hi, box.hi, 12); b1.hi[ax] = avg[ax];
    std::memcpy(b2.lo, box.lo, 12); std::memcpy(b2.hi, box.hi, 12); b2.lo[ax] = avg[ax];
    for (int32_t f : box.faces) {
      if (in_box(b1.lo, b1.hi, f)) b1.faces.push_back(f);
      if (in_box(b2.lo, b2.hi, f)) b2.faces.push_back(f);
    }
    // Every split copies a face into each child that holds one of its vertices: on a mesh whose triangles straddle the
    // planes the queued lists can double per level, and memory would run out long before the pop cap above says so.
    queued_ids += b1.faces.size() + b2.faces.size();
    queued_ids -= std::min(queued_ids, box.faces.size());
    if (queued_ids > id_cap || (b1.faces.size() == box.faces.size() && b2.faces.size() == box.faces.size() && box.faces.size() > (size_t)max_faces))
      return fail(RTX_ELIMIT, "the reference's divideBox_KD does not terminate on this mesh (its face lists stop shrinking; SURVEY F13)");
    queue.push_back(std::move(b1));
    queue.push_back(std::move(b2));
    if (head > 4096 && head * 2 > queue.size()) { queue.erase(queue.begin(), queue.begin() + (long)head); head = 0; }
  }
  s->leaf_lo = llo; s->leaf_hi = lhi; s->leaf_start = lstart; s->leaf_count = lcount; s->leaf_faces = lfaces;
  s->have_leaves = true;
  s->committed = false;
  return (int)lstart.size();
}

extern "C" int rtx_scene_get_leafboxes(const rtx_scene* s, float* bmin_xyz, float* bmax_xyz, uint32_t* start, uint32_t* count,
                                       int32_t* faces, uint32_t* nleaves, uint32_t* nfaces) {
  if (!s) return fail(RTX_EINVAL, "null scene");
  if (!s->have_leaves) return fail(RTX_ESTATE, "no leaf list (rtx_scene_build_reference_leafboxes / rtx_scene_set_leafboxes)");
  if (nleaves) *nleaves = (uint32_t)s->leaf_start.size();
  if (nfaces) *nfaces = (uint32_t)s->leaf_faces.size();
  if (bmin_xyz) std::memcpy(bmin_xyz, s->leaf_lo.data(), s->leaf_lo.size() * 4);
  if (bmax_xyz) std::memcpy(bmax_xyz, s->leaf_hi.data(), s->leaf_hi.size() * 4);
  if (start) std::memcpy(start, s->leaf_start.data(), s->leaf_start.size() * 4);
  if (count) std::memcpy(count, s->leaf_count.data(), s->leaf_count.size() * 4);
  if (faces) std::memcpy(faces, s->leaf_faces.data(), s->leaf_faces.size() * 4);
  return RTX_OK;
}

extern "C" int rtx_scene_build(rtx_scene* s) {
  if (!s) return fail(RTX_EINVAL, "null scene");
  if (s->built) return RTX_OK;
  auto t0 = std::chrono::steady_clock::now();
  const uint32_t ntris = (uint32_t)s->tri_mat.size();
  const uint32_t nverts = (uint32_t)(s->verts.size() / 3);
  // ---- flatten: per-face constants in the reference's arithmetic (rtx_math.h)
  std::unique_ptr<float4[]> tri_rec(new float4[std::max<size_t>((size_t)ntris * 4, 1)]);        // mesh order first, permuted below (every record is written by the face loop)
  std::vector<float4>& face_nm = s->h_face_nm;
  face_nm.assign(std::max<uint32_t>(ntris, 1), make_float4(0, 0, 0, 0));
  std::vector<float> lo, hi;               // padded boxes of the triangles that go into the BVH
  std::vector<uint32_t> regular, wild;     // mesh face ids
  lo.reserve((size_t)ntris * 3); hi.reserve((size_t)ntris * 3);
  float scene_abs = 0.0f;
  // Coordinates beyond kCoordMax (or not numbers) make their own triangles wild instead of stretching the padding of
  // every box: the slab arithmetic multiplies box coordinates by up to 1e18 (cull_inv) and has to stay finite in fp32.
  constexpr float kCoordMax = 1e18f;
  for (size_t i = 0; i < s->verts.size(); ++i)
    if (std::fabs(s->verts[i]) <= kCoordMax) scene_abs = std::max(scene_abs, std::fabs(s->verts[i]));
  const float pad_abs = std::max(2e-5f * scene_abs, 1e-30f);
  s->domain = scene_abs > 0.0f ? 30.0f * scene_abs : INFINITY;        // what pad_abs is good for; beyond: brute force
  const double kWildBelow = [] { const char* e = std::getenv("RTX_WILD_BELOW"); const double v = e ? std::atof(e) : 0.0; return v > 0 ? v : 2.5e-6; }();   // (tuning / test hook)
  // (test hook: RTX_TEST_PAD_SCALE < 0 shrinks the boxes, so that the hierarchy loses hits and RTX_FLAG_VERIFY has something to find)
  const float kPadScale = [] { const char* e = std::getenv("RTX_TEST_PAD_SCALE"); return e ? (float)std::atof(e) : 1.0f; }();
  auto vert = [&](uint32_t i) { return v3(s->verts[3 * i], s->verts[3 * i + 1], s->verts[3 * i + 2]); };
  const int threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  // faces [f0, f1): records into their own places, boxes and face lists appended to the caller's vectors (a chunk per thread
  // for large meshes, joined in face order below: the same vectors as one pass over all faces gives)
  auto face_range = [&](uint32_t f0, uint32_t f1, std::vector<float>& lo, std::vector<float>& hi, std::vector<uint32_t>& regular, std::vector<uint32_t>& wild) {
  for (uint32_t f = f0; f < f1; ++f) {
    const V3 a = vert(s->idx[3 * f]), b = vert(s->idx[3 * f + 1]), c = vert(s->idx[3 * f + 2]);
    const V3 nf = face_normal(a, b, c);       // tucano/mesh.hpp:461-463
    const V3 nq = normalized(nf);             // src/flyscene.cpp:839 (identity model matrix)
    const V3 u = b - a, v = c - a;            // src/Intersect.cpp:26-27
    const float D = dot(nq, a);               // src/Intersect.cpp:10
    const float uu = dot(u, u), uv = dot(u, v), vv = dot(v, v);
    tri_rec[4 * (size_t)f + 0] = make_float4(a.x, a.y, a.z, D);
    tri_rec[4 * (size_t)f + 1] = make_float4(u.x, u.y, u.z, uu);
    tri_rec[4 * (size_t)f + 2] = make_float4(v.x, v.y, v.z, uv);
    tri_rec[4 * (size_t)f + 3] = make_float4(nq.x, nq.y, nq.z, vv);
    int32_t m = s->tri_mat[f];
    float mw; std::memcpy(&mw, &m, 4);
    face_nm[f] = make_float4(nq.x, nq.y, nq.z, mw);
    // Conservative culling box.  The reference's isInTriangle (src/Intersect.cpp:24-46) works on fp32
    // barycentrics s = s1 / D, t = t1 / D with D = uv^2 - uu vv.  With q = sin^2(angle at v0):
    //  * D and the numerators lose ~15 eps / q of their value to cancellation, so the accepted set is the
    //    exact triangle scaled about v0 by up to 1 + 15 eps / q: points up to that fraction of an edge
    //    beyond the far edge, along the triangle, pass (measured 7.5e-7 / q on thin triangles, numpy
    //    replay and the oracle; tests/test_cabi_host.py aims rays there);
    //  * across the edges it reaches outside by about 40 eps / sqrt(q) * (1 + (|u| + |v|) / |v2 - v1|)
    //    * (longest edge).
    // Boxes are padded by both (the first with 1.2e-6 / q = 20 eps / q) plus margins for the plane
    // distance and slab rounding.  Below q = 2.5e-6 (D off by a third and more, then its sign) the
    // accepted set is no longer bounded: such triangles stay out of the hierarchy and are tested by
    // every ray ("wild" list), which is what the reference's brute-force scan does for all of them.
    const double du[3] = {(double)b.x - a.x, (double)b.y - a.y, (double)b.z - a.z};
    const double dv[3] = {(double)c.x - a.x, (double)c.y - a.y, (double)c.z - a.z};
    const double de[3] = {dv[0] - du[0], dv[1] - du[1], dv[2] - du[2]};
    const double duu = du[0] * du[0] + du[1] * du[1] + du[2] * du[2], dvv = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
    const double duv = du[0] * dv[0] + du[1] * dv[1] + du[2] * dv[2];
    const double dee = de[0] * de[0] + de[1] * de[1] + de[2] * de[2];
    const double q = (duu * dvv > 0) ? (duu * dvv - duv * duv) / (duu * dvv) : 0.0;
    bool in_range = true;
    for (const V3& p : {a, b, c}) in_range = in_range && std::fabs(p.x) <= kCoordMax && std::fabs(p.y) <= kCoordMax && std::fabs(p.z) <= kCoordMax;
    const bool finite_rec = in_range && std::isfinite(D) && std::isfinite(uu) && std::isfinite(vv) && std::isfinite(nq.x + nq.y + nq.z);
    if (!(q >= kWildBelow) || !(dee > 0) || !finite_rec) { wild.push_back(f); continue; }
    const double lu = std::sqrt(duu), lv = std::sqrt(dvv), le = std::sqrt(dee);
    const double longest = std::max(lu, std::max(lv, le));
    const double fuzz = 3e-6 / std::sqrt(q) * (1.0 + (lu + lv) / le);
    float pad = pad_abs + (float)(longest * (2e-3 + fuzz) + std::max(lu, lv) * (1.2e-6 / q));
    pad *= kPadScale;
    const float pa[3] = {a.x, a.y, a.z}, pb[3] = {b.x, b.y, b.z}, pc[3] = {c.x, c.y, c.z};
    float bl[3], bh[3];
    for (int k = 0; k < 3; ++k) {
      bl[k] = std::min(pa[k], std::min(pb[k], pc[k])); bh[k] = std::max(pa[k], std::max(pb[k], pc[k]));
    }
    for (int k = 0; k < 3; ++k) { lo.push_back(bl[k] - pad); hi.push_back(bh[k] + pad); }
    regular.push_back(f);
  }
  };
  auto tR = std::chrono::steady_clock::now();
  if (std::getenv("RTX_BUILD_TIMING")) std::fprintf(stderr, "before the face loop %.1f ms; ", std::chrono::duration<float, std::milli>(tR - t0).count());
  if (threads == 1 || ntris < 65536) {
    face_range(0, ntris, lo, hi, regular, wild);
  } else {
    std::vector<std::vector<float>> lo_c((size_t)threads), hi_c((size_t)threads);
    std::vector<std::vector<uint32_t>> reg_c((size_t)threads), wild_c((size_t)threads);
    const uint32_t step = (ntris + (uint32_t)threads - 1) / (uint32_t)threads;
    std::vector<std::thread> pool;
    for (int k = 0; k < threads; ++k)
      pool.emplace_back([&, k] {
        const uint32_t f0 = std::min(ntris, (uint32_t)k * step), f1 = std::min(ntris, f0 + step);
        lo_c[(size_t)k].reserve((size_t)(f1 - f0) * 3); hi_c[(size_t)k].reserve((size_t)(f1 - f0) * 3); reg_c[(size_t)k].reserve(f1 - f0);
        face_range(f0, f1, lo_c[(size_t)k], hi_c[(size_t)k], reg_c[(size_t)k], wild_c[(size_t)k]);
      });
    for (std::thread& t : pool) t.join();
    // joined in chunk order, each chunk copied to its place by a thread
    std::vector<size_t> at_reg((size_t)threads + 1, 0), at_wild((size_t)threads + 1, 0);
    for (int k = 0; k < threads; ++k) { at_reg[(size_t)k + 1] = at_reg[(size_t)k] + reg_c[(size_t)k].size(); at_wild[(size_t)k + 1] = at_wild[(size_t)k] + wild_c[(size_t)k].size(); }
    lo.resize(at_reg[(size_t)threads] * 3); hi.resize(at_reg[(size_t)threads] * 3); regular.resize(at_reg[(size_t)threads]); wild.resize(at_wild[(size_t)threads]);
    pool.clear();
    for (int k = 0; k < threads; ++k)
      pool.emplace_back([&, k] {
        const size_t r0 = at_reg[(size_t)k], nr = reg_c[(size_t)k].size(), w0 = at_wild[(size_t)k], nw = wild_c[(size_t)k].size();
        if (nr) { std::memcpy(&lo[3 * r0], lo_c[(size_t)k].data(), nr * 12); std::memcpy(&hi[3 * r0], hi_c[(size_t)k].data(), nr * 12); std::memcpy(&regular[r0], reg_c[(size_t)k].data(), nr * 4); }
        if (nw) std::memcpy(&wild[w0], wild_c[(size_t)k].data(), nw * 4);
      });
    for (std::thread& t : pool) t.join();
  }
  // ---- BVH over the regular triangles; wild ones are appended after the leaf-ordered run
  const uint32_t nreg = (uint32_t)regular.size();
  auto tA = std::chrono::steady_clock::now();
  build_bvh(lo.data(), hi.data(), nreg, threads, &s->bvh);
  auto tB = std::chrono::steady_clock::now();
  collapse_to_quads(&s->bvh);
  auto tC = std::chrono::steady_clock::now();
  if (std::getenv("RTX_BUILD_TIMING")) std::fprintf(stderr, "records %.1f ms, build_bvh %.1f ms, collapse %.1f ms\n", std::chrono::duration<float, std::milli>(tA - t0).count(),
      std::chrono::duration<float, std::milli>(tB - tA).count(), std::chrono::duration<float, std::milli>(tC - tB).count());
  for (uint32_t k = 0; k < nreg; ++k) s->bvh.order[k] = regular[s->bvh.order[k]];
  s->bvh.order.insert(s->bvh.order.end(), wild.begin(), wild.end());
  s->nwild = (uint32_t)wild.size();
  std::vector<float4>& tri_leaf = s->h_tri_leaf;
  std::vector<int32_t>& tri_face = s->h_tri_face;
  tri_leaf.resize((size_t)ntris * 4);
  tri_face.assign(std::max<uint32_t>(ntris, 1), 0);
  auto permute = [&](uint32_t k0, uint32_t k1) {
    for (uint32_t k = k0; k < k1; ++k) {
      uint32_t f = s->bvh.order[k];
      tri_face[k] = (int32_t)f;
      std::memcpy(&tri_leaf[4 * (size_t)k], &tri_rec[4 * (size_t)f], 64);
    }
  };
  if (threads == 1 || ntris < 65536) permute(0, ntris);
  else {
    const uint32_t step = (ntris + (uint32_t)threads - 1) / (uint32_t)threads;
    std::vector<std::thread> pool;
    for (int k = 0; k < threads; ++k) pool.emplace_back([&, k] { const uint32_t k0 = std::min(ntris, (uint32_t)k * step); permute(k0, std::min(ntris, k0 + step)); });
    for (std::thread& t : pool) t.join();
  }
  auto tD = std::chrono::steady_clock::now();
  if (std::getenv("RTX_BUILD_TIMING")) std::fprintf(stderr, "leaf order %.1f ms\n", std::chrono::duration<float, std::milli>(tD - tC).count());
  // Box(Tucano::Mesh) (src/box.h:41-85): max seeded with numeric_limits<float>::min()
  float mx[3] = {1.17549435e-38f, 1.17549435e-38f, 1.17549435e-38f};
  float mn[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f};
  for (uint32_t i = 0; i < nverts; ++i)
    for (int k = 0; k < 3; ++k) {
      float x = s->verts[3 * (size_t)i + k];
      if (x > mx[k]) mx[k] = x;
      if (x < mn[k]) mn[k] = x;
    }
  for (int k = 0; k < 3; ++k) { s->root_lo[k] = mn[k]; s->root_hi[k] = mx[k]; }
  auto t1 = std::chrono::steady_clock::now();
  rtx_scene_info& in = s->info;
  std::memset(&in, 0, sizeof in);
  in.ntris = ntris; in.nverts = nverts; in.nnodes = (uint32_t)s->bvh.quads.size(); in.nleaves = s->bvh.nleaves;
  in.max_depth = s->bvh.quad_depth; in.nwild = s->nwild; in.nslots = s->bvh.quad_slots;
  in.ntop_groups = s->bvh.top_groups[kTopLevels];
  in.build_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
  if (3 * s->bvh.quad_depth > (uint32_t)kStackDepth) return fail(RTX_ELIMIT, "BVH depth %u exceeds the traversal stack", s->bvh.quad_depth);
  // a node entry is a group index whose byte offset (index * 64) the kernels keep in 32 bits
  if (s->bvh.quads.size() / 4 >= (1u << 26)) return fail(RTX_ELIMIT, "BVH has %zu node groups, the device encoding holds fewer than 2^26", s->bvh.quads.size() / 4);
  s->built = true;
  return RTX_OK;
}

// ------------------------------------------------------------------ the hierarchy built / refitted on the GPU (bvh_device.h)
static bool want_device_build(const rtx_scene* s) {
  static const int env = [] { const char* e = std::getenv("RTX_BUILD"); return !e ? 0 : (std::strcmp(e, "device") == 0 ? 1 : (std::strcmp(e, "host") == 0 ? 2 : 0)); }();
  if (env) return env == 1;
  return (s->flags & RTX_FLAG_DEVICE_BUILD) != 0;
}
static double wild_below() { const char* e = std::getenv("RTX_WILD_BELOW"); const double v = e ? std::atof(e) : 0.0; return v > 0 ? v : 2.5e-6; }
static float pad_scale() { const char* e = std::getenv("RTX_TEST_PAD_SCALE"); return e ? (float)std::atof(e) : 1.0f; }
static int build_grid(size_t items) { return (int)std::max<size_t>(1, std::min<size_t>((items + 255) / 256, 4096)); }

static int upload_mesh(rtx_scene* s, hipStream_t st) {
  if (s->mesh_dev) return RTX_OK;
  if (int rc = upload(s->d_verts, s->verts, st)) return rc;
  if (int rc = upload(s->d_idx, s->idx, st)) return rc;
  if (int rc = upload(s->d_tri_mat, s->tri_mat, st)) return rc;
  s->mesh_dev = true;
  return RTX_OK;
}
static void release_build_ws(rtx_scene* s) {
  DevBuf* tmp[] = {&s->b_keys[0], &s->b_keys[1], &s->b_vals, &s->b_sort_tmp, &s->b_lr, &s->b_par_int, &s->b_par_leaf, &s->b_arrived,
                   &s->b_list[0], &s->b_list[1], &s->b_bin[0], &s->b_bin[1], &s->b_bin[2], &s->b_queue[0], &s->b_queue[1], &s->b_nodes, &s->b_quads};
  for (DevBuf* b : tmp) b->release();
}
// What the header's reductions say about the mesh as a whole: the root gate's box, the padding's validity range.
static void take_bounds(rtx_scene* s, const devbuild::Header& H) {
  for (int k = 0; k < 3; ++k) { s->root_lo[k] = devbuild::dec_f(H.root_lo[k]); s->root_hi[k] = devbuild::dec_f(H.root_hi[k]); }
  float scene_abs; std::memcpy(&scene_abs, &H.abs_bits, 4);
  s->domain = scene_abs > 0.0f ? 30.0f * scene_abs : INFINITY;
}
static int ensure_tri_buffers(rtx_scene* s, uint32_t ntris) {
  if (int rc = s->d_tris.ensure((size_t)std::max<uint32_t>(ntris, 1) * 64)) return rc;
  if (s->flags & RTX_FLAG_FAST_TRIANGLES) { if (int rc = s->d_tris_mt.ensure((size_t)std::max<uint32_t>(ntris, 1) * 48)) return rc; }
  else s->d_tris_mt.release();
  if (int rc = s->d_tri_face.ensure((size_t)std::max<uint32_t>(ntris, 1) * 4)) return rc;
  if (int rc = s->d_face_nm.ensure((size_t)std::max<uint32_t>(ntris, 1) * 16)) return rc;
  if (int rc = s->d_face_pos.ensure((size_t)std::max<uint32_t>(ntris, 1) * 4)) return rc;
  if (int rc = s->d_tri_box.ensure((size_t)std::max<uint32_t>(ntris, 1) * 32)) return rc;
  return RTX_OK;
}
// Returns RTX_OK with *fallback = true when the mesh is one the host builder has to take (too few regular triangles, a
// tree too deep for the traversal stack).  The scene's device is current.
static int device_build(rtx_scene* s, hipStream_t st, bool* fallback) {
  using namespace devbuild;
  *fallback = false;
  const uint32_t ntris = (uint32_t)s->tri_mat.size(), nverts = (uint32_t)(s->verts.size() / 3);
  if (ntris < 8) { *fallback = true; return RTX_OK; }
  auto t0 = std::chrono::steady_clock::now();
  if (int rc = upload_mesh(s, st)) return rc;
  if (int rc = s->d_hdr.ensure(sizeof(Header))) return rc;
  Header H; header_init(&H);
  HIP_TRY(hipMemcpyAsync(s->d_hdr.p, &H, sizeof H, hipMemcpyHostToDevice, st));
  Header* dH = s->d_hdr.as<Header>();
  const float ps = pad_scale(); const double wb = wild_below();
  // a leaf slot's triangle test priced against a group step in the collapse plan (measured: 0.3 / 0.6 / 1.0 / 2.0 give the 1M soup
  // 6.1 / 7.2 / 7.5 / 7.5 Grays/s, the bunny 17.5 / 17.3 / 18.0 / 17.7; the hall does not care)
  static const float tri_cost = [] { const char* e = std::getenv("RTX_LBVH_TRI_COST"); return e ? (float)std::atof(e) : 1.0f; }();
  static const int extended = [] { const char* e = std::getenv("RTX_LBVH_EXTENDED"); return e ? std::atoi(e) : 1; }();     // (A/B hook: 0 = plain Morton keys)
  if (int rc = s->b_keys[0].ensure((size_t)ntris * 8)) return rc;
  if (int rc = s->b_keys[1].ensure((size_t)ntris * 8)) return rc;
  if (int rc = s->b_vals.ensure((size_t)ntris * 4)) return rc;
  if (int rc = ensure_tri_buffers(s, ntris)) return rc;
  hipLaunchKernelGGL(k_vert_bounds, dim3(std::min(build_grid(nverts), 512)), dim3(256), 0, st, s->d_verts.as<float>(), nverts, dH);
  hipLaunchKernelGGL(k_tri_keys, dim3(build_grid(ntris)), dim3(256), 0, st, s->d_verts.as<float>(), s->d_idx.as<uint32_t>(), ntris, dH, ps, wb, extended,
                     s->b_keys[0].as<uint64_t>(), s->b_vals.as<uint32_t>());
  size_t tmp_bytes = 0;
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, s->b_keys[0].as<uint64_t>(), s->b_keys[1].as<uint64_t>(), s->b_vals.as<uint32_t>(),
                                             s->d_tri_face.as<uint32_t>(), (int)ntris, 0, 64, st));
  if (int rc = s->b_sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16))) return rc;
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(s->b_sort_tmp.p, tmp_bytes, s->b_keys[0].as<uint64_t>(), s->b_keys[1].as<uint64_t>(), s->b_vals.as<uint32_t>(),
                                             s->d_tri_face.as<uint32_t>(), (int)ntris, 0, 64, st));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(&H, s->d_hdr.p, sizeof H, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const uint32_t nwild = H.nwild, nreg = ntris - nwild;
  if (nreg < 8) { *fallback = true; return RTX_OK; }
  hipLaunchKernelGGL(k_tri_setup_leaf, dim3(build_grid(ntris)), dim3(256), 0, st, s->d_verts.as<float>(), s->d_idx.as<uint32_t>(), s->d_tri_mat.as<int>(),
                     s->d_tri_face.as<int>(), ntris, nreg, dH, ps, wb, s->d_tris.as<float4>(), s->d_tris_mt.as<float4>(), s->d_face_nm.as<float4>(),
                     s->d_face_pos.as<int>(), s->d_tri_box.as<float4>());
  // binary radix tree over the nreg sorted keys
  if (int rc = s->b_lr.ensure((size_t)nreg * 8)) return rc;
  if (int rc = s->b_par_int.ensure((size_t)nreg * 4)) return rc;
  if (int rc = s->b_par_leaf.ensure((size_t)nreg * 4)) return rc;
  if (int rc = s->b_arrived.ensure((size_t)nreg * 4)) return rc;
  for (DevBuf& b : s->b_list) if (int rc = b.ensure((size_t)nreg * 4)) return rc;
  for (DevBuf& b : s->b_bin) if (int rc = b.ensure((size_t)nreg * 16)) return rc;
  for (DevBuf& b : s->b_queue) if (int rc = b.ensure((size_t)nreg * 8)) return rc;
  if (int rc = s->b_nodes.ensure((size_t)nreg * 64)) return rc;
  if (int rc = s->b_quads.ensure((size_t)nreg * 128)) return rc;
  hipLaunchKernelGGL(k_radix_tree, dim3(build_grid(nreg)), dim3(256), 0, st, s->b_keys[1].as<uint64_t>(), (int)nreg, s->b_lr.as<uint2>(),
                     s->b_par_int.as<uint32_t>(), s->b_par_leaf.as<uint32_t>());
  HIP_TRY(hipMemsetAsync(s->b_arrived.p, 0, (size_t)nreg * 4, st));
  BinArrays B{s->b_bin[0].as<float4>(), s->b_bin[1].as<float4>(), s->b_bin[2].as<float4>(), s->b_lr.as<uint2>(), s->d_tri_box.as<float4>()};
  hipLaunchKernelGGL(k_bu_leaves, dim3(build_grid(nreg)), dim3(256), 0, st, nreg, s->b_par_leaf.as<uint32_t>(), s->b_arrived.as<uint32_t>(),
                     s->b_list[1].as<uint32_t>(), dH);
  for (int r = 1; r < kMaxRounds; ++r) {
    // (a round holds at most half the nodes of the one before in a balanced tree; the loop inside takes whatever there is)
    const size_t guess = std::max<size_t>((size_t)nreg >> std::min(r, 12), 16384);
    hipLaunchKernelGGL(k_bu_round, dim3(build_grid(guess)), dim3(256), 0, st, r, s->b_list[r & 1].as<uint32_t>(), s->b_list[(r + 1) & 1].as<uint32_t>(), B,
                       s->b_par_int.as<uint32_t>(), s->b_arrived.as<uint32_t>(), dH, tri_cost);
  }
  // groups, top-down; the expected-cost plan first, two binary levels per group if that tree is too deep for the stack
  // (test hooks, read at every build: RTX_TEST_DEVICE_LEVELS = fewer group levels allowed than the traversal stack holds, so that small
  //  meshes reach the fallbacks; RTX_TEST_DEVICE_COLLAPSE=fixed = start with the second rule)
  const int max_levels = [] { const char* e = std::getenv("RTX_TEST_DEVICE_LEVELS"); const int v = e ? std::atoi(e) : 0; return v > 0 && v < kMaxLevels ? v : kMaxLevels; }();
  const int first_rule = [] { const char* e = std::getenv("RTX_TEST_DEVICE_COLLAPSE"); return e && std::strcmp(e, "fixed") == 0 ? 1 : 0; }();
  for (int fixed = first_rule; fixed < 2; ++fixed) {
    if (fixed) {
      HIP_TRY(hipMemsetAsync((char*)s->d_hdr.p + offsetof(Header, ngroups), 0, offsetof(Header, round_count) - offsetof(Header, ngroups), st));
    }
    hipLaunchKernelGGL(k_emit_seed, dim3(1), dim3(64), 0, st, s->b_queue[0].as<uint2>(), dH);
    for (int level = 0; level < max_levels; ++level) {
      const size_t guess = level < 8 ? (size_t)1 << (2 * level + 2) : nreg;
      hipLaunchKernelGGL(k_emit_level, dim3(build_grid(std::min<size_t>(guess, nreg))), dim3(256), 0, st, level, s->b_queue[level & 1].as<uint2>(),
                         s->b_queue[(level + 1) & 1].as<uint2>(), dH, B, tri_cost, fixed, s->b_quads.as<BvhNode>(), s->b_nodes.as<uint4>());
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&H, s->d_hdr.p, sizeof H, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    uint64_t done = 0;
    for (int r = 1; r <= kMaxRounds; ++r) done += H.round_count[r];
    if (done != (uint64_t)nreg - 1) { *fallback = true; return RTX_OK; }     // a binary tree taller than the rounds provided for
    if (H.level_count[max_levels] == 0) break;
    if (fixed) { *fallback = true; return RTX_OK; }
  }
  const uint32_t ngroups = H.ngroups;
  if (ngroups >= (1u << 26)) return fail(RTX_ELIMIT, "BVH has %u node groups, the device encoding holds fewer than 2^26", ngroups);
  // exact-size homes for what stays (the build's buffers are sized for the worst case, one group per triangle)
  if (int rc = s->d_nodes.ensure((size_t)ngroups * 64)) return rc;
  if (int rc = s->d_quads.ensure((size_t)ngroups * 128)) return rc;
  HIP_TRY(hipMemcpyAsync(s->d_nodes.p, s->b_nodes.p, (size_t)ngroups * 64, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(s->d_quads.p, s->b_quads.p, (size_t)ngroups * 128, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  take_bounds(s, H);
  s->nwild = nwild;
  s->level_start.assign(1, 0);
  uint32_t depth = 0;
  for (int l = 0; l < max_levels && H.level_count[l]; ++l) { s->level_start.push_back(s->level_start.back() + H.level_count[l]); ++depth; }
  s->level_identity = true; s->refit_ready = true;
  s->bvh = BvhBuildResult();
  s->bvh.quad_depth = depth; s->bvh.quad_slots = H.nslots; s->bvh.nleaves = H.nleaves;
  for (int l = 1; l <= kTopLevels; ++l) s->bvh.top_groups[l] = s->level_start[std::min<size_t>(l, s->level_start.size() - 1)];
  auto t1 = std::chrono::steady_clock::now();
  s->device_build_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
  rtx_scene_info& in = s->info;
  const rtx_scene_info was = in;             // (a rebuild of a committed scene: spheres, lights, materials and the device footprint stay what they were)
  std::memset(&in, 0, sizeof in);
  in.nspheres = was.nspheres; in.nlights = was.nlights; in.nmaterials = was.nmaterials; in.device_bytes = was.device_bytes;
  in.ntris = ntris; in.nverts = nverts; in.nnodes = ngroups * 4; in.nleaves = H.nleaves; in.max_depth = depth; in.nwild = nwild;
  in.nslots = H.nslots; in.ntop_groups = s->bvh.top_groups[kTopLevels]; in.build_ms = s->device_build_ms;
  in.build_kind = 1; in.device_build_ms = s->device_build_ms;
  s->built = true; s->geom_dev = true; s->device_built = true; s->host_stale = true;
  if (!s->keep_build_ws) release_build_ws(s);
  return RTX_OK;
}

// Host-built trees: what a refit needs beside the packed groups -- the fp32 slots and the groups of every level.
static int refit_prepare(rtx_scene* s, hipStream_t st) {
  if (s->refit_ready) return RTX_OK;
  const uint32_t ntris = (uint32_t)s->tri_mat.size();
  if (int rc = upload_mesh(s, st)) return rc;
  if (int rc = upload(s->d_quads, s->bvh.quads, st)) return rc;
  if (int rc = s->d_tri_box.ensure((size_t)std::max<uint32_t>(ntris, 1) * 32)) return rc;
  if (int rc = s->d_hdr.ensure(sizeof(devbuild::Header))) return rc;
  const uint32_t ngroups = (uint32_t)(s->bvh.quads.size() / 4);
  std::vector<uint32_t> order; order.reserve(ngroups);
  s->level_start.assign(1, 0);
  std::vector<uint32_t> cur{0}, next;
  while (!cur.empty()) {
    next.clear();
    for (uint32_t g : cur) {
      order.push_back(g);
      for (int k = 0; k < 4; ++k) {
        const BvhNode& n = s->bvh.quads[4 * (size_t)g + k];
        if (n.count == 0 && n.leftFirst != kEmptySlot) next.push_back(n.leftFirst);
      }
    }
    s->level_start.push_back((uint32_t)order.size());
    cur.swap(next);
  }
  if (int rc = upload(s->d_level_groups, order, st)) return rc;
  HIP_TRY(hipStreamSynchronize(st));
  s->level_identity = false; s->refit_ready = true;
  return RTX_OK;
}
// New vertices are in d_verts: records, boxes, groups.  *rebuild = true when the topology cannot hold the new mesh.
static int device_refit(rtx_scene* s, hipStream_t st, bool* rebuild) {
  using namespace devbuild;
  *rebuild = false;
  const uint32_t ntris = (uint32_t)s->tri_mat.size(), nverts = (uint32_t)(s->verts.size() / 3);
  if (int rc = refit_prepare(s, st)) return rc;
  if (int rc = ensure_tri_buffers(s, ntris)) return rc;
  Header H; header_init(&H);
  HIP_TRY(hipMemcpyAsync(s->d_hdr.p, &H, sizeof H, hipMemcpyHostToDevice, st));
  Header* dH = s->d_hdr.as<Header>();
  hipLaunchKernelGGL(k_vert_bounds, dim3(std::min(build_grid(nverts), 512)), dim3(256), 0, st, s->d_verts.as<float>(), nverts, dH);
  hipLaunchKernelGGL(k_tri_setup_leaf, dim3(build_grid(ntris)), dim3(256), 0, st, s->d_verts.as<float>(), s->d_idx.as<uint32_t>(), s->d_tri_mat.as<int>(),
                     s->d_tri_face.as<int>(), ntris, ntris - s->nwild, dH, pad_scale(), wild_below(), s->d_tris.as<float4>(), s->d_tris_mt.as<float4>(),
                     s->d_face_nm.as<float4>(), s->d_face_pos.as<int>(), s->d_tri_box.as<float4>());
  if (ntris - s->nwild > 0)
    for (size_t l = s->level_start.size() - 1; l-- > 0;) {
      const uint32_t base = s->level_start[l], count = s->level_start[l + 1] - base;
      if (!count) continue;
      hipLaunchKernelGGL(k_refit_level, dim3(build_grid(count)), dim3(256), 0, st, s->level_identity ? nullptr : s->d_level_groups.as<uint32_t>(), base,
                         count, s->d_quads.as<BvhNode>(), s->d_tri_box.as<float4>(), s->d_nodes.as<uint4>());
    }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(&H, s->d_hdr.p, sizeof H, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (H.wild_changed) { *rebuild = true; return RTX_OK; }
  take_bounds(s, H);
  s->host_stale = true;
  return RTX_OK;
}

// Host copies of what the GPU built or refitted, for the exports (rtx_scene_get_*), rtx_scene_clone and a commit on another device.
static int fetch_host_mirror(rtx_scene* s) {
  if (!s->host_stale) return RTX_OK;
  if (s->device < 0 || !s->geom_dev) return fail(RTX_ESTATE, "the scene's device state is gone");
  HIP_TRY(hipSetDevice(s->device));
  const uint32_t ntris = (uint32_t)s->tri_mat.size();
  const uint32_t ngroups = s->info.nnodes / 4;
  for (rtx_scene::Slot& w : s->slots) if (w.stream) HIP_TRY(hipStreamSynchronize(w.stream));      // (non-blocking streams: a plain copy does not wait for them)
  if (s->verts_stale) { HIP_TRY(hipMemcpy(s->verts.data(), s->d_verts.p, s->verts.size() * 4, hipMemcpyDeviceToHost)); s->verts_stale = false; }
  s->bvh.quads.resize((size_t)ngroups * 4);
  HIP_TRY(hipMemcpy(s->bvh.quads.data(), s->d_quads.p, (size_t)ngroups * 128, hipMemcpyDeviceToHost));
  s->h_tri_face.resize(std::max<uint32_t>(ntris, 1));
  HIP_TRY(hipMemcpy(s->h_tri_face.data(), s->d_tri_face.p, (size_t)ntris * 4, hipMemcpyDeviceToHost));
  s->bvh.order.assign(s->h_tri_face.begin(), s->h_tri_face.begin() + ntris);
  s->h_face_nm.resize(std::max<uint32_t>(ntris, 1));
  HIP_TRY(hipMemcpy(s->h_face_nm.data(), s->d_face_nm.p, (size_t)ntris * 16, hipMemcpyDeviceToHost));
  std::vector<float4> dt((size_t)ntris * 4);
  HIP_TRY(hipMemcpy(dt.data(), s->d_tris.p, (size_t)ntris * 64, hipMemcpyDeviceToHost));
  s->h_tri_leaf.resize((size_t)ntris * 4);
  for (size_t k = 0; k + 3 < dt.size(); k += 4) {       // device order {n, D} {v0, vv} {u, uu} {v, uv} -> export order {v0, D} {u, uu} {v, uv} {n, vv}
    const float4 n = dt[k], a = dt[k + 1];
    s->h_tri_leaf[k] = make_float4(a.x, a.y, a.z, n.w);
    s->h_tri_leaf[k + 1] = dt[k + 2];
    s->h_tri_leaf[k + 2] = dt[k + 3];
    s->h_tri_leaf[k + 3] = make_float4(n.x, n.y, n.z, a.w);
  }
  s->host_stale = false;
  return RTX_OK;
}

// What the traversal and shading kernels read (a refit's fp32 slots, boxes and mesh copy -- d_quads, d_tri_box, d_verts ... -- come on top).
static uint64_t device_footprint(const rtx_scene* s) {
  return s->d_nodes.cap + s->d_tris.cap + s->d_tris_mt.cap + s->d_tri_face.cap + s->d_face_nm.cap + s->d_spheres.cap + s->d_sph_mat.cap +
         s->d_mats.cap + s->d_lights.cap + s->d_slights.cap + s->d_leafboxes.cap + s->d_leaf_faces.cap + s->d_face_pos.cap;
}

extern "C" int rtx_scene_commit(rtx_scene* s, int device) {
  if (!s) return fail(RTX_EINVAL, "null scene");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(RTX_EINVAL, "device %d out of range (%d devices)", device, ndev);
  if (s->device >= 0) {                       // calls still in flight read the buffers this is about to replace
    (void)hipSetDevice(s->device);
    for (rtx_scene::Call& c : s->calls) if (int rc = finish_call(s, c)) return rc;
  }
  // what the GPU built or refitted comes to the host before its device copy may go (another device takes it from there)
  if (s->host_stale && s->device >= 0 && s->device != device) if (int rc = fetch_host_mirror(s)) return rc;
  if (s->device >= 0 && s->device != device) release_device(s);
  HIP_TRY(hipSetDevice(device));
  s->device = device;
  // (a process has 4 hardware queues by default and streams take them in turn: the slots' main streams first, so
  // that calls in flight never share one)
  for (rtx_scene::Slot& w : s->slots) if (!w.stream) HIP_TRY(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
  for (rtx_scene::Slot& w : s->slots) if (!w.stream2) HIP_TRY(hipStreamCreateWithFlags(&w.stream2, hipStreamNonBlocking));
  if (int rc = scratch_init(device)) return rc;
  for (rtx_scene::Slot& w : s->slots) if (int rc = w.w_guard.ensure(64)) return rc;
  for (rtx_scene::Call& c : s->calls) if (!c.hg) HIP_TRY(hipHostMalloc((void**)&c.hg, 64, hipHostMallocDefault));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  s->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // (the 48-byte records of RTX_FLAG_FAST_TRIANGLES come or go: the geometry is uploaded again, from the host copies)
  if (s->geom_dev && ((s->flags & RTX_FLAG_FAST_TRIANGLES) != 0) != (s->d_tris_mt.p != nullptr)) {
    if (int rc = fetch_host_mirror(s)) return rc;
    HIP_TRY(hipSetDevice(device));
    s->geom_dev = false;
  }
  if (!s->built && want_device_build(s)) {
    bool fallback = false;
    if (int rc = device_build(s, ws(s).stream, &fallback)) return rc;
    if (fallback) { s->geom_dev = false; s->device_built = false; release_build_ws(s); }
  }
  if (int rc = rtx_scene_build(s)) return rc;
  for (rtx_scene::Slot& w : s->slots) {
    if (int rc = w.w_stack_ovf.ensure((size_t)trace_grid(s) * kBlock * (kStackDepth - kLdsStack) * 4)) return rc;
    if (int rc = w.w_stack_ovf2.ensure((size_t)trace_grid(s) * kBlock * (kStackDepth - kLdsStack) * 4)) return rc;
  }
  const uint32_t ntris = (uint32_t)s->tri_mat.size();
  const std::vector<float4>& tri_leaf = s->h_tri_leaf;
  const std::vector<int32_t>& tri_face = s->h_tri_face;
  const std::vector<float4>& face_nm = s->h_face_nm;
  // ---- upload
  std::vector<DevMat> dm(std::max<size_t>(s->mats.size(), 1));
  for (size_t i = 0; i < s->mats.size(); ++i) {
    const rtx_material& m = s->mats[i];
    DevMat& d = dm[i];
    std::memset(&d, 0, sizeof d);
    for (int k = 0; k < 3; ++k) { d.ka[k] = m.ka[k]; d.kd[k] = m.kd[k]; d.ks[k] = m.ks[k]; }
    d.ns = m.ns; d.ni = m.ni; d.d = m.d; d.illum = m.illum;
  }
  std::vector<float4> dl(std::max<size_t>(s->lights.size() / 3, 1));
  for (size_t i = 0; i < s->lights.size() / 3; ++i) dl[i] = make_float4(s->lights[3 * i], s->lights[3 * i + 1], s->lights[3 * i + 2], 0.0f);
  std::vector<float4> ds(std::max<size_t>(s->spheres.size() / 4, 1));
  for (size_t i = 0; i < s->spheres.size() / 4; ++i) ds[i] = make_float4(s->spheres[4 * i], s->spheres[4 * i + 1], s->spheres[4 * i + 2], s->spheres[4 * i + 3]);
  std::vector<int32_t> dsm = s->sph_mat; if (dsm.empty()) dsm.push_back(-1);
  // Device encoding of the groups: 64-byte PackedGroup (bvh_build.h); the exported layout of
  // rtx_scene_get_bvh stays the fp32 one {lo, ref, hi, count}.
  const bool up_geom = !s->geom_dev;         // (a hierarchy the GPU built or refitted for this mesh is already where it belongs)
  if (up_geom) {
  std::vector<PackedGroup> dnodes;
  pack_groups(s->bvh.quads, &dnodes);
  if (int rc = upload(s->d_nodes, dnodes, ws(s).stream)) return rc;
  // Device order of the triangle record: {n, D} {v0, vv} {u, uu} {v, uv}, so that the plane
  // stage of the test (most triangles stop there) needs one 16-byte load instead of two; the
  // exported layout of rtx_scene_get_tris stays {v0, D} {u, uu} {v, uv} {n, vv}.
  {
    std::vector<float4> dtris(tri_leaf.size());
    for (size_t k = 0; k + 3 < tri_leaf.size(); k += 4) {
      const float4 a = tri_leaf[k], u = tri_leaf[k + 1], v = tri_leaf[k + 2], n = tri_leaf[k + 3];
      dtris[k] = make_float4(n.x, n.y, n.z, a.w);
      dtris[k + 1] = make_float4(a.x, a.y, a.z, n.w);
      dtris[k + 2] = u;
      dtris[k + 3] = v;
    }
    if (int rc = upload(s->d_tris, dtris, ws(s).stream)) return rc;
  }
  // RTX_FLAG_FAST_TRIANGLES: the 48-byte Moeller-Trumbore records {v0, e1.x} {e1.yz, e2.xy} {e2.z, face id, 0, 0}, leaf order
  // (beside the reference records, which the brute-force guard / verification, the GI probes and the debug ray keep using)
  if (s->flags & RTX_FLAG_FAST_TRIANGLES) {
    std::vector<float4> mt((size_t)ntris * 3);
    for (uint32_t k = 0; k < ntris; ++k) {
      const float4 a = tri_leaf[4 * (size_t)k], u = tri_leaf[4 * (size_t)k + 1], v = tri_leaf[4 * (size_t)k + 2];
      float fw; const int32_t f = tri_face[k]; std::memcpy(&fw, &f, 4);
      mt[3 * (size_t)k] = make_float4(a.x, a.y, a.z, u.x);
      mt[3 * (size_t)k + 1] = make_float4(u.y, u.z, v.x, v.y);
      mt[3 * (size_t)k + 2] = make_float4(v.z, fw, 0.0f, 0.0f);
    }
    if (int rc = upload(s->d_tris_mt, mt, ws(s).stream)) return rc;
  } else {
    s->d_tris_mt.release();
  }
  if (int rc = upload(s->d_tri_face, tri_face, ws(s).stream)) return rc;
  if (int rc = upload(s->d_face_nm, face_nm, ws(s).stream)) return rc;
  s->refit_ready = false;                    // d_quads / the level lists belong to whatever tree was here before
  }
  if (int rc = upload(s->d_spheres, ds, ws(s).stream)) return rc;
  if (int rc = upload(s->d_sph_mat, dsm, ws(s).stream)) return rc;
  if (int rc = upload(s->d_mats, dm, ws(s).stream)) return rc;
  if (int rc = upload(s->d_lights, dl, ws(s).stream)) return rc;
  std::vector<float4> dsl(std::max<size_t>(s->slights.size() / 4, 1));
  for (size_t i = 0; i < s->slights.size() / 4; ++i) dsl[i] = make_float4(s->slights[4 * i], s->slights[4 * i + 1], s->slights[4 * i + 2], s->slights[4 * i + 3]);
  const size_t nleaf = s->have_leaves ? s->leaf_start.size() : 0;
  std::vector<float4> dlb(std::max<size_t>(nleaf * 2, 1));
  for (size_t i = 0; i < nleaf; ++i) {
    float sw, cw; std::memcpy(&sw, &s->leaf_start[i], 4); std::memcpy(&cw, &s->leaf_count[i], 4);
    dlb[2 * i] = make_float4(s->leaf_lo[3 * i], s->leaf_lo[3 * i + 1], s->leaf_lo[3 * i + 2], sw);
    dlb[2 * i + 1] = make_float4(s->leaf_hi[3 * i], s->leaf_hi[3 * i + 1], s->leaf_hi[3 * i + 2], cw);
  }
  std::vector<int32_t> dlf = s->leaf_faces; if (dlf.empty()) dlf.push_back(0);
  if (int rc = upload(s->d_slights, dsl, ws(s).stream)) return rc;
  if (int rc = upload(s->d_leafboxes, dlb, ws(s).stream)) return rc;
  if (int rc = upload(s->d_leaf_faces, dlf, ws(s).stream)) return rc;
  if (up_geom) {
    std::vector<int32_t> fpos(std::max<uint32_t>(ntris, 1), 0);
    for (uint32_t k = 0; k < ntris; ++k) fpos[(size_t)tri_face[k]] = (int32_t)k;
    if (int rc = upload(s->d_face_pos, fpos, ws(s).stream)) return rc;
    HIP_TRY(hipStreamSynchronize(ws(s).stream));
    s->geom_dev = true;
  }
  HIP_TRY(hipStreamSynchronize(ws(s).stream));

  DevScene& d = s->dev;
  d.nodes = s->d_nodes.as<uint4>(); d.tris = s->d_tris.as<float4>(); d.tris_mt = s->d_tris_mt.as<float4>(); d.tri_face = s->d_tri_face.as<int>();
  d.face_nm = s->d_face_nm.as<float4>(); d.spheres = s->d_spheres.as<float4>(); d.sph_mat = s->d_sph_mat.as<int>();
  d.mats = s->d_mats.as<DevMat>(); d.lights = s->d_lights.as<float4>();
  d.slights = s->d_slights.as<float4>(); d.leafboxes = s->d_leafboxes.as<float4>();
  d.leaf_faces = s->d_leaf_faces.as<int>(); d.face_pos = s->d_face_pos.as<int>();
  d.nsl = (uint32_t)(s->slights.size() / 4); d.nleafboxes = (uint32_t)nleaf;
  d.sl_sin = sinf((float)(M_PI * 2 / 4)); d.sl_cos = cosf((float)(M_PI * 2 / 4));      // Shader::getNPointsOnCircle with n = 4
  d.flags = s->flags & RTX_FLAG_NO_ROOT_GATE; d.ntop = s->bvh.top_groups[kTopLevels];
  // Shader's shadowed branch (src/flyscene.cpp:731-743) colours a blocked light with the
  // *blocker's* Ka, Kd, Ks, Ns, d.  When every face and sphere has an in-range material and all
  // those records agree in these fields, any blocker gives the same colour and shadow rays may
  // stop at the first hit before the light.
  {
    bool uniform = !s->mats.empty();
    for (size_t i = 1; uniform && i < s->mats.size(); ++i) {
      const rtx_material &a = s->mats[0], &b = s->mats[i];
      uniform = std::memcmp(a.ka, b.ka, 12) == 0 && std::memcmp(a.kd, b.kd, 12) == 0 && std::memcmp(a.ks, b.ks, 12) == 0 &&
                std::memcmp(&a.ns, &b.ns, 4) == 0 && std::memcmp(&a.d, &b.d, 4) == 0;
    }
    const int nm = (int)s->mats.size();
    for (size_t i = 0; uniform && i < s->tri_mat.size(); ++i) uniform = s->tri_mat[i] >= 0 && s->tri_mat[i] < nm;
    for (size_t i = 0; uniform && i < s->sph_mat.size(); ++i) uniform = s->sph_mat[i] >= 0 && s->sph_mat[i] < nm;
    if (uniform) d.flags |= 2u;
  }
  // skip_zero_weight asks, per hit, whether the material's reflection weight (1 - d) or refraction weight
  // (d) is exactly zero.  Where every dissolve value in use answers alike the kernels need not look the
  // material up: bit 2 no material reflects, bit 3 every one does, bit 4 none refracts, bit 5 every one does.
  {
    const int nm = (int)s->mats.size();
    bool use_default = false;                      // some face or sphere falls back to Mtl(): d = 1
    for (size_t i = 0; !use_default && i < s->tri_mat.size(); ++i) use_default = s->tri_mat[i] < 0 || s->tri_mat[i] >= nm;
    for (size_t i = 0; !use_default && i < s->sph_mat.size(); ++i) use_default = s->sph_mat[i] < 0 || s->sph_mat[i] >= nm;
    int n_all = 0, n_norefl = 0, n_norefr = 0;
    auto tally = [&](float dd) { ++n_all; n_norefl += (1.0f - dd == 0.0f); n_norefr += (dd == 0.0f); };
    for (const rtx_material& m : s->mats) tally(m.d);          // (records no face uses are counted too: conservative)
    if (use_default) tally(1.0f);
    if (n_all) {
      if (n_norefl == n_all) d.flags |= 4u; else if (n_norefl == 0) d.flags |= 8u;
      if (n_norefr == n_all) d.flags |= 16u; else if (n_norefr == 0) d.flags |= 32u;
    }
  }
  d.ntris = ntris; d.nwild = s->nwild; d.nspheres = (uint32_t)(s->spheres.size() / 4); d.nlights = (uint32_t)(s->lights.size() / 3);
  d.nmats = (uint32_t)s->mats.size();
  for (int k = 0; k < 3; ++k) { d.root_lo[k] = s->root_lo[k]; d.root_hi[k] = s->root_hi[k]; }
  d.root_entry = 0;     // group 0 holds the children of the root
  rtx_scene_info& in = s->info;
  in.nspheres = d.nspheres; in.nlights = d.nlights; in.nmaterials = d.nmats;
  in.device_bytes = device_footprint(s);
  for (auto& per_kind : s->tune) for (rtx_scene::Tune& T : per_kind) T = rtx_scene::Tune();     // another scene on this handle: the tuner starts over
  s->committed = true;
  return RTX_OK;
}

// ------------------------------------------------------------------ animated meshes
static int update_vertices(rtx_scene* s, const float* verts, uint32_t nverts, int32_t how, bool on_device) {
  if (!s || !verts) return fail(RTX_EINVAL, "null argument");
  if (!s->committed || s->device < 0) return fail(RTX_ESTATE, "scene not committed");
  if ((size_t)nverts * 3 != s->verts.size()) return fail(RTX_EINVAL, "the scene has %zu vertices, got %u", s->verts.size() / 3, nverts);
  if (how != RTX_UPDATE_REFIT && how != RTX_UPDATE_REBUILD) return fail(RTX_EINVAL, "how must be RTX_UPDATE_REFIT or RTX_UPDATE_REBUILD");
  HIP_TRY(hipSetDevice(s->device));
  for (rtx_scene::Call& c : s->calls) if (int rc = finish_call(s, c)) return rc;
  auto t0 = std::chrono::steady_clock::now();
  hipStream_t st = ws(s).stream;
  // (an error on the way leaves device buffers half written: the scene then counts as not committed, and says so, until a commit succeeds)
  s->committed = false;
  // a refit needs the old tree's fp32 slots: take them (host-built trees) before anything of the old state goes
  if (how == RTX_UPDATE_REFIT) if (int rc = refit_prepare(s, st)) return rc;
  if (int rc = upload_mesh(s, st)) return rc;
  if (on_device) {
    HIP_TRY(hipMemcpyAsync(s->d_verts.p, verts, (size_t)nverts * 12, hipMemcpyDeviceToDevice, st));
    s->verts_stale = true;
  } else {
    std::memcpy(s->verts.data(), verts, (size_t)nverts * 12);
    HIP_TRY(hipMemcpyAsync(s->d_verts.p, s->verts.data(), (size_t)nverts * 12, hipMemcpyHostToDevice, st));
    s->verts_stale = false;
  }
  // Flyscene's leaf list `boxes` is made from the vertices (divideBox_KD): it does not describe the new mesh
  s->have_leaves = false; s->leaf_lo.clear(); s->leaf_hi.clear(); s->leaf_start.clear(); s->leaf_count.clear(); s->leaf_faces.clear();
  s->dev.nleafboxes = 0;
  bool rebuild = how == RTX_UPDATE_REBUILD;
  if (!rebuild) {
    if (int rc = device_refit(s, st, &rebuild)) return rc;
    if (!rebuild) ++s->info.refits;
  }
  if (rebuild) {
    bool fallback = !want_device_build(s);
    if (!fallback) {
      s->keep_build_ws = true;
      if (int rc = device_build(s, st, &fallback)) return rc;
    }
    if (fallback) {                  // the host's builder, from the host's copy of the vertices, and everything uploaded again
      HIP_TRY(hipStreamSynchronize(st));      // (the copy into d_verts runs on a non-blocking stream)
      if (s->verts_stale) { HIP_TRY(hipMemcpy(s->verts.data(), s->d_verts.p, s->verts.size() * 4, hipMemcpyDeviceToHost)); s->verts_stale = false; }
      s->built = false; s->geom_dev = false; s->device_built = false; s->host_stale = false; s->refit_ready = false;
      if (int rc = rtx_scene_commit(s, s->device)) return rc;
    }
  }
  DevScene& d = s->dev;
  d.nodes = s->d_nodes.as<uint4>(); d.tris = s->d_tris.as<float4>(); d.tris_mt = s->d_tris_mt.as<float4>(); d.tri_face = s->d_tri_face.as<int>();
  d.face_nm = s->d_face_nm.as<float4>(); d.face_pos = s->d_face_pos.as<int>();
  d.nwild = s->nwild; d.ntop = s->bvh.top_groups[kTopLevels];
  for (int k = 0; k < 3; ++k) { d.root_lo[k] = s->root_lo[k]; d.root_hi[k] = s->root_hi[k]; }
  auto t1 = std::chrono::steady_clock::now();
  const float ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
  if (rebuild) s->info.refits = 0; else s->refit_ms = ms;
  s->info.device_bytes = device_footprint(s);
  s->committed = true;
  return RTX_OK;
}
extern "C" int rtx_scene_update_vertices(rtx_scene* s, const float* verts, uint32_t nverts, int32_t how) { return update_vertices(s, verts, nverts, how, false); }
extern "C" int rtx_scene_update_vertices_dev(rtx_scene* s, const float* verts, uint32_t nverts, int32_t how) { return update_vertices(s, verts, nverts, how, true); }

extern "C" int rtx_scene_get_info(const rtx_scene* s, rtx_scene_info* out) {
  if (!s || !out) return fail(RTX_EINVAL, "null argument");
  if (!s->built) return fail(RTX_ESTATE, "scene not built (rtx_scene_build / rtx_scene_commit)");
  *out = s->info;
  out->device_build_ms = s->device_build_ms; out->refit_ms = s->refit_ms;
  return RTX_OK;
}

extern "C" int rtx_scene_get_tris(const rtx_scene* s, float* tri16) {
  if (!s || !tri16) return fail(RTX_EINVAL, "null argument");
  if (!s->built) return fail(RTX_ESTATE, "scene not built (rtx_scene_build / rtx_scene_commit)");
  if (int rc = fetch_host_mirror(const_cast<rtx_scene*>(s))) return rc;      // (a hierarchy the GPU built or refitted: copied back on demand)
  std::memcpy(tri16, s->h_tri_leaf.data(), s->h_tri_leaf.size() * sizeof(float4));
  return RTX_OK;
}

extern "C" int rtx_scene_get_bvh(const rtx_scene* s, uint32_t* nodes8, uint32_t* tri_face) {
  if (!s) return fail(RTX_EINVAL, "null scene");
  if (!s->built) return fail(RTX_ESTATE, "scene not built (rtx_scene_build / rtx_scene_commit)");
  if (int rc = fetch_host_mirror(const_cast<rtx_scene*>(s))) return rc;      // (a hierarchy the GPU built or refitted: copied back on demand)
  if (nodes8) std::memcpy(nodes8, s->bvh.quads.data(), s->bvh.quads.size() * sizeof(BvhNode));
  if (tri_face) std::memcpy(tri_face, s->bvh.order.data(), s->bvh.order.size() * 4);
  return RTX_OK;
}

extern "C" int rtx_scene_get_packed_bvh(const rtx_scene* s, uint32_t* groups16) {
  if (!s) return fail(RTX_EINVAL, "null scene");
  if (!s->built) return fail(RTX_ESTATE, "scene not built (rtx_scene_build / rtx_scene_commit)");
  if (int rc = fetch_host_mirror(const_cast<rtx_scene*>(s))) return rc;      // (a hierarchy the GPU built or refitted: copied back on demand)
  if (!groups16) return fail(RTX_EINVAL, "null argument");
  std::vector<PackedGroup> g;
  pack_groups(s->bvh.quads, &g);
  std::memcpy(groups16, g.data(), g.size() * sizeof(PackedGroup));
  return RTX_OK;
}

// ------------------------------------------------------------------ launch helpers
static int grid_for(const rtx_scene* s, size_t items) {
  size_t blocks = (items + kBlock - 1) / kBlock;
  size_t cap = (size_t)s->ncu * 8;
  return (int)std::max<size_t>(1, std::min(blocks, cap));
}
static void span_begin(rtx_scene* s, bool trace, hipStream_t on = nullptr) {
  rtx_scene::Call& c = cur_call(s);
  hipEvent_t a = c.events.get(), b = c.events.get();
  (void)hipEventRecord(a, on ? on : ws(s).stream);
  (trace ? c.trace_spans : c.all_spans).push_back({a, b});
  if (trace) c.span_kind.push_back(-1);
}
static void span_end(rtx_scene* s, bool trace, hipStream_t on = nullptr) {
  rtx_scene::Call& c = cur_call(s);
  (void)hipEventRecord((trace ? c.trace_spans : c.all_spans).back().second, on ? on : ws(s).stream);
}
static void tune_update(rtx_scene* s, rtx_scene::Call& c);
// Waits for an enqueued call, turns its events and counters into rtx_stats, feeds the tuner.
static int finish_call(rtx_scene* s, rtx_scene::Call& c) {
  if (!c.pending) return RTX_OK;
  c.pending = false;
  HIP_TRY(hipEventSynchronize(c.all_spans[0].second));
  HIP_TRY(hipGetLastError());
  float ms = 0;
  for (size_t i = 0; i < c.trace_spans.size(); ++i) {
    float m = 0;
    (void)hipEventElapsedTime(&m, c.trace_spans[i].first, c.trace_spans[i].second);
    ms += m;
    if (i == c.camera_span) c.stats.ms_trace_camera = m;
  }
  c.stats.ms_trace = ms;
  c.stats.launches_trace = (uint32_t)c.trace_spans.size();
  float tot = 0;
  (void)hipEventElapsedTime(&tot, c.all_spans[0].first, c.all_spans[0].second);
  c.stats.ms_total = tot;
  if (c.pipeline && c.shade) {
    const uint32_t* hc = c.hc;
    for (int g = 0; g <= c.max_gen; ++g) {
      if (c.shadows) c.stats.rays_shadow += (uint64_t)hc[4 * g] * c.NLS;
      c.stats.rays_gi += (uint64_t)hc[4 * g] * c.G;
      if (g < c.max_gen) { c.stats.rays_refraction += hc[4 * g + 2]; c.stats.rays_reflection += hc[4 * g + 1] - hc[4 * g + 2]; }
    }
  }
  if (c.guard) { c.stats.rays_out_of_domain = c.hg[0]; c.stats.verify_checked = c.hg[1]; c.stats.verify_mismatches = c.hg[2]; }
  if (c.tune) tune_update(s, c);
  s->stats = c.stats;
  return RTX_OK;
}
// Every call starts here: takes the call slot that is not the most recent one (finishing what it
// still holds) and opens the span of the whole call.
static bool bucket_settled(const rtx_scene* s, int bucket);
static int size_bucket(uint64_t n0) {
  int lg = 0;
  while (lg < 62 && (2ull << lg) <= n0) ++lg;          // floor(log2(n0)), 0 for n0 <= 1
  return std::min(lg / 2, rtx_scene::kBuckets - 1);
}
static int stats_begin(rtx_scene* s, uint64_t n0 = 0) {
  // The slot used longest ago -- unless nothing is in flight or waiting to be handed out: a caller that never enqueues
  // stays in one slot (one workspace grows instead of four, and the camera tables cached there are found again).
  bool busy = false;
  for (const rtx_scene::Call& o : s->calls) busy = busy || o.pending || o.unreported;
  if (busy) s->cur = (s->cur + 1) % kSlots;
  rtx_scene::Call& c = cur_call(s);
  if (int rc = finish_call(s, c)) return rc;
  c.unreported = false;
  // The other slots' calls may still be running on their own streams: this call then runs beside them -- unless the
  // scheduler tuner may time this call's launches or those of a call in flight: such calls have the GPU to themselves.
  c.bucket = size_bucket(n0);
  c.alone = !bucket_settled(s, c.bucket);
  bool others_alone = false;
  for (int i = 1; i < kSlots; ++i) { const rtx_scene::Call& o = s->calls[(s->cur + i) % kSlots]; others_alone = others_alone || (o.pending && o.alone); }
  if (c.alone || others_alone)
    for (int i = 1; i < kSlots; ++i) if (int rc = finish_call(s, s->calls[(s->cur + i) % kSlots])) return rc;
  c.events.reset(); c.trace_spans.clear(); c.all_spans.clear(); c.camera_span = (size_t)-1;
  c.span_kind.clear();
  for (int k = 0; k < 3; ++k) { c.mode[k] = -1; c.timed[k] = false; }
  c.pipeline = c.shade = c.shadows = c.tune = c.throughput = c.tunable = c.guard = false; c.max_gen = 0; c.NLS = c.G = 0;
  std::memset(&c.stats, 0, sizeof c.stats);
  span_begin(s, false);
  return RTX_OK;
}
// ... and ends here: closes the span; the call is now in flight.
static void stats_enqueue_end(rtx_scene* s) {
  if (cur_call(s).guard) (void)hipMemcpyAsync(cur_call(s).hg, ws(s).w_guard.p, 12, hipMemcpyDeviceToHost, ws(s).stream);
  span_end(s, false);
  cur_call(s).pending = true;
}
// The synchronous form: every call in flight is finished, oldest first.
static int stats_end(rtx_scene* s) {
  stats_enqueue_end(s);
  for (int i = 1; i <= kSlots; ++i) if (int rc = finish_call(s, s->calls[(s->cur + i) % kSlots])) return rc;
  for (rtx_scene::Call& c : s->calls) c.unreported = false;     // a synchronous call finishes everything
  return RTX_OK;
}
static int check_ready(rtx_scene* s) {
  if (!s) return fail(RTX_EINVAL, "null scene");
  if (!s->committed) return fail(RTX_ESTATE, "scene not committed (call rtx_scene_commit after any change)");
  HIP_TRY(hipSetDevice(s->device));
  return RTX_OK;
}

// Traversal launches: persistent waves (grid = what is resident) over static ranges of
// `grab` rays, plus the global overflow columns of the per-lane stacks.
static uint32_t grab_size() {
  static const uint32_t g = [] {
    const char* e = std::getenv("RTX_GRAB");          // tuning hook; multiples of 64
    const long v = e ? std::atol(e) : 0;
    return (v >= 64 && v <= 65536) ? (uint32_t)(v / 64 * 64) : 64u;
  }();
  return g;
}
// Scheduler per launch kind: SCHED_GROUPS, SCHED_REFILL (a leaf phase tests whole leaves) or
// SCHED_REFILL_SPLIT (a leaf phase stops once fewer than kTriMinSplit lanes have triangles left; the
// rest stays pending).  Order of precedence: RTX_FLAG_SCHED_* of the scene, the RTX_SCHED
// environment variable ("refill" | "split" | "groups"), the tuned choice, the heuristic.
enum { KIND_CAMERA = 0, KIND_SHADOW = 1, KIND_BOUNCE = 2 };
enum { SCHED_GROUPS = 0, SCHED_REFILL = 1, SCHED_REFILL_SPLIT = 2 };
constexpr uint32_t kTriMinSplit = 24;
constexpr uint64_t kTuneMinRays = 200000;   // smaller launches are not worth timing
static int sched_heuristic(const rtx_scene* s, int kind, uint64_t rays) {
  // Measured: the refilling scheduler wins on queued rays (shadow / bounce rays:
  // incoherent) and on camera rays once a wave has many 64-ray units to draw from (4K frames);
  // group by group wins on the coherent camera rays of smaller frames.  Split leaves win on
  // triangle soups (few tests hit, nothing is lost by testing late) and lose on closed surfaces.
  if (kind != KIND_CAMERA) return SCHED_REFILL;
  return rays >= (uint64_t)trace_grid(s) * (kBlock / 64) * 64 * 8 ? SCHED_REFILL : SCHED_GROUPS;
}
static int sched_forced() {
  static const int forced = [] {
    const char* e = std::getenv("RTX_SCHED");
    if (!e) return -1;
    return std::strcmp(e, "refill") == 0 ? SCHED_REFILL : std::strcmp(e, "split") == 0 ? SCHED_REFILL_SPLIT
         : std::strcmp(e, "groups") == 0 ? SCHED_GROUPS : -1;
  }();
  return forced;
}
static int sched_pick(rtx_scene* s, int kind, uint64_t rays) {
  rtx_scene::Call& c = cur_call(s);
  if (!c.span_kind.empty()) c.span_kind.back() = kind;
  if (s->flags & RTX_FLAG_SCHED_REFILL) return (s->flags & RTX_FLAG_SCHED_SPLIT) ? SCHED_REFILL_SPLIT : SCHED_REFILL;
  if (s->flags & RTX_FLAG_SCHED_GROUPS) return SCHED_GROUPS;
  if (sched_forced() >= 0) return sched_forced();
  if (c.mode[kind] >= 0) return c.mode[kind];          // one mode per kind and call
  rtx_scene::Tune& T = s->tune[kind][c.bucket];
  int m;
  if (T.choice >= 0) m = T.choice;
  else if (!c.tunable) m = sched_heuristic(s, kind, rays);   // a query call: never accounted by tune_update, so it takes no trial
  else if (!T.started || T.too_small) { m = sched_heuristic(s, kind, rays); T.started = true; }   // first call (cold caches) and small launches: not timed
  else {
    m = 0;                                              // the mode tried least so far: groups, refill, split, twice each
    for (int k = 1; k < 3; ++k) if (T.launched[k] < T.launched[m]) m = k;
    ++T.launched[m];
    c.timed[kind] = true;
  }
  c.mode[kind] = m;
  return m;
}
// When a render call has finished: per kind, milliseconds per ray of its launches -> tuning state.
static void tune_update(rtx_scene* s, rtx_scene::Call& c) {
  if ((s->flags & (RTX_FLAG_SCHED_REFILL | RTX_FLAG_SCHED_GROUPS)) || sched_forced() >= 0) return;
  double ms[3] = {0, 0, 0};
  bool seen[3] = {false, false, false};
  for (size_t i = 0; i < c.trace_spans.size() && i < c.span_kind.size(); ++i) {
    const int k = c.span_kind[i];
    if (k < 0) continue;
    float m = 0;
    (void)hipEventElapsedTime(&m, c.trace_spans[i].first, c.trace_spans[i].second);
    ms[k] += m; seen[k] = true;
  }
  const uint64_t rays[3] = {c.stats.rays_primary, c.stats.rays_shadow, c.stats.rays_reflection + c.stats.rays_refraction};
  for (int k = 0; k < 3; ++k) {
    rtx_scene::Tune& T = s->tune[k][c.bucket];
    if (!seen[k] || rays[k] == 0) continue;
    if (T.choice >= 0) continue;
    const int m = c.mode[k];
    if (rays[k] < kTuneMinRays) {                       // not worth timing: the heuristic from now on
      T.too_small = true;
      if (c.timed[k] && m >= 0 && T.launched[m] > 0) --T.launched[m];
      continue;
    }
    T.too_small = false;
    if (!c.timed[k] || m < 0) continue;
    // the better of a mode's trials counts: the first launch of a kernel variant also pays for loading it
    const double per_ray = ms[k] / (double)rays[k];
    T.ms_per_ray[m] = T.done[m] ? std::min(T.ms_per_ray[m], per_ray) : per_ray;
    ++T.done[m];
    if (T.done[0] >= 2 && T.done[1] >= 2 && T.done[2] >= 2) {
      // the heuristic's pick stays unless another one is clearly (2 %) faster
      const int h = sched_heuristic(s, k, rays[k]);
      const double* per = T.ms_per_ray;
      int pick = h;
      for (int q = 0; q < 3; ++q)
        if (per[q] < 0.98 * per[h] && per[q] < per[pick]) pick = q;
      T.choice = pick;
      static const bool log = std::getenv("RTX_TUNE_LOG") != nullptr;
      if (log)
        std::fprintf(stderr, "[rtx] scheduler for launch kind %d (%llu rays): groups %.3g, refill %.3g, split %.3g ns/ray -> %d\n", k,
                     (unsigned long long)rays[k], 1e6 * per[0], 1e6 * per[1], 1e6 * per[2], pick);
    }
  }
}
// Overlapped bounce launches (run_pipeline): RTX_OVERLAP=0 never, =1 always; by default once the tuner has
// settled the shadow and the bounce queue of this scene (its trial frames time one launch at a time).
static bool overlap_launches(rtx_scene* s) {
  static const int forced = [] { const char* e = std::getenv("RTX_OVERLAP"); return e ? std::atoi(e) : -1; }();
  if (forced == 0) return false;
  if (forced > 0) return true;
  if ((s->flags & (RTX_FLAG_SCHED_REFILL | RTX_FLAG_SCHED_GROUPS)) || sched_forced() >= 0) return true;
  const int b = cur_call(s).bucket;
  return s->tune[KIND_SHADOW][b].choice >= 0 && s->tune[KIND_BOUNCE][b].choice >= 0;
}
// Calls in flight run side by side on their slots' streams (RTX_OVERLAP_FRAMES=0 never, =1 always): by default
// once every launch kind that calls of this size class have used is settled in the tuner, whose trials need the GPU alone.
static bool bucket_settled(const rtx_scene* s, int bucket) {
  static const int forced = [] { const char* e = std::getenv("RTX_OVERLAP_FRAMES"); return e ? std::atoi(e) : -1; }();
  if (forced == 0) return false;
  if (forced > 0) return true;
  if ((s->flags & (RTX_FLAG_SCHED_REFILL | RTX_FLAG_SCHED_GROUPS)) || sched_forced() >= 0) return true;
  bool any = false;
  for (int k = 0; k < 3; ++k) {
    const rtx_scene::Tune& T = s->tune[k][bucket];
    if (!T.started) continue;
    any = true;
    if (T.choice < 0 && !T.too_small) return false;
  }
  return any;
}
// Grid of a traversal launch over `quads` quads of 64-ray units: what is resident, or one block per quad.
// (Measured and not kept: smaller grids for enqueued calls, so that a wave draws 4-32 units instead of 2-3 when
// the launch is small and the other calls in flight take the CUs left free -- no gain on any workload: with
// fewer waves per SIMD every step waits for its loads instead.)
static uint32_t trace_blocks(rtx_scene* s, uint64_t quads) {
  uint64_t g = std::max<uint64_t>(quads, 1);
  static const uint32_t upw = [] { const char* e = std::getenv("RTX_UPW"); const long x = e ? std::atol(e) : 0; return x > 0 ? (uint32_t)x : 1u; }();   // (experiment hook)
  if (upw > 1 && cur_call(s).throughput) g = std::max<uint64_t>((g + upw - 1) / upw, std::min<uint64_t>(g, (uint64_t)s->ncu));
  return (uint32_t)std::min<uint64_t>(g, (uint64_t)trace_grid(s));
}
// Domain guard + verification behind a closest-hit launch of caller or camera rays (rtx_kernels.h, guard_and_verify).
static uint32_t verify_stride(const rtx_scene* s, uint64_t n) {
  return (s->flags & RTX_FLAG_VERIFY) ? (uint32_t)std::max<uint64_t>(1, n / 2048) : 0u;
}
static int guard_begin(rtx_scene* s) {
  rtx_scene::Call& c = cur_call(s);
  if (!c.guard) { HIP_TRY(hipMemsetAsync(ws(s).w_guard.p, 0, 16, ws(s).stream)); c.guard = true; }
  return RTX_OK;
}
static int guard_queue(rtx_scene* s, const float4* rays, const uint32_t* count, uint64_t n, uint32_t bounded, uint2* hits) {
  if (s->dev.ntris == 0 && s->dev.nspheres == 0) return RTX_OK;
  if (int rc = guard_begin(s)) return rc;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(n, 1), (uint64_t)s->ncu * 8);
  hipLaunchKernelGGL(k_guard_queue, dim3(grid), dim3(kBlock), 0, ws(s).stream, s->dev, rays, count, 1u, bounded, s->domain, verify_stride(s, n), hits,
                     ws(s).w_guard.as<uint32_t>());
  return RTX_OK;
}
static bool camera_beyond_domain(const rtx_scene* s, const DevCamera* cams, uint32_t nframes) {
  bool far = false;
  for (uint32_t f = 0; f < nframes; ++f) {
    const float* o = cams[f].origin;
    far = far || std::max(std::fabs(o[0]), std::max(std::fabs(o[1]), std::fabs(o[2]))) > s->domain;
  }
  return far;
}
// A host-side check, made before the call takes a slot or queues anything: every pixel of a camera beyond the domain is a
// brute-force scan, fine for the reference's own scenes, not for millions of triangles.
static int check_camera_domain(const rtx_scene* s, const DevCamera* cams, uint32_t nframes, uint32_t nslots) {
  if (s->dev.ntris == 0 || !camera_beyond_domain(s, cams, nframes)) return RTX_OK;
  if ((double)nslots * (double)s->dev.ntris > 7e10)
    return fail(RTX_ELIMIT, "the camera is farther than %g from the origin, beyond what this mesh's culling boxes are padded for, and %u pixels x %u "
                            "triangles is too much to answer by brute force: move the camera closer or render a smaller rectangle", (double)s->domain,
                nslots, s->dev.ntris);
  return RTX_OK;
}
static int guard_camera(rtx_scene* s, const DevCamera& C, const DevCamera* cams, uint32_t nframes, uint32_t nslots, uint2* hits) {
  const bool far = camera_beyond_domain(s, cams ? cams : &C, cams ? nframes : 1u);     // C: frame 0 with the device pointers set; cams: the frames of the batch (host copies)
  const uint32_t vs = verify_stride(s, nslots);
  if ((!far && vs == 0) || s->dev.ntris == 0) return RTX_OK;
  if (int rc = guard_begin(s)) return rc;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(nslots, (uint64_t)s->ncu * 8);
  hipLaunchKernelGGL(k_guard_camera, dim3(grid), dim3(kBlock), 0, ws(s).stream, s->dev, C, nslots, s->domain, vs, hits, ws(s).w_guard.as<uint32_t>());
  return RTX_OK;
}
static bool any_unordered() { static const bool on = [] { const char* e = std::getenv("RTX_ANY_UNORDERED"); return !e || std::atoi(e) != 0; }(); return on; }
static int launch_trace_camera(rtx_scene* s, const DevCamera& C, uint32_t tiles, uint2* hits) {
  const uint32_t grid = trace_blocks(s, tiles);
  const uint32_t grid_r = std::min<uint32_t>(grid, (uint32_t)s->ncu * RTX_BLOCKS_REFILL);
  const int mode = sched_pick(s, KIND_CAMERA, (uint64_t)tiles * kBlock);
  const bool fast = (s->flags & RTX_FLAG_FAST_TRIANGLES) != 0;
  const uint32_t tri_min = mode == SCHED_REFILL_SPLIT ? kTriMinSplit : 1u;
  uint32_t* ovf = ws(s).w_stack_ovf.as<uint32_t>();
  if (mode != SCHED_GROUPS) {
    if (fast) hipLaunchKernelGGL((k_trace_camera<true, FastTri>), dim3(grid_r), dim3(kBlock), 0, ws(s).stream, s->dev, C, tiles, grab_size(), tri_min, ovf, hits);
    else hipLaunchKernelGGL((k_trace_camera<true, RefTri>), dim3(grid_r), dim3(kBlock), 0, ws(s).stream, s->dev, C, tiles, grab_size(), tri_min, ovf, hits);
  } else {
    if (fast) hipLaunchKernelGGL((k_trace_camera<false, FastTri>), dim3(grid), dim3(kBlock), 0, ws(s).stream, s->dev, C, tiles, grab_size(), 1u, ovf, hits);
    else hipLaunchKernelGGL((k_trace_camera<false, RefTri>), dim3(grid), dim3(kBlock), 0, ws(s).stream, s->dev, C, tiles, grab_size(), 1u, ovf, hits);
  }
  return RTX_OK;
}
static int launch_trace_queue(rtx_scene* s, const float4* rays, const uint32_t* count, uint32_t stride, uint64_t bound, uint2* hits,
                              bool shadow = false, bool any_blocker = false, bool side = false, bool compact = false) {
  hipStream_t on = side ? ws(s).stream2 : ws(s).stream;                        // side launches have their own overflow columns
  uint32_t* ovf = (side ? ws(s).w_stack_ovf2 : ws(s).w_stack_ovf).as<uint32_t>();
  const uint32_t grid = trace_blocks(s, ((bound + 63) / 64 * 64 * stride + kBlock - 1) / kBlock);
  const uint32_t grid_r = std::min<uint32_t>(grid, (uint32_t)s->ncu * RTX_BLOCKS_REFILL);
  const int mode = sched_pick(s, shadow ? KIND_SHADOW : KIND_BOUNCE, bound * stride);
  const uint32_t bounded = shadow ? ((any_blocker ? 3u : 1u) | 4u | (compact ? 8u : 0u)) : 0u;      // 4: the queue is in shadow_index order; 8: one hit point per parent
  const bool fast = (s->flags & RTX_FLAG_FAST_TRIANGLES) != 0;
  const uint32_t tri_min = mode == SCHED_REFILL_SPLIT ? kTriMinSplit : 1u;
  // every ray of the launch stops at its first hit: no child sort in its node steps (RTX_ANY_UNORDERED=0: the A/B switch)
  const bool any = shadow && any_blocker && (s->dev.flags & 2u) && any_unordered();
#define RTX_LAUNCH_Q(REFILL, TRI, G, TM) do { \
    if (any) hipLaunchKernelGGL((k_trace_queue<REFILL, TRI, true>), dim3(G), dim3(kBlock), 0, on, s->dev, rays, count, stride, bounded, grab_size(), TM, ovf, hits); \
    else hipLaunchKernelGGL((k_trace_queue<REFILL, TRI, false>), dim3(G), dim3(kBlock), 0, on, s->dev, rays, count, stride, bounded, grab_size(), TM, ovf, hits); } while (0)
  if (mode != SCHED_GROUPS) {
    if (fast) RTX_LAUNCH_Q(true, FastTri, grid_r, tri_min); else RTX_LAUNCH_Q(true, RefTri, grid_r, tri_min);
  } else {
    if (fast) RTX_LAUNCH_Q(false, FastTri, grid, 1u); else RTX_LAUNCH_Q(false, RefTri, grid, 1u);
  }
#undef RTX_LAUNCH_Q
  return RTX_OK;
}

// The shadow launch of a camera frame without a parent list (run_pipeline's direct mode): every output slot x light sample.
static int launch_trace_shadow_direct(rtx_scene* s, const DevCamera* cam_dev, const uint2* hits0, uint32_t nslots, uint32_t nls, uint2* hits) {
  const uint64_t bound = (uint64_t)nslots;
  const uint32_t grid = trace_blocks(s, ((bound + 63) / 64 * 64 * nls + kBlock - 1) / kBlock);
  const uint32_t grid_r = std::min<uint32_t>(grid, (uint32_t)s->ncu * RTX_BLOCKS_REFILL);
  const int mode = sched_pick(s, KIND_SHADOW, bound * nls);
  const bool fast = (s->flags & RTX_FLAG_FAST_TRIANGLES) != 0;
  const uint32_t tri_min = mode == SCHED_REFILL_SPLIT ? kTriMinSplit : 1u;
  uint32_t* ovf = ws(s).w_stack_ovf.as<uint32_t>();
  hipStream_t on = ws(s).stream;
  const bool any = (s->dev.flags & 2u) && any_unordered();
#define RTX_LAUNCH_D(REFILL, TRI, G, TM) do { \
    if (any) hipLaunchKernelGGL((k_trace_shadow_direct<REFILL, TRI, true>), dim3(G), dim3(kBlock), 0, on, s->dev, cam_dev, hits0, nslots, nls, 1u, grab_size(), TM, ovf, hits); \
    else hipLaunchKernelGGL((k_trace_shadow_direct<REFILL, TRI, false>), dim3(G), dim3(kBlock), 0, on, s->dev, cam_dev, hits0, nslots, nls, 1u, grab_size(), TM, ovf, hits); } while (0)
  if (mode != SCHED_GROUPS) {
    if (fast) RTX_LAUNCH_D(true, FastTri, grid_r, tri_min); else RTX_LAUNCH_D(true, RefTri, grid_r, tri_min);
  } else {
    if (fast) RTX_LAUNCH_D(false, FastTri, grid, 1u); else RTX_LAUNCH_D(false, RefTri, grid, 1u);
  }
#undef RTX_LAUNCH_D
  return RTX_OK;
}

// Traces `n` caller rays held on the device (xyz arrays) into w_hits[0]; w_rays[0] keeps the records.
static int trace_rays_dev(rtx_scene* s, const float* d_o, const float* d_d, uint32_t n) {
  if (int rc = ws(s).w_rays[0].ensure((size_t)n * 32)) return rc;
  if (int rc = ws(s).w_hits[0].ensure((size_t)n * 8)) return rc;
  if (int rc = ws(s).w_counters.ensure(64)) return rc;
  HIP_TRY(hipMemcpyAsync(ws(s).w_counters.p, &n, 4, hipMemcpyHostToDevice, ws(s).stream));
  hipLaunchKernelGGL(k_make_rays, dim3((n + 255) / 256), dim3(256), 0, ws(s).stream, d_o, d_d, n, ws(s).w_rays[0].as<float4>());
  span_begin(s, true);
  if (int rc = launch_trace_queue(s, ws(s).w_rays[0].as<float4>(), ws(s).w_counters.as<uint32_t>(), 1u, n, ws(s).w_hits[0].as<uint2>())) return rc;
  span_end(s, true);
  return guard_queue(s, ws(s).w_rays[0].as<float4>(), ws(s).w_counters.as<uint32_t>(), n, 0u, ws(s).w_hits[0].as<uint2>());
}

extern "C" int rtx_trace_closest_dev(rtx_scene* s, const float* o, const float* d, uint32_t n, float* t, int32_t* face) {
  if (int rc = check_ready(s)) return rc;
  if (n == 0) return RTX_OK;
  if (!o || !d) return fail(RTX_EINVAL, "null ray arrays");
  if (n >= (1u << 31)) return fail(RTX_ELIMIT, "too many rays");
  if (int rc = stats_begin(s)) return rc;
  if (int rc = trace_rays_dev(s, o, d, n)) return rc;
  hipLaunchKernelGGL(k_unpack_hits, dim3((n + 255) / 256), dim3(256), 0, ws(s).stream, s->dev, ws(s).w_rays[0].as<float4>(),
                     ws(s).w_hits[0].as<uint2>(), n, t, face, (float*)nullptr, (float*)nullptr, (int*)nullptr);
  cur_call(s).stats.rays_primary = n;
  return stats_end(s);
}

extern "C" int rtx_trace_closest(rtx_scene* s, const float* o, const float* d, uint32_t n, float* t, int32_t* face,
                                 float* normal, float* point, int32_t* mat) {
  if (int rc = check_ready(s)) return rc;
  if (n == 0) return RTX_OK;
  if (!o || !d) return fail(RTX_EINVAL, "null ray arrays");
  if (n >= (1u << 31)) return fail(RTX_ELIMIT, "too many rays");
  if (int rc = stats_begin(s)) return rc;      // (takes the call slot: its stream and workspace from here on)
  const size_t b3 = (size_t)n * 12, b1 = (size_t)n * 4;
  if (int rc = s->w_io[0].ensure(b3)) return rc;
  if (int rc = s->w_io[1].ensure(b3)) return rc;
  if (int rc = s->w_io[2].ensure(b1)) return rc;
  if (int rc = s->w_io[3].ensure(b1)) return rc;
  if (int rc = s->w_io[4].ensure(b3 * 2)) return rc;
  if (int rc = s->w_io[5].ensure(b1)) return rc;
  HIP_TRY(hipMemcpyAsync(s->w_io[0].p, o, b3, hipMemcpyHostToDevice, ws(s).stream));
  HIP_TRY(hipMemcpyAsync(s->w_io[1].p, d, b3, hipMemcpyHostToDevice, ws(s).stream));
  if (int rc = trace_rays_dev(s, s->w_io[0].as<float>(), s->w_io[1].as<float>(), n)) return rc;
  float* d_n = s->w_io[4].as<float>();
  float* d_p = d_n + (size_t)n * 3;
  hipLaunchKernelGGL(k_unpack_hits, dim3((n + 255) / 256), dim3(256), 0, ws(s).stream, s->dev, ws(s).w_rays[0].as<float4>(),
                     ws(s).w_hits[0].as<uint2>(), n, s->w_io[2].as<float>(), s->w_io[3].as<int>(), d_n, d_p, s->w_io[5].as<int>());
  cur_call(s).stats.rays_primary = n;
  if (int rc = stats_end(s)) return rc;
  if (t) HIP_TRY(hipMemcpy(t, s->w_io[2].p, b1, hipMemcpyDeviceToHost));
  if (face) HIP_TRY(hipMemcpy(face, s->w_io[3].p, b1, hipMemcpyDeviceToHost));
  if (normal) HIP_TRY(hipMemcpy(normal, d_n, b3, hipMemcpyDeviceToHost));
  if (point) HIP_TRY(hipMemcpy(point, d_p, b3, hipMemcpyDeviceToHost));
  if (mat) HIP_TRY(hipMemcpy(mat, s->w_io[5].p, b1, hipMemcpyDeviceToHost));
  return RTX_OK;
}

extern "C" int rtx_trace_visible(rtx_scene* s, const float* light, const float* pos, uint32_t n, uint8_t* visible,
                                 int32_t* bmat) {
  if (int rc = check_ready(s)) return rc;
  if (n == 0) return RTX_OK;
  if (!light || !pos || !visible) return fail(RTX_EINVAL, "null argument");
  if (n >= (1u << 31)) return fail(RTX_ELIMIT, "too many rays");
  if (int rc = stats_begin(s)) return rc;
  const size_t b3 = (size_t)n * 12;
  if (int rc = s->w_io[0].ensure(b3)) return rc;
  if (int rc = s->w_io[1].ensure(b3)) return rc;
  if (int rc = s->w_io[2].ensure(n)) return rc;
  if (int rc = s->w_io[3].ensure((size_t)n * 4)) return rc;
  if (int rc = ws(s).w_rays[0].ensure((size_t)n * 32)) return rc;
  if (int rc = ws(s).w_hits[0].ensure((size_t)n * 8)) return rc;
  if (int rc = ws(s).w_counters.ensure(64)) return rc;
  HIP_TRY(hipMemcpyAsync(s->w_io[0].p, light, b3, hipMemcpyHostToDevice, ws(s).stream));
  HIP_TRY(hipMemcpyAsync(s->w_io[1].p, pos, b3, hipMemcpyHostToDevice, ws(s).stream));
  HIP_TRY(hipMemcpyAsync(ws(s).w_counters.p, &n, 4, hipMemcpyHostToDevice, ws(s).stream));
  hipLaunchKernelGGL(k_make_shadow_rays, dim3((n + 255) / 256), dim3(256), 0, ws(s).stream, s->w_io[0].as<float>(),
                     s->w_io[1].as<float>(), n, ws(s).w_rays[0].as<float4>());
  span_begin(s, true);
  if (int rc = launch_trace_queue(s, ws(s).w_rays[0].as<float4>(), ws(s).w_counters.as<uint32_t>(), 1u, n, ws(s).w_hits[0].as<uint2>(), true)) return rc;
  span_end(s, true);
  if (int rc = guard_queue(s, ws(s).w_rays[0].as<float4>(), ws(s).w_counters.as<uint32_t>(), n, 1u, ws(s).w_hits[0].as<uint2>())) return rc;
  hipLaunchKernelGGL(k_visible, dim3((n + 255) / 256), dim3(256), 0, ws(s).stream, s->dev, s->w_io[0].as<float>(),
                     ws(s).w_rays[0].as<float4>(), ws(s).w_hits[0].as<uint2>(), n, s->w_io[2].as<unsigned char>(), s->w_io[3].as<int>());
  cur_call(s).stats.rays_shadow = n;
  if (int rc = stats_end(s)) return rc;
  HIP_TRY(hipMemcpy(visible, s->w_io[2].p, n, hipMemcpyDeviceToHost));
  if (bmat) HIP_TRY(hipMemcpy(bmat, s->w_io[3].p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return RTX_OK;
}

// ------------------------------------------------------------------ render
static int make_camera(const rtx_camera* cam, rtx_rect r, DevCamera* C) {
  if (r.x1 <= r.x0 || r.y1 <= r.y0) return fail(RTX_EINVAL, "empty rectangle");
  if (!(cam->viewport[2] > 0) || !(cam->viewport[3] > 0)) return fail(RTX_EINVAL, "viewport must have positive size");
  if ((int64_t)(r.x1 - r.x0) * (int64_t)(r.y1 - r.y0) >= (1ll << 31)) return fail(RTX_ELIMIT, "rectangle has too many pixels");
  // Transform::inverse (Eigen/src/Geometry/Transform.h:1202-1229) and Camera::getCenter (tucano/camera.hpp:115-118)
  float lin[9], inv[9], ninv[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) lin[3 * i + j] = cam->view[4 * i + j];
  mat3_inverse(lin, inv);
  for (int k = 0; k < 9; ++k) ninv[k] = -inv[k];
  const V3 tr = v3(cam->view[3], cam->view[7], cam->view[11]);
  const V3 it = mat3_mul(ninv, tr);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C->inv34[4 * i + j] = inv[3 * i + j];
  C->inv34[3] = it.x; C->inv34[7] = it.y; C->inv34[11] = it.z;
  const V3 ctr = mat3_mul(inv, -tr);
  C->origin[0] = ctr.x; C->origin[1] = ctr.y; C->origin[2] = ctr.z;
  for (int k = 0; k < 4; ++k) C->vp[k] = cam->viewport[k];
  // getPerspectiveScale (tucano/camera.hpp:263-266) and `float scale = 1.0/...` (:165)
  const float pscale = (float)((double)1.0f / std::tan((double)(cam->fovy / 2.0f) * (M_PI / (double)180.0f)));
  const float scale = (float)(1.0 / (double)pscale);
  C->aspect_scale = cam->aspect * scale;
  C->scale = scale;
  C->sub[0] = cam->subpixel[0]; C->sub[1] = cam->subpixel[1];
  C->x0 = r.x0; C->y0 = r.y0; C->w = r.x1 - r.x0; C->h = r.y1 - r.y0;
  C->ts = 0; C->part = 0; C->nparts = 1; C->tiles_x = 0; C->ntiles_local = 0; C->tab = nullptr;
  C->nframes = 1; C->frame_slots = 0; C->frame_blocks = 0; C->frames = nullptr;
  return RTX_OK;
}

static int render_impl(rtx_scene* s, const DevCamera* cams, uint32_t nframes, uint32_t npix, uint32_t nblocks, const rtx_options* opt,
                       float* rgb, float* t, int32_t* face, bool wait = true);
static int check_render_args(rtx_scene* s, const rtx_camera* cam, const rtx_options* opt, const float* rgb);

static int check_render_args(rtx_scene* s, const rtx_camera* cam, const rtx_options* opt, const float* rgb) {
  if (!cam || !opt) return fail(RTX_EINVAL, "null camera/options");
  if (opt->max_bounces < 0 || opt->max_bounces > 16) return fail(RTX_ELIMIT, "max_bounces must be in [0,16]");
  if (opt->gi_rays < 0 || opt->gi_rays > 64) return fail(RTX_ELIMIT, "gi_rays must be in [0,64]");
  if (opt->refraction && opt->reflection && opt->max_bounces > 8) return fail(RTX_ELIMIT, "reflection + refraction trees are limited to 8 bounces");
  if (rgb && opt->gi_rays > 0 && !s->have_leaves)
    return fail(RTX_ESTATE, "gi_rays needs the reference leaf list (rtx_scene_build_reference_leafboxes) before commit");
  if (rgb && s->dev.nlights + s->dev.nsl == 0) return fail(RTX_EINVAL, "shading needs at least one light (the reference divides by the light count)");
  return RTX_OK;
}

extern "C" int rtx_render_dev(rtx_scene* s, const rtx_camera* cam, const rtx_options* opt, rtx_rect rect, float* rgb,
                              float* t, int32_t* face) {
  if (int rc = check_ready(s)) return rc;
  if (int rc = check_render_args(s, cam, opt, rgb)) return rc;
  DevCamera C{};
  if (int rc = make_camera(cam, rect, &C)) return rc;
  const uint32_t npix = (uint32_t)C.w * (uint32_t)C.h;
  const uint32_t tiles = (uint32_t)((C.w + 15) / 16) * (uint32_t)((C.h + 15) / 16);
  return render_impl(s, &C, 1, npix, tiles, opt, rgb, t, face);
}

// rtx_render_dev without the wait at the end: the call's kernels are queued on the scene's stream and
// the function returns.  rtx_render_finish waits for the oldest such call and reports its statistics.
extern "C" int rtx_render_dev_enqueue(rtx_scene* s, const rtx_camera* cam, const rtx_options* opt, rtx_rect rect, float* rgb,
                                      float* t, int32_t* face) {
  if (int rc = check_ready(s)) return rc;
  if (int rc = check_render_args(s, cam, opt, rgb)) return rc;
  DevCamera C{};
  if (int rc = make_camera(cam, rect, &C)) return rc;
  const uint32_t npix = (uint32_t)C.w * (uint32_t)C.h;
  const uint32_t tiles = (uint32_t)((C.w + 15) / 16) * (uint32_t)((C.h + 15) / 16);
  return render_impl(s, &C, 1, npix, tiles, opt, rgb, t, face, false);
}
extern "C" int rtx_render_finish(rtx_scene* s, rtx_stats* out) {
  if (!s) return fail(RTX_EINVAL, "null scene");
  if (s->device < 0) return fail(RTX_ESTATE, "no enqueued render call to finish");
  HIP_TRY(hipSetDevice(s->device));          // (a scene changed after the enqueue can still finish what it started)
  // the oldest enqueued call not handed out yet (the library may have waited for it already: calls run one at a
  // time while the scheduler tuner is timing them)
  for (int i = 1; i <= kSlots; ++i) {
    rtx_scene::Call& c = s->calls[(s->cur + i) % kSlots];
    if (!c.unreported) continue;
    if (int rc = finish_call(s, c)) return rc;
    c.unreported = false;
    if (out) *out = c.stats;
    return RTX_OK;
  }
  return fail(RTX_ESTATE, "no enqueued render call to finish");
}

extern "C" int rtx_tiles_count(rtx_rect rect, int32_t tile_size, int32_t part, int32_t nparts) {
  if (rect.x1 <= rect.x0 || rect.y1 <= rect.y0) return fail(RTX_EINVAL, "empty rectangle");
  if (tile_size < 16 || tile_size > 1024 || (tile_size & 15)) return fail(RTX_EINVAL, "tile_size must be a multiple of 16 in [16,1024]");
  if (nparts < 1 || part < 0 || part >= nparts) return fail(RTX_EINVAL, "need 0 <= part < nparts");
  const int64_t tx = (rect.x1 - rect.x0 + tile_size - 1) / tile_size, ty = (rect.y1 - rect.y0 + tile_size - 1) / tile_size;
  const int64_t total = tx * ty;
  if ((int64_t)(rect.x1 - rect.x0) * (int64_t)(rect.y1 - rect.y0) >= (1ll << 31)) return fail(RTX_ELIMIT, "rectangle has too many pixels");
  const int64_t mine = (total - part + nparts - 1) / nparts;
  if (mine >= (1ll << 31) / 256) return fail(RTX_ELIMIT, "too many tiles in a part");
  return (int)mine;
}

// rtx_render_tiles_dev's layout undone: the compact tile buffers of all parts (one after the other,
// part_stride_pixels apart) -> the row-major rectangle.  Stateless; runs on `stream` (a hipStream_t, may be
// null for the default stream) and does not wait for it.
extern "C" int rtx_tiles_assemble_bytes_dev(const void* parts, int32_t nparts, uint64_t part_stride_pixels, rtx_rect rect, int32_t tile_size,
                                            int32_t bytes_per_pixel, void* out, void* stream) {
  if (!parts || !out) return fail(RTX_EINVAL, "null buffer");
  if (bytes_per_pixel < 1 || bytes_per_pixel > 64) return fail(RTX_EINVAL, "bytes_per_pixel must be in [1,64]");
  const int kmax = rtx_tiles_count(rect, tile_size, 0, nparts);      // validates rect / tile_size / nparts
  if (kmax < 0) return kmax;
  if (part_stride_pixels < (uint64_t)kmax * tile_size * tile_size) return fail(RTX_EINVAL, "part_stride_pixels smaller than a part");
  const uint32_t w = (uint32_t)(rect.x1 - rect.x0), h = (uint32_t)(rect.y1 - rect.y0);
  if ((uint64_t)w * h >= (1ull << 31)) return fail(RTX_ELIMIT, "rectangle has too many pixels");
  const uint32_t bpp = (uint32_t)bytes_per_pixel;
  const uint64_t bytes = (uint64_t)w * h * bpp;
  const bool aligned16 = (((uintptr_t)parts | (uintptr_t)out) & 15u) == 0 && ((part_stride_pixels * bpp) & 15u) == 0;
  // (tile sizes are multiples of 16 pixels, so a full tile row is a whole number of 16-byte pieces whatever the pixel size)
  if (((uint64_t)w * bpp & 15u) == 0 && ((uint64_t)(w % (uint32_t)tile_size) * bpp & 15u) == 0 && aligned16) {
    const uint32_t grid4 = (uint32_t)std::min<uint64_t>((bytes / 16 + 255) / 256, 256u * 64u);
    hipLaunchKernelGGL(k_assemble_tiles_v4, dim3(grid4), dim3(256), 0, (hipStream_t)stream, (const uint4*)parts, (uint32_t)nparts,
                       part_stride_pixels * bpp / 16, w, h, (uint32_t)tile_size, bpp, (uint4*)out);
    HIP_TRY(hipGetLastError());
    return RTX_OK;
  }
  const bool words = (bpp & 3u) == 0 && (((uintptr_t)parts | (uintptr_t)out) & 3u) == 0;
  const uint64_t n = words ? bytes / 4 : bytes;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256u * 64u);
  if (words)
    hipLaunchKernelGGL(k_assemble_tiles<uint32_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)parts, (uint32_t)nparts,
                       part_stride_pixels * (bpp / 4), w, h, (uint32_t)tile_size, bpp / 4, (uint32_t*)out);
  else
    hipLaunchKernelGGL(k_assemble_tiles<uint8_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)parts, (uint32_t)nparts,
                       part_stride_pixels * bpp, w, h, (uint32_t)tile_size, bpp, (uint8_t*)out);
  HIP_TRY(hipGetLastError());
  return RTX_OK;
}
extern "C" int rtx_tiles_assemble_dev(const void* parts, int32_t nparts, uint64_t part_stride_pixels, rtx_rect rect, int32_t tile_size,
                                      int32_t words_per_pixel, void* out, void* stream) {
  if (words_per_pixel < 1 || words_per_pixel > 16) return fail(RTX_EINVAL, "words_per_pixel must be in [1,16]");
  return rtx_tiles_assemble_bytes_dev(parts, nparts, part_stride_pixels, rect, tile_size, 4 * words_per_pixel, out, stream);
}

// ppmIO.hpp:145's quantisation of a frame (any layout: it works value by value), so that what leaves a rank is the
// 3 bytes per pixel the reference writes to result.ppm instead of 12.  Stateless, like the assemble call.
extern "C" int rtx_rgb_to_ppm8_dev(const float* rgb, uint64_t npixels, uint8_t* out, void* stream) {
  if (npixels == 0) return RTX_OK;
  if (!rgb || !out) return fail(RTX_EINVAL, "null buffer");
  if (npixels >= (1ull << 40)) return fail(RTX_ELIMIT, "too many pixels");
  const uint64_t n = npixels * 3;
  const bool aligned = ((uintptr_t)rgb & 15u) == 0 && ((uintptr_t)out & 3u) == 0;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(((aligned ? n / 4 : n) + 255) / 256 + 1, 256u * 32u);
  if (aligned) hipLaunchKernelGGL(k_rgb_to_ppm8, dim3(grid), dim3(256), 0, (hipStream_t)stream, rgb, n, out);
  else hipLaunchKernelGGL(k_rgb_to_ppm8_unaligned, dim3(grid), dim3(256), 0, (hipStream_t)stream, rgb, n, out);
  HIP_TRY(hipGetLastError());
  return RTX_OK;
}

static int render_tiles_impl(rtx_scene* s, const rtx_camera* cams, int32_t nframes, const rtx_options* opt, rtx_rect rect, int32_t tile_size,
                             int32_t part, int32_t nparts, float* rgb, float* t, int32_t* face, bool wait, bool batch = false) {
  if (int rc = check_ready(s)) return rc;
  if (nframes < 1 || nframes > kMaxBatch) return fail(RTX_ELIMIT, "a batch holds 1 to %d frames", kMaxBatch);
  if (int rc = check_render_args(s, cams, opt, rgb)) return rc;
  int nt;
  if (tile_size == 0 && batch) {         // (batch entry points only) the whole rectangle, row-major
    if (part != 0 || nparts != 1) return fail(RTX_EINVAL, "tile_size 0 renders the whole rectangle: part 0 of 1");
    nt = 1;
  } else {
    nt = rtx_tiles_count(rect, tile_size, part, nparts);
    if (nt < 0) return nt;
  }
  DevCamera C[kMaxBatch];
  for (int32_t f = 0; f < nframes; ++f) {
    C[f] = DevCamera{};
    if (int rc = make_camera(&cams[f], rect, &C[f])) return rc;
  }
  if (nt == 0) {                       // a part without tiles: an empty call, so that enqueue / finish still pair up
    if (wait) return RTX_OK;
    if (int rc = stats_begin(s)) return rc;
    stats_enqueue_end(s);
    cur_call(s).unreported = true;
    return RTX_OK;
  }
  uint64_t npix, nblocks;
  if (tile_size == 0) {
    npix = (uint64_t)C[0].w * (uint64_t)C[0].h;
    nblocks = (uint64_t)((C[0].w + 15) / 16) * (uint64_t)((C[0].h + 15) / 16);
  } else {
    // Frames of a batch lie one largest-part apart (part 0 owns the most tiles): every part of a partition then has the
    // same layout [frame][local tile], whatever its own tile count, and the parts of all ranks gather into one buffer.
    // The slots between a part's last tile and the next frame are dead (never read or written), like those of partial
    // edge tiles.
    const int kmax = nframes > 1 ? rtx_tiles_count(rect, tile_size, 0, nparts) : nt;
    npix = (uint64_t)kmax * (uint64_t)(tile_size * tile_size);
    nblocks = (uint64_t)nt * (uint64_t)((tile_size / 16) * (tile_size / 16));
  }
  if (npix * (uint64_t)nframes >= (1ull << 31) || nblocks * (uint64_t)nframes >= (1ull << 31) / 256)
    return fail(RTX_ELIMIT, "too many pixels in this part");
  for (int32_t f = 0; f < nframes; ++f) {
    if (tile_size) {
      C[f].ts = tile_size; C[f].part = part; C[f].nparts = nparts; C[f].tiles_x = (C[f].w + tile_size - 1) / tile_size;
      C[f].ntiles_local = (uint32_t)nt;
    }
    C[f].nframes = (uint32_t)nframes; C[f].frame_slots = (uint32_t)npix; C[f].frame_blocks = (uint32_t)nblocks;
  }
  return render_impl(s, C, (uint32_t)nframes, (uint32_t)(npix * nframes), (uint32_t)(nblocks * nframes), opt, rgb, t, face, wait);
}
extern "C" int rtx_render_tiles_dev(rtx_scene* s, const rtx_camera* cam, const rtx_options* opt, rtx_rect rect,
                                    int32_t tile_size, int32_t part, int32_t nparts, float* rgb, float* t, int32_t* face) {
  return render_tiles_impl(s, cam, 1, opt, rect, tile_size, part, nparts, rgb, t, face, true);
}
extern "C" int rtx_render_tiles_dev_enqueue(rtx_scene* s, const rtx_camera* cam, const rtx_options* opt, rtx_rect rect,
                                            int32_t tile_size, int32_t part, int32_t nparts, float* rgb, float* t, int32_t* face) {
  return render_tiles_impl(s, cam, 1, opt, rect, tile_size, part, nparts, rgb, t, face, false);
}
// `nframes` cameras over one rectangle and tile partition in ONE set of launches: frame f's pixels land at
// rgb + 3 f P, t + f P, face + f P with P = rtx_tiles_count(...) * tile_size^2 (tile_size 0: the whole rectangle
// row-major, P = its pixels).  What a rank of an N-GPU run calls for its share of several frames of an animation:
// the persistent traversal kernels then draw from nframes times as many rays per launch.
extern "C" int rtx_render_tiles_batch_dev(rtx_scene* s, const rtx_camera* cams, int32_t nframes, const rtx_options* opt, rtx_rect rect,
                                          int32_t tile_size, int32_t part, int32_t nparts, float* rgb, float* t, int32_t* face) {
  if (!cams) return fail(RTX_EINVAL, "null camera/options");
  return render_tiles_impl(s, cams, nframes, opt, rect, tile_size, part, nparts, rgb, t, face, true, true);
}
extern "C" int rtx_render_tiles_batch_dev_enqueue(rtx_scene* s, const rtx_camera* cams, int32_t nframes, const rtx_options* opt, rtx_rect rect,
                                                  int32_t tile_size, int32_t part, int32_t nparts, float* rgb, float* t, int32_t* face) {
  if (!cams) return fail(RTX_EINVAL, "null camera/options");
  return render_tiles_impl(s, cams, nframes, opt, rect, tile_size, part, nparts, rgb, t, face, false, true);
}

// The generation loop of the ray queue.  Primary nodes are either the pixels of a camera
// rectangle (implicit rays) or `n0` caller rays already converted to queue records in
// w_rays[0] (Flyscene::traceRay(bounce0, ray, inside)).
struct Primary { bool camera; DevCamera C; uint32_t n0, tiles; int bounce0; const DevCamera* cams = nullptr; uint32_t nframes = 1; };

static int run_pipeline(rtx_scene* s, const Primary& pr, const rtx_options* opt, float* rgb, float* t, int32_t* face) {
  const uint32_t NLS = s->dev.nlights + 5u * s->dev.nsl;
  const uint32_t G = (uint32_t)std::max(opt->gi_rays, 0);
  const bool shade = rgb != nullptr;
  const bool tree = opt->reflection && opt->refraction;
  const bool any_child = opt->reflection || opt->refraction;
  const int max_gen = (shade && any_child) ? std::max(0, opt->max_bounces - pr.bounce0) : 0;
  const uint64_t max_nodes = tree ? ((uint64_t)pr.n0 << max_gen) : pr.n0;
  // ray indices are 32-bit in the kernels and a wave's next unit may lie up to one grid of units past the end
  if ((max_nodes + 64) * std::max<uint64_t>(std::max(NLS, G), 1) >= (1ull << 31)) return fail(RTX_ELIMIT, "ray queue would exceed 2^31 entries");
  if (int rc = ws(s).w_hits[0].ensure((size_t)pr.n0 * 8)) return rc;
  // Direct mode: a camera frame whose nodes spawn nothing but shadow rays (no reflection / refraction generations, no GI probes)
  // needs no parent list -- the shadow launch walks the output slots and takes the hit ones (DirectSource), k_resolve_direct shades
  // every slot; k_shade's two passes over all slots, the parent list and the hit-point records are not made at all.  Same pixels,
  // bit for bit (RTX_FLAG_PARENT_LIST or RTX_DIRECT=0 keep the list; tests/test_gpu_direct.py renders both ways).
  static const bool direct_ok = [] { const char* e = std::getenv("RTX_DIRECT"); return !(e && std::atoi(e) == 0); }();
  const bool direct = direct_ok && !(s->flags & RTX_FLAG_PARENT_LIST) && pr.camera && shade && max_gen == 0 && G == 0 && !tree;
  if (shade) {
    if (!direct) { if (int rc = ws(s).w_parents.ensure((size_t)max_nodes * 4)) return rc; }
    if (opt->shadows) {
      const size_t tiled = ((size_t)max_nodes + 63) / 64 * 64;          // shadow_index layout: whole groups of 64 parents
      if (!direct) { if (int rc = ws(s).w_shadow_rays.ensure(tiled * 16)) return rc; }                // one hit point per parent (QueueSource forms the rays)
      if (int rc = ws(s).w_shadow_hits.ensure(tiled * NLS * 8)) return rc;
      if (direct) { if (int rc = ws(s).w_camera.ensure(sizeof(DevCamera))) return rc; }
    }
    if (G) {
      if (int rc = ws(s).w_gi_rays.ensure((size_t)max_nodes * G * 32)) return rc;
      if (int rc = ws(s).w_gi_hits.ensure((size_t)max_nodes * G * 8)) return rc;
    }
    if (tree) {
      if (int rc = ws(s).w_acc.ensure((size_t)pr.n0 * 24)) return rc;
      if (int rc = ws(s).w_acc_flag.ensure((size_t)pr.n0 * 4)) return rc;
    }
    if (max_gen > 0) {
      if (int rc = ws(s).w_rays[0].ensure((size_t)max_nodes * 32)) return rc;
      if (int rc = ws(s).w_rays[1].ensure((size_t)max_nodes * 32)) return rc;
      if (int rc = ws(s).w_hits[1].ensure((size_t)max_nodes * 8)) return rc;
      if (max_gen > 1) { if (int rc = ws(s).w_hits[2].ensure((size_t)max_nodes * 8)) return rc; }
    }
  }
  // counters: per generation g, [4g] = #parents, [4g+1] = #nodes of generation g+1, [4g+2] = #refraction rays spawned
  if (int rc = ws(s).w_counters.ensure((size_t)(max_gen + 2) * 16 + 16)) return rc;
  HIP_TRY(hipMemsetAsync(ws(s).w_counters.p, 0, (size_t)(max_gen + 2) * 16, ws(s).stream));
  uint32_t* cnt = ws(s).w_counters.as<uint32_t>();
  uint32_t* n0_dev = cnt + (size_t)(max_gen + 2) * 4;       // device copy of n0 for the queue kernel of caller rays
  if (!pr.camera) HIP_TRY(hipMemcpyAsync(n0_dev, &pr.n0, 4, hipMemcpyHostToDevice, ws(s).stream));

  span_begin(s, true);
  if (pr.camera) {
    cur_call(s).camera_span = cur_call(s).trace_spans.size() - 1;
    if (int rc = launch_trace_camera(s, pr.C, pr.tiles, ws(s).w_hits[0].as<uint2>())) return rc;
  } else {
    if (int rc = launch_trace_queue(s, ws(s).w_rays[0].as<float4>(), n0_dev, 1u, pr.n0, ws(s).w_hits[0].as<uint2>())) return rc;
  }
  span_end(s, true);
  if (pr.camera) { if (int rc = guard_camera(s, pr.C, pr.cams, pr.nframes, pr.n0, ws(s).w_hits[0].as<uint2>())) return rc; }
  else if (int rc = guard_queue(s, ws(s).w_rays[0].as<float4>(), n0_dev, pr.n0, 0u, ws(s).w_hits[0].as<uint2>())) return rc;

  const float th_gi = G ? (float)(M_PI * 2 / (double)G) : 0.0f;      // Ray::resendRay: float theta = M_PI * 2 / n
  const float th_sl = (float)(M_PI * 2 / 4);                          // Shader::getNPointsOnCircle with n = 4
  uint64_t bound = pr.n0;                                             // host-side upper bound of the node count
  // (not for calls that run beside other calls: the GPU is full then, and the extra stream only costs -- hall -2.5 %)
  const bool overlap = max_gen > 0 && shade && !cur_call(s).throughput && overlap_launches(s);
  for (int g = 0; g <= max_gen; ++g) {
    const int cur = g & 1, nxt = cur ^ 1;
    PipeArgs A;
    A.S = s->dev; A.C = pr.C; A.bounce = pr.bounce0 + g; A.from_camera = (pr.camera && g == 0) ? 1 : 0; A.first_gen = g == 0;
    A.max_bounces = opt->max_bounces; A.reflection = opt->reflection; A.refraction = opt->refraction; A.gi_rays = (int)G;
    A.shadows = opt->shadows; A.skip_zero_weight = opt->skip_zero_weight; A.atomic_accum = tree ? 1 : 0;
    A.n0 = pr.n0;
    A.nrays = g == 0 ? nullptr : cnt + 4 * (g - 1) + 1;
    A.rays = (pr.camera && g == 0) ? nullptr : ws(s).w_rays[cur].as<float4>();
    // hits of generation g: [0] for the first, then [1] and [2] in turn (the trace of generation g+1 may run while
    // k_resolve still reads those of g)
    A.hits = ws(s).w_hits[g == 0 ? 0 : 1 + ((g - 1) & 1)].as<uint2>();
    uint2* hits_next = ws(s).w_hits[1 + (g & 1)].as<uint2>();
    A.rgb = rgb; A.out_t = t; A.out_face = face;
    A.parents = ws(s).w_parents.as<uint32_t>();
    A.shadow_rays = ws(s).w_shadow_rays.as<float4>(); A.shadow_hits = ws(s).w_shadow_hits.as<uint2>();
    A.gi_q = ws(s).w_gi_rays.as<float4>(); A.gi_hits = ws(s).w_gi_hits.as<uint2>();
    A.next_rays = ws(s).w_rays[nxt].as<float4>();
    A.counters = cnt + 4 * g;
    A.acc = ws(s).w_acc.as<long long>(); A.acc_flag = ws(s).w_acc_flag.as<uint32_t>();
    A.gi_sin = sinf(th_gi); A.gi_cos = cosf(th_gi); A.sl_sin = sinf(th_sl); A.sl_cos = cosf(th_sl);
    if (direct) {
      if (opt->shadows && NLS) {
        hipLaunchKernelGGL(k_store_camera, dim3(1), dim3(64), 0, ws(s).stream, pr.C, ws(s).w_camera.as<DevCamera>());
        span_begin(s, true);
        if (int rc = launch_trace_shadow_direct(s, ws(s).w_camera.as<DevCamera>(), A.hits, pr.n0, NLS, ws(s).w_shadow_hits.as<uint2>())) return rc;
        span_end(s, true);
      }
      hipLaunchKernelGGL(k_resolve_direct, dim3(grid_for(s, pr.n0)), dim3(kBlock), 0, ws(s).stream, A);
      break;
    }
    hipLaunchKernelGGL(k_shade, dim3(grid_for(s, bound)), dim3(kBlock), 0, ws(s).stream, A);
    if (!shade) break;
    // The bounce rays of generation g+1 and the shadow rays of g both come out of k_shade and do not depend on
    // each other: with overlap on, the bounce launch goes to stream2 and fills the CUs that the shadow launch
    // leaves as its last waves drain (and the other way round); the main stream takes the result up again
    // before the next k_shade.
    const uint64_t bound_next = tree ? bound * 2 : bound;
    hipEvent_t side_done = nullptr;
    if (overlap && g < max_gen) {
      rtx_scene::Call& cc = cur_call(s);
      hipEvent_t shaded = cc.events.get();
      side_done = cc.events.get();
      HIP_TRY(hipEventRecord(shaded, ws(s).stream));
      HIP_TRY(hipStreamWaitEvent(ws(s).stream2, shaded, 0));
      span_begin(s, true, ws(s).stream2);
      if (int rc = launch_trace_queue(s, ws(s).w_rays[nxt].as<float4>(), cnt + 4 * g + 1, 1u, bound_next, hits_next, false, false, true)) return rc;
      span_end(s, true, ws(s).stream2);
      HIP_TRY(hipEventRecord(side_done, ws(s).stream2));
    }
    if (opt->shadows && NLS) {
      span_begin(s, true);
      if (int rc = launch_trace_queue(s, ws(s).w_shadow_rays.as<float4>(), cnt + 4 * g, NLS, bound, ws(s).w_shadow_hits.as<uint2>(), true, true, false, true)) return rc;
      span_end(s, true);
    }
    if (G) {
      span_begin(s, true);
      hipLaunchKernelGGL(k_trace_gi, dim3(grid_for(s, bound * G)), dim3(kBlock), 0, ws(s).stream, s->dev, ws(s).w_gi_rays.as<float4>(),
                         cnt + 4 * g, G, ws(s).w_gi_hits.as<uint2>());
      span_end(s, true);
    }
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(s, bound)), dim3(kBlock), 0, ws(s).stream, A);
    if (tree && g == max_gen) {          // the last generation is in: fixed-point sums -> the caller's pixels
      PipeArgs F = A;
      F.from_camera = pr.camera ? 1 : 0;
      hipLaunchKernelGGL(k_acc_to_rgb, dim3(grid_for(s, pr.n0)), dim3(kBlock), 0, ws(s).stream, F);
    }
    if (g < max_gen) {
      bound = bound_next;
      if (side_done) {
        HIP_TRY(hipStreamWaitEvent(ws(s).stream, side_done, 0));
      } else {
        span_begin(s, true);
        if (int rc = launch_trace_queue(s, ws(s).w_rays[nxt].as<float4>(), cnt + 4 * g + 1, 1u, bound, hits_next)) return rc;
        span_end(s, true);
      }
    }
  }
  // the counters go to pinned host memory behind the kernels; finish_call() turns them into ray counts
  rtx_scene::Call& c = cur_call(s);
  const size_t words = (size_t)(max_gen + 2) * 4;
  if (c.hc_words < words) {
    if (c.hc) (void)hipHostFree(c.hc);
    c.hc = nullptr; c.hc_words = 0;
    HIP_TRY(hipHostMalloc((void**)&c.hc, (words + 64) * 4, hipHostMallocDefault));
    c.hc_words = words + 64;
  }
  HIP_TRY(hipMemcpyAsync(c.hc, ws(s).w_counters.p, words * 4, hipMemcpyDeviceToHost, ws(s).stream));
  c.pipeline = true; c.shade = shade; c.shadows = opt->shadows != 0; c.max_gen = max_gen; c.NLS = NLS; c.G = G;
  return RTX_OK;
}

static int render_impl(rtx_scene* s, const DevCamera* cams, uint32_t nframes, uint32_t npix, uint32_t tiles, const rtx_options* opt,
                       float* rgb, float* t, int32_t* face, bool wait) {
  if (int rc = check_camera_domain(s, cams, nframes, npix)) return rc;
  if (int rc = stats_begin(s, npix)) return rc;
  cur_call(s).throughput = !wait && !cur_call(s).alone;
  cur_call(s).tunable = true;
  const DevCamera& C = cams[0];
  const size_t tab_cap = ws(s).w_cam_tab.cap, fr_cap = ws(s).w_frames.cap;
  const size_t tab_words = (size_t)(C.w + C.h);
  if (int rc = ws(s).w_cam_tab.ensure(tab_words * nframes * 4)) return rc;
  if (nframes > 1) { if (int rc = ws(s).w_frames.ensure((size_t)nframes * 64)) return rc; }
  if (ws(s).w_cam_tab.cap != tab_cap || ws(s).w_frames.cap != fr_cap) ws(s).cam_tab_ok = false;      // a new allocation (maybe at the old address)
  Primary pr; pr.camera = true; pr.C = C; pr.n0 = npix; pr.tiles = tiles; pr.bounce0 = 0;
  pr.C.tab = ws(s).w_cam_tab.as<float>();
  pr.C.frames = nframes > 1 ? ws(s).w_frames.as<float>() : nullptr;
  // the tables depend on the cameras and the rectangle only: frames of the same cameras reuse them
  std::vector<DevCamera>& held = ws(s).cam_tab_of;
  if (!ws(s).cam_tab_ok || held.size() != nframes || std::memcmp(held.data(), cams, sizeof(DevCamera) * nframes) != 0) {
    for (uint32_t f = 0; f < nframes; ++f)
      hipLaunchKernelGGL(k_camera_tables, dim3((uint32_t)(tab_words + 255) / 256), dim3(256), 0, ws(s).stream, cams[f],
                         ws(s).w_cam_tab.as<float>() + tab_words * f);
    for (uint32_t f0 = 0; nframes > 1 && f0 < nframes; f0 += kFramesPerStore) {
      FrameBatch B;
      std::memset(&B, 0, sizeof B);
      const uint32_t nf = std::min<uint32_t>(kFramesPerStore, nframes - f0);
      for (uint32_t f = 0; f < nf; ++f) {
        std::memcpy(B.rec[f], cams[f0 + f].inv34, 12 * sizeof(float));
        std::memcpy(B.rec[f] + 12, cams[f0 + f].origin, 3 * sizeof(float));
      }
      hipLaunchKernelGGL(k_store_frames, dim3(1), dim3(16 * kFramesPerStore), 0, ws(s).stream, B, nf, ws(s).w_frames.as<float>() + 16u * f0);
    }
    held.assign(cams, cams + nframes); ws(s).cam_tab_ok = true;
  }
  pr.cams = cams; pr.nframes = nframes;
  if (int rc = run_pipeline(s, pr, opt, rgb, t, face)) return rc;
  uint64_t live = C.frame_slots ? C.frame_slots : npix;
  if (C.ts) {   // partial edge tiles: count only slots that are pixels of the rectangle
    live = 0;
    const int tiles_y = (C.h + C.ts - 1) / C.ts;
    for (uint32_t k = 0; k < C.ntiles_local; ++k) {
      const uint32_t g = k * (uint32_t)C.nparts + (uint32_t)C.part;
      const int gx = (int)(g % (uint32_t)C.tiles_x), gy = (int)(g / (uint32_t)C.tiles_x);
      if (gy >= tiles_y) continue;
      live += (uint64_t)std::min(C.ts, C.w - gx * C.ts) * (uint64_t)std::min(C.ts, C.h - gy * C.ts);
    }
  }
  cur_call(s).stats.rays_primary = live * nframes;
  cur_call(s).tune = true;
  if (wait) return stats_end(s);
  stats_enqueue_end(s);
  cur_call(s).unreported = true;
  return RTX_OK;
}

// Flyscene::traceRay(bounce, Ray(o, d), insideObject) for n caller rays (src/flyscene.cpp:587-601).
extern "C" int rtx_trace_shade(rtx_scene* s, const rtx_options* opt, const float* o, const float* d, uint32_t n, int32_t bounce,
                               int32_t inside, float* rgb) {
  if (int rc = check_ready(s)) return rc;
  if (n == 0) return RTX_OK;
  if (!opt || !o || !d || !rgb) return fail(RTX_EINVAL, "null argument");
  if (n >= (1u << 31)) return fail(RTX_ELIMIT, "too many rays");
  if (bounce < 0) return fail(RTX_EINVAL, "bounce must be >= 0");
  rtx_camera dummy{};
  if (int rc = check_render_args(s, &dummy, opt, rgb)) return rc;
  const size_t b3 = (size_t)n * 12;
  const bool tree = opt->reflection && opt->refraction;
  const int max_gen = std::max(0, opt->max_bounces - bounce);
  const uint64_t max_nodes = tree ? ((uint64_t)n << max_gen) : n;
  // the limit first, so that an oversize request is RTX_ELIMIT rather than a failed allocation
  const uint64_t per_node = std::max<uint64_t>(std::max<uint64_t>(s->dev.nlights + 5u * s->dev.nsl, (uint64_t)std::max(opt->gi_rays, 0)), 1);
  if (max_nodes * per_node >= (1ull << 31)) return fail(RTX_ELIMIT, "ray queue would exceed 2^31 entries");
  if (int rc = stats_begin(s)) return rc;
  if (int rc = s->w_io[0].ensure(b3)) return rc;
  if (int rc = s->w_io[1].ensure(b3)) return rc;
  if (int rc = s->w_io[2].ensure(b3)) return rc;
  if (int rc = ws(s).w_rays[0].ensure((size_t)max_nodes * 32)) return rc;
  HIP_TRY(hipMemcpyAsync(s->w_io[0].p, o, b3, hipMemcpyHostToDevice, ws(s).stream));
  HIP_TRY(hipMemcpyAsync(s->w_io[1].p, d, b3, hipMemcpyHostToDevice, ws(s).stream));
  hipLaunchKernelGGL(k_make_nodes, dim3((n + 255) / 256), dim3(256), 0, ws(s).stream, s->w_io[0].as<float>(), s->w_io[1].as<float>(), n,
                     inside ? 1 : 0, ws(s).w_rays[0].as<float4>());
  Primary pr; pr.camera = false; std::memset(&pr.C, 0, sizeof pr.C); pr.n0 = n; pr.tiles = 0; pr.bounce0 = bounce;
  if (int rc = run_pipeline(s, pr, opt, s->w_io[2].as<float>(), nullptr, nullptr)) return rc;
  cur_call(s).stats.rays_primary = n;
  if (int rc = stats_end(s)) return rc;
  HIP_TRY(hipMemcpy(rgb, s->w_io[2].p, b3, hipMemcpyDeviceToHost));
  return RTX_OK;
}

// Camera::screenToWorld / getCenter for n raster positions (tucano/camera.hpp:115-118,155-173).
extern "C" int rtx_camera_points(rtx_scene* s, const rtx_camera* cam, const float* ij, uint32_t n, float* world, float* center) {
  if (int rc = check_ready(s)) return rc;
  if (!cam || (n && (!ij || !world))) return fail(RTX_EINVAL, "null argument");
  DevCamera C;
  rtx_rect r{0, 0, 1, 1};
  if (int rc = make_camera(cam, r, &C)) return rc;
  if (center) { center[0] = C.origin[0]; center[1] = C.origin[1]; center[2] = C.origin[2]; }
  if (n == 0) return RTX_OK;
  if (int rc = s->w_io[0].ensure((size_t)n * 8)) return rc;
  if (int rc = s->w_io[1].ensure((size_t)n * 12)) return rc;
  HIP_TRY(hipMemcpyAsync(s->w_io[0].p, ij, (size_t)n * 8, hipMemcpyHostToDevice, ws(s).stream));
  hipLaunchKernelGGL(k_camera_points, dim3((n + 255) / 256), dim3(256), 0, ws(s).stream, C, s->w_io[0].as<float>(), n, s->w_io[1].as<float>());
  HIP_TRY(hipStreamSynchronize(ws(s).stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(world, s->w_io[1].p, (size_t)n * 12, hipMemcpyDeviceToHost));
  return RTX_OK;
}

// Flyscene::intersectBox(box, origin, dest) (src/flyscene.cpp:906-972) for n rays against one box:
// t of Intersect::BoxObject::intersects (may be negative; INFINITY = miss).
extern "C" int rtx_intersect_box(rtx_scene* s, const float* bmin, const float* bmax, const float* o, const float* d, uint32_t n,
                                 float* t) {
  if (int rc = check_ready(s)) return rc;
  if (n == 0) return RTX_OK;
  if (!bmin || !bmax || !o || !d || !t) return fail(RTX_EINVAL, "null argument");
  const size_t b3 = (size_t)n * 12;
  if (int rc = s->w_io[0].ensure(b3)) return rc;
  if (int rc = s->w_io[1].ensure(b3)) return rc;
  if (int rc = s->w_io[2].ensure((size_t)n * 4)) return rc;
  HIP_TRY(hipMemcpyAsync(s->w_io[0].p, o, b3, hipMemcpyHostToDevice, ws(s).stream));
  HIP_TRY(hipMemcpyAsync(s->w_io[1].p, d, b3, hipMemcpyHostToDevice, ws(s).stream));
  hipLaunchKernelGGL(k_ref_box, dim3((n + 255) / 256), dim3(256), 0, ws(s).stream, s->w_io[0].as<float>(), s->w_io[1].as<float>(), n,
                     bmin[0], bmin[1], bmin[2], bmax[0], bmax[1], bmax[2], s->w_io[2].as<float>());
  HIP_TRY(hipStreamSynchronize(ws(s).stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(t, s->w_io[2].p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return RTX_OK;
}

// The hit search of Flyscene::ReflectDebugRay (src/flyscene.cpp:324-357) for n rays.
extern "C" int rtx_debug_hit(rtx_scene* s, const float* o, const float* d, uint32_t n, float* t, int32_t* face, float* normal,
                             float* point, uint8_t* box_flags) {
  if (int rc = check_ready(s)) return rc;
  if (!s->have_leaves) return fail(RTX_ESTATE, "the debug ray walks the reference leaf list (rtx_scene_build_reference_leafboxes) before commit");
  if (n == 0) return RTX_OK;
  if (!o || !d || !t || !face || !normal || !point) return fail(RTX_EINVAL, "null argument");
  const size_t b3 = (size_t)n * 12, nleaf = s->dev.nleafboxes;
  if (int rc = s->w_io[0].ensure(b3)) return rc;
  if (int rc = s->w_io[1].ensure(b3)) return rc;
  if (int rc = s->w_io[2].ensure((size_t)n * 4)) return rc;
  if (int rc = s->w_io[3].ensure((size_t)n * 4)) return rc;
  if (int rc = s->w_io[4].ensure(b3)) return rc;
  if (int rc = s->w_io[5].ensure(std::max<size_t>(1, (size_t)n * nleaf))) return rc;
  if (int rc = ws(s).w_rays[0].ensure((size_t)n * 32)) return rc;
  HIP_TRY(hipMemcpyAsync(s->w_io[0].p, o, b3, hipMemcpyHostToDevice, ws(s).stream));
  HIP_TRY(hipMemcpyAsync(s->w_io[1].p, d, b3, hipMemcpyHostToDevice, ws(s).stream));
  if (box_flags) HIP_TRY(hipMemsetAsync(s->w_io[5].p, 0, std::max<size_t>(1, (size_t)n * nleaf), ws(s).stream));
  hipLaunchKernelGGL(k_make_rays, dim3((n + 255) / 256), dim3(256), 0, ws(s).stream, s->w_io[0].as<float>(), s->w_io[1].as<float>(), n,
                     ws(s).w_rays[0].as<float4>());
  // outputs: t -> io[2], face -> io[3], normal -> io[0], point -> io[4] (the xyz inputs are consumed by k_make_rays first)
  hipLaunchKernelGGL(k_debug_hit, dim3((n + 255) / 256), dim3(256), 0, ws(s).stream, s->dev, ws(s).w_rays[0].as<float4>(), n,
                     s->w_io[2].as<float>(), s->w_io[3].as<int>(), s->w_io[0].as<float>(), s->w_io[4].as<float>(),
                     box_flags ? s->w_io[5].as<unsigned char>() : nullptr);
  HIP_TRY(hipStreamSynchronize(ws(s).stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(t, s->w_io[2].p, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(face, s->w_io[3].p, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(normal, s->w_io[0].p, b3, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(point, s->w_io[4].p, b3, hipMemcpyDeviceToHost));
  if (box_flags && nleaf) HIP_TRY(hipMemcpy(box_flags, s->w_io[5].p, (size_t)n * nleaf, hipMemcpyDeviceToHost));
  return RTX_OK;
}

extern "C" int rtx_render(rtx_scene* s, const rtx_camera* cam, const rtx_options* opt, rtx_rect rect, float* rgb, float* t,
                          int32_t* face) {
  if (int rc = check_ready(s)) return rc;
  if (rect.x1 <= rect.x0 || rect.y1 <= rect.y0) return fail(RTX_EINVAL, "empty rectangle");
  const size_t npix = (size_t)(rect.x1 - rect.x0) * (size_t)(rect.y1 - rect.y0);
  if (rgb) if (int rc = s->w_io[0].ensure(npix * 12)) return rc;
  if (t) if (int rc = s->w_io[1].ensure(npix * 4)) return rc;
  if (face) if (int rc = s->w_io[2].ensure(npix * 4)) return rc;
  int rc = rtx_render_dev(s, cam, opt, rect, rgb ? s->w_io[0].as<float>() : nullptr, t ? s->w_io[1].as<float>() : nullptr,
                          face ? s->w_io[2].as<int32_t>() : nullptr);
  if (rc) return rc;
  if (rgb) HIP_TRY(hipMemcpy(rgb, s->w_io[0].p, npix * 12, hipMemcpyDeviceToHost));
  if (t) HIP_TRY(hipMemcpy(t, s->w_io[1].p, npix * 4, hipMemcpyDeviceToHost));
  if (face) HIP_TRY(hipMemcpy(face, s->w_io[2].p, npix * 4, hipMemcpyDeviceToHost));
  return RTX_OK;
}

// Flyscene::trace_global_illum(Ray(o, d)) for n caller rays (src/flyscene.cpp:631-669): {Kd of the closest hit over the
// spheres and the reference's flat leaf list, t}; zeros when nothing is hit.
extern "C" int rtx_trace_gi(rtx_scene* s, const float* o, const float* d, uint32_t n, float* kd_t4) {
  if (int rc = check_ready(s)) return rc;
  if (!s->have_leaves) return fail(RTX_ESTATE, "trace_global_illum walks the reference leaf list (rtx_scene_build_reference_leafboxes) before commit");
  if (n == 0) return RTX_OK;
  if (!o || !d || !kd_t4) return fail(RTX_EINVAL, "null argument");
  if (n >= (1u << 31)) return fail(RTX_ELIMIT, "too many rays");
  if (int rc = stats_begin(s)) return rc;
  const size_t b3 = (size_t)n * 12;
  if (int rc = s->w_io[0].ensure(b3)) return rc;
  if (int rc = s->w_io[1].ensure(b3)) return rc;
  if (int rc = s->w_io[2].ensure((size_t)n * 8)) return rc;
  if (int rc = ws(s).w_rays[0].ensure((size_t)n * 32)) return rc;
  if (int rc = ws(s).w_counters.ensure(64)) return rc;
  HIP_TRY(hipMemcpyAsync(s->w_io[0].p, o, b3, hipMemcpyHostToDevice, ws(s).stream));
  HIP_TRY(hipMemcpyAsync(s->w_io[1].p, d, b3, hipMemcpyHostToDevice, ws(s).stream));
  HIP_TRY(hipMemcpyAsync(ws(s).w_counters.p, &n, 4, hipMemcpyHostToDevice, ws(s).stream));
  hipLaunchKernelGGL(k_make_rays, dim3((n + 255) / 256), dim3(256), 0, ws(s).stream, s->w_io[0].as<float>(), s->w_io[1].as<float>(), n,
                     ws(s).w_rays[0].as<float4>());
  span_begin(s, true);
  hipLaunchKernelGGL(k_trace_gi, dim3(grid_for(s, n)), dim3(kBlock), 0, ws(s).stream, s->dev, ws(s).w_rays[0].as<float4>(),
                     ws(s).w_counters.as<uint32_t>(), 1u, s->w_io[2].as<uint2>());
  span_end(s, true);
  cur_call(s).stats.rays_gi = n;
  if (int rc = stats_end(s)) return rc;
  std::vector<uint2> res(n);
  HIP_TRY(hipMemcpy(res.data(), s->w_io[2].p, (size_t)n * 8, hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n; ++i) {
    float* out = kd_t4 + 4 * (size_t)i;
    float t; std::memcpy(&t, &res[i].x, 4);
    if (res[i].x == 0u) { out[0] = out[1] = out[2] = out[3] = 0.0f; continue; }
    const int m = (int)res[i].y;
    // materials[-1] is out of bounds in the reference; the library's convention everywhere: the default Mtl
    const float dflt[3] = {0.5f, 0.5f, 0.5f};
    const float* kd = (m >= 0 && m < (int)s->mats.size()) ? s->mats[m].kd : dflt;
    out[0] = kd[0]; out[1] = kd[1]; out[2] = kd[2]; out[3] = t;
  }
  return RTX_OK;
}

// Intersect::{Triangle, Plane, BoxObject, Sphere}::intersects(Ray&) (src/Intersect.h:8-89, src/Intersect.cpp:6-193):
// n Ray objects against one primitive, no scene needed.  Host pointers; a one-off query (it allocates and frees
// its own device buffers).
extern "C" int rtx_intersect_primitive(int device, int32_t kind, const float* prim, const float* o, const float* d, uint32_t n,
                                       float* t, int32_t* face_id, float* normal, float* point) {
  if (kind < RTX_PRIM_TRIANGLE || kind > RTX_PRIM_SPHERE) return fail(RTX_EINVAL, "unknown primitive kind %d", kind);
  if (n == 0) return RTX_OK;
  if (!prim || !o || !d || !t) return fail(RTX_EINVAL, "null argument");
  if (n >= (1u << 31)) return fail(RTX_ELIMIT, "too many rays");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(RTX_EINVAL, "device %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  const int nprim = kind == RTX_PRIM_TRIANGLE ? 12 : kind == RTX_PRIM_SPHERE ? 4 : 6;
  const size_t b3 = (size_t)n * 12;
  DevBuf buf;                                     // prim | o | d | t | face | normal | point
  const size_t off_o = 256, off_d = off_o + b3, off_t = off_d + b3, off_f = off_t + (size_t)n * 4, off_n = off_f + (size_t)n * 4,
               off_p = off_n + b3;
  if (int rc = buf.ensure(off_p + b3)) return rc;
  char* base = buf.as<char>();
  int rc = RTX_OK;
  auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == RTX_OK) rc = fail(RTX_EHIP, "%s: %s", what, hipGetErrorString(e)); };
  chk(hipMemcpy(base, prim, (size_t)nprim * 4, hipMemcpyHostToDevice), "hipMemcpy(prim)");
  chk(hipMemcpy(base + off_o, o, b3, hipMemcpyHostToDevice), "hipMemcpy(o)");
  chk(hipMemcpy(base + off_d, d, b3, hipMemcpyHostToDevice), "hipMemcpy(d)");
  if (rc == RTX_OK) {
    hipLaunchKernelGGL(k_primitive, dim3((n + 255) / 256), dim3(256), 0, nullptr, (int)kind, (const float*)base, (const float*)(base + off_o),
                       (const float*)(base + off_d), n, (float*)(base + off_t), (int*)(base + off_f), (float*)(base + off_n),
                       (float*)(base + off_p));
    chk(hipDeviceSynchronize(), "k_primitive");
    chk(hipGetLastError(), "k_primitive");
  }
  if (rc == RTX_OK) chk(hipMemcpy(t, base + off_t, (size_t)n * 4, hipMemcpyDeviceToHost), "hipMemcpy(t)");
  if (rc == RTX_OK && face_id) chk(hipMemcpy(face_id, base + off_f, (size_t)n * 4, hipMemcpyDeviceToHost), "hipMemcpy(face)");
  if (rc == RTX_OK && normal) chk(hipMemcpy(normal, base + off_n, b3, hipMemcpyDeviceToHost), "hipMemcpy(normal)");
  if (rc == RTX_OK && point) chk(hipMemcpy(point, base + off_p, b3, hipMemcpyDeviceToHost), "hipMemcpy(point)");
  buf.release();
  return rc;
}

// ------------------------------------------------------------------ RCCL, loaded on demand
// rtx_render_multi moves the tile parts of the other GPUs to the first scene's device with grouped ncclSend / ncclRecv
// over xGMI (north star: "tiles gathered over xGMI with RCCL").  librccl is opened with dlopen the first time two
// handles of one call sit on different devices -- not linked: a process that also holds PyTorch already has a copy of
// the library loaded (torch ships its own), and the one dlopen finds by SONAME is that same copy, where a link-time
// dependency would bring /opt/rocm's second one into the process.  Where the library or a communicator cannot be had
// (or RTX_MULTI_GATHER=peer is set) the parts travel by hipMemcpyPeerAsync as before; results are identical.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <map>
#include <mutex>
namespace {
struct Rccl {
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
  std::string why;
};
std::mutex g_rccl_mu;
Rccl* rccl_api() {                     // (callers hold g_rccl_mu)
  static Rccl R;
  static bool tried = false;
  if (tried) return &R;
  tried = true;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
  if (!h) { R.why = std::string("dlopen(librccl): ") + (dlerror() ? dlerror() : "not found"); return &R; }
  auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p && R.why.empty()) R.why = std::string("librccl lacks ") + n; return p; };
  R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
  R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
  R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
  R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
  R.Send = (decltype(R.Send))sym("ncclSend");
  R.Recv = (decltype(R.Recv))sym("ncclRecv");
  R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
  R.ok = R.why.empty();
  return &R;
}
// One communicator per set of devices (rank r of the set = its r-th device in ascending order), kept for the life of
// the process: creating one costs hundreds of milliseconds.
struct CommSet { std::vector<int> devs; std::vector<ncclComm_t> comms; };
std::map<std::vector<int>, CommSet> g_comm_sets;
CommSet* rccl_comms(const std::vector<int>& devs, std::string* why) {      // (callers hold g_rccl_mu)
  Rccl* R = rccl_api();
  if (!R->ok) { *why = R->why; return nullptr; }
  auto it = g_comm_sets.find(devs);
  if (it != g_comm_sets.end()) return it->second.comms.empty() ? nullptr : &it->second;
  CommSet cs; cs.devs = devs; cs.comms.assign(devs.size(), nullptr);
  const ncclResult_t rc = R->CommInitAll(cs.comms.data(), (int)devs.size(), devs.data());
  if (rc != ncclSuccess) { *why = std::string("ncclCommInitAll: ") + R->GetErrorString(rc); cs.comms.clear(); }
  CommSet* out = &(g_comm_sets[devs] = cs);
  return out->comms.empty() ? nullptr : out;
}
bool multi_wants_rccl() {
  static const bool peer = [] { const char* e = std::getenv("RTX_MULTI_GATHER"); return e && std::strcmp(e, "peer") == 0; }();
  return !peer;
}
// (test hook) RTX_MULTI_GATHER=rccl-self: handles that share the first scene's device render into buffers of their own
// and their parts go through a ncclSend / ncclRecv pair of that device's rank to itself -- the RCCL path (dlopen,
// communicator, grouped calls, stream order) exercised on a box with one GPU.
bool multi_self_test() {
  static const bool self = [] { const char* e = std::getenv("RTX_MULTI_GATHER"); return e && std::strcmp(e, "rccl-self") == 0; }();
  return self;
}
}  // namespace

// ------------------------------------------------------------------ several GPUs, one process
// A copy of a scene's host description and host-built acceleration data (no device state): what a caller commits
// on another device without flattening the mesh and building the BVH again.
extern "C" int rtx_scene_clone(const rtx_scene* src, rtx_scene** out) {
  if (!src || !out) return fail(RTX_EINVAL, "null argument");
  if (int rc = fetch_host_mirror(const_cast<rtx_scene*>(src))) return rc;      // (what the GPU built or refitted travels as its host copy)
  rtx_scene* s = new rtx_scene();
  s->verts = src->verts; s->idx = src->idx; s->tri_mat = src->tri_mat; s->mats = src->mats;
  s->spheres = src->spheres; s->sph_mat = src->sph_mat; s->lights = src->lights; s->slights = src->slights;
  s->leaf_lo = src->leaf_lo; s->leaf_hi = src->leaf_hi; s->leaf_start = src->leaf_start; s->leaf_count = src->leaf_count;
  s->leaf_faces = src->leaf_faces; s->have_leaves = src->have_leaves; s->flags = src->flags;
  s->built = src->built; s->h_tri_leaf = src->h_tri_leaf; s->h_face_nm = src->h_face_nm; s->h_tri_face = src->h_tri_face;
  std::memcpy(s->root_lo, src->root_lo, sizeof s->root_lo); std::memcpy(s->root_hi, src->root_hi, sizeof s->root_hi);
  s->nwild = src->nwild; s->bvh = src->bvh; s->info = src->info; s->domain = src->domain;
  s->device_build_ms = src->device_build_ms; s->refit_ms = src->refit_ms;
  *out = s;
  return RTX_OK;
}

// Flyscene::raytraceScene's decomposition (src/flyscene.cpp:417-430,460: rows strided over 16 threads) over the
// GPUs of one process: scene i renders part i of nscenes of the tile grid on its own device, all at once (each
// scene has its own streams); the compact parts of the other devices travel to the first scene's device over xGMI --
// grouped ncclSend / ncclRecv (RCCL, above), peer copies where RCCL is not to be had; handles that share the first
// scene's device render straight into the gather buffer -- and are put back into row-major order there.
static int render_multi(rtx_scene* const* scenes, int32_t n, const rtx_camera* cam, const rtx_options* opt, rtx_rect rect,
                        int32_t tile_size, float* rgb, float* t, int32_t* face, bool host_out) {
  if (!scenes || n < 1) return fail(RTX_EINVAL, "need at least one scene");
  if (n > 64) return fail(RTX_ELIMIT, "at most 64 scenes");
  for (int i = 0; i < n; ++i) {
    if (int rc = check_ready(scenes[i])) return rc;
    for (int j = 0; j < i; ++j) if (scenes[j] == scenes[i]) return fail(RTX_EINVAL, "scene %d is scene %d: every part needs a scene handle of its own (rtx_scene_clone)", i, j);
  }
  const int kmax = rtx_tiles_count(rect, tile_size, 0, n);
  if (kmax < 0) return kmax;
  const uint64_t stride = (uint64_t)kmax * (uint64_t)tile_size * (uint64_t)tile_size;      // pixels per part
  if (stride >= (1ull << 31)) return fail(RTX_ELIMIT, "too many pixels in a part");
  void* outs[3] = {rgb, t, face};
  const int words[3] = {3, 1, 1};
  rtx_scene* s0 = scenes[0];
  const size_t npix = (size_t)(rect.x1 - rect.x0) * (size_t)(rect.y1 - rect.y0);
  // buffers: part i of every wanted output on scene i's device; all parts on the first scene's device
  HIP_TRY(hipSetDevice(s0->device));
  for (int c = 0; c < 3; ++c) {
    if (!outs[c]) continue;
    if (int rc = s0->m_gather[c].ensure((size_t)n * stride * words[c] * 4)) return rc;
    if (host_out) if (int rc = s0->w_io[c].ensure(npix * words[c] * 4)) return rc;
  }
  // scene 0 (and any scene on its device) renders straight into its slot of the gather buffer; the others into a
  // buffer on their own device, which is then moved
  const bool self_test = multi_self_test();
  auto moves = [&](int i) { return i > 0 && (scenes[i]->device != s0->device || self_test); };
  for (int i = 1; i < n; ++i) {
    if (!moves(i)) continue;
    HIP_TRY(hipSetDevice(scenes[i]->device));
    for (int c = 0; c < 3; ++c) if (outs[c]) if (int rc = scenes[i]->m_part[c].ensure(stride * words[c] * 4)) return rc;
  }
  auto part_ptr = [&](int i, int c) -> char* {
    if (!outs[c]) return nullptr;
    if (!moves(i)) return s0->m_gather[c].as<char>() + (size_t)i * stride * words[c] * 4;
    return scenes[i]->m_part[c].as<char>();
  };
  // every part at once: the calls only queue work
  int first_err = RTX_OK;
  int queued = 0;
  int slot_of[64];                                     // the call slot part i went to on its scene
  for (int i = 0; i < n && first_err == RTX_OK; ++i) {
    first_err = render_tiles_impl(scenes[i], cam, 1, opt, rect, tile_size, i, n, (float*)part_ptr(i, 0), (float*)part_ptr(i, 1),
                                  (int32_t*)part_ptr(i, 2), false);
    if (first_err == RTX_OK) { slot_of[i] = scenes[i]->cur; ++queued; }
  }
  std::string err = first_err ? g_err : std::string();
  // Wait for exactly the calls queued above (also after an error: nothing of this function stays in flight).  Not
  // rtx_render_finish: that hands out the OLDEST unreported call of a handle, which is the caller's own when the caller
  // still has an enqueued render on it -- the part would then be copied before it was rendered, and the caller's
  // enqueue / finish pairs would be off by one from here on.
  for (int i = 0; i < queued; ++i) {
    rtx_scene::Call& c = scenes[i]->calls[slot_of[i]];
    int rc = RTX_OK;
    if (hipSetDevice(scenes[i]->device) != hipSuccess) rc = fail(RTX_EHIP, "hipSetDevice(%d)", scenes[i]->device);
    else rc = finish_call(scenes[i], c);
    c.unreported = false;
    if (rc && !first_err) { first_err = rc; err = g_err; }
  }
  if (first_err) { g_err = err; return first_err; }
  HIP_TRY(hipSetDevice(s0->device));
  hipStream_t st = scenes[0]->slots[slot_of[0]].stream;
  // ---- the parts of the other devices -> the first scene's device: RCCL send / recv pairs in one group (every part on
  // its own link into device 0), else peer copies
  bool any_move = false;
  for (int i = 1; i < n; ++i) any_move = any_move || moves(i);
  bool moved = false;
  if (any_move && multi_wants_rccl()) {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    std::vector<int> devs;
    for (int i = 0; i < n; ++i) devs.push_back(scenes[i]->device);
    std::sort(devs.begin(), devs.end());
    devs.erase(std::unique(devs.begin(), devs.end()), devs.end());
    std::string why;
    CommSet* cs = rccl_comms(devs, &why);
    if (cs) {
      Rccl* R = rccl_api();
      auto rank_of = [&](int dev) { return (int)(std::lower_bound(devs.begin(), devs.end(), dev) - devs.begin()); };
      const int root = rank_of(s0->device);
      // one stream per communicator: the root's is `st`; a sending device's is the stream of its first handle (the parts
      // are finished -- the host has waited for them above -- so no stream has to wait for another, and the final
      // synchronise of `st` below covers the senders: a part that has arrived has been read)
      std::vector<hipStream_t> stream_of(devs.size(), nullptr);
      stream_of[root] = st;
      for (int i = 1; i < n; ++i) { const int r = rank_of(scenes[i]->device); if (!stream_of[r]) stream_of[r] = scenes[i]->slots[slot_of[i]].stream; }
      ncclResult_t rc = R->GroupStart();
      for (int i = 1; i < n && rc == ncclSuccess; ++i) {
        if (!moves(i)) continue;
        const int r = rank_of(scenes[i]->device);
        for (int c = 0; c < 3 && rc == ncclSuccess; ++c) {
          if (!outs[c]) continue;
          const size_t bytes = (size_t)rtx_tiles_count(rect, tile_size, i, n) * tile_size * tile_size * words[c] * 4;
          if (!bytes) continue;
          rc = R->Send(scenes[i]->m_part[c].p, bytes, ncclChar, root, cs->comms[r], stream_of[r]);
          if (rc == ncclSuccess)
            rc = R->Recv(s0->m_gather[c].as<char>() + (size_t)i * stride * words[c] * 4, bytes, ncclChar, r, cs->comms[root], st);
        }
      }
      const ncclResult_t rc_end = R->GroupEnd();
      if (rc == ncclSuccess) rc = rc_end;
      if (rc != ncclSuccess) return fail(RTX_EHIP, "RCCL gather of the tile parts: %s", R->GetErrorString(rc));
      HIP_TRY(hipSetDevice(s0->device));
      moved = true;
    } else {
      static bool told = false;
      if (!told) { told = true; std::fprintf(stderr, "[rtx] rtx_render_multi: no RCCL (%s); the tile parts travel by peer copies\n", why.c_str()); }
    }
  }
  for (int i = 1; i < n && !moved; ++i) {
    if (!moves(i)) continue;
    for (int c = 0; c < 3; ++c) {
      if (!outs[c]) continue;
      const int nt = rtx_tiles_count(rect, tile_size, i, n);
      const size_t bytes = (size_t)nt * tile_size * tile_size * words[c] * 4;
      if (bytes) HIP_TRY(hipMemcpyPeerAsync(s0->m_gather[c].as<char>() + (size_t)i * stride * words[c] * 4, s0->device, scenes[i]->m_part[c].p,
                                            scenes[i]->device, bytes, st));
    }
  }
  for (int c = 0; c < 3; ++c) {
    if (!outs[c]) continue;
    void* dst = host_out ? s0->w_io[c].p : outs[c];
    if (int rc = rtx_tiles_assemble_dev(s0->m_gather[c].p, n, stride, rect, tile_size, words[c], dst, (void*)st)) return rc;
  }
  HIP_TRY(hipStreamSynchronize(st));
  if (host_out)
    for (int c = 0; c < 3; ++c) if (outs[c]) HIP_TRY(hipMemcpy(outs[c], s0->w_io[c].p, npix * words[c] * 4, hipMemcpyDeviceToHost));
  return RTX_OK;
}
extern "C" int rtx_render_multi(rtx_scene* const* scenes, int32_t nscenes, const rtx_camera* cam, const rtx_options* opt, rtx_rect rect,
                                int32_t tile_size, float* rgb, float* t, int32_t* face) {
  return render_multi(scenes, nscenes, cam, opt, rect, tile_size, rgb, t, face, true);
}
extern "C" int rtx_render_multi_dev(rtx_scene* const* scenes, int32_t nscenes, const rtx_camera* cam, const rtx_options* opt, rtx_rect rect,
                                    int32_t tile_size, float* rgb, float* t, int32_t* face) {
  return render_multi(scenes, nscenes, cam, opt, rect, tile_size, rgb, t, face, false);
}

// ------------------------------------------------------------------ the entry point that allocates nothing
// rtx_trace_closest_dev for callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that
// forbid hipMalloc): the same kernels, with the ray records, the hit records, the ray count and the overflow
// columns of the traversal stacks in a file-scope device array instead of the scene's growable workspace.
// Every GPU the library runs on holds its own copy of the array until the process ends; g_scratch_ptr[device]
// is that copy's device address (hipGetSymbolAddress at rtx_scene_commit, with the device current) -- the
// array's name is only a host-side stand-in and is never handed to a kernel or a memset.
constexpr uint32_t kNoAllocMaxRays = RTX_NOALLOC_MAX_RAYS;
constexpr uint32_t kNoAllocBlocks = 256;                                  // grid cap: sizes the overflow columns
constexpr size_t kScratchRaysOff = 0;                                     // 32 B per ray
constexpr size_t kScratchHitsOff = kScratchRaysOff + (size_t)kNoAllocMaxRays * 32;     // 8 B per ray
constexpr size_t kScratchCountOff = kScratchHitsOff + (size_t)kNoAllocMaxRays * 8;     // the ray count the queue kernel reads
constexpr size_t kScratchOvfOff = kScratchCountOff + 256;
constexpr size_t kScratchBytes = kScratchOvfOff + (size_t)kNoAllocBlocks * kBlock * (kStackDepth - kLdsStack) * 4;   // ~11.7 MB
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kMaxDevices = 64;
static unsigned char* g_scratch_ptr[kMaxDevices] = {};

// Called by rtx_scene_commit with `device` current: once per device.
static int scratch_init(int device) {
  if (device < 0 || device >= kMaxDevices) return fail(RTX_ELIMIT, "device index %d beyond %d", device, kMaxDevices);
  if (g_scratch_ptr[device]) return RTX_OK;
  void* p = nullptr;
  HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
  g_scratch_ptr[device] = (unsigned char*)p;
  return RTX_OK;
}

__global__ void k_store_u32(uint32_t* p, uint32_t v) { if (blockIdx.x == 0 && threadIdx.x == 0) *p = v; }

extern "C" int rtx_trace_closest_noalloc(rtx_scene* s, const float* o, const float* d, uint32_t n, float* t, int32_t* face) {
  if (int rc = check_ready(s)) return rc;
  if (n == 0) return RTX_OK;
  if (!o || !d) return fail(RTX_EINVAL, "null ray arrays");
  if (n > kNoAllocMaxRays)
    return fail(RTX_ELIMIT, "rtx_trace_closest_noalloc takes at most %u rays per call (got %u): its scratch buffer is a fixed device array", kNoAllocMaxRays, n);
  unsigned char* base = g_scratch_ptr[s->device];
  if (!base) return fail(RTX_ESTATE, "no scratch address for device %d (rtx_scene_commit looks it up)", s->device);
  float4* rays = (float4*)(base + kScratchRaysOff);
  uint2* hits = (uint2*)(base + kScratchHitsOff);
  uint32_t* count = (uint32_t*)(base + kScratchCountOff);
  uint32_t* ovf = (uint32_t*)(base + kScratchOvfOff);
  // Nothing here needs to start at zero: the count and the records of rays [0, n) are written before they are read,
  // every ray writes its hit record, and a stack column is written before it is read.  What the last call left
  // beyond n is never looked at.
  hipStream_t st = ws(s).stream;
  const uint32_t blocks = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_store_u32, dim3(1), dim3(64), 0, st, count, n);
  hipLaunchKernelGGL(k_make_rays, dim3(blocks), dim3(256), 0, st, o, d, n, rays);
  const uint32_t grid = std::min<uint32_t>(blocks, kNoAllocBlocks);
  // the scheduler: what the scene's flags force, else the heuristic's pick for queued rays (no tuner trial here)
  const bool fast = (s->flags & RTX_FLAG_FAST_TRIANGLES) != 0;
  const uint32_t tri_min = (s->flags & RTX_FLAG_SCHED_SPLIT) ? kTriMinSplit : 1u;
  if (s->flags & RTX_FLAG_SCHED_GROUPS) {
    if (fast) hipLaunchKernelGGL((k_trace_queue<false, FastTri>), dim3(grid), dim3(kBlock), 0, st, s->dev, rays, count, 1u, 0u, grab_size(), 1u, ovf, hits);
    else hipLaunchKernelGGL((k_trace_queue<false, RefTri>), dim3(grid), dim3(kBlock), 0, st, s->dev, rays, count, 1u, 0u, grab_size(), 1u, ovf, hits);
  } else {
    if (fast) hipLaunchKernelGGL((k_trace_queue<true, FastTri>), dim3(grid), dim3(kBlock), 0, st, s->dev, rays, count, 1u, 0u, grab_size(), tri_min, ovf, hits);
    else hipLaunchKernelGGL((k_trace_queue<true, RefTri>), dim3(grid), dim3(kBlock), 0, st, s->dev, rays, count, 1u, 0u, grab_size(), tri_min, ovf, hits);
  }
  // the domain guard as in rtx_trace_closest_dev (its counters, never read back, live in the scratch array too)
  if (s->dev.ntris != 0 || s->dev.nspheres != 0)
    hipLaunchKernelGGL(k_guard_queue, dim3(std::min<uint32_t>(n, kNoAllocBlocks * 8u)), dim3(kBlock), 0, st, s->dev, (const float4*)rays, (const uint32_t*)count, 1u, 0u,
                       s->domain, 0u, hits, count + 16);
  hipLaunchKernelGGL(k_unpack_hits, dim3(blocks), dim3(256), 0, st, s->dev, rays, hits, n, t, face, (float*)nullptr, (float*)nullptr,
                     (int*)nullptr);
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return RTX_OK;
}

extern "C" int rtx_get_stats(const rtx_scene* s, rtx_stats* out) {
  if (!s || !out) return fail(RTX_EINVAL, "null argument");
  *out = s->stats;
  return RTX_OK;
}
