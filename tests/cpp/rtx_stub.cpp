#include <cstdlib>
// A CPU stand-in for librtx's C ABI (include/rtx.h) -- TEST INFRASTRUCTURE for tests/cpp/facade_threads.cpp only, never
// shipped, never linked into anything that claims parity.  It lets the facade's threading be exercised under
// ThreadSanitizer on a machine without a GPU: every entry point that takes an rtx_scene touches a plain (non-atomic)
// field of it on the way in and out, so two calls that overlap on one handle -- which include/rtx.h forbids -- are a data
// race TSan reports.  Results are simple functions of the inputs, so that the test can tell whose answer it got.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtx.h"

struct rtx_scene {
  int in_call = 0;          // plain on purpose (see above)
  long calls = 0;
  int renders = 0, commits = 0;
  uint32_t ntris = 0;
  bool committed = false;
};
namespace {
struct Enter { rtx_scene* s; explicit Enter(rtx_scene* s_) : s(s_) { s->in_call = s->in_call + 1; s->calls = s->calls + 1; } ~Enter() { s->in_call = s->in_call - 1; } };
thread_local std::string g_err;
}
extern "C" {
const char* rtx_last_error(void) { return g_err.c_str(); }
int rtx_device_count(void) { return 1; }
int rtx_host_alloc(uint64_t bytes, void** out) { *out = std::malloc(bytes ? bytes : 1); return *out ? 0 : -1; }
int rtx_host_free(void* p) { std::free(p); return 0; }
int rtx_scene_create(const float*, uint32_t, const uint32_t*, const int32_t*, uint32_t ntris, const rtx_material*, uint32_t, rtx_scene** out) {
  *out = new rtx_scene(); (*out)->ntris = ntris; return 0;
}
int rtx_scene_destroy(rtx_scene* s) { delete s; return 0; }
int rtx_scene_add_material(rtx_scene* s, const rtx_material*) { Enter e(s); s->committed = false; return 0; }
int rtx_scene_set_spheres(rtx_scene* s, const float*, const int32_t*, uint32_t) { Enter e(s); s->committed = false; return 0; }
int rtx_scene_set_lights(rtx_scene* s, const float*, uint32_t, const float*, uint32_t) { Enter e(s); s->committed = false; return 0; }
int rtx_scene_set_flags(rtx_scene* s, uint32_t) { Enter e(s); return 0; }
int rtx_scene_build_reference_leafboxes(rtx_scene* s, int32_t, int64_t) { Enter e(s); return 1; }
int rtx_scene_get_leafboxes(const rtx_scene*, float* lo, float* hi, uint32_t* st, uint32_t* ct, int32_t* faces, uint32_t* nl, uint32_t* nf) {
  if (nl) *nl = 1; if (nf) *nf = 1;
  if (lo) { lo[0] = lo[1] = lo[2] = -1; } if (hi) { hi[0] = hi[1] = hi[2] = 1; }
  if (st) st[0] = 0; if (ct) ct[0] = 1; if (faces) faces[0] = 0;
  return 0;
}
int rtx_scene_build(rtx_scene* s) { Enter e(s); return 0; }
int rtx_scene_commit(rtx_scene* s, int) { Enter e(s); s->committed = true; s->commits = s->commits + 1; return 0; }
int rtx_scene_get_info(const rtx_scene* s, rtx_scene_info* i) { std::memset(i, 0, sizeof *i); i->ntris = s->ntris; return 0; }
int rtx_scene_clone(const rtx_scene* src, rtx_scene** out) { *out = new rtx_scene(*src); (*out)->in_call = 0; return 0; }
int rtx_get_stats(const rtx_scene*, rtx_stats* st) { std::memset(st, 0, sizeof *st); return 0; }
// t = |o| + 10 |d.x|, face = the integer part of o.y, material = face + 1: anything that depends on the ray alone
int rtx_trace_closest(rtx_scene* s, const float* o, const float* d, uint32_t n, float* t, int32_t* face, float* nrm, float* pt, int32_t* mat) {
  Enter e(s);
  for (uint32_t i = 0; i < n; ++i) {
    t[i] = std::sqrt(o[3 * i] * o[3 * i] + o[3 * i + 1] * o[3 * i + 1] + o[3 * i + 2] * o[3 * i + 2]) + 10.0f * std::fabs(d[3 * i]);
    face[i] = 0;
    if (nrm) { nrm[3 * i] = d[3 * i]; nrm[3 * i + 1] = d[3 * i + 1]; nrm[3 * i + 2] = d[3 * i + 2]; }
    if (pt) { pt[3 * i] = o[3 * i]; pt[3 * i + 1] = o[3 * i + 1]; pt[3 * i + 2] = o[3 * i + 2]; }
    if (mat) mat[i] = (int32_t)o[3 * i + 1];
  }
  return 0;
}
int rtx_trace_shade(rtx_scene* s, const rtx_options*, const float* o, const float* d, uint32_t n, int32_t bounce, int32_t inside, float* rgb) {
  Enter e(s);
  for (uint32_t i = 0; i < n; ++i) { rgb[3 * i] = o[3 * i] + d[3 * i]; rgb[3 * i + 1] = (float)bounce; rgb[3 * i + 2] = (float)inside + o[3 * i + 2]; }
  return 0;
}
int rtx_trace_visible(rtx_scene* s, const float* light, const float* pos, uint32_t n, uint8_t* vis, int32_t* bm) {
  Enter e(s);
  for (uint32_t i = 0; i < n; ++i) { vis[i] = light[3 * i] > pos[3 * i] ? 1 : 0; if (bm) bm[i] = -1; }
  return 0;
}
int rtx_camera_points(rtx_scene* s, const rtx_camera* cam, const float* ij, uint32_t n, float* world, float* center) {
  Enter e(s);
  if (center) { center[0] = -cam->view[3]; center[1] = -cam->view[7]; center[2] = -cam->view[11]; }
  for (uint32_t i = 0; i < n; ++i) { world[3 * i] = ij[2 * i]; world[3 * i + 1] = ij[2 * i + 1]; world[3 * i + 2] = -1.0f; }
  return 0;
}
int rtx_render(rtx_scene* s, const rtx_camera*, const rtx_options*, rtx_rect r, float* rgb, float*, int32_t*) {
  Enter e(s);
  s->renders = s->renders + 1;
  const int W = r.x1 - r.x0, H = r.y1 - r.y0;
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) { float* p = rgb + 3 * ((size_t)y * W + x); p[0] = (float)x; p[1] = (float)y; p[2] = (float)(x * 1000 + y); }
  return 0;
}
int rtx_render_multi(rtx_scene* const* scenes, int32_t, const rtx_camera* c, const rtx_options* o, rtx_rect r, int32_t, float* rgb, float* t, int32_t* f) {
  return rtx_render(scenes[0], c, o, r, rgb, t, f);
}
// test hooks
int rtx_stub_renders(const rtx_scene* s) { return s->renders; }
}
