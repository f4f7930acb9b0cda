// Binned-SAH builder for the 32-byte-node linear BVH (see bvh_build.h).
// Replaces Flyscene::divideBox_KD (src/flyscene.cpp:1110-1185), whose tree is never even
// attached to the root box in the reference; only the *results* of closest-hit
// queries have to agree with the reference's brute-force scan, which holds for any
// conservative hierarchy because hits are reduced with a (t, faceId) lexicographic min.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace rtx {
namespace {

constexpr int kBins = 16;
// SAH cost of a split relative to one triangle test.  Measured on the GPU (RTX_TRAV_COST is the tuning
// hook): against 1.2, 0.6 (leaves of mostly one triangle) gives the soup +7 %, the hall the same, the bunny
// -8 %; 0.8 gives +4 % / same / same (leaf steps run at the lowest lane utilisation, so small leaves
// pay on incoherent rays and cost node steps on coherent ones).
static const float kTravCost = [] { const char* e = std::getenv("RTX_TRAV_COST"); return e ? (float)std::atof(e) : 0.8f; }();

struct Box3 {
  float lo[3], hi[3];
  void reset() { for (int k = 0; k < 3; ++k) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; } }
  void grow(const float* l, const float* h) {
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]); }
  }
  void grow(const Box3& b) { grow(b.lo, b.hi); }
  float half_area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
  }
};

// A triangle as the builder moves it about: its padded box and its input index, 32 bytes.  The passes of a node's range
// read and partition THESE, in place -- the first version permuted an index array and gathered boxes and centroids
// through it, one cache miss per triangle and pass once the mesh outgrew the caches.
struct Prim {
  float lo[3], hi[3]; uint32_t id, pad_;
  float cen(int a) const { return 0.5f * lo[a] + 0.5f * hi[a]; }
};
struct Ctx {
  Prim* prim;
};

// A range of the order array waiting to become a node.  `known`: its bounds came with it (the union of the parent's bins on
// the side of the split: min / max are exact, so these are the very numbers a pass over the range would give).
struct Task { uint32_t node, b, e; int depth; bool known; Box3 bb, cb; };

// Bins of one axis: triangle boxes, centroid boxes and counts.
struct AxisBins {
  Box3 box[kBins], cen[kBins]; uint32_t cnt[kBins];
  void reset() { for (int i = 0; i < kBins; ++i) { box[i].reset(); cen[i].reset(); cnt[i] = 0; } }
  void merge(const AxisBins& o) { for (int i = 0; i < kBins; ++i) { box[i].grow(o.box[i]); cen[i].grow(o.cen[i]); cnt[i] += o.cnt[i]; } }
};
struct Bins3 { AxisBins ax[3]; };

// n - 1 helper threads that sleep between jobs: run(f) executes f(k) for k = 0 .. n-1, k = 0 on the caller, and returns when
// all are done.  The top of the tree hands out several hundred short jobs (bin a range, count, scatter): starting threads for
// each would cost as much as the work.
class Pool {
 public:
  explicit Pool(int n) : n_(n < 1 ? 1 : n) {
    for (int k = 1; k < n_; ++k) th_.emplace_back([this, k] { loop(k); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (std::thread& t : th_) t.join();
  }
  int size() const { return n_; }
  void run(const std::function<void(int)>& f) {
    if (n_ == 1) { f(0); return; }
    { std::lock_guard<std::mutex> g(mu_); job_ = &f; left_ = n_ - 1; ++gen_; }
    cv_.notify_all();
    f(0);
    std::unique_lock<std::mutex> g(mu_);
    done_.wait(g, [this] { return left_ == 0; });
    job_ = nullptr;
  }
 private:
  void loop(int k) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* f;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        f = job_;
      }
      (*f)(k);
      { std::lock_guard<std::mutex> g(mu_); if (--left_ == 0) done_.notify_one(); }
    }
  }
  int n_; std::vector<std::thread> th_;
  std::mutex mu_; std::condition_variable cv_, done_;
  const std::function<void(int)>* job_ = nullptr; int left_ = 0; uint64_t gen_ = 0; bool stop_ = false;
};

struct SubtreeBuilder {
  const Ctx& c;
  std::vector<BvhNode>& nodes;   // nodes[root] must exist
  uint32_t nleaves = 0, max_depth = 0;
  Pool* pool = nullptr;          // set: ranges of kParallelMin triangles and more are binned and partitioned by its threads
  int threads = 1;               // (= pool->size())
  Prim* scratch = nullptr;       // the parallel partition's copy of a range (room for the whole mesh, owned by build_bvh)
  static constexpr uint32_t kSmall = 16;   // ranges of up to this many triangles: only the bins they touch are looked at
  Bins3 small_bins; bool small_ready = false;   // (kept clean between nodes: a small node resets what it touched)
  static constexpr uint32_t kParallelMin = 1u << 13;
  SubtreeBuilder(const Ctx& ctx, std::vector<BvhNode>& n) : c(ctx), nodes(n) {}

  static int levels_needed(uint32_t count) {
    uint32_t leaves = (count + kMaxLeafTris - 1) / kMaxLeafTris;
    int l = 0;
    while ((1u << l) < leaves) ++l;
    return l;
  }

  void make_leaf(uint32_t node, uint32_t b, uint32_t e, const Box3& bb, int depth) {
    BvhNode& n = nodes[node];
    std::memcpy(n.lo, bb.lo, 12); std::memcpy(n.hi, bb.hi, 12);
    n.leftFirst = b; n.count = e - b;
    // keep triangles of a leaf in ascending input-id order (cheap, helps tie handling stay local)
    std::sort(c.prim + b, c.prim + e, [](const Prim& x, const Prim& y) { return x.id < y.id; });
    ++nleaves;
    max_depth = std::max<uint32_t>(max_depth, depth);
  }

  void clean_small(bool small, const uint8_t touched[3][kSmall], const int ntouched[3]) {
    if (!small) return;
    for (int a = 0; a < 3; ++a)
      for (int j = 0; j < ntouched[a]; ++j) { const int i = touched[a][j]; AxisBins& B = small_bins.ax[a]; B.box[i].reset(); B.cen[i].reset(); B.cnt[i] = 0; }
  }
  static int bin_of(float cen, float lo, float scale) {
    int bi = (int)((cen - lo) * scale);
    return bi < 0 ? 0 : (bi >= kBins ? kBins - 1 : bi);
  }
  // One pass over ord[b, e): the bins of every axis with a centroid extent (the three passes of the first version read
  // every triangle's box three times; the boxes are gathered through the order array, so a pass is cache misses).
  void bin_range(uint32_t b, uint32_t e, const Box3& cb, const bool use[3], const float scale[3], Bins3* out) const {
    for (int a = 0; a < 3; ++a) if (use[a]) out->ax[a].reset();
    for (uint32_t i = b; i < e; ++i) {
      const Prim& P = c.prim[i];
      const float* tl = P.lo; const float* thi = P.hi;
      const float tc[3] = {P.cen(0), P.cen(1), P.cen(2)};
      for (int a = 0; a < 3; ++a) {
        if (!use[a]) continue;
        const int bi = bin_of(tc[a], cb.lo[a], scale[a]);
        AxisBins& B = out->ax[a];
        B.box[bi].grow(tl, thi); B.cen[bi].grow(tc, tc); ++B.cnt[bi];
      }
    }
  }

  // Returns the split position (b < mid < e) or 0 when the range should become a leaf.  On a binned split the bounds of
  // the two sides are handed back (*sides = true).
  uint32_t split(uint32_t b, uint32_t e, const Box3& bb, const Box3& cb, int depth, bool* sides, Box3 side_bb[2], Box3 side_cb[2]) {
    const uint32_t count = e - b;
    *sides = false;
    const bool must_balance = depth + levels_needed(count) >= kMaxDepth;
    const bool wide = pool && threads > 1 && count >= kParallelMin;
    int best_axis = -1, best_bin = -1;
    float best_cost = FLT_MAX;
    Bins3 bins_local;
    Bins3* binsp = &bins_local;
    bool use[3] = {false, false, false}; float scale[3] = {0, 0, 0};
    uint8_t touched[3][kSmall]; int ntouched[3] = {0, 0, 0};
    const bool small = !must_balance && count <= kSmall;
    if (small) {
      // Nine nodes in ten hold a handful of triangles: their 16 bins per axis are mostly empty, and resetting and scanning
      // all of them was most of the builder's time.  The occupied bins, in ascending order, give the same prefix and suffix
      // unions (an empty bin adds nothing), the same costs at the positions behind an occupied bin, and the positions behind
      // an empty bin repeat the cost before them -- never strictly smaller, so never chosen: same split, same tie-breaks.
      if (!small_ready) { for (int a = 0; a < 3; ++a) small_bins.ax[a].reset(); small_ready = true; }
      binsp = &small_bins;
      for (int axis = 0; axis < 3; ++axis) {
        const float ext = cb.hi[axis] - cb.lo[axis];
        use[axis] = ext > 0.0f;
        if (use[axis]) scale[axis] = kBins / ext;
      }
      for (uint32_t i = b; i < e; ++i) {
        const Prim& P = c.prim[i];
        const float tc[3] = {P.cen(0), P.cen(1), P.cen(2)};
        for (int a = 0; a < 3; ++a) {
          if (!use[a]) continue;
          const int bi = bin_of(tc[a], cb.lo[a], scale[a]);
          AxisBins& B = small_bins.ax[a];
          if (B.cnt[bi] == 0) touched[a][ntouched[a]++] = (uint8_t)bi;
          B.box[bi].grow(P.lo, P.hi); B.cen[bi].grow(tc, tc); ++B.cnt[bi];
        }
      }
      for (int axis = 0; axis < 3; ++axis) {
        if (!use[axis]) continue;
        uint8_t* tb = touched[axis]; const int m = ntouched[axis];
        for (int i = 1; i < m; ++i) { const uint8_t v = tb[i]; int j = i; while (j > 0 && tb[j - 1] > v) { tb[j] = tb[j - 1]; --j; } tb[j] = v; }
        const AxisBins& B = small_bins.ax[axis];
        float right_area[kSmall]; uint32_t right_cnt[kSmall];
        Box3 acc; acc.reset(); uint32_t n = 0;
        for (int j = m - 1; j > 0; --j) { acc.grow(B.box[tb[j]]); n += B.cnt[tb[j]]; right_area[j] = acc.half_area(); right_cnt[j] = n; }
        acc.reset(); n = 0;
        for (int j = 0; j + 1 < m; ++j) {
          acc.grow(B.box[tb[j]]); n += B.cnt[tb[j]];
          float cost = acc.half_area() * n + right_area[j + 1] * right_cnt[j + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = tb[j]; }
        }
      }
    } else
    if (!must_balance) {
      bool any = false;
      for (int axis = 0; axis < 3; ++axis) {
        const float ext = cb.hi[axis] - cb.lo[axis];
        use[axis] = ext > 0.0f;
        if (use[axis]) { scale[axis] = kBins / ext; any = true; }
      }
      if (any) {
        if (wide) {
          std::vector<Bins3> part((size_t)threads);
          const uint32_t step = (count + (uint32_t)threads - 1) / (uint32_t)threads;
          pool->run([&](int k) {
            const uint32_t cb0 = std::min(e, b + (uint32_t)k * step), ce0 = std::min(e, cb0 + step);
            bin_range(cb0, ce0, cb, use, scale, &part[(size_t)k]);
          });
          for (int a = 0; a < 3; ++a) if (use[a]) { bins_local.ax[a] = part[0].ax[a]; for (int k = 1; k < threads; ++k) bins_local.ax[a].merge(part[(size_t)k].ax[a]); }
        } else {
          bin_range(b, e, cb, use, scale, &bins_local);
        }
      }
      for (int axis = 0; axis < 3; ++axis) {
        if (!use[axis]) continue;
        const AxisBins& B = bins_local.ax[axis];
        float right_area[kBins]; uint32_t right_cnt[kBins];
        Box3 acc; acc.reset(); uint32_t n = 0;
        for (int i = kBins - 1; i > 0; --i) { acc.grow(B.box[i]); n += B.cnt[i]; right_area[i] = acc.half_area(); right_cnt[i] = n; }
        acc.reset(); n = 0;
        for (int i = 0; i < kBins - 1; ++i) {
          acc.grow(B.box[i]); n += B.cnt[i];
          if (n == 0 || right_cnt[i + 1] == 0) continue;
          float cost = acc.half_area() * n + right_area[i + 1] * right_cnt[i + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = i; }
        }
      }
    }
    if (best_axis >= 0) {
      float parent = bb.half_area();
      float split_cost = kTravCost + (parent > 0 ? best_cost / parent : FLT_MAX);
      if (count <= (uint32_t)kMaxLeafTris && split_cost >= (float)count) { clean_small(small, touched, ntouched); return 0; }
      const float sc = scale[best_axis], lo = cb.lo[best_axis];
      const int ax = best_axis, bin = best_bin;
      auto left_side = [&](const Prim& P) { return bin_of(P.cen(ax), lo, sc) <= bin; };
      uint32_t m;
      if (wide) {
        // every thread sorts its chunk of a copy into "left" and "right", then writes both where the prefix sums say: the two
        // sides hold the same SETS as std::partition's, in another order -- and nothing below depends on the order inside a
        // range (bins are unions and counts, leaves sort their triangles, the median fallback breaks ties by index)
        const uint32_t step = (count + (uint32_t)threads - 1) / (uint32_t)threads;
        std::vector<uint32_t> nl((size_t)threads, 0);
        pool->run([&](int k) {
          const uint32_t c0 = std::min(count, (uint32_t)k * step), c1 = std::min(count, c0 + step);
          std::memcpy(scratch + c0, c.prim + b + c0, (size_t)(c1 - c0) * sizeof(Prim));
          uint32_t n = 0;
          for (uint32_t i = c0; i < c1; ++i) n += left_side(scratch[i]) ? 1u : 0u;
          nl[(size_t)k] = n;
        });
        std::vector<uint32_t> lpos((size_t)threads), rpos((size_t)threads);
        uint32_t total_left = 0;
        for (int k = 0; k < threads; ++k) total_left += nl[(size_t)k];
        uint32_t lp = 0, rp = total_left;
        for (int k = 0; k < threads; ++k) {
          const uint32_t c0 = std::min(count, (uint32_t)k * step), c1 = std::min(count, c0 + step);
          lpos[(size_t)k] = lp; rpos[(size_t)k] = rp;
          lp += nl[(size_t)k]; rp += (c1 - c0) - nl[(size_t)k];
        }
        pool->run([&](int k) {
          const uint32_t c0 = std::min(count, (uint32_t)k * step), c1 = std::min(count, c0 + step);
          uint32_t l = lpos[(size_t)k], r = rpos[(size_t)k];
          for (uint32_t i = c0; i < c1; ++i) { const Prim& t = scratch[i]; if (left_side(t)) c.prim[b + l++] = t; else c.prim[b + r++] = t; }
        });
        m = b + total_left;
      } else {
        Prim* mid = std::partition(c.prim + b, c.prim + e, left_side);
        m = (uint32_t)(mid - c.prim);
      }
      if (m > b && m < e) {
        const AxisBins& B = binsp->ax[best_axis];
        for (int sd = 0; sd < 2; ++sd) { side_bb[sd].reset(); side_cb[sd].reset(); }
        if (small) {
          for (int j = 0; j < ntouched[best_axis]; ++j) { const int i = touched[best_axis][j]; const int sd = i <= best_bin ? 0 : 1; side_bb[sd].grow(B.box[i]); side_cb[sd].grow(B.cen[i]); }
        } else {
          for (int i = 0; i < kBins; ++i) { const int sd = i <= best_bin ? 0 : 1; side_bb[sd].grow(B.box[i]); side_cb[sd].grow(B.cen[i]); }
        }
        *sides = true;
        clean_small(small, touched, ntouched);
        return m;
      }
    }
    clean_small(small, touched, ntouched);
    if (count <= (uint32_t)kMaxLeafTris) return 0;
    // balanced fallback: median along the widest centroid axis (also the depth guard)
    int axis = 0;
    for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
    uint32_t m = b + count / 2;
    std::nth_element(c.prim + b, c.prim + m, c.prim + e, [&](const Prim& x, const Prim& y) {
      float cx = x.cen(axis), cy = y.cen(axis);
      return cx < cy || (cx == cy && x.id < y.id);
    });
    return m;
  }

  void bounds(uint32_t b, uint32_t e, Box3* bb, Box3* cb) const {
    bb->reset(); cb->reset();
    for (uint32_t i = b; i < e; ++i) {
      const Prim& P = c.prim[i];
      const float pc[3] = {P.cen(0), P.cen(1), P.cen(2)};
      bb->grow(P.lo, P.hi);
      cb->grow(pc, pc);
    }
  }

  // Expands node `node` over ord[b,e).  When `stop` is non-null, ranges <= stop_size are
  // not expanded but recorded as tasks (top phase).
  void build(const Task& root, std::vector<Task>* stop, uint32_t stop_size) {
    std::vector<Task> stack;
    stack.push_back(root);
    while (!stack.empty()) {
      Task t = stack.back(); stack.pop_back();
      if (stop && t.e - t.b <= stop_size) { stop->push_back(t); continue; }
      if (!t.known) {
        if (pool && threads > 1 && t.e - t.b >= kParallelMin) {
          std::vector<Box3> pb((size_t)threads), pc((size_t)threads);
          const uint32_t count = t.e - t.b, step = (count + (uint32_t)threads - 1) / (uint32_t)threads;
          pool->run([&](int k) {
            const uint32_t c0 = std::min(t.e, t.b + (uint32_t)k * step), c1 = std::min(t.e, c0 + step);
            bounds(c0, c1, &pb[(size_t)k], &pc[(size_t)k]);
          });
          t.bb = pb[0]; t.cb = pc[0];
          for (int k = 1; k < threads; ++k) { t.bb.grow(pb[(size_t)k]); t.cb.grow(pc[(size_t)k]); }
        } else {
          bounds(t.b, t.e, &t.bb, &t.cb);
        }
        t.known = true;
      }
      bool sides = false; Box3 sbb[2], scb[2];
      uint32_t mid = split(t.b, t.e, t.bb, t.cb, t.depth, &sides, sbb, scb);
      if (mid == 0) { make_leaf(t.node, t.b, t.e, t.bb, t.depth); continue; }
      uint32_t left = (uint32_t)nodes.size();
      nodes.emplace_back(); nodes.emplace_back();
      BvhNode& n = nodes[t.node];
      std::memcpy(n.lo, t.bb.lo, 12); std::memcpy(n.hi, t.bb.hi, 12);
      n.leftFirst = left; n.count = 0;
      // right pushed first so the left subtree is laid out right after its parent pair
      Task r{left + 1, mid, t.e, t.depth + 1, sides, sbb[1], scb[1]};
      Task l{left, t.b, mid, t.depth + 1, sides, sbb[0], scb[0]};
      stack.push_back(r);
      stack.push_back(l);
    }
  }
};

}  // namespace

void build_bvh(const float* tri_lo, const float* tri_hi, uint32_t ntris, int threads, BvhBuildResult* out) {
  const auto t_begin = std::chrono::steady_clock::now();
  out->nodes.clear(); out->order.clear(); out->nleaves = 0; out->max_depth = 0;
  out->order.resize(ntris);
  out->nodes.reserve((size_t)ntris * 2 + 2);
  out->nodes.emplace_back();
  if (ntris == 0) {
    // Empty mesh: a lone root with an inverted (empty) box.  The traversal kernels never
    // descend when the scene has no triangles (DevScene::ntris == 0).
    BvhNode& n = out->nodes[0];
    for (int k = 0; k < 3; ++k) { n.lo[k] = FLT_MAX; n.hi[k] = -FLT_MAX; }
    n.leftFirst = 0; n.count = 0;
    return;
  }
  if (threads < 1) threads = 1;
  uint32_t stop_size = 0;
  if (threads > 1 && ntris > 65536) stop_size = std::max<uint32_t>(ntris / (uint32_t)(threads * 8), 8192);
  std::unique_ptr<Pool> helpers(stop_size ? new Pool(threads) : nullptr);     // the top of the tree works on the whole mesh: its passes are shared out as well
  std::unique_ptr<Prim[]> prims(new Prim[ntris]), scratch(stop_size ? new Prim[ntris] : nullptr);
  auto fill = [&](uint32_t i0, uint32_t i1) {
    for (uint32_t i = i0; i < i1; ++i) {
      Prim& P = prims[i];
      for (int k = 0; k < 3; ++k) { P.lo[k] = tri_lo[3 * (size_t)i + k]; P.hi[k] = tri_hi[3 * (size_t)i + k]; }
      P.id = i; P.pad_ = 0;
    }
  };
  if (helpers) {
    const uint32_t step = (ntris + (uint32_t)threads - 1) / (uint32_t)threads;
    helpers->run([&](int k) { const uint32_t i0 = std::min(ntris, (uint32_t)k * step); fill(i0, std::min(ntris, i0 + step)); });
  } else fill(0, ntris);
  Ctx ctx;
  ctx.prim = prims.get();
  struct WriteOrder {        // the order array is the ids in their final places
    const Prim* p; std::vector<uint32_t>& o;
    ~WriteOrder() { for (size_t i = 0; i < o.size(); ++i) o[i] = p[i].id; }
  } write_order{prims.get(), out->order};

  std::vector<Task> tasks;
  SubtreeBuilder top(ctx, out->nodes);
  top.pool = helpers.get(); top.threads = stop_size ? threads : 1; top.scratch = scratch.get();
  Task root_task{0u, 0u, ntris, 0, false, Box3(), Box3()};
  top.build(root_task, stop_size ? &tasks : nullptr, stop_size);
  out->nleaves = top.nleaves; out->max_depth = top.max_depth;
  helpers.reset();
  const auto t_top = std::chrono::steady_clock::now();
  if (tasks.empty()) return;

  struct Local { std::vector<BvhNode> nodes; uint32_t nleaves = 0, max_depth = 0; };
  std::vector<Local> locals(tasks.size());
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= tasks.size()) break;
      Local& L = locals[i];
      L.nodes.reserve((size_t)(tasks[i].e - tasks[i].b) + 2);
      L.nodes.emplace_back();
      SubtreeBuilder sb(ctx, L.nodes);
      Task local = tasks[i];
      local.node = 0;
      sb.build(local, nullptr, 0);
      L.nleaves = sb.nleaves; L.max_depth = sb.max_depth;
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
  for (auto& t : pool) t.join();
  const auto t_tasks = std::chrono::steady_clock::now();
  struct Report { std::chrono::steady_clock::time_point a, b, c; size_t n; ~Report() { if (std::getenv("RTX_BUILD_TIMING")) std::fprintf(stderr, "[build_bvh: top %.1f ms, %zu subtrees %.1f ms, stitch %.1f ms] ",
      std::chrono::duration<float, std::milli>(b - a).count(), n, std::chrono::duration<float, std::milli>(c - b).count(), std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - c).count()); } } report{t_begin, t_top, t_tasks, tasks.size()};
  // stitch in task order: local node 0 -> reserved global slot, local nodes 1.. appended (places from a prefix sum, copies on threads)
  std::vector<uint32_t> off(tasks.size() + 1);
  off[0] = (uint32_t)out->nodes.size();
  for (size_t i = 0; i < tasks.size(); ++i) off[i + 1] = off[i] + (uint32_t)(locals[i].nodes.size() - 1);
  out->nodes.resize(off[tasks.size()]);
  std::atomic<size_t> next_stitch{0};
  auto stitch = [&]() {
    for (;;) {
      const size_t i = next_stitch.fetch_add(1);
      if (i >= tasks.size()) break;
      Local& L = locals[i];
      const uint32_t o = off[i];
      auto fix = [&](BvhNode n) { if (n.count == 0) n.leftFirst = o + n.leftFirst - 1; return n; };
      const bool root_internal = L.nodes.size() > 1;
      out->nodes[tasks[i].node] = root_internal ? fix(L.nodes[0]) : L.nodes[0];
      // a leaf has count > 0 here (ntris > 0), so count == 0 means internal
      for (size_t k = 1; k < L.nodes.size(); ++k) { const BvhNode n = L.nodes[k]; out->nodes[o + k - 1] = n.count == 0 ? fix(n) : n; }
    }
  };
  pool.clear();
  for (int t = 0; t < threads; ++t) pool.emplace_back(stitch);
  for (auto& t : pool) t.join();
  for (size_t i = 0; i < tasks.size(); ++i) { out->nleaves += locals[i].nleaves; out->max_depth = std::max(out->max_depth, locals[i].max_depth); }
}


namespace {
// Fills group g with the (up to 4) grandchildren of binary node b; returns how many slots are in use and, per slot,
// the binary node it came from.
// The (up to) four binary nodes that become the slots of b's group.
//  greedy: start from b's two children and keep replacing the internal candidate with the largest surface by its
//    own two children until four slots are taken -- a leaf child leaves its slot's worth of depth to the other side,
//    and the big boxes (the ones most rays enter) are the ones opened;
//  fixed: the children's children (two binary levels per group step, whatever they hold).
// The collapse that minimises the expected number of group steps exactly (round 3; the default): a ray enters a
// group when it hits the box of the binary node the group was made from, so the expectation is the sum of those nodes'
// surface areas, and the best way to spend the k slots a subtree is given follows by dynamic programming:
//   cost[n][1] = area(n) + min_k D(n, k) for an inner node (it becomes a group of its own), 0 for a leaf;
//   cost[n][k >= 2] = D(n, k) = min_{a + b = k} cost[left][a] + cost[right][b]   (n is opened inside its parent's group).
// Against the greedy rule below (open the largest candidate first), CPU mirror: soup 30.25 -> 29.82 group steps per ray,
// hall 9.93 -> 9.44, bunny 7.52 -> 7.44, 13-23 % fewer groups (3.4-3.7 slots in use per group instead of 3.0-3.1); on
// the GPU hall +1 %, soup and bunny within noise.
struct DpPlan {
  std::vector<float> cost;         // 4 per node: k = 1..4
  std::vector<uint8_t> left_k;     // 4 per node: slots given to the left child for k = 2..4 (index k-1); [0] = the k a group of its own takes
  bool on = false;
};
thread_local DpPlan g_dp;
void dp_plan(const std::vector<BvhNode>& bin) {
  const size_t n = bin.size();
  g_dp.cost.assign(4 * n, FLT_MAX); g_dp.left_k.assign(4 * n, 0);
  float* cost = g_dp.cost.data(); uint8_t* left_k = g_dp.left_k.data();      // (g_dp is this thread's; the helpers below get the arrays)
  auto node_dp = [&bin, cost, left_k](uint32_t x) {
    float* c = &cost[4 * (size_t)x];
    if (bin[x].count != 0) { c[0] = 0.0f; return; }
    const uint32_t l = bin[x].leftFirst, r = l + 1;
    const float* cl = &cost[4 * (size_t)l]; const float* cr = &cost[4 * (size_t)r];
    for (int k = 2; k <= 4; ++k)
      for (int a = 1; a < k; ++a) {
        if (cl[a - 1] == FLT_MAX || cr[k - a - 1] == FLT_MAX) continue;
        const float v = cl[a - 1] + cr[k - a - 1];
        if (v < c[k - 1]) { c[k - 1] = v; left_k[4 * (size_t)x + k - 1] = (uint8_t)a; }
      }
    const float ex = bin[x].hi[0] - bin[x].lo[0], ey = bin[x].hi[1] - bin[x].lo[1], ez = bin[x].hi[2] - bin[x].lo[2];
    const float area = ex * ey + ey * ez + ez * ex;
    int bk = 2;
    for (int k = 3; k <= 4; ++k) if (c[k - 1] < c[bk - 1]) bk = k;
    c[0] = area + c[bk - 1];
    left_k[4 * (size_t)x] = (uint8_t)bk;
  };
  // children before parents: the subtree of `root`, parents after their children
  auto subtree = [&bin, &node_dp](uint32_t root) {
    std::vector<uint32_t> order, stack{root};
    while (!stack.empty()) { const uint32_t x = stack.back(); stack.pop_back(); order.push_back(x); if (bin[x].count == 0) { stack.push_back(bin[x].leftFirst); stack.push_back(bin[x].leftFirst + 1); } }
    for (size_t i = order.size(); i-- > 0;) node_dp(order[i]);
  };
  const int threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  if (threads == 1 || n < (1u << 17)) { subtree(0); return; }
  // large trees: the subtrees seven levels down on threads (a node's plan depends on its own subtree only), the top after them
  std::vector<uint32_t> top, roots, level{0};
  for (int d = 0; d < 7 && !level.empty(); ++d) {
    std::vector<uint32_t> next;
    for (uint32_t x : level) { top.push_back(x); if (bin[x].count == 0) { next.push_back(bin[x].leftFirst); next.push_back(bin[x].leftFirst + 1); } }
    level.swap(next);
  }
  roots.swap(level);
  std::atomic<size_t> cursor{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&] { for (;;) { const size_t i = cursor.fetch_add(1); if (i >= roots.size()) break; subtree(roots[i]); } });
  for (std::thread& t : pool) t.join();
  for (size_t i = top.size(); i-- > 0;) node_dp(top[i]);
}
void dp_expand(const std::vector<BvhNode>& bin, uint32_t x, int k, uint32_t cand[4], int* nc) {
  if (k == 1) { cand[(*nc)++] = x; return; }
  const int a = g_dp.left_k[4 * (size_t)x + k - 1];
  dp_expand(bin, bin[x].leftFirst, a, cand, nc);
  dp_expand(bin, bin[x].leftFirst + 1, k - a, cand, nc);
}
// (the choice travels as an argument: scenes are built independently, possibly on several threads)
int group_candidates(const std::vector<BvhNode>& bin, uint32_t b, bool greedy, uint32_t cand[4]) {
  int nc = 0;
  if (greedy && g_dp.on) { dp_expand(bin, b, g_dp.left_k[4 * (size_t)b], cand, &nc); return nc; }
  const uint32_t kids[2] = {bin[b].leftFirst, bin[b].leftFirst + 1};
  if (!greedy) {
    for (int k = 0; k < 2; ++k) {
      const BvhNode& c = bin[kids[k]];
      if (c.count == 0) { cand[nc++] = c.leftFirst; cand[nc++] = c.leftFirst + 1; }
      else cand[nc++] = kids[k];
    }
    return nc;
  }
  cand[nc++] = kids[0]; cand[nc++] = kids[1];
  while (nc < 4) {
    int pick = -1; float best = -1.0f;
    for (int k = 0; k < nc; ++k) {
      const BvhNode& c = bin[cand[k]];
      if (c.count != 0) continue;
      const float ex = c.hi[0] - c.lo[0], ey = c.hi[1] - c.lo[1], ez = c.hi[2] - c.lo[2];
      const float area = ex * ey + ey * ez + ez * ex;
      if (area > best) { best = area; pick = k; }
    }
    if (pick < 0) break;
    const uint32_t l = bin[cand[pick]].leftFirst;
    for (int k = nc; k > pick + 1; --k) cand[k] = cand[k - 1];      // the two children take the place of their parent
    cand[pick] = l; cand[pick + 1] = l + 1;
    ++nc;
  }
  return nc;
}
int fill_group(const std::vector<BvhNode>& bin, uint32_t b, bool greedy, std::vector<BvhNode>& q, uint32_t g, uint32_t depth, BvhBuildResult* out,
               uint32_t cand[4]) {
  const int nc = group_candidates(bin, b, greedy, cand);
  out->quad_depth = std::max(out->quad_depth, depth);
  for (int k = 0; k < 4; ++k) {
    BvhNode s;
    if (k < nc) {
      s = bin[cand[k]];
      ++out->quad_slots;
    } else {
      for (int a = 0; a < 3; ++a) { s.lo[a] = FLT_MAX; s.hi[a] = -FLT_MAX; }
      s.leftFirst = kEmptySlot; s.count = 0;
    }
    q[4 * (size_t)g + k] = s;
  }
  return nc;
}
// Depth first below the top levels: a subtree's groups stay together in memory.
uint32_t collapse_node(const std::vector<BvhNode>& bin, uint32_t b, bool greedy, std::vector<BvhNode>& q, uint32_t depth, BvhBuildResult* out) {
  uint32_t cand[4];
  const uint32_t g = (uint32_t)(q.size() / 4);
  q.resize(q.size() + 4);
  const int nc = fill_group(bin, b, greedy, q, g, depth, out, cand);
  for (int k = 0; k < nc; ++k) {
    if (bin[cand[k]].count == 0) {
      const uint32_t child = collapse_node(bin, cand[k], greedy, q, depth + 1, out);
      q[4 * (size_t)g + k].leftFirst = child;
    }
  }
  return g;
}
}  // namespace

static void collapse_once(BvhBuildResult* out, bool greedy);
void collapse_to_quads(BvhBuildResult* out) {
  static const bool fixed = [] { const char* e = std::getenv("RTX_COLLAPSE"); return e && std::strcmp(e, "fixed") == 0; }();   // tuning hook
  static const bool dp = [] { const char* e = std::getenv("RTX_COLLAPSE"); return !e || std::strcmp(e, "dp") == 0; }();   // default; "greedy" / "fixed": the earlier choices
  g_dp.on = dp && out->nodes.size() >= 3 && out->nodes[0].count == 0;
  if (g_dp.on) dp_plan(out->nodes);
  collapse_once(out, !fixed);
  g_dp.on = false;
  // the traversal stack holds 3 entries per group level (rtx.hip checks 3 * quad_depth against the same constant): a
  // tree the greedy choice makes too deep in groups is collapsed two binary levels per group instead
  if (!fixed && 3 * out->quad_depth > (uint32_t)kTraversalStackDepth) collapse_once(out, false);
}
static void collapse_once(BvhBuildResult* out, bool greedy) {
  out->quads.clear(); out->quad_depth = 0; out->quad_slots = 0;
  for (uint32_t& t : out->top_groups) t = 0;
  const std::vector<BvhNode>& bin = out->nodes;
  const BvhNode& root = bin[0];
  if (root.count == 0 && bin.size() >= 3) {        // internal root
    std::vector<BvhNode>& q = out->quads;
    q.reserve(bin.size() * 2);
    // The first kTopLevels levels breadth first, so that their groups are 0 .. top_groups[level]-1 (the kernels keep
    // them in LDS); everything below depth first.
    struct Item { uint32_t bin_node, group, depth; };
    std::vector<Item> fifo;
    q.resize(4);
    fifo.push_back({0u, 0u, 1u});
    for (size_t head = 0; head < fifo.size(); ++head) {
      const Item it = fifo[head];
      uint32_t cand[4];
      const int nc = fill_group(bin, it.bin_node, greedy, q, it.group, it.depth, out, cand);
      for (int k = 0; k < nc; ++k) {
        if (bin[cand[k]].count != 0) continue;
        if (it.depth < (uint32_t)kTopLevels) {
          const uint32_t child = (uint32_t)(q.size() / 4);
          q.resize(q.size() + 4);
          q[4 * (size_t)it.group + k].leftFirst = child;
          fifo.push_back({cand[k], child, it.depth + 1});
        }
      }
    }
    for (const Item& it : fifo) for (int l = (int)it.depth; l <= kTopLevels; ++l) ++out->top_groups[l];
    // (all top-level groups are numbered now; their deeper children follow, subtree by subtree)
    // the groups of the deepest top level: their internal slots (candidates recomputed as fill_group found them)
    for (const Item& it : fifo) {
      if (it.depth < (uint32_t)kTopLevels) continue;
      uint32_t cand[4];
      const int nc = group_candidates(bin, it.bin_node, greedy, cand);
      for (int k = 0; k < nc; ++k)
        if (bin[cand[k]].count == 0) {
          const uint32_t child = collapse_node(bin, cand[k], greedy, q, it.depth + 1, out);
          q[4 * (size_t)it.group + k].leftFirst = child;
        }
    }
    return;
  }
  // leaf root (<= 4 triangles) or empty mesh: one group holding the root itself / nothing
  out->quads.resize(4);
  for (int k = 0; k < 4; ++k) {
    BvhNode s;
    for (int a = 0; a < 3; ++a) { s.lo[a] = FLT_MAX; s.hi[a] = -FLT_MAX; }
    s.leftFirst = kEmptySlot; s.count = 0;
    out->quads[k] = s;
  }
  out->quad_depth = 1;
  for (int l = 1; l <= kTopLevels; ++l) out->top_groups[l] = 1;
  if (root.count > 0) { out->quads[0] = root; out->quad_slots = 1; }
}

void pack_groups(const std::vector<BvhNode>& quads, std::vector<PackedGroup>* out) {
  const size_t ng = quads.size() / 4;
  out->assign(ng, PackedGroup{});
  for (size_t g = 0; g < ng; ++g) {
    const BvhNode* q = &quads[4 * g];
    PackedGroup& P = (*out)[g];
    float glo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, ghi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    bool any = false;
    for (int k = 0; k < 4; ++k) {
      if (q[k].leftFirst == kEmptySlot && q[k].count == 0) continue;
      any = true;
      for (int a = 0; a < 3; ++a) { glo[a] = std::min(glo[a], q[k].lo[a]); ghi[a] = std::max(ghi[a], q[k].hi[a]); }
    }
    uint32_t planes[6] = {0, 0, 0, 0, 0, 0};   // lox loy loz hix hiy hiz
    float steps[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) {
      if (!any) { glo[a] = 0.0f; ghi[a] = 0.0f; }
      P.origin[a] = glo[a];
      // smallest power-of-two step with origin + 255 * step >= ghi (in the kernel's fp32 arithmetic)
      int e = 27;                                              // 2^-100: flat groups
      const float ext = ghi[a] - glo[a];
      if (ext > 0.0f) {
        int ex; std::frexp((double)ext / 255.0, &ex);          // ext/255 = m * 2^ex, m in [0.5, 1)
        e = std::max(1, std::min(254, ex + 127));
        auto top = [&](int ee) { float st; uint32_t b = (uint32_t)ee << 23; std::memcpy(&st, &b, 4); return glo[a] + 255.0f * st; };
        while (e < 254 && top(e) < ghi[a]) ++e;
      }
      float step; { uint32_t b = (uint32_t)e << 23; std::memcpy(&step, &b, 4); }
      steps[a] = step;
      for (int k = 0; k < 4; ++k) {
        uint32_t ql = 255, qh = 0;                              // unused slot (the entry says so)
        if (!(q[k].leftFirst == kEmptySlot && q[k].count == 0)) {
          double fl = std::floor(((double)q[k].lo[a] - (double)glo[a]) / (double)step);
          double fh = std::ceil(((double)q[k].hi[a] - (double)glo[a]) / (double)step);
          int il = (int)std::max(0.0, std::min(255.0, fl)), ih = (int)std::max(0.0, std::min(255.0, fh));
          // make sure of the direction in fp32, as the kernel decodes: origin + q * step
          while (il > 0 && glo[a] + (float)il * step > q[k].lo[a]) --il;
          while (ih < 255 && glo[a] + (float)ih * step < q[k].hi[a]) ++ih;
          ql = (uint32_t)il; qh = (uint32_t)ih;
        }
        planes[a] |= ql << (8 * k);
        planes[3 + a] |= qh << (8 * k);
      }
    }
    P.step_x = steps[0]; P.step_y = steps[1]; P.step_z = steps[2];
    P.lox = planes[0]; P.loy = planes[1]; P.loz = planes[2]; P.hix = planes[3]; P.hiy = planes[4]; P.hiz = planes[5];
    for (int k = 0; k < 4; ++k) {
      const BvhNode& nd = q[k];
      if (nd.leftFirst == kEmptySlot && nd.count == 0) P.entry[k] = kEmptyPacked;
      else if (nd.count) P.entry[k] = 0x80000000u | (nd.count << 28) | nd.leftFirst;
      else P.entry[k] = nd.leftFirst;
    }
  }
}

}  // namespace rtx
