// dr_bvh_build.cpp -- host-side BVHAccel constructor for the standalone host
// (a Dart caller marshals the BVHAccel it already built instead).
//
// Follows accelerators/bvh_accel.dart:41-91 (constructor), :228-417
// (_recursiveBuild, SPLIT_SAH with 12 buckets, f32 bucket costs, right child
// built first) and :419-437 (_flattenBVHTree), with core/common.dart:256-297
// (partition, nth_element == full sort) and core/bbox.dart.  The numerics
// contract is the reference's: f32 storage, f64 expressions.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <atomic>
#include <cstdlib>
#include <thread>
#include <chrono>
#include <cstdio>
#include <vector>

#include "../../include/dartray_hip.h"

#include "dr_options.h"  // dr_opt: dr_set_option's value, else the environment's (by value)

namespace {

// dart:math min / max (BBox.union / unionPoint, bbox.dart:135-155): the lesser / greater value, -0.0 below +0.0 --
// std::min / std::max would keep whichever zero came first, and the result would depend on the order of the items
static inline float dartMin(float a, float b) {
  if (a > b) return b;
  if (a < b) return a;
  if (a == 0.0f) return (float)(((double)a + (double)b) * (double)a * (double)b);
  return b != b ? b : a;
}
static inline float dartMax(float a, float b) {
  if (a > b) return a;
  if (a < b) return b;
  if (a == 0.0f) return a + b;
  return b != b ? b : a;
}

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::numeric_limits<float>::infinity();
      hi[k] = -std::numeric_limits<float>::infinity();
    }
  }
  void grow(const Box& b) {
    for (int k = 0; k < 3; ++k) {
      lo[k] = dartMin(lo[k], b.lo[k]);
      hi[k] = dartMax(hi[k], b.hi[k]);
    }
  }
  void growPoint(const float* p) {
    for (int k = 0; k < 3; ++k) {
      lo[k] = dartMin(lo[k], p[k]);
      hi[k] = dartMax(hi[k], p[k]);
    }
  }
  // BBox.surfaceArea (bbox.dart:163-166): d = pMax - pMin is a Vector (f32)
  double area() const {
    double dx = (double)(float)((double)hi[0] - (double)lo[0]);
    double dy = (double)(float)((double)hi[1] - (double)lo[1]);
    double dz = (double)(float)((double)hi[2] - (double)lo[2]);
    return 2.0 * (dx * dy + dx * dz + dy * dz);
  }
  // BBox.maximumExtent (bbox.dart:173-182)
  int widestAxis() const {
    float dx = (float)((double)hi[0] - (double)lo[0]);
    float dy = (float)((double)hi[1] - (double)lo[1]);
    float dz = (float)((double)hi[2] - (double)lo[2]);
    if (dx > dy && dx > dz) return 0;
    return dy > dz ? 1 : 2;
  }
};

struct Item {  // _BVHPrimitiveInfo (bvh_accel.dart:490-501)
  uint32_t prim;
  float c[3];
  Box b;
};

struct TNode {  // trivially constructible: the scratch tree (2n slots) is allocated without initialisation
  Box b;
  int32_t left, right;
  int32_t itemStart, itemEnd;  // leaf: its primitives are items[itemStart, itemEnd)
  uint32_t nPrims;
  uint32_t axis;
  uint32_t subNodes;   // nodes in this sub-tree
  uint32_t subPrims;   // primitives in this sub-tree
  uint32_t subDepth;   // height of this sub-tree
};

// The recursion of bvh_accel.dart:228-417.  Sub-trees only communicate through `orderedPrims` and the
// node numbering, and both follow from the finished tree (right-child-first leaf order, left-first
// depth-first numbering), so sub-trees are built by concurrent tasks and the two numberings are
// assigned afterwards.  Inside one node the bounds / centroid-bounds / SAH-bucket passes are
// order-independent reductions (min, max, integer counts) and run on several threads for big nodes;
// the reference's two-pointer `partition` (common.dart:256-287) decides the order of the items
// inside each half and is therefore kept serial.  The result is bit-identical for any thread count.
class SahBuilder {
 public:
  std::vector<Item> items;
  TNode* tree = nullptr;
  ~SahBuilder() { free(tree); }
  std::atomic<int> running{1};
  std::atomic<uint32_t> maxLeaf{1};  // largest multi-primitive leaf (rare: coincident centroids)
  int maxPrims = 4;
  int maxThreads = 1;

  // Scratch-tree slots are handed out arithmetically, not through a shared counter: a sub-tree over n
  // items has at most 2n-1 nodes, so the node of [start,end) sits at `self`, its left sub-tree starts at
  // self+1 and its right sub-tree at self + 2*|left| -- no atomics on the hot path, depth-first locality.
  int build(int start, int end, int self) {
    const int n = end - start;
    Box bb, cb;
    bounds(start, end, &bb, &cb);
    if (n == 1) return leaf(self, start, end, bb);
    const int dim = cb.widestAxis();
    if (cb.hi[dim] == cb.lo[dim]) return leaf(self, start, end, bb);  // bvh_accel.dart:265-274
    int mid = (start + end) / 2;
    if (n <= 4) {
      sortRange(start, end, dim);  // nth_element (common.dart:289-297)
    } else {
      constexpr int NB = 12;
      const double cmin = cb.lo[dim], cmax = cb.hi[dim];
      auto bucketOf = [&](const Item& it) {
        int b = (int)(NB * (((double)it.c[dim] - cmin) / (cmax - cmin)));
        return b == NB ? NB - 1 : b;
      };
      int cnt[NB];
      Box bk[NB];
      buckets(start, end, bucketOf, cnt, bk);
      float cost[NB - 1];  // Float32List (bvh_accel.dart:345)
      const double total = bb.area();
      for (int s = 0; s < NB - 1; ++s) {
        Box l, r;
        l.reset();
        r.reset();
        int nl = 0, nr = 0;
        for (int j = 0; j <= s; ++j) { l.grow(bk[j]); nl += cnt[j]; }
        for (int j = s + 1; j < NB; ++j) { r.grow(bk[j]); nr += cnt[j]; }
        cost[s] = (float)(0.125 + (nl * l.area() + nr * r.area()) / total);
      }
      int best = 0;
      double bestCost = cost[0];
      for (int s = 1; s < NB - 1; ++s)
        if ((double)cost[s] < bestCost) { bestCost = cost[s]; best = s; }
      if (n > maxPrims || bestCost < n) {
        mid = split(start, end, [&](const Item& it) { return bucketOf(it) <= best; });
      } else {
        return leaf(self, start, end, bb);
      }
    }
    // Right child first (bvh_accel.dart:407-411) -- as a concurrent task when the range is big.
    int r, l;
    const int leftSlot = self + 1, rightSlot = self + 2 * (mid - start);
    if (n > kTaskCutoff && acquireThread()) {
      std::thread t([&] {
        r = build(mid, end, rightSlot);
        running.fetch_sub(1);
      });
      l = build(start, mid, leftSlot);
      t.join();
    } else {
      r = build(mid, end, rightSlot);
      l = build(start, mid, leftSlot);
    }
    TNode& t = tree[self];
    t.left = l;
    t.right = r;
    t.b = tree[l].b;
    t.b.grow(tree[r].b);
    t.axis = (uint32_t)dim;
    t.nPrims = 0;
    t.itemStart = t.itemEnd = 0;
    t.subNodes = 1 + tree[l].subNodes + tree[r].subNodes;
    t.subPrims = tree[l].subPrims + tree[r].subPrims;
    t.subDepth = 1 + std::max(tree[l].subDepth, tree[r].subDepth);
    return self;
  }

  // orderedPrims + _flattenBVHTree (bvh_accel.dart:407-411,419-437) in one top-down pass: a node's index is
  // its parent's + 1 (first child) or + 1 + |left sub-tree| (second child, stored in the parent's `offset`);
  // the right sub-tree's primitives precede the left one's because the right child is built first.
  void emit(int n, uint32_t index, uint32_t primBase, DrBvhNode* out, uint32_t* order) {
    const TNode& t = tree[n];
    DrBvhNode& o = out[index];
    for (int k = 0; k < 3; ++k) {
      o.bmin[k] = t.b.lo[k];
      o.bmax[k] = t.b.hi[k];
    }
    o.pad = 0;
    if (t.nPrims > 0) {
      o.offset = primBase;
      o.nprims = (uint16_t)t.nPrims;
      o.axis = 0;
      for (int i = t.itemStart; i < t.itemEnd; ++i) order[primBase + (uint32_t)(i - t.itemStart)] = items[i].prim;
      return;
    }
    const uint32_t li = index + 1, ri = index + 1 + tree[t.left].subNodes;
    o.offset = ri;
    o.nprims = 0;
    o.axis = (uint8_t)t.axis;
    const uint32_t rightBase = primBase, leftBase = primBase + tree[t.right].subPrims;
    if (t.subNodes > (uint32_t)kTaskCutoff && acquireThread()) {
      std::thread th([&] {
        emit(t.right, ri, rightBase, out, order);
        running.fetch_sub(1);
      });
      emit(t.left, li, leftBase, out, order);
      th.join();
    } else {
      emit(t.left, li, leftBase, out, order);
      emit(t.right, ri, rightBase, out, order);
    }
  }

 private:
  static constexpr int kTaskCutoff = 1 << 15;   // ranges above this may be built by their own task
  static constexpr int kSliceCutoff = 1 << 20;  // ranges above this run their reductions on several threads

  bool acquireThread() {
    if (running.fetch_add(1) < maxThreads) return true;
    running.fetch_sub(1);
    return false;
  }
  // fn(lo, hi, slot): slot indexes a per-slice partial result
  template <class F>
  void slices(int start, int end, int* nSlices, F fn) {
    int want = 1;
    if (end - start > kSliceCutoff) want = std::max(1, std::min(16, maxThreads - running.load() + 1));
    *nSlices = want;
    if (want == 1) {
      fn(start, end, 0);
      return;
    }
    std::vector<std::thread> th;
    const int64_t n = end - start;
    for (int k = 1; k < want; ++k)
      th.emplace_back([=] { fn(start + (int)(n * k / want), start + (int)(n * (k + 1) / want), k); });
    fn(start, start + (int)(n / want), 0);
    for (auto& t : th) t.join();
  }
  void bounds(int start, int end, Box* bb, Box* cb) {
    if (end - start <= kSliceCutoff) {
      bb->reset();
      cb->reset();
      for (int i = start; i < end; ++i) {
        bb->grow(items[i].b);
        cb->growPoint(items[i].c);
      }
      return;
    }
    Box pb[16], pc[16];
    int ns = 1;
    slices(start, end, &ns, [&](int lo, int hi, int k) {
      Box b, c;
      b.reset();
      c.reset();
      for (int i = lo; i < hi; ++i) {
        b.grow(items[i].b);
        c.growPoint(items[i].c);
      }
      pb[k] = b;
      pc[k] = c;
    });
    *bb = pb[0];
    *cb = pc[0];
    for (int k = 1; k < ns; ++k) {
      bb->grow(pb[k]);
      cb->grow(pc[k]);
    }
  }
  template <class B>
  void buckets(int start, int end, B bucketOf, int* cnt, Box* bk) {
    constexpr int NB = 12;
    if (end - start <= kSliceCutoff) {
      for (int b = 0; b < NB; ++b) {
        cnt[b] = 0;
        bk[b].reset();
      }
      for (int i = start; i < end; ++i) {
        int b = bucketOf(items[i]);
        cnt[b]++;
        bk[b].grow(items[i].b);
      }
      return;
    }
    std::vector<int> pcnt(16 * NB, 0);
    std::vector<Box> pbk(16 * NB);
    for (auto& b : pbk) b.reset();
    int ns = 1;
    slices(start, end, &ns, [&](int lo, int hi, int k) {
      int* c = &pcnt[k * NB];
      Box* bx = &pbk[k * NB];
      for (int i = lo; i < hi; ++i) {
        int b = bucketOf(items[i]);
        c[b]++;
        bx[b].grow(items[i].b);
      }
    });
    for (int b = 0; b < NB; ++b) {
      cnt[b] = 0;
      bk[b].reset();
      for (int k = 0; k < ns; ++k) {
        cnt[b] += pcnt[k * NB + b];
        bk[b].grow(pbk[k * NB + b]);
      }
    }
  }
  int leaf(int self, int start, int end, const Box& bb) {
    TNode& t = tree[self];
    t.left = t.right = -1;
    t.axis = 0;
    t.itemStart = start;
    t.itemEnd = end;
    t.nPrims = (uint32_t)(end - start);
    if (t.nPrims > 1) {
      for (uint32_t cur = maxLeaf.load(); t.nPrims > cur && !maxLeaf.compare_exchange_weak(cur, t.nPrims);) {
      }
    }
    t.b = bb;
    t.subNodes = 1;
    t.subPrims = t.nPrims;
    t.subDepth = 0;
    return self;
  }
  // partition (common.dart:256-287)
  template <class P>
  int split(int first, int last, P pred) {
    while (first < last) {
      while (pred(items[first])) {
        if (++first == last) return first;
      }
      do {
        if (--last == first) return first;
      } while (!pred(items[last]));
      std::swap(items[first], items[last]);
      ++first;
    }
    return first;
  }
  // List.sort with comparator (a,b) => a.c[dim] < b.c[dim] ? -1 : 1.  The Dart
  // SDK sorts fewer than 32 elements by insertion sort; only ranges of <= 4
  // elements reach this in SAH mode.
  void sortRange(int first, int last, int dim) {
    for (int i = first + 1; i < last; ++i) {
      Item el = items[i];
      int j = i;
      while (j > first && !(items[j - 1].c[dim] < el.c[dim])) {
        items[j] = items[j - 1];
        --j;
      }
      items[j] = el;
    }
  }
};

}  // namespace

extern "C" int dr_bvh_build(const float* verts, uint64_t nverts, const uint32_t* tri_idx, uint64_t ntris,
                            int32_t max_prims_in_node, DrBvhNode* nodes_out, uint64_t* nnodes_out, uint32_t* order_out,
                            uint32_t* depth_out) {
  return dr_bvh_build_mixed(verts, nverts, tri_idx, ntris, nullptr, 0, max_prims_in_node, nodes_out, nnodes_out, order_out,
                            depth_out);
}

extern "C" int dr_bvh_build_mixed(const float* verts, uint64_t nverts, const uint32_t* tri_idx, uint64_t ntris,
                                  const float* quadric_bounds, uint64_t nquadrics, int32_t max_prims_in_node,
                                  DrBvhNode* nodes_out, uint64_t* nnodes_out, uint32_t* order_out, uint32_t* depth_out) {
  if (!nnodes_out) return DR_ERR_INVALID;
  *nnodes_out = 0;
  if (depth_out) *depth_out = 0;
  if (ntris == 0) return DR_OK;
  if (!tri_idx || !nodes_out || !order_out || ntris >= (1ull << 30)) return DR_ERR_INVALID;
  for (uint64_t i = 0; i < ntris; ++i) {
    if (tri_idx[3 * i] == DR_PRIM_QUADRIC) {
      if (!quadric_bounds || tri_idx[3 * i + 1] >= nquadrics) return DR_ERR_INVALID;
      continue;
    }
    if (!verts) return DR_ERR_INVALID;
    for (int k = 0; k < 3; ++k)
      if (tri_idx[3 * i + k] >= nverts) return DR_ERR_INVALID;
  }
  SahBuilder b;
  b.maxPrims = std::min(255, max_prims_in_node > 0 ? max_prims_in_node : 4);  // bvh_accel.dart:44
  const DrOpt env = dr_opt("DARTRAY_BUILD_THREADS");
  int hw = (int)std::thread::hardware_concurrency();
  b.maxThreads = env ? std::max(1, env.toInt(1)) : std::max(1, std::min(hw, 64));
  const bool dbg = dr_opt("DARTRAY_VERBOSE").toInt(0) >= 2;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  b.items.resize(ntris);
  auto fill = [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) {
      Item& it = b.items[i];
      it.prim = (uint32_t)i;
      it.b.reset();
      if (tri_idx[3 * i] == DR_PRIM_QUADRIC) {  // Shape.worldBound of a quadric, computed by the caller (shape.dart:37-39)
        const float* qb = quadric_bounds + 6 * (size_t)tri_idx[3 * i + 1];
        for (int k = 0; k < 3; ++k) { it.b.lo[k] = qb[k]; it.b.hi[k] = qb[3 + k]; }
      } else {
        for (int k = 0; k < 3; ++k) it.b.growPoint(verts + 3 * (size_t)tri_idx[3 * i + k]);  // Triangle.worldBound (triangle.dart:39-42)
      }
      for (int k = 0; k < 3; ++k)  // BBox.center: (pMin*0.5) + (pMax*0.5), each a Point (bbox.dart:66)
        it.c[k] = (float)((double)(float)((double)it.b.lo[k] * 0.5) + (double)(float)((double)it.b.hi[k] * 0.5));
    }
  };
  {
    const int nt = ntris > (1u << 20) ? std::min(b.maxThreads, 16) : 1;
    std::vector<std::thread> th;
    for (int k = 1; k < nt; ++k) th.emplace_back(fill, ntris * k / nt, ntris * (k + 1) / nt);
    fill(0, ntris / nt);
    for (auto& t : th) t.join();
  }
  const double t1 = now();
  b.tree = (TNode*)malloc(2 * ntris * sizeof(TNode));  // upper bound on the node count (one-primitive leaves)
  if (!b.tree) return DR_ERR_INVALID;
  const double t2 = now();
  const int root = b.build(0, (int)ntris, 0);
  const double t3 = now();
  if (b.maxLeaf.load() > 65535) return DR_ERR_UNSUPPORTED;
  b.emit(root, 0, 0, nodes_out, order_out);
  *nnodes_out = b.tree[root].subNodes;
  if (dbg)
    fprintf(stderr, "dr_bvh_build: %d threads, fill %.3f s, alloc %.3f s, build %.3f s, emit %.3f s\n", b.maxThreads, t1 - t0,
            t2 - t1, t3 - t2, now() - t3);
  if (depth_out) *depth_out = b.tree[root].subDepth;
  return DR_OK;
}
