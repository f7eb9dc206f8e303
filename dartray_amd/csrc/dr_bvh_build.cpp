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
#include <vector>

#include "../../include/dartray_hip.h"

namespace {

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
      lo[k] = std::min(lo[k], b.lo[k]);
      hi[k] = std::max(hi[k], b.hi[k]);
    }
  }
  void growPoint(const float* p) {
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::min(lo[k], p[k]);
      hi[k] = std::max(hi[k], p[k]);
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

struct TNode {
  Box b;
  int32_t left = -1, right = -1;
  uint32_t firstPrim = 0, nPrims = 0;
  uint8_t axis = 0;
};

class SahBuilder {
 public:
  std::vector<Item> items;
  std::vector<TNode> tree;
  std::vector<uint32_t> order;
  int maxPrims = 4;
  uint32_t depth = 0;

  int build(int start, int end, uint32_t d) {
    depth = std::max(depth, d);
    const int self = (int)tree.size();
    tree.emplace_back();
    Box bb;
    bb.reset();
    for (int i = start; i < end; ++i) bb.grow(items[i].b);
    const int n = end - start;
    if (n == 1) return leaf(self, start, end, bb);
    Box cb;
    cb.reset();
    for (int i = start; i < end; ++i) cb.growPoint(items[i].c);
    const int dim = cb.widestAxis();
    if (cb.hi[dim] == cb.lo[dim]) return leaf(self, start, end, bb);  // bvh_accel.dart:265-274
    int mid = (start + end) / 2;
    if (n <= 4) {
      sortRange(start, end, dim);  // nth_element (common.dart:289-297)
    } else {
      constexpr int NB = 12;
      int cnt[NB] = {0};
      Box bk[NB];
      for (auto& b : bk) b.reset();
      const double cmin = cb.lo[dim], cmax = cb.hi[dim];
      auto bucketOf = [&](const Item& it) {
        int b = (int)(NB * (((double)it.c[dim] - cmin) / (cmax - cmin)));
        return b == NB ? NB - 1 : b;
      };
      for (int i = start; i < end; ++i) {
        int b = bucketOf(items[i]);
        cnt[b]++;
        bk[b].grow(items[i].b);
      }
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
    const int r = build(mid, end, d + 1);  // right child first (bvh_accel.dart:407-411)
    const int l = build(start, mid, d + 1);
    TNode& t = tree[self];
    t.left = l;
    t.right = r;
    t.b = tree[l].b;
    t.b.grow(tree[r].b);
    t.axis = (uint8_t)dim;
    t.nPrims = 0;
    return self;
  }

 private:
  int leaf(int self, int start, int end, const Box& bb) {
    TNode& t = tree[self];
    t.firstPrim = (uint32_t)order.size();
    t.nPrims = (uint32_t)(end - start);
    t.b = bb;
    for (int i = start; i < end; ++i) order.push_back(items[i].prim);
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

uint32_t flatten(const std::vector<TNode>& tree, int n, DrBvhNode* out, uint32_t* next) {
  const uint32_t me = (*next)++;
  const TNode& t = tree[n];
  DrBvhNode& o = out[me];
  for (int k = 0; k < 3; ++k) {
    o.bmin[k] = t.b.lo[k];
    o.bmax[k] = t.b.hi[k];
  }
  o.pad = 0;
  if (t.nPrims > 0) {
    o.offset = t.firstPrim;
    o.nprims = (uint16_t)t.nPrims;
    o.axis = 0;
  } else {
    o.nprims = 0;
    o.axis = t.axis;
    flatten(tree, t.left, out, next);
    out[me].offset = flatten(tree, t.right, out, next);
  }
  return me;
}

}  // namespace

extern "C" int dr_bvh_build(const float* verts, uint64_t nverts, const uint32_t* tri_idx, uint64_t ntris,
                            int32_t max_prims_in_node, DrBvhNode* nodes_out, uint64_t* nnodes_out, uint32_t* order_out,
                            uint32_t* depth_out) {
  if (!nnodes_out) return DR_ERR_INVALID;
  *nnodes_out = 0;
  if (depth_out) *depth_out = 0;
  if (ntris == 0) return DR_OK;
  if (!verts || !tri_idx || !nodes_out || !order_out || ntris >= (1ull << 31)) return DR_ERR_INVALID;
  SahBuilder b;
  b.maxPrims = std::min(255, max_prims_in_node > 0 ? max_prims_in_node : 4);  // bvh_accel.dart:44
  b.items.resize(ntris);
  for (uint64_t i = 0; i < ntris; ++i) {
    Item& it = b.items[i];
    it.prim = (uint32_t)i;
    it.b.reset();
    for (int k = 0; k < 3; ++k) {
      if (tri_idx[3 * i + k] >= nverts) return DR_ERR_INVALID;
      it.b.growPoint(verts + 3 * (size_t)tri_idx[3 * i + k]);  // Triangle.worldBound (triangle.dart:39-42)
    }
    for (int k = 0; k < 3; ++k)  // BBox.center: (pMin*0.5) + (pMax*0.5), each a Point (bbox.dart:66)
      it.c[k] = (float)((double)(float)((double)it.b.lo[k] * 0.5) + (double)(float)((double)it.b.hi[k] * 0.5));
  }
  b.tree.reserve(2 * ntris);
  b.order.reserve(ntris);
  b.build(0, (int)ntris, 0);
  for (const TNode& t : b.tree)
    if (t.nPrims > 65535) return DR_ERR_UNSUPPORTED;
  uint32_t next = 0;
  flatten(b.tree, 0, nodes_out, &next);
  *nnodes_out = next;
  memcpy(order_out, b.order.data(), ntris * sizeof(uint32_t));
  if (depth_out) *depth_out = b.depth;
  return DR_OK;
}
