// dr_bvh_device.hip -- BVHAccel's constructor on the GPU: the SAH build of
// accelerators/bvh_accel.dart:41-91,228-437 (12 buckets, f32 bucket costs :345, right child first :407-411, flattening
// :419-437) with core/common.dart:256-297 (`partition`, sort-based `nth_element`), byte-identical to the host builder
// dr_bvh_build (dr_bvh_build.cpp) and hence to the oracle's serial restatement and the Python one.
//
// Why the result does not depend on the order of evaluation:
//  * a node's bounds, centroid bounds and the 12 buckets' counts / boxes are min / max / integer sums over the node's
//    items (Dart's Math.min / Math.max order -0.0 below +0.0, which is what the integer keys below do);
//  * the reference's two-pointer partition exchanges the k-th item of the left part that fails the predicate with the
//    k-th item from the END that passes it: with T = number of passing items, the left part is [start, start + T), the
//    failing items of it in ascending order are L[0..m), the passing items behind it in DESCENDING order are R[0..m),
//    and the partition is exactly the m swaps L[k] <-> R[k] -- ranks that follow from one prefix sum of the predicate;
//  * sub-trees only meet in `orderedPrims` and the node numbering, both functions of the finished tree.
// Structure: (1) item boxes / centroids, one thread per primitive; (2) the top of the tree level by level: every
// segment (node) above DR_BUILD_SMALL items is worked on by all the threads its items take -- wave-level reductions
// into per-segment accumulators, a device-wide prefix sum, pair swaps; (3) every segment of at most DR_BUILD_SMALL
// items is finished by ONE thread running the reference's recursion literally (serial partition, insertion sort);
// (4) sub-tree sizes bottom-up, node indices / primitive offsets top-down, nodes and order written out.  Scratch-tree
// slots are arithmetic as in the host builder: the node of [start, end) at `self`, its left sub-tree from self + 1,
// its right one from self + 2 (mid - start).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dartray_hip.h"

#include "dr_options.h"  // dr_opt: dr_set_option's value, else the environment's (by value)
#include "dr_device.h"

int dr_fail(int code, const std::string& msg);  // dr_api.hip

#ifndef DR_BUILD_SMALL
#define DR_BUILD_SMALL 64  // segments of at most this many items are finished by one thread
#endif
#define BNB 12              // SAH buckets (bvh_accel.dart:319)
#define B_INVALID 0xffffffffu

namespace {

// ---- order-preserving integer keys of f32 values: -inf < ... < -0.0 < +0.0 < ... < +inf -----------------------------
DR_DEV int fkey(float x) {
  const int b = __float_as_int(x);
  return b ^ ((b >> 31) & 0x7fffffff);
}
DR_DEV float fval(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
// Math.min / Math.max of dart:math on doubles holding f32 values: the lesser / greater, -0.0 below +0.0
DR_DEV float dmin(float a, float b) { return fkey(b) < fkey(a) ? b : a; }
DR_DEV float dmax(float a, float b) { return fkey(b) > fkey(a) ? b : a; }

struct Items {  // _BVHPrimitiveInfo (bvh_accel.dart:490-501), structure of arrays
  uint32_t* prim;
  float* f[9];  // centroid xyz, bounds.pMin xyz, bounds.pMax xyz
};
struct DBox {
  float lo[3], hi[3];
  DR_DEV void reset() {
    for (int k = 0; k < 3; ++k) {
      lo[k] = __int_as_float(0x7f800000);
      hi[k] = __int_as_float((int)0xff800000);
    }
  }
  DR_DEV void grow(const DBox& b) {
    for (int k = 0; k < 3; ++k) {
      lo[k] = dmin(lo[k], b.lo[k]);
      hi[k] = dmax(hi[k], b.hi[k]);
    }
  }
  DR_DEV double area() const {  // BBox.surfaceArea (bbox.dart:163-166): d = pMax - pMin is a Vector (f32)
    const double dx = (double)(float)((double)hi[0] - (double)lo[0]);
    const double dy = (double)(float)((double)hi[1] - (double)lo[1]);
    const double dz = (double)(float)((double)hi[2] - (double)lo[2]);
    return 2.0 * (dx * dy + dx * dz + dy * dz);
  }
  DR_DEV int widestAxis() const {  // BBox.maximumExtent (bbox.dart:173-182)
    const float dx = (float)((double)hi[0] - (double)lo[0]);
    const float dy = (float)((double)hi[1] - (double)lo[1]);
    const float dz = (float)((double)hi[2] - (double)lo[2]);
    if (dx > dy && dx > dz) return 0;
    return dy > dz ? 1 : 2;
  }
};
struct TNode {
  DBox b;
  int32_t left, right;         // scratch-tree slots of the children (-1: leaf)
  int32_t itemStart, itemEnd;  // leaf: its primitives are items[itemStart, itemEnd)
  uint32_t nPrims, axis;
  uint32_t subNodes, subPrims, subDepth;
  uint32_t index, primBase;    // position in the flattened array / in orderedPrims
};
struct Seg {  // a node of the level-by-level part
  uint32_t start, end, slot;
  uint32_t mid;       // after the partition
  uint32_t dim;
  uint32_t action;    // 0 leaf, 1 split
  uint32_t best;      // SAH bucket
  uint32_t child[2];  // entries of the next level's list (B_INVALID: small sub-tree or none)
  float cmin, cmax;
  uint32_t m;         // swaps of the partition
  uint32_t nTrue;
};
#define ACC_WORDS 12             // bb.lo, bb.hi, cb.lo, cb.hi as keys
#define SAH_WORDS (BNB * 7)      // per bucket: count, box keys

DR_DEV int bucketOf(float c, double cmin, double cmax) {  // bvh_accel.dart:323-327
  int b = (int)(BNB * (((double)c - cmin) / (cmax - cmin)));
  return b == BNB ? BNB - 1 : b;
}

// ---- (1) items -------------------------------------------------------------------------------------------------
__global__ void kb_fill(const float* verts, const uint32_t* idx, const float* qb, uint32_t n, Items it) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  DBox b;
  b.reset();
  const uint32_t a = idx[3 * (size_t)i];
  if (a == DR_PRIM_QUADRIC) {  // Shape.worldBound of a quadric, computed by the caller (shape.dart:37-39)
    const float* q = qb + 6 * (size_t)idx[3 * (size_t)i + 1];
    for (int k = 0; k < 3; ++k) {
      b.lo[k] = q[k];
      b.hi[k] = q[3 + k];
    }
  } else {
    for (int v = 0; v < 3; ++v) {  // Triangle.worldBound (triangle.dart:39-42)
      const float* p = verts + 3 * (size_t)idx[3 * (size_t)i + v];
      for (int k = 0; k < 3; ++k) {
        b.lo[k] = dmin(b.lo[k], p[k]);
        b.hi[k] = dmax(b.hi[k], p[k]);
      }
    }
  }
  it.prim[i] = i;
  for (int k = 0; k < 3; ++k) {
    // BBox.center: (pMin * 0.5) + (pMax * 0.5), each a Point (bbox.dart:66)
    it.f[k][i] = (float)((double)(float)((double)b.lo[k] * 0.5) + (double)(float)((double)b.hi[k] * 0.5));
    it.f[3 + k][i] = b.lo[k];
    it.f[6 + k][i] = b.hi[k];
  }
}

// ---- (2) the level-by-level part ----------------------------------------------------------------------------------
#define ROWS 32  // rows of 64 items a wave walks per launch (kb_bounds / kb_buckets)
__global__ void kb_init_acc(int* acc, int* sah, uint32_t nseg) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nseg * ACC_WORDS) {
    const uint32_t w = i % ACC_WORDS;
    acc[i] = ((w / 3) & 1) ? fkey(__int_as_float((int)0xff800000)) : fkey(__int_as_float(0x7f800000));
  }
  if (i < nseg * SAH_WORDS) {
    const uint32_t w = i % 7;
    sah[i] = w == 0 ? 0 : (w <= 3 ? fkey(__int_as_float(0x7f800000)) : fkey(__int_as_float((int)0xff800000)));
  }
}
// bounds and centroid bounds of every listed segment (bvh_accel.dart:235-241,254-258).  A lane keeps the running
// minima / maxima of the segment its items belong to and hands them over (12 atomics) when the segment changes; at
// the end a wave whose lanes all hold the same segment -- the usual case above a few thousand items -- combines them
// first, so that the big segments of the top levels receive one set of atomics per wave, not per item.
__global__ void __launch_bounds__(256) kb_bounds(Items it, const uint32_t* seg, uint32_t n, int* acc) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  const uint64_t base = (uint64_t)wave * (64 * ROWS);
  uint32_t cur = B_INVALID;
  int v[ACC_WORDS];
  auto flush = [&]() {
    if (cur == B_INVALID) return;
    int* a = acc + (size_t)cur * ACC_WORDS;
    for (int k = 0; k < 3; ++k) {
      atomicMin(a + k, v[k]);
      atomicMax(a + 3 + k, v[3 + k]);
      atomicMin(a + 6 + k, v[6 + k]);
      atomicMax(a + 9 + k, v[9 + k]);
    }
  };
  for (int r = 0; r < ROWS; ++r) {
    const uint64_t p = base + (uint64_t)r * 64 + lane;
    if (p >= n) break;
    const uint32_t s = seg[p];
    if (s == B_INVALID) continue;
    if (s != cur) {
      flush();
      cur = s;
      for (int k = 0; k < 3; ++k) {
        v[k] = v[6 + k] = fkey(__int_as_float(0x7f800000));
        v[3 + k] = v[9 + k] = fkey(__int_as_float((int)0xff800000));
      }
    }
    for (int k = 0; k < 3; ++k) {
      const int lo = fkey(it.f[3 + k][p]), hi = fkey(it.f[6 + k][p]), c = fkey(it.f[k][p]);
      v[k] = min(v[k], lo);
      v[3 + k] = max(v[3 + k], hi);
      v[6 + k] = min(v[6 + k], c);
      v[9 + k] = max(v[9 + k], c);
    }
  }
  // all lanes on one segment: combine across the wave, one lane reports
  const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
  if (__all(cur == first) && first != B_INVALID) {
    for (int off = 32; off > 0; off >>= 1)
      for (int k = 0; k < ACC_WORDS; ++k) {
        const int o = __shfl_xor(v[k], off);
        v[k] = ((k / 3) & 1) ? max(v[k], o) : min(v[k], o);
      }
    if (lane == 0) flush();
  } else {
    flush();
  }
}
// per segment: the decisions of bvh_accel.dart:245-316 that do not need the buckets
__global__ void kb_decide(Seg* segs, uint32_t nseg, const int* acc, TNode* tree) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nseg) return;
  Seg& s = segs[e];
  const int* a = acc + (size_t)e * ACC_WORDS;
  DBox bb, cb;
  for (int k = 0; k < 3; ++k) {
    bb.lo[k] = fval(a[k]);
    bb.hi[k] = fval(a[3 + k]);
    cb.lo[k] = fval(a[6 + k]);
    cb.hi[k] = fval(a[9 + k]);
  }
  TNode& t = tree[s.slot];
  t.b = bb;
  const int dim = cb.widestAxis();
  s.dim = (uint32_t)dim;
  s.cmin = cb.lo[dim];
  s.cmax = cb.hi[dim];
  s.action = (cb.hi[dim] == cb.lo[dim]) ? 0u : 1u;  // all centroids coincide on the widest axis: a leaf (:265-274)
  s.child[0] = s.child[1] = B_INVALID;
  s.mid = s.start;
  s.m = 0;
}
// the 12 buckets' counts and bounds (bvh_accel.dart:319-341): rows of 64 items that lie in one segment go through the
// wave's LDS copy of that segment's buckets (ds atomics), which is handed over when the segment changes
__global__ void __launch_bounds__(256) kb_buckets(Items it, const uint32_t* seg, uint32_t n, const Seg* segs, int* sah) {
  __shared__ int s_bk[4][SAH_WORDS];
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t base = (uint64_t)wave * (64 * ROWS);
  int* lb = s_bk[wv];
  uint32_t cur = B_INVALID;  // wave-uniform: the segment the LDS copy belongs to
  auto clearLds = [&]() {
    for (uint32_t i = lane; i < SAH_WORDS; i += 64u) {
      const uint32_t w = i % 7;
      lb[i] = w == 0 ? 0 : (w <= 3 ? fkey(__int_as_float(0x7f800000)) : fkey(__int_as_float((int)0xff800000)));
    }
  };
  auto flushLds = [&]() {
    if (cur == B_INVALID) return;
    int* g = sah + (size_t)cur * SAH_WORDS;
    for (uint32_t i = lane; i < SAH_WORDS; i += 64u) {
      const uint32_t w = i % 7;
      const int x = lb[i];
      if (w == 0) {
        if (x) atomicAdd(g + i, x);
      } else if (w <= 3) {
        atomicMin(g + i, x);
      } else {
        atomicMax(g + i, x);
      }
    }
  };
  for (int r = 0; r < ROWS; ++r) {
    const uint64_t p0 = base + (uint64_t)r * 64;
    if (p0 >= n) break;
    const uint64_t p = p0 + lane;
    const uint32_t s = p < n ? seg[p] : B_INVALID;
    const bool on = s != B_INVALID && segs[s].action == 1u;
    const uint32_t sFirst = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
    const bool uniform = __all(on && s == sFirst);
    if (uniform) {
      if (sFirst != cur) {
        flushLds();
        cur = sFirst;
        clearLds();
      }
    }
    if (!on) continue;
    const Seg& sg = segs[s];
    const int b = bucketOf(it.f[sg.dim][p], (double)sg.cmin, (double)sg.cmax);
    int* dst = uniform ? lb + b * 7 : sah + (size_t)s * SAH_WORDS + b * 7;
    atomicAdd(dst, 1);
    for (int k = 0; k < 3; ++k) {
      atomicMin(dst + 1 + k, fkey(it.f[3 + k][p]));
      atomicMax(dst + 4 + k, fkey(it.f[6 + k][p]));
    }
  }
  flushLds();
}
// per segment: bucket costs, the cheapest split, split or leaf (bvh_accel.dart:343-385)
__global__ void kb_cost(Seg* segs, uint32_t nseg, const int* sah, const TNode* tree, int maxPrims) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nseg) return;
  Seg& s = segs[e];
  if (s.action != 1u) return;
  const int* g = sah + (size_t)e * SAH_WORDS;
  int cnt[BNB];
  DBox bk[BNB];
  for (int b = 0; b < BNB; ++b) {
    cnt[b] = g[b * 7];
    for (int k = 0; k < 3; ++k) {
      bk[b].lo[k] = fval(g[b * 7 + 1 + k]);
      bk[b].hi[k] = fval(g[b * 7 + 4 + k]);
    }
  }
  const int n = (int)(s.end - s.start);
  const double total = tree[s.slot].b.area();
  float cost[BNB - 1];  // Float32List (bvh_accel.dart:345)
  for (int sp = 0; sp < BNB - 1; ++sp) {
    DBox l, r;
    l.reset();
    r.reset();
    int nl = 0, nr = 0;
    for (int j = 0; j <= sp; ++j) {
      l.grow(bk[j]);
      nl += cnt[j];
    }
    for (int j = sp + 1; j < BNB; ++j) {
      r.grow(bk[j]);
      nr += cnt[j];
    }
    cost[sp] = (float)(0.125 + (nl * l.area() + nr * r.area()) / total);
  }
  int best = 0;
  double bestCost = cost[0];
  for (int sp = 1; sp < BNB - 1; ++sp)
    if ((double)cost[sp] < bestCost) {
      bestCost = cost[sp];
      best = sp;
    }
  s.best = (uint32_t)best;
  if (!(n > maxPrims || bestCost < n)) s.action = 0u;  // a leaf (:377-385)
}
// the predicate of the partition (bucket <= best, bvh_accel.dart:379-381) for every item of a splitting segment
__global__ void kb_flags(Items it, const uint32_t* seg, uint32_t n, const Seg* segs, uint32_t* flags) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n) return;
  uint32_t f = 0;
  if (p < n) {
    const uint32_t s = seg[p];
    if (s != B_INVALID && segs[s].action == 1u) {
      const Seg& sg = segs[s];
      f = (uint32_t)bucketOf(it.f[sg.dim][p], (double)sg.cmin, (double)sg.cmax) <= sg.best ? 1u : 0u;
    }
  }
  flags[p] = f;  // flags[n] = 0: the scan's last entry is the total
}
__global__ void kb_mid(Seg* segs, uint32_t nseg, const uint32_t* pre) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nseg) return;
  Seg& s = segs[e];
  if (s.action != 1u) return;
  const uint32_t T = pre[s.end] - pre[s.start];
  s.nTrue = T;
  s.mid = s.start + T;
  s.m = T - (pre[s.mid] - pre[s.start]);  // failing items of the left part == passing items behind it
}
// partition (common.dart:256-287) as ranks: the failing items of the left part in ascending order, the passing items
// behind it in ascending order (the swaps pair them in opposite directions)
__global__ void kb_rank(const uint32_t* seg, uint32_t n, const Seg* segs, const uint32_t* flags, const uint32_t* pre, uint32_t* tmpL,
                        uint32_t* tmpR) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t s = seg[p];
  if (s == B_INVALID) return;
  const Seg& sg = segs[s];
  if (sg.action != 1u || sg.m == 0u) return;
  const uint32_t tb = pre[p] - pre[sg.start];
  if (p < sg.mid) {
    if (!flags[p]) tmpL[sg.start + ((p - sg.start) - tb)] = p;
  } else if (flags[p]) {
    tmpR[sg.start + (tb - (sg.nTrue - sg.m))] = p;
  }
}
__global__ void kb_swap(Items it, const uint32_t* seg, uint32_t n, const Seg* segs, const uint32_t* tmpL, const uint32_t* tmpR) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t s = seg[p];
  if (s == B_INVALID) return;
  const Seg& sg = segs[s];
  if (sg.action != 1u) return;
  const uint32_t k = p - sg.start;
  if (k >= sg.m) return;
  const uint32_t a = tmpL[sg.start + k], b = tmpR[sg.start + (sg.m - 1u - k)];
  const uint32_t pa = it.prim[a];
  it.prim[a] = it.prim[b];
  it.prim[b] = pa;
  for (int j = 0; j < 9; ++j) {
    const float x = it.f[j][a];
    it.f[j][a] = it.f[j][b];
    it.f[j][b] = x;
  }
}
// per segment: its scratch-tree node; its children join the next level's list or the list of small sub-trees
__global__ void kb_children(Seg* segs, uint32_t nseg, TNode* tree, Seg* next, uint32_t* nNext, uint3* small, uint32_t* nSmall,
                            uint32_t* maxLeaf) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nseg) return;
  Seg& s = segs[e];
  TNode& t = tree[s.slot];
  if (s.action == 0u) {
    t.left = t.right = -1;
    t.axis = 0;
    t.itemStart = (int32_t)s.start;
    t.itemEnd = (int32_t)s.end;
    t.nPrims = s.end - s.start;
    t.subNodes = 1;
    t.subPrims = t.nPrims;
    t.subDepth = 0;
    atomicMax(maxLeaf, t.nPrims);
    return;
  }
  const uint32_t leftSlot = s.slot + 1u, rightSlot = s.slot + 2u * (s.mid - s.start);
  t.left = (int32_t)leftSlot;
  t.right = (int32_t)rightSlot;
  t.axis = s.dim;
  t.nPrims = 0;
  t.itemStart = t.itemEnd = 0;
  const uint32_t cs[2] = {s.start, s.mid}, ce[2] = {s.mid, s.end}, sl[2] = {leftSlot, rightSlot};
  for (int c = 0; c < 2; ++c) {
    if (ce[c] - cs[c] > (uint32_t)DR_BUILD_SMALL) {
      const uint32_t j = atomicAdd(nNext, 1u);
      Seg& d = next[j];
      d.start = cs[c];
      d.end = ce[c];
      d.slot = sl[c];
      s.child[c] = j;
    } else {
      const uint32_t j = atomicAdd(nSmall, 1u);
      small[j] = make_uint3(cs[c], ce[c], sl[c]);
      s.child[c] = B_INVALID;
    }
  }
}
__global__ void kb_reassign(uint32_t* seg, uint32_t n, const Seg* segs) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t s = seg[p];
  if (s == B_INVALID) return;
  const Seg& sg = segs[s];
  seg[p] = sg.action == 1u ? sg.child[p < sg.mid ? 0 : 1] : B_INVALID;
}

// ---- (3) a small sub-tree, by one thread: the recursion of bvh_accel.dart:228-417 as it stands ----------------------
struct Frame {
  uint32_t start, end, self, mid;
  uint32_t state;
};
DR_DEV void item_swap(const Items& it, uint32_t a, uint32_t b) {
  const uint32_t pa = it.prim[a];
  it.prim[a] = it.prim[b];
  it.prim[b] = pa;
  for (int j = 0; j < 9; ++j) {
    const float x = it.f[j][a];
    it.f[j][a] = it.f[j][b];
    it.f[j][b] = x;
  }
}
DR_DEV void make_leaf(TNode& t, uint32_t start, uint32_t end, const DBox& bb, uint32_t* maxLeaf) {
  t.left = t.right = -1;
  t.axis = 0;
  t.itemStart = (int32_t)start;
  t.itemEnd = (int32_t)end;
  t.nPrims = end - start;
  t.b = bb;
  t.subNodes = 1;
  t.subPrims = t.nPrims;
  t.subDepth = 0;
  if (t.nPrims > 1u) atomicMax(maxLeaf, t.nPrims);
}
__global__ void __launch_bounds__(64) kb_small(Items it, const uint3* small, uint32_t nSmall, TNode* tree, int maxPrims, uint32_t* maxLeaf) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nSmall) return;
  Frame st[DR_BUILD_SMALL + 2];
  int sp = 0;
  st[0] = Frame{small[i].x, small[i].y, small[i].z, 0u, 0u};
  while (sp >= 0) {
    Frame& f = st[sp];
    TNode& t = tree[f.self];
    if (f.state == 0u) {
      const uint32_t start = f.start, end = f.end, n = end - start;
      DBox bb, cb;
      bb.reset();
      cb.reset();
      for (uint32_t p = start; p < end; ++p)
        for (int k = 0; k < 3; ++k) {
          bb.lo[k] = dmin(bb.lo[k], it.f[3 + k][p]);
          bb.hi[k] = dmax(bb.hi[k], it.f[6 + k][p]);
          cb.lo[k] = dmin(cb.lo[k], it.f[k][p]);
          cb.hi[k] = dmax(cb.hi[k], it.f[k][p]);
        }
      if (n == 1u) {
        make_leaf(t, start, end, bb, maxLeaf);
        --sp;
        continue;
      }
      const int dim = cb.widestAxis();
      if (cb.hi[dim] == cb.lo[dim]) {
        make_leaf(t, start, end, bb, maxLeaf);
        --sp;
        continue;
      }
      uint32_t mid = (start + end) / 2u;
      const float* cd = it.f[dim];
      if (n <= 4u) {
        // nth_element == List.sort with (a, b) => a.c[dim] < b.c[dim] ? -1 : 1: an insertion sort below 32 elements
        // (common.dart:289-297); an element goes in front of the equal ones before it
        for (uint32_t a = start + 1u; a < end; ++a) {
          uint32_t j = a;
          while (j > start && !(cd[j - 1u] < cd[j])) {
            item_swap(it, j - 1u, j);
            --j;
          }
        }
      } else {
        const double cmin = cb.lo[dim], cmax = cb.hi[dim];
        int cnt[BNB];
        DBox bk[BNB];
        for (int b = 0; b < BNB; ++b) {
          cnt[b] = 0;
          bk[b].reset();
        }
        for (uint32_t p = start; p < end; ++p) {
          const int b = bucketOf(cd[p], cmin, cmax);
          cnt[b]++;
          for (int k = 0; k < 3; ++k) {
            bk[b].lo[k] = dmin(bk[b].lo[k], it.f[3 + k][p]);
            bk[b].hi[k] = dmax(bk[b].hi[k], it.f[6 + k][p]);
          }
        }
        float cost[BNB - 1];
        const double total = bb.area();
        for (int s = 0; s < BNB - 1; ++s) {
          DBox l, r;
          l.reset();
          r.reset();
          int nl = 0, nr = 0;
          for (int j = 0; j <= s; ++j) {
            l.grow(bk[j]);
            nl += cnt[j];
          }
          for (int j = s + 1; j < BNB; ++j) {
            r.grow(bk[j]);
            nr += cnt[j];
          }
          cost[s] = (float)(0.125 + (nl * l.area() + nr * r.area()) / total);
        }
        int best = 0;
        double bestCost = cost[0];
        for (int s = 1; s < BNB - 1; ++s)
          if ((double)cost[s] < bestCost) {
            bestCost = cost[s];
            best = s;
          }
        if ((int)n > maxPrims || bestCost < (double)n) {
          // partition (common.dart:256-287), literally
          uint32_t first = start, last = end;
          auto pred = [&](uint32_t p) { return bucketOf(cd[p], cmin, cmax) <= best; };
          bool doneP = false;
          while (first < last && !doneP) {
            while (pred(first)) {
              if (++first == last) {
                doneP = true;
                break;
              }
            }
            if (doneP) break;
            for (;;) {
              if (--last == first) {
                doneP = true;
                break;
              }
              if (pred(last)) break;
            }
            if (doneP) break;
            item_swap(it, first, last);
            ++first;
          }
          mid = first;
        } else {
          make_leaf(t, start, end, bb, maxLeaf);
          --sp;
          continue;
        }
      }
      t.b = bb;
      t.axis = (uint32_t)dim;
      f.mid = mid;
      f.state = 1u;
      st[++sp] = Frame{mid, end, f.self + 2u * (mid - start), 0u, 0u};  // the right child first (:407-411)
    } else if (f.state == 1u) {
      f.state = 2u;
      st[++sp] = Frame{f.start, f.mid, f.self + 1u, 0u, 0u};
    } else {
      const uint32_t l = f.self + 1u, r = f.self + 2u * (f.mid - f.start);
      t.left = (int32_t)l;
      t.right = (int32_t)r;
      t.nPrims = 0;
      t.itemStart = t.itemEnd = 0;
      t.subNodes = 1u + tree[l].subNodes + tree[r].subNodes;
      t.subPrims = tree[l].subPrims + tree[r].subPrims;
      t.subDepth = 1u + max(tree[l].subDepth, tree[r].subDepth);
      --sp;
    }
  }
}

// ---- (4) sizes bottom-up, numbering top-down, output ---------------------------------------------------------------
__global__ void kb_sizes(const Seg* segs, uint32_t nseg, TNode* tree) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nseg) return;
  const Seg& s = segs[e];
  if (s.action != 1u) return;
  TNode& t = tree[s.slot];
  const TNode &l = tree[t.left], &r = tree[t.right];
  t.subNodes = 1u + l.subNodes + r.subNodes;
  t.subPrims = l.subPrims + r.subPrims;
  t.subDepth = 1u + max(l.subDepth, r.subDepth);
}
DR_DEV void write_node(const TNode& t, const Items& it, DrBvhNode* out, uint32_t* order) {
  // _flattenBVHTree (bvh_accel.dart:419-437) + orderedPrims (:407-411: the right sub-tree's primitives come first)
  DrBvhNode o;
  for (int k = 0; k < 3; ++k) {
    o.bmin[k] = t.b.lo[k];
    o.bmax[k] = t.b.hi[k];
  }
  o.pad = 0;
  if (t.nPrims > 0u) {
    o.offset = t.primBase;
    o.nprims = (uint16_t)t.nPrims;
    o.axis = 0;
    for (int32_t i = t.itemStart; i < t.itemEnd; ++i) order[t.primBase + (uint32_t)(i - t.itemStart)] = it.prim[i];
  } else {
    o.offset = 0;  // second child: set by the caller
    o.nprims = 0;
    o.axis = (uint8_t)t.axis;
  }
  out[t.index] = o;
}
DR_DEV void number_children(TNode* tree, const TNode& t, DrBvhNode* out) {
  TNode &l = tree[t.left], &r = tree[t.right];
  l.index = t.index + 1u;
  r.index = t.index + 1u + l.subNodes;
  r.primBase = t.primBase;
  l.primBase = t.primBase + r.subPrims;
  out[t.index].offset = r.index;
}
__global__ void kb_number(const Seg* segs, uint32_t nseg, TNode* tree, Items it, DrBvhNode* out, uint32_t* order) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nseg) return;
  const TNode& t = tree[segs[e].slot];
  write_node(t, it, out, order);
  if (t.nPrims == 0u) number_children(tree, t, out);
}
__global__ void __launch_bounds__(64) kb_emit_small(const uint3* small, uint32_t nSmall, TNode* tree, Items it, DrBvhNode* out, uint32_t* order) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nSmall) return;
  uint32_t st[DR_BUILD_SMALL + 2];
  int sp = 0;
  st[0] = small[i].z;
  while (sp >= 0) {
    const TNode& t = tree[st[sp--]];
    write_node(t, it, out, order);
    if (t.nPrims == 0u) {
      number_children(tree, t, out);
      st[++sp] = (uint32_t)t.right;
      st[++sp] = (uint32_t)t.left;
    }
  }
}

template <class T>
struct Buf {
  T* p = nullptr;
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
  ~Buf() {
    if (p) (void)hipFree(p);
  }
};
#define BT(x)                                                                                          \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) return dr_fail(DR_ERR_HIP, std::string("dr_bvh_build_device: ") + hipGetErrorString(e_)); \
  } while (0)

}  // namespace

extern "C" int dr_bvh_build_device(const float* verts, uint64_t nverts, const uint32_t* tri_idx, uint64_t ntris, const float* quadric_bounds,
                                   uint64_t nquadrics, int32_t max_prims_in_node, DrBvhNode* nodes_out, uint64_t* nnodes_out,
                                   uint32_t* order_out, uint32_t* depth_out) {
  if (!nnodes_out) return dr_fail(DR_ERR_INVALID, "dr_bvh_build_device: nnodes_out is null");
  *nnodes_out = 0;
  if (depth_out) *depth_out = 0;
  if (ntris == 0) return DR_OK;
  if (!tri_idx || !nodes_out || !order_out || ntris >= (1ull << 30)) return dr_fail(DR_ERR_INVALID, "dr_bvh_build_device: bad argument");
  for (uint64_t i = 0; i < ntris; ++i) {
    if (tri_idx[3 * i] == DR_PRIM_QUADRIC) {
      if (!quadric_bounds || tri_idx[3 * i + 1] >= nquadrics) return dr_fail(DR_ERR_INVALID, "dr_bvh_build_device: quadric index out of range");
      continue;
    }
    if (!verts) return dr_fail(DR_ERR_INVALID, "dr_bvh_build_device: verts is null");
    for (int k = 0; k < 3; ++k)
      if (tri_idx[3 * i + k] >= nverts) return dr_fail(DR_ERR_INVALID, "dr_bvh_build_device: vertex index out of range");
  }
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return dr_fail(DR_ERR_NO_DEVICE, "dr_bvh_build_device before dr_init");
  const int maxPrims = std::min(255, max_prims_in_node > 0 ? max_prims_in_node : 4);  // bvh_accel.dart:44
  const bool dbg = dr_opt("DARTRAY_VERBOSE").toInt(0) >= 2;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  const uint32_t n = (uint32_t)ntris;
  hipStream_t s = 0;

  Buf<float> dV, dQ, dF;
  Buf<uint32_t> dI, dPrim, dSeg, dFlags, dPre, dTmpL, dTmpR, dCounters, dOrder;
  Buf<TNode> dTree;
  Buf<DrBvhNode> dOut;
  BT(dV.alloc(3 * std::max<uint64_t>(nverts, 1)));
  BT(dI.alloc(3 * (size_t)n));
  BT(dQ.alloc(6 * std::max<uint64_t>(nquadrics, 1)));
  if (nverts) BT(hipMemcpyAsync(dV.p, verts, 3 * nverts * sizeof(float), hipMemcpyHostToDevice, s));
  BT(hipMemcpyAsync(dI.p, tri_idx, 3 * (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  if (nquadrics) BT(hipMemcpyAsync(dQ.p, quadric_bounds, 6 * nquadrics * sizeof(float), hipMemcpyHostToDevice, s));
  BT(dF.alloc(9 * (size_t)n));
  BT(dPrim.alloc(n));
  BT(dSeg.alloc(n));
  BT(dFlags.alloc((size_t)n + 1));
  BT(dPre.alloc((size_t)n + 1));
  BT(dTmpL.alloc(n));
  BT(dTmpR.alloc(n));
  BT(dCounters.alloc(8));
  BT(dTree.alloc(2 * (size_t)n));
  Items it;
  it.prim = dPrim.p;
  for (int k = 0; k < 9; ++k) it.f[k] = dF.p + (size_t)k * n;
  const unsigned gItems = (n + 255) / 256, gItems1 = (n + 256) / 256;
  const unsigned gWaves = (unsigned)(((uint64_t)n + 64 * ROWS - 1) / (64 * ROWS) * 64 + 255) / 256;
  hipLaunchKernelGGL(kb_fill, dim3(gItems), dim3(256), 0, s, dV.p, dI.p, dQ.p, n, it);
  BT(hipMemsetAsync(dCounters.p, 0, 8 * sizeof(uint32_t), s));
  uint32_t* const dNNext = dCounters.p;
  uint32_t* const dNSmall = dCounters.p + 1;
  uint32_t* const dMaxLeaf = dCounters.p + 2;
  const double t1 = now();

  // small sub-trees: at most one per DR_BUILD_SMALL / 2 items... bounded by n (every item in at most one)
  Buf<uint3> dSmall;
  BT(dSmall.alloc(n));
  // the per-level lists are kept: sizes run bottom-up, numbering top-down
  struct LevelLists : std::vector<Seg*> {  // freed on every return path
    ~LevelLists() {
      for (Seg* p : *this) (void)hipFree(p);
    }
  } levels;
  std::vector<uint32_t> levelCount;
  auto freeLevels = [&]() {
    for (Seg* p : levels) (void)hipFree(p);
    levels.clear();
  };
  size_t scanBytes = 0;
  BT(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, dFlags.p, dPre.p, (int)(n + 1), s));
  Buf<unsigned char> dScanTmp;
  BT(dScanTmp.alloc(scanBytes));

  uint32_t nseg = 0, nSmallHost = 0;
  if (n > (uint32_t)DR_BUILD_SMALL) {
    Seg root;
    memset(&root, 0, sizeof(root));
    root.start = 0;
    root.end = n;
    root.slot = 0;
    Seg* l0 = nullptr;
    BT(hipMalloc((void**)&l0, sizeof(Seg)));
    levels.push_back(l0);
    BT(hipMemcpyAsync(l0, &root, sizeof(Seg), hipMemcpyHostToDevice, s));
    BT(hipMemsetAsync(dSeg.p, 0, (size_t)n * sizeof(uint32_t), s));  // every item in entry 0
    nseg = 1;
  } else {
    const uint3 one = make_uint3(0u, n, 0u);
    BT(hipMemcpyAsync(dSmall.p, &one, sizeof(one), hipMemcpyHostToDevice, s));
    nSmallHost = 1;
    const uint32_t ns = 1;
    BT(hipMemcpyAsync(dNSmall, &ns, sizeof(ns), hipMemcpyHostToDevice, s));
  }
  Buf<int> dAcc, dSah;
  size_t accCap = 0;
  while (nseg > 0) {
    if (levels.size() > 4096) {
      freeLevels();
      return dr_fail(DR_ERR_UNSUPPORTED, "dr_bvh_build_device: tree deeper than 4096 levels");
    }
    Seg* cur = levels.back();
    levelCount.push_back(nseg);
    if (nseg > accCap) {
      if (dAcc.p) (void)hipFree(dAcc.p);
      if (dSah.p) (void)hipFree(dSah.p);
      dAcc.p = nullptr;
      dSah.p = nullptr;
      accCap = (size_t)nseg * 2;
      BT(dAcc.alloc(accCap * ACC_WORDS));
      BT(dSah.alloc(accCap * SAH_WORDS));
    }
    const unsigned gSeg = (nseg + 255) / 256;
    hipLaunchKernelGGL(kb_init_acc, dim3((nseg * SAH_WORDS + 255) / 256), dim3(256), 0, s, dAcc.p, dSah.p, nseg);
    hipLaunchKernelGGL(kb_bounds, dim3(gWaves), dim3(256), 0, s, it, dSeg.p, n, dAcc.p);
    hipLaunchKernelGGL(kb_decide, dim3(gSeg), dim3(256), 0, s, cur, nseg, dAcc.p, dTree.p);
    hipLaunchKernelGGL(kb_buckets, dim3(gWaves), dim3(256), 0, s, it, dSeg.p, n, cur, dSah.p);
    hipLaunchKernelGGL(kb_cost, dim3(gSeg), dim3(256), 0, s, cur, nseg, dSah.p, dTree.p, maxPrims);
    hipLaunchKernelGGL(kb_flags, dim3(gItems1), dim3(256), 0, s, it, dSeg.p, n, cur, dFlags.p);
    BT(hipcub::DeviceScan::ExclusiveSum(dScanTmp.p, scanBytes, dFlags.p, dPre.p, (int)(n + 1), s));
    hipLaunchKernelGGL(kb_mid, dim3(gSeg), dim3(256), 0, s, cur, nseg, dPre.p);
    hipLaunchKernelGGL(kb_rank, dim3(gItems), dim3(256), 0, s, dSeg.p, n, cur, dFlags.p, dPre.p, dTmpL.p, dTmpR.p);
    hipLaunchKernelGGL(kb_swap, dim3(gItems), dim3(256), 0, s, it, dSeg.p, n, cur, dTmpL.p, dTmpR.p);
    Seg* nxt = nullptr;
    BT(hipMalloc((void**)&nxt, (size_t)nseg * 2 * sizeof(Seg)));
    BT(hipMemsetAsync(dNNext, 0, sizeof(uint32_t), s));
    hipLaunchKernelGGL(kb_children, dim3(gSeg), dim3(256), 0, s, cur, nseg, dTree.p, nxt, dNNext, dSmall.p, dNSmall, dMaxLeaf);
    hipLaunchKernelGGL(kb_reassign, dim3(gItems), dim3(256), 0, s, dSeg.p, n, cur);
    uint32_t cnt[2] = {0, 0};
    BT(hipMemcpyAsync(cnt, dCounters.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    BT(hipStreamSynchronize(s));
    nseg = cnt[0];
    nSmallHost = cnt[1];
    if (nseg) levels.push_back(nxt);
    else (void)hipFree(nxt);
  }
  const double t2 = now();
  if (nSmallHost) hipLaunchKernelGGL(kb_small, dim3((nSmallHost + 63) / 64), dim3(64), 0, s, it, dSmall.p, nSmallHost, dTree.p, maxPrims, dMaxLeaf);
  for (size_t L = levels.size(); L-- > 0;)
    if (L < levelCount.size())
      hipLaunchKernelGGL(kb_sizes, dim3((levelCount[L] + 255) / 256), dim3(256), 0, s, levels[L], levelCount[L], dTree.p);
  TNode rootNode;
  uint32_t maxLeaf = 0;
  BT(hipMemcpyAsync(&rootNode, dTree.p, sizeof(TNode), hipMemcpyDeviceToHost, s));
  BT(hipMemcpyAsync(&maxLeaf, dMaxLeaf, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  BT(hipStreamSynchronize(s));
  const double t3 = now();
  if (maxLeaf > 65535) {
    freeLevels();
    return dr_fail(DR_ERR_UNSUPPORTED, "dr_bvh_build_device: a leaf with more than 65535 primitives");
  }
  const uint32_t nnodes = rootNode.subNodes;
  BT(dOut.alloc(nnodes));
  BT(dOrder.alloc(n));
  {  // the root's number and primitive offset; then level by level, then the small sub-trees
    BT(hipMemsetAsync((char*)dTree.p + offsetof(TNode, index), 0, 2 * sizeof(uint32_t), s));
  }
  for (size_t L = 0; L < levelCount.size(); ++L)
    hipLaunchKernelGGL(kb_number, dim3((levelCount[L] + 255) / 256), dim3(256), 0, s, levels[L], levelCount[L], dTree.p, it, dOut.p, dOrder.p);
  if (nSmallHost) hipLaunchKernelGGL(kb_emit_small, dim3((nSmallHost + 63) / 64), dim3(64), 0, s, dSmall.p, nSmallHost, dTree.p, it, dOut.p, dOrder.p);
  BT(hipGetLastError());
  BT(hipMemcpyAsync(nodes_out, dOut.p, (size_t)nnodes * sizeof(DrBvhNode), hipMemcpyDeviceToHost, s));
  BT(hipMemcpyAsync(order_out, dOrder.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  BT(hipStreamSynchronize(s));
  freeLevels();
  *nnodes_out = nnodes;
  if (depth_out) *depth_out = rootNode.subDepth;
  if (dbg)
    fprintf(stderr, "dr_bvh_build_device: %u primitives, %u nodes, %zu levels by all threads, %u small sub-trees; upload + items %.3f s, levels %.3f s, "
            "small sub-trees + sizes %.3f s, numbering + download %.3f s\n", n, nnodes, levelCount.size(), nSmallHost, t1 - t0, t2 - t1, t3 - t2,
            now() - t3);
  return DR_OK;
}
