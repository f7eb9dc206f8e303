// dr_scene_prep.hip -- what dr_scene_create does with the marshalled tree, on the device.
//
// A foreign host's BVHAccel.nodes (the 32-byte _LinearBVHNode records, accelerators/bvh_accel.dart:533-538) are INPUT:
// every node is validated (a malformed node must come back as DR_ERR_INVALID, not as an out-of-bounds device read or an
// endless traversal), the tree's height is measured (it bounds the traversal stack), the sibling-pair records of the v3
// traversal (dr_device.h) are built in their memory order, and every box is checked to be the union of its children's
// boxes / of its triangles' vertices (initInterior bvh_accel.dart:518-524, Triangle.worldBound :238-241: what lets the
// v3 kernels re-derive a node's own box from data they fetch anyway).  Round 3 ran all of this as serial host loops over
// the node array: 0.6 s for C4's 20 M nodes, against a 0.26 s render and a 0.14 s device BVH build.  Here: one thread per
// node / primitive, a level-synchronous walk from the root for the height (one small read-back per level, like the
// builder), one device-wide prefix sum for the pair numbering.  The results -- error or not, height, every pair record --
// are those of the host loops (tests/test_gpu_scene_prep.py compares them byte for byte).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/dartray_hip.h"
#include "dr_scene_prep.h"

namespace {

enum { ST_FIRST_BAD = 0, ST_BAD_KIND = 1, ST_BIG_LEAF = 2, ST_NOT_UNION = 3, ST_OVERFLOW = 4, ST_BAD_TRI = 5, ST_WORDS = 8 };
enum { BAD_INTERIOR = 1, BAD_LEAF_RANGE = 2 };

__device__ __forceinline__ uint32_t node_offset(const uint4* nodes, uint32_t i) { return nodes[2 * (size_t)i + 1].z; }
__device__ __forceinline__ uint32_t node_meta(const uint4* nodes, uint32_t i) { return nodes[2 * (size_t)i + 1].w; }

// every node: an interior node's second child follows its first sub-tree (offset > i + 1: the depth-first numbering of
// bvh_accel.dart:419-437 -- so every walk terminates), axis 0..2; a leaf's primitives exist
__global__ void kp_validate_nodes(const uint4* nodes, uint32_t nnodes, uint64_t ntris, uint32_t* status) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnodes) return;
  const uint32_t off = node_offset(nodes, i), meta = node_meta(nodes, i);
  const uint32_t nprims = meta & 0xffffu, axis = (meta >> 16) & 0xffu;
  uint32_t bad = 0;
  if (nprims == 0) {
    if (off <= i + 1u || off >= nnodes || axis > 2u) bad = BAD_INTERIOR;
  } else {
    if ((uint64_t)off + nprims > ntris) bad = BAD_LEAF_RANGE;
    if (nprims > 31u) status[ST_BIG_LEAF] = 1u;  // packed references carry at most 31 primitives per leaf: v2 kernel only
  }
  if (bad) {
    const uint32_t old = atomicMin(&status[ST_FIRST_BAD], i);
    if (i < old) status[ST_BAD_KIND] = bad;  // (racy between equal-rank writers only; the host re-reads the node itself)
  }
}

// primitives: vertex indices inside the vertex array, quadric rows inside the quadric table, material and light ids
__global__ void kp_validate_prims(const uint32_t* triIdx, const uint32_t* mat, const int32_t* light, uint64_t ntris, uint64_t nverts,
                                  uint32_t nquadrics, uint32_t nmaterials, uint32_t nlights, uint32_t* status) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntris) return;
  const uint32_t a = triIdx[3 * i], b = triIdx[3 * i + 1], c = triIdx[3 * i + 2];
  uint32_t bad = 0;
  if (a == DR_PRIM_QUADRIC) {
    if (b >= nquadrics) bad = 2;
  } else if (a >= nverts || b >= nverts || c >= nverts) {
    bad = 1;
  }
  if (!bad && mat[i] >= nmaterials) bad = 3;
  if (!bad && light[i] >= (int32_t)nlights) bad = 4;
  if (bad) atomicMax(&status[ST_BAD_TRI], bad);
}

// one level of the walk from the root: the children of the interior nodes of `in`
__global__ void kp_expand(const uint4* nodes, const uint32_t* in, uint32_t nIn, uint32_t* out, uint32_t cap, uint32_t* nOut, uint32_t* status) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t i = 0;
  bool interior = false;
  if (t < nIn) {
    i = in[t];
    interior = (node_meta(nodes, i) & 0xffffu) == 0u;
  }
  const unsigned long long m = __ballot(interior);
  if (m == 0ull) return;
  const int lane = (int)(threadIdx.x & 63u);
  uint32_t base = 0;
  if (lane == __ffsll((long long)m) - 1) base = atomicAdd(nOut, 2u * (uint32_t)__popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1);
  if (interior) {
    const uint32_t p = base + 2u * (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (p + 1u < cap) {
      out[p] = i + 1u;
      out[p + 1u] = node_offset(nodes, i);
    } else {
      status[ST_OVERFLOW] = 1u;  // more nodes reached than the array holds: some node has two parents
    }
  }
}

// is every box the union of what is below it?  (f32 min / max of the stored values: exact)
__global__ void kp_union_check(const uint4* nodes, uint32_t nnodes, const float* verts, const uint32_t* triIdx, uint32_t* status) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnodes) return;
  const uint4 a = nodes[2 * (size_t)i], b = nodes[2 * (size_t)i + 1];
  const uint32_t nprims = b.w & 0xffffu;
  float lo[3], hi[3];
  if (nprims == 0) {
    const uint4 a0 = nodes[2 * (size_t)(i + 1)], b0 = nodes[2 * (size_t)(i + 1) + 1];
    const uint4 a1 = nodes[2 * (size_t)b.z], b1 = nodes[2 * (size_t)b.z + 1];
    lo[0] = fminf(__uint_as_float(a0.x), __uint_as_float(a1.x));
    lo[1] = fminf(__uint_as_float(a0.y), __uint_as_float(a1.y));
    lo[2] = fminf(__uint_as_float(a0.z), __uint_as_float(a1.z));
    hi[0] = fmaxf(__uint_as_float(a0.w), __uint_as_float(a1.w));
    hi[1] = fmaxf(__uint_as_float(b0.x), __uint_as_float(b1.x));
    hi[2] = fmaxf(__uint_as_float(b0.y), __uint_as_float(b1.y));
  } else {
    for (int k = 0; k < 3; ++k) {
      lo[k] = __uint_as_float(0x7f800000u);
      hi[k] = -lo[k];
    }
    for (uint32_t t = 0; t < nprims; ++t)
      for (int v = 0; v < 3; ++v) {
        const uint32_t vi = triIdx[3 * ((size_t)b.z + t) + v];
        for (int k = 0; k < 3; ++k) {
          const float x = verts[3 * (size_t)vi + k];
          lo[k] = fminf(lo[k], x);
          hi[k] = fmaxf(hi[k], x);
        }
      }
  }
  // (compared as values like the host loop did: -0.0 == +0.0)
  const bool same = lo[0] == __uint_as_float(a.x) && lo[1] == __uint_as_float(a.y) && lo[2] == __uint_as_float(a.z) &&
                    hi[0] == __uint_as_float(a.w) && hi[1] == __uint_as_float(b.x) && hi[2] == __uint_as_float(b.y);
  if (!same) status[ST_NOT_UNION] = 1u;
}

struct IsInterior {
  const uint4* nodes;
  __host__ __device__ uint32_t operator()(uint32_t i) const { return (nodes[2 * (size_t)i + 1].w & 0xffffu) == 0u ? 1u : 0u; }
};

// memory slot of interior node i's pair record: the top nodes (sorted by node index, with their slots) first, everyone
// else in node (= depth-first) order behind them
__device__ __forceinline__ uint32_t pair_slot(uint32_t i, const uint32_t* rank, const uint32_t* topNode, const uint32_t* topSlot, uint32_t ntop) {
  // number of top nodes with index < i, and whether i is one of them
  uint32_t lo = 0, hi = ntop;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (topNode[mid] < i) lo = mid + 1u;
    else hi = mid;
  }
  if (lo < ntop && topNode[lo] == i) return topSlot[lo];
  return ntop + rank[i] - lo;
}

__global__ void kp_build_pairs(const uint4* nodes, uint32_t nnodes, const uint32_t* rank, const uint32_t* topNode, const uint32_t* topSlot,
                               uint32_t ntop, uint4* pairs) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnodes) return;
  const uint4 b = nodes[2 * (size_t)i + 1];
  if ((b.w & 0xffffu) != 0u) return;
  const uint32_t slot = pair_slot(i, rank, topNode, topSlot, ntop);
  const uint32_t c[2] = {i + 1u, b.z};
  for (int k = 0; k < 2; ++k) {
    const uint4 ca = nodes[2 * (size_t)c[k]];
    uint4 cb = nodes[2 * (size_t)c[k] + 1];
    if ((cb.w & 0xffffu) == 0u) cb.z = pair_slot(c[k], rank, topNode, topSlot, ntop);  // an interior child: its own pair's slot
    pairs[4 * (size_t)slot + 2 * k] = ca;
    pairs[4 * (size_t)slot + 2 * k + 1] = cb;
  }
}

#define PREP_TRY(expr)                                                                  \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      out->message = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
      return DR_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

template <class T>
struct Buf {
  T* p = nullptr;
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
  ~Buf() {
    if (p) (void)hipFree(p);
  }
};

}  // namespace

int scene_prepare_device(const ScenePrepIn& in, ScenePrepOut* out) {
  out->depth = 0;
  out->npairs = out->topPairs = 0;
  out->pairsOk = false;
  out->message.clear();
  const uint32_t nnodes = (uint32_t)in.nnodes;
  Buf<uint32_t> status;
  PREP_TRY(status.alloc(ST_WORDS));
  {
    uint32_t init[ST_WORDS] = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0};
    PREP_TRY(hipMemcpy(status.p, init, sizeof(init), hipMemcpyHostToDevice));
  }
  const dim3 B(256);
  if (in.ntris) {
    hipLaunchKernelGGL(kp_validate_prims, dim3((unsigned)((in.ntris + 255) / 256)), B, 0, 0, in.triIdx, in.triMaterial, in.triLight, in.ntris, in.nverts,
                       in.nquadrics, in.nmaterials, in.nlights, status.p);
  }
  if (nnodes) hipLaunchKernelGGL(kp_validate_nodes, dim3((nnodes + 255) / 256), B, 0, 0, in.nodes, nnodes, in.ntris, status.p);
  uint32_t st[ST_WORDS];
  PREP_TRY(hipMemcpy(st, status.p, sizeof(st), hipMemcpyDeviceToHost));
  if (st[ST_BAD_TRI]) {
    static const char* what[5] = {"", "vertex index out of range", "quadric index out of range", "material index out of range", "light index out of range"};
    out->message = what[std::min<uint32_t>(st[ST_BAD_TRI], 4u)];
    return DR_ERR_INVALID;
  }
  if (st[ST_FIRST_BAD] != 0xffffffffu) {
    const DrBvhNode& n = in.hostNodes[st[ST_FIRST_BAD]];
    out->message = n.nprims == 0 ? "malformed BVH node (interior node: second child must follow the first sub-tree, axis 0..2)" : "leaf primitive range";
    return DR_ERR_INVALID;
  }
  if (!nnodes) return DR_OK;

  // ---- height: a level-synchronous walk from the root ----
  Buf<uint32_t> fa, fb, cnt;
  PREP_TRY(fa.alloc(nnodes));
  PREP_TRY(fb.alloc(nnodes));
  PREP_TRY(cnt.alloc(1));
  const uint32_t zero = 0;
  PREP_TRY(hipMemcpy(fa.p, &zero, sizeof(uint32_t), hipMemcpyHostToDevice));
  uint32_t nIn = 1, level = 0;
  uint64_t reached = 1;
  uint32_t *fin = fa.p, *fout = fb.p;
  for (;;) {
    PREP_TRY(hipMemsetAsync(cnt.p, 0, sizeof(uint32_t), 0));
    hipLaunchKernelGGL(kp_expand, dim3((nIn + 255) / 256), B, 0, 0, in.nodes, fin, nIn, fout, nnodes, cnt.p, status.p);
    uint32_t nOut = 0;
    PREP_TRY(hipMemcpy(&nOut, cnt.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (nOut == 0) break;
    ++level;
    reached += nOut;
    if (level > DR_PREP_MAX_STACK) {
      out->message = "BVH deeper than the traversal stack";
      return DR_ERR_UNSUPPORTED;
    }
    if (nOut > nnodes || reached > nnodes) {
      out->message = "malformed BVH (a node is the child of two nodes)";
      return DR_ERR_INVALID;
    }
    std::swap(fin, fout);
    nIn = nOut;
  }
  out->depth = level;
  if (!in.wantPairs) return DR_OK;

  // ---- sibling pairs ----
  PREP_TRY(hipMemcpy(st, status.p, sizeof(st), hipMemcpyDeviceToHost));
  if (st[ST_BIG_LEAF] || st[ST_OVERFLOW]) return DR_OK;  // the v2 kernel only
  hipLaunchKernelGGL(kp_union_check, dim3((nnodes + 255) / 256), B, 0, 0, in.nodes, nnodes, in.verts, in.triIdx, status.p);
  // the top `topLevels` levels breadth-first (the host walks them: at most 2^T - 1 records), sorted by (level, node index)
  std::vector<uint32_t> topNode, topSlot;
  if (in.topLevels > 0) {
    std::vector<std::pair<uint32_t, uint32_t>> byNode;  // (node, slot)
    std::vector<uint32_t> cur(1, 0u), nxt;
    uint32_t slots = 0;
    for (int lev = 0; lev < in.topLevels && !cur.empty(); ++lev) {
      std::sort(cur.begin(), cur.end());
      nxt.clear();
      for (uint32_t i : cur) {
        const DrBvhNode& n = in.hostNodes[i];
        if (n.nprims != 0) continue;
        byNode.push_back({i, slots++});
        nxt.push_back(i + 1u);
        nxt.push_back(n.offset);
      }
      cur.swap(nxt);
    }
    std::sort(byNode.begin(), byNode.end());
    for (auto& e : byNode) {
      topNode.push_back(e.first);
      topSlot.push_back(e.second);
    }
  }
  const uint32_t ntop = (uint32_t)topNode.size();
  Buf<uint32_t> rank, dTopNode, dTopSlot;
  Buf<unsigned char> tmp;
  PREP_TRY(rank.alloc((size_t)nnodes + 1));
  PREP_TRY(dTopNode.alloc(ntop));
  PREP_TRY(dTopSlot.alloc(ntop));
  if (ntop) {
    PREP_TRY(hipMemcpy(dTopNode.p, topNode.data(), ntop * sizeof(uint32_t), hipMemcpyHostToDevice));
    PREP_TRY(hipMemcpy(dTopSlot.p, topSlot.data(), ntop * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  {
    hipcub::CountingInputIterator<uint32_t> idx(0u);
    hipcub::TransformInputIterator<uint32_t, IsInterior, hipcub::CountingInputIterator<uint32_t>> flags(idx, IsInterior{in.nodes});
    size_t bytes = 0;
    PREP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, flags, rank.p, (int)nnodes));
    PREP_TRY(tmp.alloc(bytes));
    PREP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, flags, rank.p, (int)nnodes));
  }
  // number of interior nodes = rank[nnodes - 1] + interior(nnodes - 1): the last node of a depth-first array is a leaf
  uint32_t lastRank = 0;
  PREP_TRY(hipMemcpy(&lastRank, rank.p + (nnodes - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
  const uint32_t np = lastRank + (in.hostNodes[nnodes - 1].nprims == 0 ? 1u : 0u);
  PREP_TRY(hipMemcpy(st, status.p, sizeof(st), hipMemcpyDeviceToHost));
  if (st[ST_NOT_UNION] || np >= (1u << 29)) return DR_OK;
  if ((size_t)np > in.pairsCap) {
    out->message = "pair buffer too small";
    return DR_ERR_INVALID;
  }
  if (np) hipLaunchKernelGGL(kp_build_pairs, dim3((nnodes + 255) / 256), B, 0, 0, in.nodes, nnodes, rank.p, dTopNode.p, dTopSlot.p, ntop, in.pairsOut);
  PREP_TRY(hipDeviceSynchronize());
  out->npairs = np;
  out->topPairs = in.topLevels > 0 ? ntop : 0u;
  out->pairsOk = true;
  return DR_OK;
}
