// dr_wave.h -- 64-lane wavefront helpers shared by the kernels.
#ifndef DR_WAVE_H
#define DR_WAVE_H

#include "dr_device.h"

DR_DEV int lane_id() { return (int)(threadIdx.x & 63); }
DR_DEV uint32_t wave_bcast_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Append `val` for every lane with `pred` to a device queue: one ballot, one
// prefix popcount and ONE atomic per wave.  Must be reached by the whole wave.
DR_DEV void wave_push(uint32_t* q, uint32_t* count, bool pred, uint32_t val) {
  unsigned long long m = __ballot(pred);
  if (m == 0ull) return;
  int lane = lane_id();
  int leader = __ffsll((long long)m) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, leader);
  if (pred) q[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = val;
}
// Block-aggregated append to the three stage queues: ONE atomic per workgroup per counter instead
// of one per wave (same-address atomics run at ~90 per microsecond chip-wide, which is what bounded
// the shade kernel when every wave issued its own).  closestQ receives continuation entries first,
// then MIS entries.  Must be reached by every thread of the workgroup (<= 16 waves).
struct PushScratch {
  uint32_t cnt[4][16];
  uint32_t base[3];
};
DR_DEV void block_push(PushScratch& sm, uint32_t* closestQ, uint32_t* nClosest, uint32_t* anyQ, uint32_t* nAny,
                       uint32_t* activeQ, uint32_t* nActive, bool pCont, bool pMis, bool pAny, bool pAct, uint32_t slot,
                       uint32_t misBit) {
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63) >> 6);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned long long m0 = __ballot(pCont), m1 = __ballot(pMis), m2 = __ballot(pAny), m3 = __ballot(pAct);
  if (lane == 0) {
    sm.cnt[0][wave] = (uint32_t)__popcll(m0);
    sm.cnt[1][wave] = (uint32_t)__popcll(m1);
    sm.cnt[2][wave] = (uint32_t)__popcll(m2);
    sm.cnt[3][wave] = (uint32_t)__popcll(m3);
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    uint32_t tot = 0;
    if (threadIdx.x == 0) {
      for (int w = 0; w < nw; ++w) tot += sm.cnt[0][w] + sm.cnt[1][w];
      sm.base[0] = tot ? atomicAdd(nClosest, tot) : 0u;
    } else if (threadIdx.x == 1) {
      for (int w = 0; w < nw; ++w) tot += sm.cnt[2][w];
      sm.base[1] = tot ? atomicAdd(nAny, tot) : 0u;
    } else {
      for (int w = 0; w < nw; ++w) tot += sm.cnt[3][w];
      sm.base[2] = tot ? atomicAdd(nActive, tot) : 0u;
    }
  }
  __syncthreads();
  uint32_t pre0 = 0, pre1 = 0, pre2 = 0, pre3 = 0, all0 = 0;
  for (int w = 0; w < nw; ++w) {
    if (w < wave) {
      pre0 += sm.cnt[0][w];
      pre1 += sm.cnt[1][w];
      pre2 += sm.cnt[2][w];
      pre3 += sm.cnt[3][w];
    }
    all0 += sm.cnt[0][w];
  }
  if (pCont) closestQ[sm.base[0] + pre0 + (uint32_t)__popcll(m0 & lt)] = slot;
  if (pMis) closestQ[sm.base[0] + all0 + pre1 + (uint32_t)__popcll(m1 & lt)] = slot | misBit;
  if (pAny) anyQ[sm.base[1] + pre2 + (uint32_t)__popcll(m2 & lt)] = slot;
  if (pAct) activeQ[sm.base[2] + pre3 + (uint32_t)__popcll(m3 & lt)] = slot;
  __syncthreads();  // sm is reused by the next iteration
}
DR_DEV unsigned long long wave_sum(uint32_t v) {
  unsigned long long x = v;
  for (int off = 32; off > 0; off >>= 1) {
    unsigned int lo = (unsigned int)__shfl_xor((int)(uint32_t)(x & 0xffffffffull), off);
    unsigned int hi = (unsigned int)__shfl_xor((int)(uint32_t)(x >> 32), off);
    x += ((unsigned long long)hi << 32) | lo;
  }
  return x;
}


#endif
