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
// Workgroup-staged append to the three stage queues.  Same-address atomics run at ~130 per microsecond chip-wide
// and a returning atomic takes microseconds under that load, so a shade workgroup (the only one on its CU) must
// neither issue many nor wait for them often: entries are compacted into LDS for DR_PUSH_ITERS iterations of the
// grid-stride loop (ballot + per-wave counts, no global traffic), then flushed with ONE atomic per counter and
// coalesced copies.  closestQ receives the continuation entries of a flush first, then its MIS entries.
// Every function here must be reached by every thread of the workgroup (<= 16 waves).
#ifndef DR_PUSH_ITERS
#define DR_PUSH_ITERS 8
#endif
struct PushStage {  // head of the dynamic LDS block; followed by 4 arrays (cont, mis, any, active) of capQ entries
  uint32_t cnt[4][16];
  uint32_t base[3];
  uint32_t pad;
};
struct PushCtx {  // workgroup-uniform registers
  uint32_t n[4];
  uint32_t iters;
};
inline size_t push_stage_bytes(uint32_t blockDimX) { return sizeof(PushStage) + 4 * (size_t)blockDimX * DR_PUSH_ITERS * sizeof(uint32_t); }
DR_DEV void stage_push(PushStage& sm, PushCtx& c, bool pCont, bool pMis, bool pAny, bool pAct, uint32_t slot, uint32_t misBit) {
  uint32_t* buf = (uint32_t*)(&sm + 1);
  const uint32_t capQ = blockDim.x * DR_PUSH_ITERS;
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
  uint32_t pre[4] = {0, 0, 0, 0}, tot[4] = {0, 0, 0, 0};
  for (int w = 0; w < nw; ++w)
    for (int j = 0; j < 4; ++j) {
      const uint32_t v = sm.cnt[j][w];
      if (w < wave) pre[j] += v;
      tot[j] += v;
    }
  if (pCont) buf[c.n[0] + pre[0] + (uint32_t)__popcll(m0 & lt)] = slot;
  if (pMis) buf[capQ + c.n[1] + pre[1] + (uint32_t)__popcll(m1 & lt)] = slot | misBit;
  if (pAny) buf[2 * capQ + c.n[2] + pre[2] + (uint32_t)__popcll(m2 & lt)] = slot;
  if (pAct) buf[3 * capQ + c.n[3] + pre[3] + (uint32_t)__popcll(m3 & lt)] = slot;
  for (int j = 0; j < 4; ++j) c.n[j] += tot[j];
  ++c.iters;
  __syncthreads();  // sm.cnt is reused by the next iteration; the staged entries are visible to a flush
}
DR_DEV void stage_flush(PushStage& sm, PushCtx& c, uint32_t* closestQ, uint32_t* nClosest, uint32_t* anyQ, uint32_t* nAny,
                        uint32_t* activeQ, uint32_t* nActive, unsigned long long* nCont = nullptr,
                        unsigned long long* nMis = nullptr, unsigned long long* nShadow = nullptr) {
  const uint32_t* buf = (const uint32_t*)(&sm + 1);
  const uint32_t capQ = blockDim.x * DR_PUSH_ITERS;
  if (threadIdx.x == 0) sm.base[0] = (c.n[0] + c.n[1]) ? atomicAdd(nClosest, c.n[0] + c.n[1]) : 0u;
  else if (threadIdx.x == 64) sm.base[1] = c.n[2] ? atomicAdd(nAny, c.n[2]) : 0u;
  else if (threadIdx.x == 128) sm.base[2] = c.n[3] ? atomicAdd(nActive, c.n[3]) : 0u;
  else if (threadIdx.x == 192 && nCont) {  // statistics only: no-return atomics, nobody waits for them
    if (c.n[0]) atomicAdd(nCont, (unsigned long long)c.n[0]);
    if (c.n[1]) atomicAdd(nMis, (unsigned long long)c.n[1]);
    if (c.n[2]) atomicAdd(nShadow, (unsigned long long)c.n[2]);
  }
  __syncthreads();
  const uint32_t b0 = sm.base[0], b1 = sm.base[1], b2 = sm.base[2];
  for (uint32_t i = threadIdx.x; i < c.n[0]; i += blockDim.x) closestQ[b0 + i] = buf[i];
  for (uint32_t i = threadIdx.x; i < c.n[1]; i += blockDim.x) closestQ[b0 + c.n[0] + i] = buf[capQ + i];
  for (uint32_t i = threadIdx.x; i < c.n[2]; i += blockDim.x) anyQ[b1 + i] = buf[2 * capQ + i];
  for (uint32_t i = threadIdx.x; i < c.n[3]; i += blockDim.x) activeQ[b2 + i] = buf[3 * capQ + i];
  c.n[0] = c.n[1] = c.n[2] = c.n[3] = 0;
  c.iters = 0;
  __syncthreads();  // the staging arrays and sm.base are reused
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
