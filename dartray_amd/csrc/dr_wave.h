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
// Staged append to the three stage queues.  Same-address atomics run at ~90-130 per microsecond chip-wide and a
// returning atomic takes microseconds under that load, so a shade workgroup (the only one on its CU) must neither
// issue many nor wait for them often: every WAVE compacts its entries into its own LDS region for DR_PUSH_ITERS
// iterations of the grid-stride loop (ballot + popcount, wave-uniform counts, no barrier, no global traffic); then the
// workgroup meets ONCE, one thread per queue reserves the workgroup's range with one atomic, and every wave copies
// its region out with coalesced stores.  (Round 1 staged per workgroup with two barriers in EVERY iteration: the 12
// waves of the CU then ran in lock step -- all loading, all waiting, all computing together -- and the VALU idled
// 60 % of the time.  Between two flushes the waves now drift apart and cover each other's memory waits.)
// closestQ receives the continuation entries of a flush first, then its MIS entries.
// stage_flush must be reached by every thread of the workgroup (<= 16 waves).
#ifndef DR_PUSH_ITERS
#define DR_PUSH_ITERS 8
#endif
struct TraceCounters;
struct PushStage {  // head of the dynamic LDS block; followed by one region per wave: 4 arrays (cont, mis, any, active) of 64 * DR_PUSH_ITERS entries
  uint32_t cnt[4][16];
  uint32_t base[3];
  uint32_t pad;
  uint32_t nVert[16];  // per wave: path vertices set up so far (statistics; kept in LDS, not in a register)
#ifdef DR_SHADE_PROF
  unsigned long long prof[16][10];
#endif
  uint32_t sobol[256];  // Sobol2's low-byte table (dr_kernels.hip): a per-lane look-up, ds_read instead of a global round trip
};
struct PushCtx {  // wave-uniform registers
  uint32_t n[4];
  uint32_t iters;
};
#define DR_PUSH_CAP (64 * DR_PUSH_ITERS)  // entries per queue and wave
__host__ __device__ inline size_t push_stage_bytes(uint32_t blockDimX) { return sizeof(PushStage) + 4 * (size_t)blockDimX * DR_PUSH_ITERS * sizeof(uint32_t); }
DR_DEV void stage_push(PushStage& sm, PushCtx& c, bool pCont, bool pMis, bool pAny, bool pAct, uint32_t slot, uint32_t misBit,
                       bool pVert = false) {
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
  uint32_t* buf = (uint32_t*)(&sm + 1) + (size_t)wave * 4 * DR_PUSH_CAP;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned long long m0 = __ballot(pCont), m1 = __ballot(pMis), m2 = __ballot(pAny), m3 = __ballot(pAct);
  const unsigned long long m4 = __ballot(pVert);
  if (lane == 0) sm.nVert[wave] += (uint32_t)__popcll(m4);
  if (pCont) buf[c.n[0] + (uint32_t)__popcll(m0 & lt)] = slot;
  if (pMis) buf[DR_PUSH_CAP + c.n[1] + (uint32_t)__popcll(m1 & lt)] = slot | misBit;
  if (pAny) buf[2 * DR_PUSH_CAP + c.n[2] + (uint32_t)__popcll(m2 & lt)] = slot;
  if (pAct) buf[3 * DR_PUSH_CAP + c.n[3] + (uint32_t)__popcll(m3 & lt)] = slot;
  c.n[0] += (uint32_t)__popcll(m0);
  c.n[1] += (uint32_t)__popcll(m1);
  c.n[2] += (uint32_t)__popcll(m2);
  c.n[3] += (uint32_t)__popcll(m3);
  ++c.iters;
}
DR_DEV void stage_flush(PushStage& sm, PushCtx& c, uint32_t* closestQ, uint32_t* nClosest, uint32_t* anyQ, uint32_t* nAny,
                        uint32_t* activeQ, uint32_t* nActive, unsigned long long* stats = nullptr) {
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63) >> 6);
  const uint32_t* buf = (const uint32_t*)(&sm + 1) + (size_t)wave * 4 * DR_PUSH_CAP;
  if (lane == 0) {
    sm.cnt[0][wave] = c.n[0];
    sm.cnt[1][wave] = c.n[1];
    sm.cnt[2][wave] = c.n[2];
    sm.cnt[3][wave] = c.n[3];
  }
  __syncthreads();
  uint32_t pre[4] = {0, 0, 0, 0}, tot[4] = {0, 0, 0, 0};
  for (int w = 0; w < nw; ++w)
    for (int j = 0; j < 4; ++j) {
      const uint32_t v = sm.cnt[j][w];
      if (w < wave) pre[j] += v;
      tot[j] += v;
    }
  if (threadIdx.x == 0) sm.base[0] = (tot[0] + tot[1]) ? atomicAdd(nClosest, tot[0] + tot[1]) : 0u;
  else if (threadIdx.x == 64) sm.base[1] = tot[2] ? atomicAdd(nAny, tot[2]) : 0u;
  else if (threadIdx.x == 128) sm.base[2] = tot[3] ? atomicAdd(nActive, tot[3]) : 0u;
  else if (threadIdx.x == 192 && stats) {  // statistics only (shade_cont, shade_mis, shade_shadow): no-return atomics, nobody waits for them
    if (tot[0]) atomicAdd(stats + 0, (unsigned long long)tot[0]);
    if (tot[1]) atomicAdd(stats + 1, (unsigned long long)tot[1]);
    if (tot[2]) atomicAdd(stats + 2, (unsigned long long)tot[2]);
  }
  __syncthreads();
  const uint32_t b0 = sm.base[0], b1 = sm.base[1], b2 = sm.base[2];
  for (uint32_t i = (uint32_t)lane; i < c.n[0]; i += 64u) closestQ[b0 + pre[0] + i] = buf[i];
  for (uint32_t i = (uint32_t)lane; i < c.n[1]; i += 64u) closestQ[b0 + tot[0] + pre[1] + i] = buf[DR_PUSH_CAP + i];
  for (uint32_t i = (uint32_t)lane; i < c.n[2]; i += 64u) anyQ[b1 + pre[2] + i] = buf[2 * DR_PUSH_CAP + i];
  for (uint32_t i = (uint32_t)lane; i < c.n[3]; i += 64u) activeQ[b2 + pre[3] + i] = buf[3 * DR_PUSH_CAP + i];
  c.n[0] = c.n[1] = c.n[2] = c.n[3] = 0;
  c.iters = 0;
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
