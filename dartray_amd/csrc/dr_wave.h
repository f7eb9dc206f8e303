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
// issue many nor wait for them: every WAVE compacts its entries into its own LDS region (ballot + popcount,
// wave-uniform counts, no barrier, no global traffic) for one ROUND of DR_PUSH_ITERS iterations of the grid-stride
// loop.  Rounds are double buffered and nobody waits at a round's end: each wave publishes its counts and takes an LDS
// ticket; the wave that arrives LAST reserves the workgroup's ranges (one global atomic per queue) and marks the round
// ready; every wave then copies out its region of the PREVIOUS round -- whose ranges were reserved a whole round ago --
// with coalesced stores and starts the next round in the other buffer.  A wave can run up to two rounds ahead of the
// slowest one before it has to wait (a buffer's counts must have been consumed before they are overwritten).
// History: round 1 staged per workgroup with two barriers in EVERY iteration (the 12 waves of the CU ran in lock
// step: VALU 40 % busy); the first version of this round met at a barrier once per flush, where the in-kernel
// profile (-DDR_SHADE_PROF) still showed 12 % of the waves' time.
// closestQ receives the continuation entries of a round first, then its MIS entries.  Every spin is bounded: a
// protocol error traps instead of hanging the GPU.
#ifndef DR_PUSH_ITERS
#define DR_PUSH_ITERS 4
#endif
struct TraceCounters;
struct PushStage {  // head of the dynamic LDS block; followed by [2 buffers][wave] regions: 4 arrays (cont, mis, any, active) of 64 * DR_PUSH_ITERS entries
  uint32_t cnt[2][4][16];  // per buffer, queue, wave: entries staged in the round that last used the buffer
  uint32_t arrived[2];     // tickets taken on the buffer (monotonic: nw per round)
  uint32_t ready[2];       // round + 1 of the round whose base[] is valid
  uint32_t copied[2];      // copy-outs finished on the buffer (monotonic: nw per round)
  uint32_t base[2][3];     // reserved ranges of the round: closestQ, anyQ, activeQ
  uint32_t nVert[16];      // per wave: path vertices set up so far (statistics; kept in LDS, not in a register)
#ifdef DR_SHADE_PROF
  unsigned long long prof[16][10];
#endif
  uint32_t sobol[256];  // Sobol2's low-byte table (dr_kernels.hip): a per-lane look-up, ds_read instead of a global round trip
};
struct PushCtx {  // wave-uniform registers
  uint32_t n[4];
  uint32_t iters;
  uint32_t round;
};
#define DR_PUSH_CAP (64 * DR_PUSH_ITERS)  // entries per queue, wave and buffer
__host__ __device__ inline size_t push_stage_bytes(uint32_t blockDimX) {
  return sizeof(PushStage) + 2 * 4 * (size_t)blockDimX * DR_PUSH_ITERS * sizeof(uint32_t);
}
DR_DEV void stage_init(PushStage& sm) {  // once per kernel, by every thread, before the first stage_push
  if (threadIdx.x < 2) sm.arrived[threadIdx.x] = sm.ready[threadIdx.x] = sm.copied[threadIdx.x] = 0u;
  if (threadIdx.x < 16) sm.nVert[threadIdx.x] = 0u;
  __syncthreads();
}
DR_DEV uint32_t* stage_region(PushStage& sm, uint32_t buffer) {
  const uint32_t nw = (blockDim.x + 63u) >> 6, wave = threadIdx.x >> 6;
  return (uint32_t*)(&sm + 1) + ((size_t)buffer * nw + wave) * 4 * DR_PUSH_CAP;
}
DR_DEV void stage_push(PushStage& sm, PushCtx& c, bool pCont, bool pMis, bool pAny, bool pAct, uint32_t slot, uint32_t misBit,
                       bool pVert = false, uint32_t actBits = 0u) {
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
  uint32_t* buf = stage_region(sm, c.round & 1u);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned long long m0 = __ballot(pCont), m1 = __ballot(pMis), m2 = __ballot(pAny), m3 = __ballot(pAct);
  const unsigned long long m4 = __ballot(pVert);
  if (lane == 0) sm.nVert[wave] += (uint32_t)__popcll(m4);
  if (pCont) buf[c.n[0] + (uint32_t)__popcll(m0 & lt)] = slot;
  if (pMis) buf[DR_PUSH_CAP + c.n[1] + (uint32_t)__popcll(m1 & lt)] = slot | misBit;
  if (pAny) buf[2 * DR_PUSH_CAP + c.n[2] + (uint32_t)__popcll(m2 & lt)] = slot;
  if (pAct) buf[3 * DR_PUSH_CAP + c.n[3] + (uint32_t)__popcll(m3 & lt)] = slot | actBits;
  c.n[0] += (uint32_t)__popcll(m0);
  c.n[1] += (uint32_t)__popcll(m1);
  c.n[2] += (uint32_t)__popcll(m2);
  c.n[3] += (uint32_t)__popcll(m3);
  ++c.iters;
}
// wait (bounded) until the LDS word reaches `want`; acquire at workgroup scope
DR_DEV void stage_wait(uint32_t* word, uint32_t want) {
  for (uint32_t spin = 0; spin < (1u << 24); ++spin) {
    if (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) return;
    __builtin_amdgcn_s_sleep(2);
  }
  __builtin_trap();  // a protocol error must fail loudly, not hang the GPU
}
// this wave's region of round q goes to the ranges the round's last arriver reserved
DR_DEV void stage_copy_out(PushStage& sm, uint32_t q, uint32_t* closestQ, uint32_t* anyQ, uint32_t* activeQ) {
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63) >> 6);
  const uint32_t b = q & 1u;
  stage_wait(&sm.ready[b], q + 1u);
  uint32_t pre[4] = {0, 0, 0, 0}, mine[4] = {0, 0, 0, 0}, tot0 = 0;
  for (int w = 0; w < nw; ++w)
    for (int j = 0; j < 4; ++j) {
      const uint32_t v = sm.cnt[b][j][w];
      if (w < wave) pre[j] += v;
      if (w == wave) mine[j] = v;
      if (j == 0) tot0 += v;
    }
  const uint32_t* buf = stage_region(sm, b);
  const uint32_t b0 = sm.base[b][0], b1 = sm.base[b][1], b2 = sm.base[b][2];
  for (uint32_t i = (uint32_t)lane; i < mine[0]; i += 64u) closestQ[b0 + pre[0] + i] = buf[i];
  for (uint32_t i = (uint32_t)lane; i < mine[1]; i += 64u) closestQ[b0 + tot0 + pre[1] + i] = buf[DR_PUSH_CAP + i];
  for (uint32_t i = (uint32_t)lane; i < mine[2]; i += 64u) anyQ[b1 + pre[2] + i] = buf[2 * DR_PUSH_CAP + i];
  for (uint32_t i = (uint32_t)lane; i < mine[3]; i += 64u) activeQ[b2 + pre[3] + i] = buf[3 * DR_PUSH_CAP + i];
  // the region and the round's counts have been consumed by this wave (release: the reads above come first)
  if (lane == 0) __hip_atomic_fetch_add(&sm.copied[b], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// End of a round (every DR_PUSH_ITERS iterations, and after the last iteration): reached by every wave of the workgroup.
DR_DEV void stage_flush(PushStage& sm, PushCtx& c, uint32_t* closestQ, uint32_t* nClosest, uint32_t* anyQ, uint32_t* nAny,
                        uint32_t* activeQ, uint32_t* nActive, unsigned long long* stats = nullptr) {
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63) >> 6);
  const uint32_t r = c.round, b = r & 1u, k = r >> 1;  // the k-th use of buffer b
  // the previous use of this buffer (round r - 2) must have been copied out by every wave before its counts go
  if (k > 0) stage_wait(&sm.copied[b], (uint32_t)nw * k);
  uint32_t ticket = 0;
  if (lane == 0) {
    sm.cnt[b][0][wave] = c.n[0];
    sm.cnt[b][1][wave] = c.n[1];
    sm.cnt[b][2][wave] = c.n[2];
    sm.cnt[b][3][wave] = c.n[3];
    ticket = __hip_atomic_fetch_add(&sm.arrived[b], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  ticket = wave_bcast_first(ticket);
  if (ticket == (uint32_t)nw * (k + 1u) - 1u) {  // the last wave to arrive reserves the round's ranges
    uint32_t tot[4] = {0, 0, 0, 0};
    for (int w = 0; w < nw; ++w)
      for (int j = 0; j < 4; ++j) tot[j] += sm.cnt[b][j][w];
    if (lane == 0) {
      sm.base[b][0] = (tot[0] + tot[1]) ? atomicAdd(nClosest, tot[0] + tot[1]) : 0u;
      sm.base[b][1] = tot[2] ? atomicAdd(nAny, tot[2]) : 0u;
      sm.base[b][2] = tot[3] ? atomicAdd(nActive, tot[3]) : 0u;
      if (stats) {  // statistics only (shade_cont, shade_mis, shade_shadow): no-return atomics
        if (tot[0]) atomicAdd(stats + 0, (unsigned long long)tot[0]);
        if (tot[1]) atomicAdd(stats + 1, (unsigned long long)tot[1]);
        if (tot[2]) atomicAdd(stats + 2, (unsigned long long)tot[2]);
      }
      __hip_atomic_store(&sm.ready[b], r + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if (r > 0) stage_copy_out(sm, r - 1u, closestQ, anyQ, activeQ);
  c.n[0] = c.n[1] = c.n[2] = c.n[3] = 0;
  c.iters = 0;
  c.round = r + 1u;
}
// After the loop: the last round is still in its buffer.
DR_DEV void stage_finish(PushStage& sm, PushCtx& c, uint32_t* closestQ, uint32_t* anyQ, uint32_t* activeQ) {
  if (c.round > 0) stage_copy_out(sm, c.round - 1u, closestQ, anyQ, activeQ);
  __syncthreads();  // (the statistics below read every wave's LDS counters)
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
