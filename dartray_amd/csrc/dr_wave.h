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
// Staged append to the three stage queues.  Every WAVE compacts its entries into its own LDS region (ballot +
// popcount, wave-uniform counts, no barrier, no global traffic) for one ROUND of DR_PUSH_ITERS iterations, then reserves
// its own ranges -- one atomic per queue counter, all in flight together, one wait -- and copies the region out with
// coalesced stores.  No word is shared between waves and nobody waits for anybody.
// What the numbers said on the way here (C2, shade ms per step): one atomic per wave, queue and ITERATION on adjacent
// counters: 726 (same-line atomics are a chip-wide serial resource, ~100 per microsecond); per-workgroup staging with two
// barriers per iteration: 153; per-wave regions, one workgroup-level reservation per 4 iterations by the last wave to
// arrive, double buffered: 114 -- with 14 % of the waves' time (-DDR_SHADE_PROF) spent waiting for the slowest wave of
// the workgroup, because a buffer can only be reused once EVERY wave has copied it out; per-wave reservations every 8
// iterations with the four counters of a stage in ONE cache line: 130 (4 iterations: 204); the same with the counters
// ~1 KB apart (dr_api.hip) and work handed out in chunks from a counter: 100.
// closestQ receives the continuation entries of a wave's round first, then its MIS entries.
#ifndef DR_PUSH_ITERS
#define DR_PUSH_ITERS 8
#endif
struct TraceCounters;
struct PushStage {  // head of the dynamic LDS block; followed by one region per wave: 4 arrays (cont, mis, any, active) of 64 * DR_PUSH_ITERS entries
  uint32_t nVert[16];  // per wave: path vertices set up so far (statistics; kept in LDS, not in a register)
  unsigned long long nStat[16][3];  // per wave: continuation, MIS and shadow rays queued so far (statistics)
#ifdef DR_SHADE_PROF
  unsigned long long prof[16][10];
#endif
  uint32_t sobol[256];  // Sobol2's low-byte table (dr_kernels.hip): a per-lane look-up, ds_read instead of a global round trip
};
struct PushCtx {  // wave-uniform registers
  uint32_t n[4];
  uint32_t iters;
  uint32_t round;
  uint32_t nEnv;  // NQ == 5: entries of the fifth (environment-map) list
};
#define DR_PUSH_CAP (64 * DR_PUSH_ITERS)  // entries per queue and wave
// NQ = lists per wave: 4 (continuation, MIS, shadow, active) or 5 (+ the environment-map items of k_shade_path, k_env's input)
__host__ __device__ inline size_t push_stage_bytes(uint32_t blockDimX, int nq = 4) {
  return sizeof(PushStage) + (size_t)nq * (size_t)blockDimX * DR_PUSH_ITERS * sizeof(uint32_t);
}
DR_DEV void stage_init(PushStage& sm) {  // once per kernel, by every thread, before the first stage_push
  if (threadIdx.x < 16) {
    sm.nVert[threadIdx.x] = 0u;
    sm.nStat[threadIdx.x][0] = sm.nStat[threadIdx.x][1] = sm.nStat[threadIdx.x][2] = 0ull;
  }
  __syncthreads();
}
template <int NQ = 4>
DR_DEV uint32_t* stage_region(PushStage& sm) {
  return (uint32_t*)(&sm + 1) + (size_t)(threadIdx.x >> 6) * NQ * DR_PUSH_CAP;
}
template <int NQ = 4>
DR_DEV void stage_push(PushStage& sm, PushCtx& c, bool pCont, bool pMis, bool pAny, bool pAct, uint32_t slot, uint32_t misBit,
                       bool pVert = false, uint32_t actBits = 0u, bool pEnv = false, uint32_t envBits = 0u) {
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
  uint32_t* buf = stage_region<NQ>(sm);
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
  if constexpr (NQ == 5) {
    const unsigned long long m5 = __ballot(pEnv);
    if (pEnv) buf[4 * DR_PUSH_CAP + c.nEnv + (uint32_t)__popcll(m5 & lt)] = slot | envBits;
    c.nEnv += (uint32_t)__popcll(m5);
  }
  ++c.iters;
}
// End of a wave's round (every DR_PUSH_ITERS iterations, and after its last iteration).  closestQ receives the round's
// continuation entries first, then its MIS entries.
// `work` (optional): the kernel's chunk counter; the wave's next chunk of work is taken in the same round trip and
// returned (dynamic distribution: a wave that runs ahead simply takes more chunks).
template <int NQ = 4>
DR_DEV uint32_t stage_flush(PushStage& sm, PushCtx& c, uint32_t* closestQ, uint32_t* nClosest, uint32_t* anyQ, uint32_t* nAny,
                            uint32_t* activeQ, uint32_t* nActive, unsigned long long* stats = nullptr, uint32_t* work = nullptr,
                            uint32_t* envQ = nullptr, uint32_t* nEnvQ = nullptr) {
  const int lane = lane_id();
  const uint32_t n0 = c.n[0], n1 = c.n[1], n2 = c.n[2], n3 = c.n[3];
  uint32_t grabbed = 0u;
  uint32_t n4 = 0u;
  if constexpr (NQ == 5) n4 = c.nEnv;  // the environment-map list: a fifth counter, in its own cache line
  if (work && (n0 | n1 | n2 | n3 | n4) == 0u) {
    if (lane == 0) grabbed = atomicAdd(work, 1u);
    grabbed = wave_bcast_first(grabbed);
  }
  if (NQ == 5 && n4 != 0u && (n0 | n1 | n2 | n3) == 0u) {  // only environment-map entries this round (e.g. a chunk of sky)
    uint32_t a4 = 0u;
    if (lane == 0) {
      a4 = atomicAdd(nEnvQ, n4);
      if (work) grabbed = atomicAdd(work, 1u);
    }
    a4 = wave_bcast_first(a4);
    if (work) grabbed = wave_bcast_first(grabbed);
    const uint32_t* buf = stage_region<NQ>(sm);
    for (uint32_t i = (uint32_t)lane; i < n4; i += 64u) envQ[a4 + i] = buf[4 * DR_PUSH_CAP + i];
  }
  if ((n0 | n1 | n2 | n3) != 0u) {
    // three independent round trips in flight together, ONE wait.  Written out because the compiler's atomic optimizer
    // wraps every atomicAdd in its own readfirstlane and so waits for each before it issues the next.  Every lane gets
    // the same values back from lane 0 through readfirstlane (adding 0 returns the counter and costs the same).
    uint32_t a0 = 0u, a1 = 0u, a2 = 0u, a4 = 0u;
    if (lane == 0) {
      const uint32_t zero = 0u;
      if (NQ == 5 && work) {  // five round trips in flight together (k_shade_path's ENVQ instantiations)
        const uint32_t one = 1u;
        asm volatile(
            "global_atomic_add %0, %5, %6, %11 sc0\n\t"
            "global_atomic_add %1, %5, %7, %12 sc0\n\t"
            "global_atomic_add %2, %5, %8, %13 sc0\n\t"
            "global_atomic_add %3, %5, %9, %14 sc0\n\t"
            "global_atomic_add %4, %5, %10, %15 sc0\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(grabbed), "=&v"(a4)
            : "v"(zero), "v"(n0 + n1), "v"(n2), "v"(n3), "v"(one), "v"(n4), "s"(nClosest), "s"(nAny), "s"(nActive), "s"(work), "s"(nEnvQ)
            : "memory");
      } else if (work) {
        const uint32_t one = 1u;
        asm volatile(
            "global_atomic_add %0, %4, %5, %9 sc0\n\t"
            "global_atomic_add %1, %4, %6, %10 sc0\n\t"
            "global_atomic_add %2, %4, %7, %11 sc0\n\t"
            "global_atomic_add %3, %4, %8, %12 sc0\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(grabbed)
            : "v"(zero), "v"(n0 + n1), "v"(n2), "v"(n3), "v"(one), "s"(nClosest), "s"(nAny), "s"(nActive), "s"(work)
            : "memory");
      } else {
        asm volatile(
            "global_atomic_add %0, %3, %4, %7 sc0\n\t"
            "global_atomic_add %1, %3, %5, %8 sc0\n\t"
            "global_atomic_add %2, %3, %6, %9 sc0\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(a0), "=&v"(a1), "=&v"(a2)
            : "v"(zero), "v"(n0 + n1), "v"(n2), "v"(n3), "s"(nClosest), "s"(nAny), "s"(nActive)
            : "memory");
      }
      const int wave = (int)(threadIdx.x >> 6);  // statistics (shade_cont, shade_mis, shade_shadow): summed at the end
      sm.nStat[wave][0] += n0;
      sm.nStat[wave][1] += n1;
      sm.nStat[wave][2] += n2;
    }
    const uint32_t b0 = wave_bcast_first(a0), b1 = wave_bcast_first(a1), b2 = wave_bcast_first(a2);
    const uint32_t* buf = stage_region<NQ>(sm);
    for (uint32_t i = (uint32_t)lane; i < n0; i += 64u) closestQ[b0 + i] = buf[i];
    for (uint32_t i = (uint32_t)lane; i < n1; i += 64u) closestQ[b0 + n0 + i] = buf[DR_PUSH_CAP + i];
    for (uint32_t i = (uint32_t)lane; i < n2; i += 64u) anyQ[b1 + i] = buf[2 * DR_PUSH_CAP + i];
    for (uint32_t i = (uint32_t)lane; i < n3; i += 64u) activeQ[b2 + i] = buf[3 * DR_PUSH_CAP + i];
    if constexpr (NQ == 5) {
      if (n4 != 0u) {
        if (!work) {  // (no caller: the five-list form is k_shade_path's, which always hands its work counter in)
          if (lane == 0) a4 = atomicAdd(nEnvQ, n4);
        }
        const uint32_t b4 = wave_bcast_first(a4);
        for (uint32_t i = (uint32_t)lane; i < n4; i += 64u) envQ[b4 + i] = buf[4 * DR_PUSH_CAP + i];
      }
    }
    if (work) grabbed = wave_bcast_first(grabbed);
  }
  c.n[0] = c.n[1] = c.n[2] = c.n[3] = 0;
  c.nEnv = 0u;
  c.iters = 0;
  c.round += 1u;
  return grabbed;
}
// After the loop: every round has been flushed (the loop flushes after its last iteration).
DR_DEV void stage_finish(PushStage& sm, PushCtx& c, uint32_t* closestQ, uint32_t* anyQ, uint32_t* activeQ,
                         unsigned long long* stats = nullptr) {
  __syncthreads();  // (the statistics read every wave's LDS counters)
  if (stats && threadIdx.x < 3) {  // one no-return atomic per workgroup and counter
    unsigned long long v = 0;
    for (int w = 0; w < 16; ++w) v += sm.nStat[w][threadIdx.x];
    if (v) atomicAdd(stats + threadIdx.x, v);
  }
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
