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
