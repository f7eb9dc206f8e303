// dr_rng.h -- random streams shared by host and device code.
//
// DartRandom restates the Dart VM's dart:math Random (the generator behind
// lib/core/rng.dart:27-43).  The SDK is not vendored in the reference; the
// algorithm (multiply-with-carry, A = 0xffffda61, Thomas-Wang seeding, four
// warm-up steps) is the published one.
//
// counter_key() is this library's own construction: the reference consumes one
// serial stream per task (sampler_renderer.dart:137), which cannot be split
// across GPU lanes, so the on-device sampler gives every (pixel, LD block)
// and every (pixel, sample) its own DartRandom stream.
#ifndef DR_RNG_H
#define DR_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define DR_HD __host__ __device__ inline
#else
#define DR_HD inline
#endif

DR_HD uint64_t dr_mix64(uint64_t n) {
  n = (~n) + (n << 21);
  n = n ^ (n >> 24);
  n = n * 265;
  n = n ^ (n >> 14);
  n = n * 21;
  n = n ^ (n >> 28);
  n = n + (n << 31);
  return n;
}

struct DartRandom {
  uint32_t lo, hi;
  DR_HD void seed(int64_t s) {
    uint64_t hash = dr_mix64((uint64_t)s);
    if (hash == 0) hash = 0x5A17;
    lo = (uint32_t)(hash & 0xffffffffu);
    hi = (uint32_t)(hash >> 32);
    step();
    step();
    step();
    step();
  }
  DR_HD void step() {
    uint64_t s = 0xffffda61ULL * (uint64_t)lo + (uint64_t)hi;
    lo = (uint32_t)(s & 0xffffffffu);
    hi = (uint32_t)(s >> 32);
  }
  // Random.nextInt(0xffffffff) (rng.dart:40-42): `result = rnd32 % max` with the retry test
  // `rnd32 - result + max > 2^32`; for max = 2^32 - 1 the result is lo itself and the only rejected draw is
  // lo == 0xffffffff.
  DR_HD uint32_t randomUint() {
    do step();
    while (lo == 0xffffffffu);
    return lo;
  }
  // Random.nextDouble() (rng.dart:36-38): 26 + 27 bits from two steps.
  DR_HD double randomFloat() {
    step();
    double a = (double)(lo & ((1u << 26) - 1));
    step();
    double b = (double)(lo & ((1u << 27) - 1));
    return (a * 134217728.0 + b) / 9007199254740992.0;
  }
};

// kind 1: LD block stream of (pixel, block); kind 2: in-Li stream of (pixel, sample).
DR_HD int64_t dr_counter_key(uint64_t seed, uint64_t a, uint64_t b, uint64_t kind) {
  uint64_t h = dr_mix64(seed ^ 0x9E3779B97F4A7C15ULL);
  h = dr_mix64(h ^ (a * 0xD1B54A32D192ED03ULL + kind));
  h = dr_mix64(h ^ (b * 0x8CB92BA72F3D8DD7ULL + 0x5851F42D4C957F2DULL));
  return (int64_t)(h & 0x7fffffffffffffffULL);
}

#endif
