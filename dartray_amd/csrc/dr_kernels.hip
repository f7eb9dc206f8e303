// dr_kernels.hip -- hand-written gfx950 kernels of the DartRay hot path (the traversal kernels are in dr_trace.hip):
//   k_shade_path     PathIntegrator.Li vertex step      (surface_integrators/path_integrator.dart:29-122,
//                                                        core/integrator.dart:79-185)
//   k_shade_direct   DirectLightingIntegrator.Li         (surface_integrators/direct_lighting_integrator.dart:30-68)
//   k_gen_samples(_lm / _pc / _multi)  LDPixelSample     (core/montecarlo.dart:407-551; _pc: two waves per 64 pixels, 512+ spp)
//   k_raygen         Perspective / Orthographic / EnvironmentCamera.generateRay (cameras/*.dart)
//   k_film / k_film_resolve  ImageFilm.addSample / writeImage (film/image_film.dart:99-185,268-299)
//
// One path per lane, 64-lane waves, path state in 64-slot tiles (BatchState, dr_kernels.h).  The shade kernels run
// one large workgroup per CU whose waves take chunks of the active list and stage their queue entries in LDS; what bounds
// each kernel is in DESIGN.md section 3 and MEASUREMENTS.md.  The path is about 1 flop per byte: no MFMA.
//
// Compiled with -ffp-contract=off: the Dart VM never fuses a*b+c.
#include <type_traits>

#include "dr_kernels.h"
#include "dr_rng.h"
#include "dr_wave.h"

#ifdef DR_NS  // a second instantiation of this file (another state layout, -DDR_SUB=...): every symbol in its own namespace
namespace DR_NS {
#endif

// ---------------------------------------------------------------------------
// scene upload: gather each primitive's vertices into its 48-byte record
// ---------------------------------------------------------------------------
__global__ void k_gather_tris(const float* verts, const uint32_t* idx, const uint32_t* mat, const int32_t* light,
                              const uint8_t* rev, float4* out, uint64_t ntris) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntris) return;
  uint32_t a = idx[3 * i], b = idx[3 * i + 1], c = idx[3 * i + 2];
  // rev[i]: bit 0 = reverseOrientation, bits 1.. = quadric kind (set by the host for DR_PRIM_QUADRIC rows)
  const uint32_t flags = ((uint32_t)rev[i] & 1u) | (((uint32_t)rev[i] >> 1) << 8);
  float4 q0, q1, q2;
  if (a == DR_PRIM_QUADRIC) {
    q0 = make_float4(__uint_as_float(b), 0.f, 0.f, 0.f);
    q1 = make_float4(0.f, 0.f, 0.f, 0.f);
    q2 = make_float4(0.f, __uint_as_float(mat[i]), __uint_as_float((uint32_t)light[i]), __uint_as_float(flags));
  } else {
    q0 = make_float4(verts[3 * (size_t)a], verts[3 * (size_t)a + 1], verts[3 * (size_t)a + 2], verts[3 * (size_t)b]);
    q1 = make_float4(verts[3 * (size_t)b + 1], verts[3 * (size_t)b + 2], verts[3 * (size_t)c], verts[3 * (size_t)c + 1]);
    q2 = make_float4(verts[3 * (size_t)c + 2], __uint_as_float(mat[i]), __uint_as_float((uint32_t)light[i]),
                     __uint_as_float(flags));
  }
  out[3 * i] = q0;
  out[3 * i + 1] = q1;
  out[3 * i + 2] = q2;
}

// Shading records (ShTri, dr_device.h) of a plain-triangle scene: dg.nn and the BSDF frame's sn of every primitive,
// with the functions the shade kernels would otherwise run per path vertex.
__global__ void k_make_shtris(DScene sc, float4* out, uint64_t ntris) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntris) return;
  const Tri tr = load_tri(sc, (uint32_t)i);
  DGeo dg;
  tri_dg(tr.p1, tr.p2, tr.p3, tr.reverse, F3{0, 0, 0}, F3{0, 0, 0}, 0.0, &dg);
  const F3 sn = vnormalize(dg.dpdu);  // bsdf.dart:45-51
  out[2 * i] = make_float4(dg.nn.x, dg.nn.y, dg.nn.z, sn.x);
  out[2 * i + 1] = make_float4(sn.y, sn.z, __uint_as_float(tr.mat), __uint_as_float((uint32_t)tr.light));
}

// ---------------------------------------------------------------------------
// LD sampler (montecarlo.dart:407-551), counter mode: lane = (pixel, LD block)
// ---------------------------------------------------------------------------
DR_DEV float VanDerCorput(uint32_t n, uint32_t scramble) {  // montecarlo.dart:495-504
  n = __brev(n);
  n ^= scramble;
  // min(((n>>8)&0xffffff)/2^24, ONE_MINUS_EPSILON): the quotient never exceeds 1-2^-24; a 24-bit integer times
  // 2^-24 is exact in f32, so this equals the reference's f64 quotient stored to f32
  return (float)(n >> 8) * 5.9604644775390625e-8f;
}
// Sobol2 (montecarlo.dart:486-493) XORs the direction number v_b = v_{b-1} ^ (v_{b-1} >> 1), v_0 = 2^31, into the
// scramble for every set bit b of n: linear over GF(2), so the low byte of n is one table look-up.
struct SobolTable {
  uint32_t lo[256];
  uint32_t v8;  // direction number of bit 8
  constexpr SobolTable() : lo(), v8(0) {
    for (uint32_t n = 0; n < 256; ++n) {
      uint32_t r = 0, v = 1u << 31;
      for (uint32_t m = n; m != 0; m >>= 1, v ^= v >> 1)
        if (m & 1u) r ^= v;
      lo[n] = r;
    }
    uint32_t v = 1u << 31;
    for (int b = 0; b < 8; ++b) v ^= v >> 1;
    v8 = v;
  }
};
__constant__ SobolTable c_sobol = SobolTable();
DR_DEV float Sobol2(uint32_t n, uint32_t scramble) {
  scramble ^= c_sobol.lo[n & 255u];
  n >>= 8;
  for (uint32_t v = c_sobol.v8; n != 0; n >>= 1, v ^= v >> 1)
    if (n & 1u) scramble ^= v;
  return (float)(scramble >> 8) * 5.9604644775390625e-8f;
}

// the same with the low-byte table read from LDS (the shade kernels copy it there: shade_count_init)
DR_DEV float Sobol2_lds(const uint32_t* tab, uint32_t n, uint32_t scramble) {
  scramble ^= tab[n & 255u];
  n >>= 8;
  for (uint32_t v = c_sobol.v8; n != 0; n >>= 1, v ^= v >> 1)
    if (n & 1u) scramble ^= v;
  return (float)(scramble >> 8) * 5.9604644775390625e-8f;
}

// ---- camera-sample vector access (see BatchState): float form or compact (index, scramble) form ----
// LD block of float field f (Appendix B layout: image, lens, time, n1D 1-D slots, then the 2-D slots)
DR_DEV int sv_block(const RenderParams& rp, int f) {
  return f < 5 ? (f >> 1) : (f < 5 + rp.n1D ? f - 2 : 3 + rp.n1D + ((f - 5 - rp.n1D) >> 1));
}
// raw fetch of 1-D field f: float bits, or the permuted index + its scramble
DR_DEV void sv_fetch1(const RenderParams& rp, const BatchState& st, uint32_t slot, int f, uint32_t* raw, uint32_t* scr) {
  if (st.svFloat) {
    *raw = __float_as_uint(LDS_STREAM(st.sv() + TI64(st.tileStride, slot) + (size_t)f * 64));
    *scr = 0u;
    return;
  }
  const int k = sv_block(rp, f);
  const uint8_t* q = st.svIdx() + (size_t)(slot >> 6) * st.tileStride * 4 + (((size_t)k * 64 + (slot & 63u)) << st.idxShift);
  *raw = st.idxShift ? (uint32_t)*(const uint16_t*)q : (uint32_t)*q;
  *scr = st.svScr[(size_t)(2 * k) * st.pixCap + (slot >> rp.sppShift)];
}
// raw fetch of the 2-D entry whose first float is field f
DR_DEV void sv_fetch2(const RenderParams& rp, const BatchState& st, uint32_t slot, int f, uint32_t* raw, uint32_t* scr) {
  if (st.svFloat) {
    raw[0] = __float_as_uint(LDS_STREAM(st.sv() + TI64(st.tileStride, slot) + (size_t)f * 64));
    raw[1] = __float_as_uint(LDS_STREAM(st.sv() + TI64(st.tileStride, slot) + (size_t)(f + 1) * 64));
    scr[0] = scr[1] = 0u;
    return;
  }
  const int k = sv_block(rp, f);
  const uint8_t* q = st.svIdx() + (size_t)(slot >> 6) * st.tileStride * 4 + (((size_t)k * 64 + (slot & 63u)) << st.idxShift);
  raw[0] = raw[1] = st.idxShift ? (uint32_t)*(const uint16_t*)q : (uint32_t)*q;
  scr[0] = st.svScr[(size_t)(2 * k) * st.pixCap + (slot >> rp.sppShift)];
  scr[1] = st.svScr[(size_t)(2 * k + 1) * st.pixCap + (slot >> rp.sppShift)];
}
// value of a fetched entry; second: the y half of a 2-D entry (Sample02, montecarlo.dart:480-484)
DR_DEV float sv_value(const BatchState& st, uint32_t raw, uint32_t scr, bool second) {
  if (st.svFloat) return __uint_as_float(raw);
  return second ? Sobol2(raw, scr) : VanDerCorput(raw, scr);
}
DR_DEV float sv_one(const RenderParams& rp, const BatchState& st, uint32_t slot, int f) {
  uint32_t raw, scr;
  sv_fetch1(rp, st, slot, f, &raw, &scr);
  return sv_value(st, raw, scr, false);
}
DR_DEV void sv_pair(const RenderParams& rp, const BatchState& st, uint32_t slot, int f, float* x, float* y) {
  uint32_t raw[2], scr[2];
  sv_fetch2(rp, st, slot, f, raw, scr);
  *x = sv_value(st, raw[0], scr[0], false);
  *y = sv_value(st, raw[1], scr[1], true);
}

// PT = uint8_t while spp <= 256 (half the LDS => twice the waves per CU), uint16_t above.  Rows of the
// permutation table are padded (68 B / 132 B) so that both phases are bank-conflict free: Fisher-Yates
// (lane = pixel, random row) and the write-out (consecutive lanes = consecutive rows of ONE pixel), which
// makes every global store a fully coalesced 256-byte wave store.
template <class PT, int ROW>
__global__ void __launch_bounds__(64) k_gen_samples(RenderParams rp, BatchState st, uint32_t npix) {
  extern __shared__ __align__(16) unsigned char s_raw[];
  PT* s_perm = (PT*)s_raw;                                      // [spp][ROW]
  uint32_t* s_scr = (uint32_t*)(s_raw + (size_t)rp.spp * ROW * sizeof(PT));  // [2][64]
  uint32_t* s_magic = s_scr + 128;  // [spp + 1]: floor(2^32 / m), the reciprocal of the Fisher-Yates divisors
  const int lane = threadIdx.x;
  const uint32_t p0 = blockIdx.x * 64u;
  const uint32_t p = p0 + lane;
  const int k = blockIdx.y;  // LD block: image, lens, time, 1-D slots, 2-D slots (montecarlo.dart:437-448)
  const int spp = rp.spp;
  bool is2D;
  int dst;
  if (rp.blocks) {
    const LdBlock b = rp.blocks[k];
    if (b.n != 1) return;  // k_gen_samples_multi's
    is2D = b.is2D != 0;
    dst = b.dst;
  }
  else if (k == 0) { is2D = true; dst = 0; }
  else if (k == 1) { is2D = true; dst = 2; }
  else if (k == 2) { is2D = false; dst = 4; }
  else if (k < 3 + rp.n1D) { is2D = false; dst = 5 + (k - 3); }
  else { is2D = true; dst = 5 + rp.n1D + 2 * (k - 3 - rp.n1D); }
  for (int m = 1 + lane; m <= spp; m += 64) s_magic[m] = m == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint32_t)m);
  __syncthreads();
  if (p < npix) {
    const int2 xy = st.pix[p];
    const uint64_t pixelIndex = (uint64_t)(xy.y - rp.extY0) * (uint64_t)rp.extW + (uint64_t)(xy.x - rp.extX0);
    DartRandom rng;
    rng.seed(dr_counter_key(rp.seed, pixelIndex, (uint64_t)k, 1));
    // LDShuffleScrambled1D/2D with nSamples == 1 (montecarlo.dart:524-551)
    s_scr[lane] = rng.randomUint();
    s_scr[64 + lane] = is2D ? rng.randomUint() : 0u;
    for (int i = 0; i < spp; ++i) (void)rng.randomUint();  // Shuffle of ONE entry: other = i + r % 1 (:294-303)
    for (int i = 0; i < spp; ++i) s_perm[i * ROW + lane] = (PT)i;
    // Fisher-Yates (montecarlo.dart:294-303).  The only serial chain left is the RNG: entry i + 1 is fetched before the
    // swap of step i is stored (and patched if that swap hits it), so no LDS round trip sits between two steps.
    PT a = s_perm[lane];
    for (int i = 0; i < spp; ++i) {
      // r % m for the wave-uniform divisor m = spp - i: q = mulhi(r, floor(2^32 / m)) is r / m or one less
      const uint32_t r = rng.randomUint(), m = (uint32_t)(spp - i);
      uint32_t rem = r - __umulhi(r, s_magic[m]) * m;
      if (rem >= m) rem -= m;
      const int other = i + (int)rem;
      const PT ahead = s_perm[(i + 1 < spp ? i + 1 : i) * ROW + lane];
      const PT b = s_perm[other * ROW + lane];
      s_perm[i * ROW + lane] = b;
      s_perm[other * ROW + lane] = a;
      a = (other == i + 1) ? a : ahead;
    }
  }
  __syncthreads();
  // write-out: element e of this wave's 64*spp contiguous outputs belongs to pixel e >> sppShift
  const uint32_t nOut = min(64u, npix - p0) * (uint32_t)spp;
  const uint32_t slot0 = p0 * (uint32_t)spp;
  if (!st.svFloat) {  // compact form: the permuted indices and the scrambles (block k == blockIdx.y: rp.blocks is null)
    if (p < npix) {
      st.svScr[(size_t)(2 * k) * st.pixCap + p] = s_scr[lane];
      st.svScr[(size_t)(2 * k + 1) * st.pixCap + p] = s_scr[64 + lane];
    }
    for (uint32_t e = lane; e < nOut; e += 64u) {
      const uint32_t pl = e >> rp.sppShift, j = e & (uint32_t)(spp - 1), slot = slot0 + e;
      PT* o = (PT*)(st.svIdx() + (size_t)(slot >> 6) * st.tileStride * 4) + (size_t)k * 64 + (slot & 63u);
      *o = s_perm[j * ROW + pl];
    }
    return;
  }
  float* out0 = st.sv() + (size_t)dst * 64;  // field `dst` of the sample vector inside each tile
  for (uint32_t e = lane; e < nOut; e += 64u) {
    const uint32_t pl = e >> rp.sppShift, j = e & (uint32_t)(spp - 1);
    const uint32_t idx = s_perm[j * ROW + pl];
    float* o = out0 + TI64(st.tileStride, slot0 + e);
    o[0] = VanDerCorput(idx, s_scr[pl]);
    if (is2D) o[64] = Sobol2(idx, s_scr[64 + pl]);
  }
}

// Compact form, spp >= 64 (the common case): the same shuffle with a LANE-MAJOR table -- entry i of lane l lives in
// dword column l, [i / EPW][l][i % EPW] with EPW entries per dword -- so every lane only ever touches its own LDS
// bank whatever row the random swap partner is in (the row-major table above takes ~6-way bank conflicts there), and
// a lane's 64 consecutive entries are one tile's whole 64-entry index run: it stores them itself, 16 bytes at a time.
// The workgroup has LN = blockDim.x <= 64 lanes (pixels): what limits this kernel is the LDS footprint of a shuffle
// (spp entries per pixel) and the latency of its dependent steps, so the 160 KB of a CU are better spent on many
// narrow waves than on few full ones -- 64 lanes up to 128 spp, 32 at 256, 16 above (launch_gen_samples).
// The LD block a workgroup row produces: blockIdx.y counts the blocks of rp.genMask (every (pixel, block) pair has its
// own keyed stream, so a block nobody reads can be left out without touching the others).
DR_DEV int gen_block(const RenderParams& rp) {
  if (rp.genMask == 0ull) return (int)blockIdx.y;
  unsigned long long m = rp.genMask;
  for (uint32_t i = 0; i < blockIdx.y; ++i) m &= m - 1ull;
  return __ffsll((long long)m) - 1;
}
// The part of a (pixel, LD block) stream in front of its shuffle -- seeding, the one or two scramble words, and the spp draws of
// the one-entry Shuffles (montecarlo.dart:294-303,524-551: `other = i + r % 1`, values that only advance the generator) -- needs no
// table.  In the shuffle kernels it ran at the occupancy a CU's LDS leaves them (one wave per SIMD at 512 spp) and was half of
// their generator steps; here it runs one thread per stream at full occupancy and leaves the generator state behind.
__global__ void __launch_bounds__(256) k_gen_burnin(RenderParams rp, BatchState st, uint32_t npix) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  if (st.genAlive && !st.genAlive[p >> 6]) return;  // no path of this 64-pixel group reads this bounce's blocks
  const int k = gen_block(rp);
  const bool is2D = k < 2 || k >= 3 + rp.n1D;
  const int2 xy = st.pix[p];
  const uint64_t pixelIndex = (uint64_t)(xy.y - rp.extY0) * (uint64_t)rp.extW + (uint64_t)(xy.x - rp.extX0);
  DartRandom rng;
  rng.seed(dr_counter_key(rp.seed, pixelIndex, (uint64_t)k, 1));
  st.svScr[(size_t)(2 * k) * st.pixCap + p] = rng.randomUint();
  st.svScr[(size_t)(2 * k + 1) * st.pixCap + p] = is2D ? rng.randomUint() : 0u;
  // spp draws: plain steps; a lane that met the one value in 2^32 that Random.nextInt redraws does them again the slow way
  const DartRandom saved = rng;
  bool rare = rp.genSlowDraws != 0;
  for (int i = 0; i < rp.spp; ++i) {
    rng.step();
    rare |= rng.lo == 0xffffffffu;
  }
  if (rare) {
    rng = saved;
    for (int i = 0; i < rp.spp; ++i) (void)rng.randomUint();
  }
  st.genState[(size_t)k * st.pixCap + p] = make_uint2(rng.lo, rng.hi);
}

template <class PT>
__global__ void __launch_bounds__(64) k_gen_samples_lm(RenderParams rp, BatchState st, uint32_t npix) {
  extern __shared__ __align__(16) unsigned char s_raw[];
  constexpr int EPW = 4 / (int)sizeof(PT);  // entries per dword
  const int LN = (int)blockDim.x;
  if (st.genAlive && !st.genAlive[(blockIdx.x * (uint32_t)LN) >> 6]) return;  // (LN divides 64: the workgroup's pixels are in one group; uniform exit)
  PT* s_perm = (PT*)s_raw;                  // [spp / EPW][LN][EPW]
  uint32_t* s_magic = (uint32_t*)(s_raw + (size_t)rp.spp * LN * sizeof(PT));  // [spp + 1]: floor(2^32 / m)
  const int lane = threadIdx.x;
  const uint32_t p = blockIdx.x * (uint32_t)LN + lane;
  const int k = gen_block(rp);  // LD block: image, lens, time, 1-D slots, 2-D slots (montecarlo.dart:437-448)
  const int spp = rp.spp;
  const bool is2D = k < 2 || k >= 3 + rp.n1D;
  auto at = [&](int i) -> PT& { return s_perm[(i / EPW) * (LN * EPW) + lane * EPW + (i % EPW)]; };
  for (int m = 1 + lane; m <= spp; m += LN) s_magic[m] = m == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint32_t)m);
  __syncthreads();
  if (p >= npix) return;
  const int2 xy = st.pix[p];
  const uint64_t pixelIndex = (uint64_t)(xy.y - rp.extY0) * (uint64_t)rp.extW + (uint64_t)(xy.x - rp.extX0);
  DartRandom rng;
  if (st.genState) {  // k_gen_burnin has seeded the stream, drawn its scrambles and made its burn-in draws
    const uint2 gs = st.genState[(size_t)k * st.pixCap + p];
    rng.lo = gs.x;
    rng.hi = gs.y;
  } else {
    rng.seed(dr_counter_key(rp.seed, pixelIndex, (uint64_t)k, 1));
    // LDShuffleScrambled1D/2D with nSamples == 1 (montecarlo.dart:524-551)
    st.svScr[(size_t)(2 * k) * st.pixCap + p] = rng.randomUint();
    st.svScr[(size_t)(2 * k + 1) * st.pixCap + p] = is2D ? rng.randomUint() : 0u;
  }
  // randomUint() is one generator step unless the step lands on 0xffffffff (it then steps again: once in 2^32).  A loop
  // around every step is a branch per step in a serial chain; instead a GROUP of draws is made with plain steps and
  // redone the slow way from the saved state if any lane of the wave saw the rare value.
  auto draws = [&](uint32_t* r, int n) {  // r == nullptr: the draws are discarded
    const DartRandom saved = rng;
    bool rare = rp.genSlowDraws != 0;  // (tests: DARTRAY_GEN_SLOW_DRAWS=1 takes the slow path everywhere; same streams)
    for (int j = 0; j < n; ++j) {
      rng.step();
      rare |= rng.lo == 0xffffffffu;
      if (r) r[j] = rng.lo;
    }
    if (__ballot(rare) != 0ull) {
      rng = saved;
      for (int j = 0; j < n; ++j) {
        const uint32_t v = rng.randomUint();
        if (r) r[j] = v;
      }
    }
  };
  if (!st.genState)
    for (int i = 0; i < spp; i += 16) draws(nullptr, 16);  // Shuffle of ONE entry: other = i + r % 1 (:294-303), spp times
  for (int i = 0; i < spp; ++i) at(i) = (PT)i;
  // Fisher-Yates (montecarlo.dart:294-303), FOUR steps at a time.  One lane's shuffle is a serial chain (generator ->
  // partner -> LDS read -> LDS writes) and at 512 / 1024 spp only one or two waves fit a CU's LDS, so nothing hides the
  // chain's latency.  Per group: the four partners (the only truly serial part: four generator steps) are computed one
  // group AHEAD, beside the LDS round trip of the current group; ALL the group's reads are issued together, the four
  // swaps replayed in registers -- a position a previous step of the group wrote is patched from that step's value,
  // newest first -- then the eight writes go out in step order (LDS executes a wave's accesses in order, so aliasing
  // writes end up right).
  auto partners = [&](int i, int* o) {
    uint32_t r[4];
    draws(r, 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t m = (uint32_t)(spp - i - j);
      uint32_t rem = r[j] - __umulhi(r[j], s_magic[m]) * m;  // r % m for the wave-uniform divisor: mulhi(r, floor(2^32 / m)) is r / m or one less
      if (rem >= m) rem -= m;
      o[j] = i + j + (int)rem;
    }
  };
  int oNext[4];
  partners(0, oNext);
  for (int i = 0; i < spp; i += 4) {
    int o[4];
    PT R[4], A[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = oNext[j];
      R[j] = at(o[j]);
      A[j] = at(i + j);
    }
    if (i + 4 < spp) partners(i + 4, oNext);
    PT av[4], bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      PT aj = A[j], bj = R[j];
#pragma unroll
      for (int k = 0; k < j; ++k) {  // oldest first: the newest write wins
        if (o[k] == i + j) aj = av[k];
        if (o[k] == o[j]) bj = av[k];
      }
      av[j] = aj;
      bv[j] = bj;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      at(i + j) = bv[j];
      at(o[j]) = av[j];
    }
  }
  // this lane's spp entries are spp / 64 whole index runs: slot p * spp + i is entry i & 63 of tile (p * spp + i) >> 6
  const uint32_t* cols = (const uint32_t*)s_raw + lane;  // dword d of this lane's column is cols[d * LN]
  const size_t tile0 = ((size_t)p * (size_t)spp) >> 6;
  constexpr int DPR = 64 / EPW;  // dwords per 64-entry run
  for (int t = 0; t < spp / 64; ++t) {
    uint4* o = (uint4*)(st.svIdx() + (tile0 + t) * (size_t)st.tileStride * 4 + (size_t)k * 64 * sizeof(PT));
    for (int q = 0; q < DPR / 4; ++q) {
      const int d = t * DPR + 4 * q;
      o[q] = make_uint4(cols[(d + 0) * LN], cols[(d + 1) * LN], cols[(d + 2) * LN], cols[(d + 3) * LN]);
    }
  }
}

// 512 / 1024 spp: the same shuffle by TWO waves per 64 pixels.  A table of spp 16-bit entries per pixel lets one
// (1024 spp) or two (512 spp) groups of 64 pixels fit a CU's LDS, so a single wave per group leaves three SIMDs idle and
// pays every instruction of the serial chain in full.  Wave 0 runs the generator side (seed, scrambles, the burn-in
// draws, then the partners of every step, four steps per group) and hands the partners over through a small LDS ring;
// wave 1 initialises the table meanwhile and performs the swaps; both write the result out.  The two instruction
// streams run on different SIMDs side by side: the time per step is the longer of the two instead of their sum.
// Hand-over: `prod` / `cons` count finished groups (published every 4 groups, bounded spins, a protocol error traps).
#define DR_GEN_RING 16  // groups of 4 partners per lane in the ring
// (MEASUREMENTS.md 5.6, what the loop keeps of round 5's variants: the group's magic numbers requested before the draws; table rows of 64
// entries per index -- one shift-add per address; the four partners in whole registers.  Requesting the next ring entry a group ahead was slower.)
DR_DEV void gen_wait(uint32_t* word, uint32_t want) {
  for (uint32_t spin = 0; spin < (1u << 22); ++spin) {
    if (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) return;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_trap();
}
__global__ void __launch_bounds__(128) k_gen_samples_pc(RenderParams rp, BatchState st, uint32_t npix) {
  extern __shared__ __align__(16) unsigned char s_raw[];
  typedef uint16_t PT;
  if (st.genAlive && !st.genAlive[blockIdx.x]) return;  // (the workgroup is one 64-pixel group; uniform exit)
  const int spp = rp.spp, G = spp / 4;
  PT* s_perm = (PT*)s_raw;                                                   // [spp / 2][64][2]
  uint32_t* s_magic = (uint32_t*)(s_raw + (size_t)spp * 64 * sizeof(PT));   // [spp + 1]: floor(2^32 / m)
  uint2* s_ring = (uint2*)(s_magic + ((spp + 2) & ~1));                      // [DR_GEN_RING][64]: four 16-bit partners
  uint32_t* s_flag = (uint32_t*)(s_ring + DR_GEN_RING * 64);                // prod, cons
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  const uint32_t p = blockIdx.x * 64u + (uint32_t)lane;
  const bool valid = p < npix;
  const int k = gen_block(rp);
  const bool is2D = k < 2 || k >= 3 + rp.n1D;
  auto at = [&](int i) -> PT& { return s_perm[i * 64 + lane]; };
  for (int m = 1 + (int)threadIdx.x; m <= spp; m += 128) s_magic[m] = m == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint32_t)m);
  if (threadIdx.x < 2) s_flag[threadIdx.x] = 0u;
  __syncthreads();
  if (role == 0) {
    // ---- generator side ----
    const int2 xy = st.pix[valid ? p : npix - 1u];
    const uint64_t pixelIndex = (uint64_t)(xy.y - rp.extY0) * (uint64_t)rp.extW + (uint64_t)(xy.x - rp.extX0);
    DartRandom rng;
    if (st.genState) {  // (k_gen_burnin: seed, scrambles and burn-in draws done)
      const uint2 gs = st.genState[(size_t)k * st.pixCap + (valid ? p : npix - 1u)];
      rng.lo = gs.x;
      rng.hi = gs.y;
    } else {
      rng.seed(dr_counter_key(rp.seed, pixelIndex, (uint64_t)k, 1));
      const uint32_t scr0 = rng.randomUint(), scr1 = is2D ? rng.randomUint() : 0u;  // LDShuffleScrambled1D/2D (montecarlo.dart:524-551)
      if (valid) {
        st.svScr[(size_t)(2 * k) * st.pixCap + p] = scr0;
        st.svScr[(size_t)(2 * k + 1) * st.pixCap + p] = scr1;
      }
    }
    auto draws = [&](uint32_t* r, int n) {  // as in k_gen_samples_lm: plain steps, redone the slow way after a rare value
      const DartRandom saved = rng;
      bool rare = rp.genSlowDraws != 0;
      for (int j = 0; j < n; ++j) {
        rng.step();
        rare |= rng.lo == 0xffffffffu;
        if (r) r[j] = rng.lo;
      }
      if (__ballot(rare) != 0ull) {
        rng = saved;
        for (int j = 0; j < n; ++j) {
          const uint32_t v = rng.randomUint();
          if (r) r[j] = v;
        }
      }
    };
    if (!st.genState)
      for (int i = 0; i < spp; i += 16) draws(nullptr, 16);  // Shuffle of ONE entry, spp times (:294-303)
    for (int g = 0; g < G; ++g) {
      if ((g & 3) == 0 && g + 4 > DR_GEN_RING) {  // room for four more groups
        gen_wait(&s_flag[1], (uint32_t)(g + 4 - DR_GEN_RING));
      }
      // (the group's four magic numbers are requested BEFORE the draws: a single wave per SIMD pays every LDS round trip in full,
      // this one now runs under the generator steps)
      uint32_t mg[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) mg[j] = s_magic[spp - 4 * g - j];
      uint32_t r[4];
      draws(r, 4);
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t m = (uint32_t)(spp - 4 * g - j);
        uint32_t rem = r[j] - __umulhi(r[j], mg[j]) * m;
        if (rem >= m) rem -= m;
        o[j] = (uint32_t)(4 * g + j) + rem;
      }
      s_ring[(g % DR_GEN_RING) * 64 + lane] = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
      if ((g & 3) == 3 && lane == 0) __hip_atomic_store(&s_flag[0], (uint32_t)(g + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  } else {
    // ---- table side ----
    for (int d = 0; d < spp; ++d) at(d) = (PT)d;
    auto waitProd = [&](uint32_t want) {
      gen_wait(&s_flag[0], want);
    };
    // The ring entry of group g + 1 is requested at the top of group g, so that its LDS round trip runs under g's table reads
    // (LDS operations of a wave complete in order: the wait for g's reads covers it).  The producer publishes every four groups:
    // before the last group of a quad the next quad is waited for (it is at most 8 of the ring's 16 groups ahead of `cons`).
    for (int g = 0; g < G; ++g) {
      if ((g & 3) == 0) waitProd((uint32_t)(g + 4));
      const uint2 q = s_ring[(g % DR_GEN_RING) * 64 + lane];
      const int i = 4 * g;
      int o[4] = {(int)(q.x & 0xffffu), (int)(q.x >> 16), (int)(q.y & 0xffffu), (int)(q.y >> 16)};
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(o[j]));
      PT R[4], A[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        R[j] = at(o[j]);
        A[j] = at(i + j);
      }
      PT av[4], bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        PT aj = A[j], bj = R[j];
#pragma unroll
        for (int kk = 0; kk < j; ++kk) {  // oldest first: the newest write wins
          if (o[kk] == i + j) aj = av[kk];
          if (o[kk] == o[j]) bj = av[kk];
        }
        av[j] = aj;
        bv[j] = bj;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        at(i + j) = bv[j];
        at(o[j]) = av[j];
      }
      if ((g & 3) == 3 && lane == 0) __hip_atomic_store(&s_flag[1], (uint32_t)(g + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  // write-out: a lane's spp entries are spp / 64 whole 64-entry index runs (128 B each); the two waves take alternate runs
  if (valid) {
    const uint32_t* cols = (const uint32_t*)s_raw + lane;
    const size_t tile0 = ((size_t)p * (size_t)spp) >> 6;
    for (int t = role; t < spp / 64; t += 2) {
      uint4* o = (uint4*)(st.svIdx() + (tile0 + t) * (size_t)st.tileStride * 4 + (size_t)k * 64 * sizeof(PT));
      for (int q = 0; q < 8; ++q) {
        const int e = t * 64 + 8 * q;
        auto two = [&](int i) { return (uint32_t)s_perm[i * 64 + lane] | ((uint32_t)s_perm[(i + 1) * 64 + lane] << 16); };
        o[q] = make_uint4(two(e), two(e + 2), two(e + 4), two(e + 6));
      }
    }
  }
}

// Blocks with n > 1 entries per pixel sample (DirectLighting with light nsamples > 1): lane = pixel, the
// n*spp values are generated and shuffled in place in the sample-vector arrays.  Element (sample i, entry j,
// dim d) lives at sv[(dst + dims*j + d)][p*spp + i].  Rare configuration: correctness over speed.
__global__ void __launch_bounds__(64) k_gen_samples_multi(RenderParams rp, BatchState st, uint32_t npix) {
  const uint32_t p = blockIdx.x * 64u + threadIdx.x;
  const int k = blockIdx.y;
  const LdBlock b = rp.blocks[k];
  if (b.n == 1 || p >= npix) return;
  const int spp = rp.spp, n = b.n, dims = b.is2D ? 2 : 1;
  const int2 xy = st.pix[p];
  const uint64_t pixelIndex = (uint64_t)(xy.y - rp.extY0) * (uint64_t)rp.extW + (uint64_t)(xy.x - rp.extX0);
  DartRandom rng;
  rng.seed(dr_counter_key(rp.seed, pixelIndex, (uint64_t)k, 1));
  const uint32_t s0 = rng.randomUint();
  const uint32_t s1 = b.is2D ? rng.randomUint() : 0u;
  float* base = st.sv() + (size_t)b.dst * 64;
  const uint32_t slot0 = p * (uint32_t)spp;
  auto at = [&](int i, int j, int d) -> float& { return base[TI64(st.tileStride, slot0 + (uint32_t)i) + (size_t)(dims * j + d) * 64]; };
  for (int i = 0; i < spp; ++i)
    for (int j = 0; j < n; ++j) {
      const uint32_t e = (uint32_t)(i * n + j);
      at(i, j, 0) = VanDerCorput(e, s0);
      if (b.is2D) at(i, j, 1) = Sobol2(e, s1);
    }
  for (int i = 0; i < spp; ++i)      // Shuffle(samples + i*n, n, dims) (montecarlo.dart:294-303,531-533)
    for (int j = 0; j < n; ++j) {
      const int other = j + (int)(rng.randomUint() % (uint32_t)(n - j));
      for (int d = 0; d < dims; ++d) {
        const float t = at(i, j, d);
        at(i, j, d) = at(i, other, d);
        at(i, other, d) = t;
      }
    }
  for (int i = 0; i < spp; ++i) {    // Shuffle(samples, spp, n*dims)
    const int other = i + (int)(rng.randomUint() % (uint32_t)(spp - i));
    for (int j = 0; j < n; ++j)
      for (int d = 0; d < dims; ++d) {
        const float t = at(i, j, d);
        at(i, j, d) = at(other, j, d);
        at(other, j, d) = t;
      }
  }
}

// Lazy sample generation: the 64-pixel groups that have an entry in a stage's active list
__global__ void __launch_bounds__(256) k_mark_alive(const uint32_t* list, const uint32_t* nList, uint32_t shift, uint8_t* alive) {
  const uint32_t n = *nList;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) alive[(list[i] & 0x3fffffffu) >> shift] = 1;
}
// ... and what that came to: pixels of the marked groups x the blocks of each bounce (DrScene's sampler statistics)
__global__ void __launch_bounds__(256) k_sum_alive(const uint8_t* alive, uint32_t nGroups, uint32_t npix, uint32_t nb0, uint32_t nb1, uint32_t nb2,
                                                   TraceCounters* ctr) {
  uint32_t mine = 0u;  // (a lane sees a few hundred groups at most: 64 pixels x <= 7 blocks each)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < 3u * nGroups; i += gridDim.x * blockDim.x) {
    const uint32_t b = i / nGroups, g = i - b * nGroups;
    if (alive[i]) mine += min(64u, npix - 64u * g) * (b == 0 ? nb0 : (b == 1 ? nb1 : nb2));
  }
  const unsigned long long sum = wave_sum(mine);
  if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&ctr->gen_pixel_blocks, sum);
}
// Host-buffer mode: [n][stride] -> [nFloats][cap]
__global__ void k_transpose_samples(const float* aos, int stride, BatchState st, int nFloats) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= st.nslots) return;
  for (int k = 0; k < nFloats; ++k) st.sv()[TI64(st.tileStride, s) + (size_t)k * 64] = aos[(size_t)s * stride + k];
}

// 3-vector / colour fields: `cap` in the callers below is the tile stride (see BatchState)
DR_DEV F3 ld3(const float* a, uint32_t cap, uint32_t s) {
  a += TI(cap, s);
  return F3{LDS_STREAM(a), LDS_STREAM(a + DR_SUB), LDS_STREAM(a + 2 * DR_SUB)};
}
DR_DEV void st3(float* a, uint32_t cap, uint32_t s, F3 v) {
  a += TI(cap, s);
  STS_STREAM(a, v.x);
  STS_STREAM(a + DR_SUB, v.y);
  STS_STREAM(a + 2 * DR_SUB, v.z);
}
DR_DEV C3 ldc(const float* a, uint32_t cap, uint32_t s) {
  a += TI(cap, s);
  return C3{LDS_STREAM(a), LDS_STREAM(a + DR_SUB), LDS_STREAM(a + 2 * DR_SUB)};
}
DR_DEV void stc(float* a, uint32_t cap, uint32_t s, C3 v) {
  a += TI(cap, s);
  STS_STREAM(a, v.r);
  STS_STREAM(a + DR_SUB, v.g);
  STS_STREAM(a + 2 * DR_SUB, v.b);
}

// ---------------------------------------------------------------------------
// camera (perspective_camera.dart:93-132; transform.dart:110-144)
// ---------------------------------------------------------------------------
DR_DEV F3 xf_point(const float* m, F3 p) {
  double x = p.x, y = p.y, z = p.z;
  F3 o = f3((double)m[0] * x + (double)m[1] * y + (double)m[2] * z + (double)m[3],
            (double)m[4] * x + (double)m[5] * y + (double)m[6] * z + (double)m[7],
            (double)m[8] * x + (double)m[9] * y + (double)m[10] * z + (double)m[11]);
  double w = (double)m[12] * x + (double)m[13] * y + (double)m[14] * z + (double)m[15];
  if (w != 1.0) o = f3((double)o.x / w, (double)o.y / w, (double)o.z / w);  // invScale
  return o;
}
DR_DEV F3 xf_vector(const float* m, F3 p) {
  double x = p.x, y = p.y, z = p.z;
  return f3((double)m[0] * x + (double)m[1] * y + (double)m[2] * z, (double)m[4] * x + (double)m[5] * y + (double)m[6] * z,
            (double)m[8] * x + (double)m[9] * y + (double)m[10] * z);
}

__global__ void __launch_bounds__(256) k_raygen(RenderParams rp, BatchState st) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= st.nslots) return;
  const uint32_t cap = st.tileStride;  // words per 64-slot tile
  const int2 xy = st.pix[s >> rp.sppShift];
  float sx, sy;
  sv_pair(rp, st, s, 0, &sx, &sy);
  const double imageX = (double)xy.x + (double)sx;               // montecarlo.dart:451-452
  const double imageY = (double)xy.y + (double)sy;
  F3 o = F3{0.f, 0.f, 0.f}, d;
  if (rp.cameraType == DR_CAMERA_ENVIRONMENT) {
    // EnvironmentCamera.generateRay (environment_camera.dart:42-52)
    const double theta = DR_PI * imageY / (double)rp.yres;
    const double phi = 2 * DR_PI * imageX / (double)rp.xres;
    d = f3(sin(theta) * cos(phi), cos(theta), sin(theta) * sin(phi));
  } else {
    F3 Pras = f3(imageX, imageY, 0.0);
    F3 Pcamera = xf_point(rp.r2c, Pras);
    if (rp.cameraType == DR_CAMERA_ORTHOGRAPHIC) {  // orthographic_camera.dart:52-80
      o = Pcamera;
      d = F3{0.f, 0.f, 1.f};
    } else {
      d = vnormalize(Pcamera);
    }
    if (rp.lensRadius > 0.0) {
      double lu, lv;
      sv_pair(rp, st, s, 2, &sx, &sy);
      ConcentricSampleDisk((double)sx, (double)sy, &lu, &lv);
      lu *= (double)rp.lensRadius;
      lv *= (double)rp.lensRadius;
      double ft = (double)rp.focalDistance / (double)d.z;
      F3 Pfocus = vadd(o, vmul(d, ft));
      o = f3(lu, lv, 0.0);   // (the orthographic camera, too, REPLACES the origin: orthographic_camera.dart:73)
      d = vnormalize(vsub(Pfocus, o));
    }
  }
  o = xf_point(rp.c2w, o);
  d = xf_vector(rp.c2w, d);
  st3(st.ro(), cap, s, o);
  st3(st.rd(), cap, s, d);
  st.rtmin()[TD(cap, s)] = 0.0;
  // hprim is written by the camera trace for every slot; pathThroughput = 1, L = 0 and flags = PF_HAS_CONT are what the
  // first shading stage assumes instead of reading them (52 B per camera sample less to write here and to read there)
}

// ---------------------------------------------------------------------------
// shading
// ---------------------------------------------------------------------------

// EstimateDirect's contribution of the pending NEE rays (integrator.dart:135-145,169-180).
// ENV: the scene has an InfiniteAreaLight (compiled out otherwise: the area-light-only path keeps its registers)
// QUAD: the scene has sphere / disk primitives (likewise)
template <bool ENV, bool QUAD, class LV>
DR_DEV C3 resolve_nee(const DScene& sc, const LV& lv, const SlotRef& sr, uint32_t flags, int shOcc, C3 Ld1) {
  C3 Ld = C3{0.f, 0.f, 0.f};
  if ((flags & PF_HAS_SH) && shOcc == 0) Ld = cadd(Ld, Ld1);
  if (flags & PF_HAS_MIS) {
    const int prim = sr.i32<F_MISPRIM>();
    const int li = sr.i32<F_MISLIGHT>();
    if (ENV && lv.light(li).kind == DR_LIGHT_INFINITE) {
      // the MIS ray escaped: Li = light.Le(ray) (integrator.dart:173-175), folded into Ld2 at set-up
      if (prim < 0) Ld = cadd(Ld, ldcf<F_LD2>(sr));
    } else if (!QUAD && prim >= 0) {
      const ShTri tr = load_shtri(sc, (uint32_t)prim);
      if (tr.light == li) {  // lightIsect.primitive.getAreaLight() == light (integrator.dart:170-172)
        C3 Li = light_L(lv.light(li), tr.nn, vneg(ld3f<F_MISD>(sr)));
        if (!cblack(Li)) Ld = cadd(Ld, ldcf<F_LD2>(sr));
      }
    } else if (prim >= 0) {
      Tri tr = load_tri(sc, (uint32_t)prim);
      if (tr.light == li) {  // lightIsect.primitive.getAreaLight() == light (integrator.dart:170-172)
        const F3 wi = ld3f<F_MISD>(sr);
        DGeo dg;
        if (QUAD && tr.kind) {
          // dg.nn of a quadric depends on the hit point: repeat the (deterministic) test of the MIS ray
          const DQuadric& qd = sc.quads[tr.quad];
          double th;
          F3 phit;
          (void)quadric_hit(qd, ld3f<F_RO>(sr), wi, sr.f64<F_RTMIN>(), DR_INF, &th, &phit);
          quadric_dg(qd, phit, &dg);
        } else if (QUAD && sc.srec && (__float_as_uint(sc.srec[7 * (size_t)prim + 6].x) & DR_SHADING_UV)) {
          const ShadeRec sr = load_srec(sc, (uint32_t)prim);
          tri_dg_uv(tr.p1, tr.p2, tr.p3, sr.uv, tr.reverse, F3{0, 0, 0}, wi, 0.0, 0.0, 0.0, &dg);  // only nn is used
        } else {
          tri_dg(tr.p1, tr.p2, tr.p3, tr.reverse, F3{0, 0, 0}, wi, 0.0, &dg);  // only nn is used
        }
        C3 Li = light_L(lv.light(li), dg.nn, vneg(wi));
        if (!cblack(Li)) Ld = cadd(Ld, ldcf<F_LD2>(sr));
      }
    }
  }
  return Ld;
}

// EstimateDirect up to the points where it must trace (integrator.dart:119-185):
// writes the shadow ray / MIS ray and their candidate contributions.
// PRE (the path kernel): the common outcome -- a shadow ray, no MIS ray -- stores the FINISHED contribution
// pathThroughput * (Ld1 * nLights) in Ld1 (the reference's own roundings: Ld = 0 + Ld1; Ld * nLights; pathThroughput *
// that, integrator.dart:113-116 / path_integrator.dart:56-68), so that the next stage reads 12 bytes instead of 24 and
// nobody writes betaNee; the two halves are evaluated BSDF half first (they are independent) so that the light half
// knows whether a MIS ray is pending.  Every other outcome keeps the raw terms and sets PF_RAW_NEE.
// ENVONLY (k_env): the light is known to be the infinite one -- only that branch is compiled.
template <bool ENV, bool QUAD, bool NI, bool PRE, class LV, bool ENVONLY = false>
DR_DEV uint32_t setup_nee(const DScene& sc, const LV& lv, const SlotRef& sr, int lightNum, const Bsdf& bsdf, F3 p, F3 n,
                          F3 wo, double ls0, double ls1, double lsc, double bs0, double bs1, double bsc,
                          C3 beta = C3{0.f, 0.f, 0.f}, double nLights = 0.0) {
  DLight light;
  if constexpr (!ENVONLY) light = lv.light(lightNum);
  const int flags = BSDF_ALL & ~BSDF_SPECULAR;
  uint32_t pf = 0;
  const bool infinite = ENVONLY || (ENV && light.kind == DR_LIGHT_INFINITE);
  F3 wi = F3{0, 0, 0}, ps = F3{0, 0, 0};
  double lightPdf = 0.0;
  C3 Li;
  if (QUAD && light.kind >= DR_LIGHT_POINT) {
    // Delta lights -- one shadow ray, no MIS weight and no BSDF-sampling half (integrator.dart:146-150,153).
    // PointLight.sampleLAtPoint (point_light.dart:41-47), SpotLight (spot_light.dart:54-85), DistantLight
    // (distant_light.dart:54-61)
    const F3 lp = F3{light.pos[0], light.pos[1], light.pos[2]};
    const F3 seg = vsub(lp, p);
    const bool distant = light.kind == DR_LIGHT_DISTANT;
    if (distant) {
      wi = lp;
      Li = C3{light.L[0], light.L[1], light.L[2]};
    } else {
      wi = vnormalize(seg);
      C3 I = C3{light.L[0], light.L[1], light.L[2]};
      if (light.kind == DR_LIGHT_SPOT) {
        const F3 wl = vnormalize(q_vector(light.w2l, vneg(wi)));
        const double costheta = wl.z;
        double fo;
        if (costheta < light.cosTotalWidth) fo = 0.0;
        else if (costheta > light.cosFalloffStart) fo = 1.0;
        else {
          const double delta = (costheta - light.cosTotalWidth) / (light.cosFalloffStart - light.cosTotalWidth);
          fo = delta * delta * delta * delta;
        }
        I = cmulD(I, fo);
      }
      Li = cdivD(I, vlen2(seg));
    }
    if (!cblack(Li)) {
      C3 f = bsdf_f(bsdf, wo, wi, flags);
      if (!cblack(f)) {
        if (distant) {  // VisibilityTester.setRay(p, eps, wi)
          st3f<F_SHD>(sr, wi);
          sr.f64<F_SHTMAX>() = DR_INF;
        } else {        // VisibilityTester.setSegment(p, eps, lightPos, 0)
          const double dist = vlen(seg);
          st3f<F_SHD>(sr, vdiv(seg, dist));
          sr.f64<F_SHTMAX>() = dist * (1.0 - 0.0);
        }
        stcf<F_LD1>(sr, cmulD(cmul(f, Li), (fabs(vdot(wi, n)) / 1.0)));
        pf |= PF_HAS_SH | (PRE ? PF_RAW_NEE : 0u);
      }
    }
    return pf;
  }
  // the BSDF-sampling half (integrator.dart:152-182)
  auto bsdfHalf = [&]() {
    F3 wi2 = F3{0, 0, 0};
    double bsdfPdf = 0.0;
    C3 f = bsdf_sample_f(bsdf, wo, &wi2, bs0, bs1, bsc, &bsdfPdf, flags);
    if (!cblack(f) && bsdfPdf > 0.0) {
      double lightPdf2;
      if constexpr (ENVONLY) lightPdf2 = env_pdf(sc.env, wi2);
      else lightPdf2 = infinite ? env_pdf_x<NI>(sc.env, wi2) : shapeset_pdf<QUAD>(sc, lv, light, p, wi2);
      if (lightPdf2 != 0.0) {
        double weight = PowerHeuristic(bsdfPdf, lightPdf2);
        // the radiance the MIS ray returns IF it reaches the light: Lemit of the sampled area light (its
        // front face is checked at resolve), or the map along wi2 if the ray escapes (light.Le(ray))
        C3 Lhit;
        if constexpr (ENVONLY) Lhit = env_Le(sc.env, wi2);
        else Lhit = infinite ? env_Le_x<NI>(sc.env, wi2) : C3{light.L[0], light.L[1], light.L[2]};
        st3f<F_MISD>(sr, wi2);
        stcf<F_LD2>(sr, cmulD(cmul(f, Lhit), (fabs(vdot(wi2, n)) * weight / bsdfPdf)));
        sr.i32<F_MISLIGHT>() = lightNum;
        pf |= PF_HAS_MIS;  // scene.intersect is called before `if (!Li.isBlack())` (integrator.dart:169-177)
      }
    }
  };
  // the light-sampling half (integrator.dart:128-150)
  auto lightHalf = [&](bool raw) {
    if constexpr (ENVONLY) {
      Li = env_sample_m(sc.env, lv.margFunc, lv.margCdf, ls0, ls1, &wi, &lightPdf);  // (k_env: the marginal distribution from LDS)
    } else if (!infinite) {
      // DiffuseAreaLight.sampleLAtPoint (diffuse_area_light.dart:60-70)
      F3 ns;
      ps = shapeset_sample<QUAD>(sc, lv, light, ls0, ls1, lsc, &ns, p);
      wi = vnormalize(vsub(ps, p));
      lightPdf = shapeset_pdf<QUAD>(sc, lv, light, p, wi);
      Li = light_L(light, ns, vneg(wi));
    } else {
      // InfiniteAreaLight.sampleLAtPoint (infinite_area_light.dart:92-131)
      Li = env_sample_x<NI>(sc.env, ls0, ls1, &wi, &lightPdf);
    }
    if (lightPdf > 0.0 && !cblack(Li)) {
      C3 f = bsdf_f(bsdf, wo, wi, flags);
      if (!cblack(f)) {
        if (!infinite) {
          // VisibilityTester.setSegment (visibility_tester.dart:26-29)
          F3 seg = vsub(ps, p);
          double dist = vlen(seg);
          st3f<F_SHD>(sr, vdiv(seg, dist));
          sr.f64<F_SHTMAX>() = dist * (1.0 - 1.0e-3);
        } else {
          // VisibilityTester.setRay (visibility_tester.dart:31-33)
          st3f<F_SHD>(sr, wi);
          sr.f64<F_SHTMAX>() = DR_INF;
        }
        double bsdfPdf = bsdf_pdf(bsdf, wo, wi, flags);
        double weight = PowerHeuristic(lightPdf, bsdfPdf);
        C3 X = cmulD(cmul(f, Li), (fabs(vdot(wi, n)) * weight / lightPdf));
        if (!raw) X = cmul(beta, cmulD(cadd(C3{0.f, 0.f, 0.f}, X), nLights));  // Ld = 0 + Ld1; * nLights; pathThroughput * ...
        stcf<F_LD1>(sr, X);
        pf |= PF_HAS_SH;
      }
    }
  };
  if (PRE) {
    bsdfHalf();
    const bool raw = (pf & PF_HAS_MIS) != 0 || !(isfinite(beta.r) && isfinite(beta.g) && isfinite(beta.b));
    lightHalf(raw);
    if (raw && pf != 0) pf |= PF_RAW_NEE;
  } else {
    lightHalf(true);
    bsdfHalf();
  }
  return pf;
}

// In-Li random floats (rng.randomFloat() inside PathIntegrator.Li): recorded
// values in host-buffer mode, the (pixel, sample) stream in counter mode.
struct TailSrc {
  const double* rec;
  DartRandom rng;
  int pos;
  DR_DEV void init(const RenderParams& rp, const BatchState& st, uint32_t slot, int startPos) {
    pos = startPos;
    if (st.tail) {
      rec = st.tailOff ? st.tail + (size_t)(st.tailOff[slot] - st.tailBase) : st.tail + (size_t)slot * rp.maxTail;
    } else {
      rec = nullptr;
      const int2 xy = st.pix[slot >> rp.sppShift];
      const uint64_t pixelIndex = (uint64_t)(xy.y - rp.extY0) * (uint64_t)rp.extW + (uint64_t)(xy.x - rp.extX0);
      rng.seed(dr_counter_key(rp.seed, pixelIndex, (uint64_t)(slot & (uint32_t)(rp.spp - 1)), 2));
      for (int i = 0; i < 2 * startPos; ++i) rng.step();
    }
  }
  // (the packed form's run length is re-read from the offsets instead of being held in a register: host-buffer replays only)
  DR_DEV double next(const RenderParams& rp, const BatchState& st, uint32_t slot) {
    if (rec) {
      bool in = pos < rp.maxTail;
      if (in && st.tailOff) in = (unsigned long long)pos < st.tailOff[slot + 1u] - st.tailOff[slot];
      double v = in ? rec[pos] : 0.0;
      ++pos;
      return v;
    }
    ++pos;
    return rng.randomFloat();
  }
};

// Everything k_shade_path reads that is indexed by the slot alone: fetched with independent loads (ONE memory
// round trip).  Arrays that were never written for this slot yield garbage that is never used.
struct ShadeIn {
  uint32_t slot, flags;
  int hprim, shOcc;
  double t;
  C3 L, beta, Ld1;
  F3 o, d;
  // this bounce's sample-vector entries (Appendix B), as fetched (sv_fetch1/2): lightNum, light comp, light pos,
  // bsdf dir, path dir, the two uComponents; evaluated when the item is shaded
  uint32_t raw[10], scr[10];
  bool valid;
  SlotRef sr;
};
template <bool QUAD>
DR_DEV void load_shade_in(const BatchState& st, const RenderParams& rp, int bounce, uint32_t entry, bool valid, ShadeIn* in) {
  const uint32_t slot = entry & ~Q_RESOLVE_BIT;
  in->valid = valid;
  in->slot = slot;
  if (!valid) return;
  const SlotRef sr = SlotRef::of(st, slot);
  in->sr = sr;
  in->flags = bounce == 0 ? PF_HAS_CONT : sr.u32<F_FLAGS>();
  if (entry & Q_RESOLVE_BIT) {
    // the path ended at the previous vertex (the entry says so, one iteration ahead of the state): only its light
    // estimate is pending -- flags, the occlusion result, L and the estimate are all this item reads
    in->hprim = -1;
    in->t = 0.0;
    in->shOcc = sr.i32<F_SHOCC>();
    in->L = ldcf<F_L>(sr);
    in->Ld1 = ldcf<F_LD1>(sr);
    in->beta = C3{0.f, 0.f, 0.f};
    in->o = in->d = F3{0.f, 0.f, 0.f};
    for (int k = 0; k < 10; ++k) in->raw[k] = in->scr[k] = 0u;
    return;
  }
  in->hprim = sr.i32<F_HPRIM>();
  in->t = sr.f64<F_HT>();
  if (bounce == 0) {  // the camera vertex: nothing pending, pathThroughput = 1, L = 0 (k_raygen does not store them)
    in->shOcc = 0;
    in->L = in->Ld1 = C3{0.f, 0.f, 0.f};
    in->beta = C3{1.f, 1.f, 1.f};
  } else {
    in->shOcc = sr.i32<F_SHOCC>();
    in->L = ldcf<F_L>(sr);
    in->beta = ldcf<F_BETA>(sr);
    in->Ld1 = ldcf<F_LD1>(sr);
  }
  in->o = ld3f<F_RO>(sr);
  in->d = ld3f<F_RD>(sr);
  if (bounce < 3) {
    sv_fetch1(rp, st, slot, 5 + 4 * bounce + 1, &in->raw[0], &in->scr[0]);
    sv_fetch1(rp, st, slot, 5 + 4 * bounce + 0, &in->raw[1], &in->scr[1]);
#pragma unroll
    for (int k = 0; k < 3; ++k) sv_fetch2(rp, st, slot, 5 + rp.n1D + 2 * (3 * bounce + k), &in->raw[2 + 2 * k], &in->scr[2 + 2 * k]);
    if (QUAD) {
      sv_fetch1(rp, st, slot, 5 + 4 * bounce + 3, &in->raw[8], &in->scr[8]);  // path-sample uComponent
      sv_fetch1(rp, st, slot, 5 + 4 * bounce + 2, &in->raw[9], &in->scr[9]);  // BSDF-sample uComponent of the light estimate
    }
  }
}

// DrRenderStats.shade_items / shade_vertices: the per-wave vertex counts stage_push kept in LDS, one no-return atomic
// per workgroup at the end of a shade kernel
DR_DEV void shade_count_init(PushStage& sm) {
  if (threadIdx.x < 256) sm.sobol[threadIdx.x] = c_sobol.lo[threadIdx.x];
  stage_init(sm);
}
DR_DEV void shade_count(PushStage& sm, TraceCounters* ctr, uint32_t nIn) {
  if (threadIdx.x == 0) {
    unsigned long long v = 0;
    for (int w = 0; w < 16; ++w) v += sm.nVert[w];
    if (v) atomicAdd(&ctr->shade_vertices, v);
    if (blockIdx.x == 0) atomicAdd(&ctr->shade_items, (unsigned long long)nIn);
  }
}

// -DDR_SHADE_PROF: a diagnostic build that stamps s_memtime between the phases of k_shade_path and sums, per phase,
// the cycles the waves spent there (tools/shade_prof.py prints them).  No stamp executes in the product build.
#ifdef DR_SHADE_PROF
__device__ unsigned long long g_shadeProf[16];
DR_DEV unsigned long long prof_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define PROF_DECL                                                                     \
  unsigned long long profT = prof_now();                                              \
  if (lane_id() == 0)                                                                 \
    for (int i_ = 0; i_ < 10; ++i_) s_push.prof[threadIdx.x >> 6][i_] = 0
#define PROF(i)                                                   \
  do {                                                            \
    const unsigned long long n_ = prof_now();                     \
    if (lane_id() == 0) s_push.prof[threadIdx.x >> 6][i] += n_ - profT; \
    profT = n_;                                                   \
  } while (0)
#define PROF_FLUSH                                                                                  \
  do {                                                                                              \
    if (lane_id() == 0)                                                                             \
      for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&g_shadeProf[i_], s_push.prof[threadIdx.x >> 6][i_]); \
  } while (0)
void shade_prof_dump() {
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_shadeProf), sizeof(h)) != hipSuccess) return;
  double tot = 0;
  for (int i = 0; i < 10; ++i) tot += (double)h[i];
  static const char* names[10] = {"state loads arrive", "resolve previous NEE", "primitive record + geometry + BSDF", "samples / RNG tail",
                                  "(state loads issued)", "NEE: both halves", "continuation sample + RR", "stores", "queue push", "queue flush"};
  for (int i = 0; i < 10; ++i) fprintf(stderr, "shade_prof %-38s %6.2f %%  (%.3g wave-cycles)\n", names[i], 100.0 * h[i] / tot, (double)h[i]);
  for (int i = 0; i < 16; ++i) h[i] = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_shadeProf), h, sizeof(h));
}
#else
#define PROF_DECL
#define PROF(i)
#define PROF_FLUSH
void shade_prof_dump() {}
#endif

// One vertex of PathIntegrator.Li (path_integrator.dart:44-119).
// Launch shape (measured, MEASUREMENTS.md): ONE 768-thread workgroup per CU = 3 waves per SIMD (168 VGPRs each);
// the waves are independent of each other (own staging region, own reservations, work taken in chunks), 4 waves per
// SIMD spill.
#ifndef DR_SHADE_WAVES
#define DR_SHADE_WAVES 3
#endif
#ifndef DR_SHADE_BLOCK
#define DR_SHADE_BLOCK 768
#endif
#ifndef DR_SHADE_GRID_PER_CU
#define DR_SHADE_GRID_PER_CU 1
#endif
#define DR_SHADE_GRID(numCU) ((numCU) * DR_SHADE_GRID_PER_CU)
// the general-material variants need more registers (188+ spilled VGPRs even at 2 waves per SIMD): 512 threads x 2
// waves per SIMD is ~20 % faster for them; the env-map path variant fits 3 waves once its env-map functions are
// called out of line (env_*_ni in dr_device.h)
#ifndef DR_SHADE_BLOCK_GEN
#define DR_SHADE_BLOCK_GEN 512
#endif
#ifndef DR_SHADE_WAVES_GEN
#define DR_SHADE_WAVES_GEN 2
#endif
#define SHADE_BLOCK_OF(general) ((general) ? DR_SHADE_BLOCK_GEN : DR_SHADE_BLOCK)
#define SHADE_WAVES_OF(general) ((general) ? DR_SHADE_WAVES_GEN : DR_SHADE_WAVES)
// LDS copy of the light tables and (when it is small too) the material table, behind the queue-staging block of the
// dynamic LDS; a workgroup-wide copy + barrier.
#define DR_LDS_LIGHT_BYTES (24 * 1024)  // light tables up to this size are staged in LDS (the queue staging takes ~96 KB of the 160)
#define DR_LDS_MAT_BYTES (24 * 1024)    // likewise the material table (64 B per material)
__host__ __device__ inline size_t light_table_bytes(const DScene& sc) {  // (+ 48 B per triangle: its f64 edges, LdsLights::ledges)
  return (size_t)sc.nlights * sizeof(DLight) + (size_t)sc.nltris * (sizeof(DLightTri) + 48) + (size_t)sc.ncdf * 4;
}
__host__ __device__ inline size_t mat_table_bytes(const DScene& sc) {
  const size_t b = (size_t)sc.nmats * 64;
  return b <= DR_LDS_MAT_BYTES ? b : 0;
}
DR_DEV LdsLights stage_lights(const DScene& sc, unsigned char* dyn, size_t pushBytes) {
  uint32_t* dst = (uint32_t*)(dyn + pushBytes);
  const uint32_t nL = sc.nlights * (uint32_t)(sizeof(DLight) / 4), nT = sc.nltris * (uint32_t)(sizeof(DLightTri) / 4), nC = sc.ncdf;
  const uint32_t nE = sc.nltris * 12u;  // f64 edges; nL and nT are even (8-byte multiples), so they start 8-byte aligned
  const uint32_t nM = (uint32_t)(mat_table_bytes(sc) / 4);
  const uint32_t* srcL = (const uint32_t*)sc.lights;
  const uint32_t* srcT = (const uint32_t*)sc.ltris;
  const uint32_t* srcC = (const uint32_t*)sc.lcdf;
  const uint32_t* srcM = (const uint32_t*)sc.mats;
  for (uint32_t i = threadIdx.x; i < nL; i += blockDim.x) dst[i] = srcL[i];
  for (uint32_t i = threadIdx.x; i < nT; i += blockDim.x) dst[nL + i] = srcT[i];
  for (uint32_t i = threadIdx.x; i < sc.nltris * 6u; i += blockDim.x) {
    // e1 = p2 - p1, e2 = p3 - p1 as Triangle.intersect forms them (triangle.dart:52-57): f64 differences of f32 values
    const uint32_t t = i / 6u, k = i % 6u;
    const float* p = sc.ltris[t].p;
    const double e = (double)p[3 + k] - (double)p[k % 3u];
    dst[nL + nT + 2u * i] = (uint32_t)__double2loint(e);
    dst[nL + nT + 2u * i + 1u] = (uint32_t)__double2hiint(e);
  }
  for (uint32_t i = threadIdx.x; i < nC; i += blockDim.x) dst[nL + nT + nE + i] = srcC[i];
  for (uint32_t i = threadIdx.x; i < nM; i += blockDim.x) dst[nL + nT + nE + nC + i] = srcM[i];
  __syncthreads();
  LdsLights lv;
  lv.lights = (lds_cu32*)dst;
  lv.ltris = (lds_cu32*)(dst + nL);
  lv.ledges = (lds_cu32*)(dst + nL + nT);
  lv.lcdf = (lds_cu32*)(dst + nL + nT + nE);
  lv.mats = nM ? (lds_cu32*)(dst + nL + nT + nE + nC) : (lds_cu32*)nullptr;
  lv.gmats = sc.mats;
  return lv;
}

template <bool ENV, bool QUAD, bool LLDS>
__global__ void __launch_bounds__(SHADE_BLOCK_OF(QUAD), SHADE_WAVES_OF(QUAD)) k_shade_path(DScene sc, RenderParams rp, BatchState st, StageQueues q, int bounce) {
  extern __shared__ __align__(16) unsigned char s_dyn[];
  PushStage& s_push = *(PushStage*)s_dyn;
  PushCtx pctx = {{0, 0, 0, 0}, 0, 0};
  // ENVQ (plain-triangle scenes under an environment map): nothing that touches the map is evaluated here.  A lane
  // whose light estimate picked the infinite light, or whose camera ray escaped, parks its inputs in the slot and
  // appends the slot to a fifth list (q.envQ); k_env works that list off at full lane width before the stage's rays
  // are traced.  One lane in nLights picks the map, so inline the wave paid the map's two dependent CDF searches, four
  // sin / cos and an acos / atan2 pair with ~1/9 of its lanes active, and the kernel carried ~110 spilled registers
  // per lane for them (C5: shade 776 ms per step, 34 % of the waves' time waiting behind the spill traffic).
  constexpr bool ENVQ = ENV && !QUAD;
  constexpr int NQ = ENVQ ? 5 : 4;
  using LV = typename std::conditional<LLDS, LdsLights, GlobalLights>::type;
  LV lv;
  if constexpr (LLDS) lv = stage_lights(sc, s_dyn, push_stage_bytes(SHADE_BLOCK_OF(QUAD), NQ));
  else lv = GlobalLights{sc.lights, sc.ltris, sc.lcdf, sc.mats};
  const uint32_t nIn = q.nActiveIn ? *q.nActiveIn : st.nslots;
  // Every iteration fetches its active-list entry, then the whole slot state with independent loads.  Prefetching the
  // next item's state did not pay (MEASUREMENTS.md row i: vmcnt retires in order, so a prefetch issued before the
  // shading code is waited for at its first load; issued after it, it costs 16 spilled registers).
  auto slotOf = [&](uint32_t i) -> uint32_t { return i < nIn ? (q.activeIn ? q.activeIn[i] : i) : 0u; };
  shade_count_init(s_push);
  // Work is handed out dynamically: a wave takes CHUNKS of 64 * DR_PUSH_ITERS consecutive active-list entries from the
  // launch's counter (q.work), one staging round each; the next chunk's number comes back with the round's queue
  // reservations (stage_flush), one round ahead of its use, so that a wave always knows its next entry.
  // (round 5) a THIN list -- fewer than a full chunk per resident wave -- is cut into shorter chunks, so that every wave takes one
  // and walks it in `ipc` iterations instead of a few waves walking DR_PUSH_ITERS iterations each while the others idle
  // (C5's stage 6: 26 K entries = 52 chunks of 512 for 3 072 waves)
  const uint32_t nWavesAll = gridDim.x * (blockDim.x >> 6);
  const uint32_t ipc = nIn >= nWavesAll * 64u * DR_PUSH_ITERS ? (uint32_t)DR_PUSH_ITERS
                                                              : max(1u, min((uint32_t)DR_PUSH_ITERS, (nIn + nWavesAll * 64u - 1u) / (nWavesAll * 64u)));
  const uint32_t CH = 64u * ipc;
  const uint32_t nChunks = (nIn + CH - 1u) / CH;
  const uint32_t lane = (uint32_t)lane_id();
  uint32_t cCur = 0u;
  if (lane == 0u) cCur = atomicAdd(q.work, 2u);
  cCur = wave_bcast_first(cCur);
  uint32_t cNext = cCur + 1u;
  uint32_t slotNext = slotOf(cCur * CH + lane);
  PROF_DECL;
  while (cCur < nChunks) {
    ShadeIn cur;
    const uint32_t slotCur = slotNext;
    const uint32_t idx = cCur * CH + pctx.iters * 64u + lane;
    const bool lastOfChunk = pctx.iters + 1u == ipc;
    slotNext = slotOf(lastOfChunk ? cNext * CH + lane : idx + 64u);
    load_shade_in<QUAD>(st, rp, bounce, slotCur, idx < nIn, &cur);
    PROF(4);
    const bool valid = cur.valid;
    uint32_t slot = cur.slot, pf = 0;
    bool pushCont = false, vert = false, deferred = false;
    bool envNee = false, envMiss = false;  // ENVQ: this lane's light estimate / escaped camera ray goes to k_env
    if (valid) {
      const SlotRef sr = cur.sr;
      const uint32_t flags = cur.flags;
      const int hprimIn = cur.hprim;
      const double t = cur.t;
      const int shOccIn = cur.shOcc;
      C3 L = cur.L;
      C3 beta = cur.beta;
      const C3 Ld1In = cur.Ld1;
      bool Lchanged = bounce == 0;  // L is stored when this stage changed it (the camera stage initialises it)
      const F3 o = cur.o, d = cur.d;
      auto su = [&](int k) -> float {
        if (st.svFloat) return __uint_as_float(cur.raw[k]);
        return (k >= 2 && k < 8 && (k & 1)) ? Sobol2_lds(s_push.sobol, cur.raw[k], cur.scr[k]) : VanDerCorput(cur.raw[k], cur.scr[k]);
      };
#ifdef DR_SHADE_PROF
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      PROF(0);
      if (bounce > 0 && rp.nLights > 0) {
        // L += pathThroughput * UniformSampleOneLight(...) of the previous vertex (path_integrator.dart:56-68)
        if (flags & PF_RAW_NEE) {
          C3 Ld = resolve_nee<ENV, QUAD>(sc, lv, sr, flags, shOccIn, Ld1In);
          C3 tot = cmulD(Ld, (double)rp.nLights);  // integrator.dart:113-116
          L = cadd(L, cmul(ldcf<F_BETANEE>(sr), tot));
          Lchanged = true;
        } else if ((flags & PF_HAS_SH) && shOccIn == 0) {
          L = cadd(L, Ld1In);  // the finished term (setup_nee, PRE); an occluded or absent estimate adds pathThroughput * 0
          Lchanged = true;
        }
      }
      PROF(1);
      const int prim = (flags & PF_HAS_CONT) ? hprimIn : -1;
      if (ENV && bounce == 0 && prim < 0 && sc.hasEnv) {
        // the camera ray escaped: Li = sum over lights of light.Le(ray) (sampler_renderer.dart:87-92); area
        // lights return 0 (light.dart:70-72), the infinite light its map
        if constexpr (ENVQ) {
          envMiss = true;      // k_env stores L = 0 + Le(ray); rd is still the camera ray's direction
          Lchanged = false;
        } else {
          L = cadd(L, env_Le(sc.env, d));
          Lchanged = true;
        }
      }
      if (ENV && QUAD && bounce > 0 && prim < 0 && (flags & PF_HAS_CONT) && (flags & PF_SPECULAR) && sc.hasEnv) {
        // a ray that left the scene after a specular bounce still sees the lights (path_integrator.dart:107-111)
        L = cadd(L, cmul(beta, env_Le(sc.env, d)));
        Lchanged = true;
      }
      if (prim >= 0 && bounce <= rp.maxDepth) {
        Bsdf bsdf;
        bool isQuad = false;
        const F3 wo = vneg(d);
        if constexpr (!QUAD) {
          // plain triangles, matte materials: the ray-independent part of the vertex comes from the 32-byte shading
          // record (dg.nn, the frame's sn, material, light), the rest is pointAt(t) and one cross product
          const ShTri tr = load_shtri(sc, (uint32_t)prim);
          if (bounce == 0) {  // bounces == 0 (path_integrator.dart:46)
            C3 Le = tr.light >= 0 ? light_L(lv.light(tr.light), tr.nn, wo) : C3{0.f, 0.f, 0.f};  // intersection.dart:60-63
            L = cadd(L, cmul(beta, Le));
            Lchanged = true;
          }
          bsdf = make_bsdf_pre(lv, vadd(o, vmul(d, t)), tr.nn, tr.sn, tr.mat);  // Ray.pointAt ray.dart:66-67
        } else {
          Tri tr = load_tri(sc, (uint32_t)prim);
          DGeo dg;
          isQuad = tr.kind != 0;
          DGeo dgs;  // shading geometry (GeometricPrimitive.getBSDF -> Shape.getShadingGeometry)
          if (isQuad) {
            quadric_dg_at(sc.quads[tr.quad], o, d, t, &dg);
            dgs = dg;
          } else if (sc.srec && __float_as_uint(sc.srec[7 * (size_t)prim + 6].x) != 0u) {
            const ShadeRec sr = load_srec(sc, (uint32_t)prim);
            tri_dg_srec(tr, sr, o, d, t, &dg);
            if (sr.flags & (DR_SHADING_N | DR_SHADING_S)) shading_geometry(sc, sr, tr.reverse, dg, &dgs);
            else dgs = dg;
          } else {
            tri_dg(tr.p1, tr.p2, tr.p3, tr.reverse, o, d, t, &dg);
            dgs = dg;
          }
          if (bounce == 0 || (flags & PF_SPECULAR)) {  // bounces == 0 || specularBounce (path_integrator.dart:46)
            C3 Le = tr.light >= 0 ? light_L(lv.light(tr.light), dg.nn, wo) : C3{0.f, 0.f, 0.f};  // intersection.dart:60-63
            L = cadd(L, cmul(beta, Le));
            Lchanged = true;
          }
          bsdf = make_bsdf<QUAD>(lv, dgs, tr.mat);
          bsdf.ng = dg.nn;  // BSDF(dgs, dgGeom.nn)
        }
        const F3 p = bsdf.p, n = bsdf.nn;
        const double eps = (isQuad ? 5.0e-4 : 1.0e-3) * t;  // triangle.dart:157; sphere.dart:169, disk.dart:98
        PROF(2);
        TailSrc ts;
        const int perNee = rp.nLights > 0 ? 7 : 0;
        if (bounce >= 3) ts.init(rp, st, slot, (bounce - 3) * (perNee + 3) + (bounce > 4 ? bounce - 4 : 0));
        if (rp.nLights > 0) {
          double lu, ls0, ls1, lsc, bs0, bs1, bsc = 0.0;
          if (bounce < 3) {  // SAMPLE_DEPTH (path_integrator.dart:139), slots of Appendix B
            lu = su(0);
            lsc = su(1);
            ls0 = su(2);
            ls1 = su(3);
            bs0 = su(4);
            bs1 = su(5);
            if (QUAD) bsc = su(9);
          } else {
            lu = ts.next(rp, st, slot);                                   // integrator.dart:96
            ls0 = (float)ts.next(rp, st, slot); ls1 = (float)ts.next(rp, st, slot); // LightSample.random light_sample.dart:46-51
            lsc = ts.next(rp, st, slot);
            bs0 = (float)ts.next(rp, st, slot); bs1 = (float)ts.next(rp, st, slot); // BSDFSample.random bsdf_sample.dart:37-42
            bsc = ts.next(rp, st, slot);                                  // uComponent
          }
          int lightNum = (int)floor(lu * rp.nLights);
          lightNum = lightNum < rp.nLights - 1 ? lightNum : rp.nLights - 1;
          PROF(3);
          if (ENVQ && lv.light(lightNum).kind == DR_LIGHT_INFINITE) {
            // parked for k_env (fields it overwrites with the estimate's results, or that only a raw estimate reads):
            // wo, the throughput the estimate is weighted with, the light and BSDF samples, the light's number
            envNee = true;
            st3f<F_MISD>(sr, wo);
            stcf<F_BETANEE>(sr, beta);
            sr.f32<F_LD1>(0) = (float)ls0;
            sr.f32<F_LD1>(1) = (float)ls1;
            sr.f32<F_LD2>(0) = (float)bs0;
            sr.f32<F_LD2>(1) = (float)bs1;
            sr.i32<F_MISLIGHT>() = lightNum;
          } else {
            pf |= setup_nee<(ENV && !ENVQ), QUAD, false, true>(sc, lv, sr, lightNum, bsdf, p, n, wo, ls0, ls1, lsc, bs0, bs1, bsc, beta,
                                                                (double)rp.nLights);
            if (pf & PF_RAW_NEE) stcf<F_BETANEE>(sr, beta);
          }
        }
        PROF(5);
        // Sample BSDF to get the new path direction (path_integrator.dart:70-90)
        double o0, o1, oc = 0.0;
        if (bounce < 3) {
          o0 = su(6);
          o1 = su(7);
          if (QUAD) oc = su(8);
        } else {
          o0 = (float)ts.next(rp, st, slot); o1 = (float)ts.next(rp, st, slot);
          oc = ts.next(rp, st, slot);
        }
        F3 wi = F3{0, 0, 0};
        double pdf = 0.0;
        const bool specular = QUAD && bsdf.mtype != DR_MATERIAL_MATTE;
        C3 f = specular ? spec_sample_f(bsdf, wo, &wi, oc, &pdf) : bsdf_sample_f(bsdf, wo, &wi, o0, o1, oc, &pdf, BSDF_ALL);
        bool alive = !(cblack(f) || pdf == 0.0);
        if (alive) {
          beta = cmul(beta, cdivD(cmulD(f, fabs(vdot(wi, n))), pdf));
          if (bounce > 3) {  // Russian roulette, drawn BEFORE the maxDepth test (path_integrator.dart:93-103)
            double lum = clum(beta);
            double cp = (lum != lum) ? lum : (lum < 0.5 ? lum : 0.5);  // Math.min(0.5, lum) propagates NaN
            if (ts.next(rp, st, slot) > cp) alive = false;
            else beta = cdivD(beta, cp);
          }
        }
        if (alive && bounce != rp.maxDepth) {
          st3f<F_RD>(sr, wi);
          stcf<F_BETA>(sr, beta);
          pf |= PF_HAS_CONT;
          if (specular) pf |= PF_SPECULAR;  // specularBounce (path_integrator.dart:87)
          pushCont = true;
        }
        PROF(6);
        st3f<F_RO>(sr, p);
        sr.f64<F_RTMIN>() = eps;
        vert = true;
      }
      if (Lchanged) stcf<F_L>(sr, L);
      // The path ended here with a finished light term pending: nothing is added to L after it, so k_film adds it
      // (L + Ld1 unless the shadow ray is blocked: the same f32 sum, one stage's round trip less).  Raw terms (a MIS
      // ray to resolve) still take the next stage.
      deferred = !pushCont && (pf & PF_HAS_SH) && !(pf & (PF_RAW_NEE | PF_HAS_MIS));
      if (deferred) pf |= PF_DEFERRED;
      // the camera stage initialises every slot's flags (k_film reads them); later stages need not store pf == 0:
      // the slot is in no queue, no later stage visits it, and the flags it keeps carry no PF_DEFERRED.  (A parked
      // estimate: k_env reads the continuation bit here and writes the finished flags.)
      if (bounce == 0 || pf || envNee) sr.u32<F_FLAGS>() = pf;
      PROF(7);
    }
    // A slot with a parked estimate: when its path CONTINUES it is in the next stage's list whatever the estimate turns out to be, so
    // its entry is pushed here, in place among its neighbours (round 5: pushed by k_env, these entries -- one vertex in nLights --
    // formed a second, nine times thinner sequence behind the list, and the stage after that split both again: a dense list cost
    // 100 -> 290 ps per entry over four stages in the 64-slot layout, MEASUREMENTS.md 5.8).  Only a parked slot whose path ENDS here
    // still enters the list in k_env, once its flags say whether a light term is left to fold in.
    stage_push<NQ>(s_push, pctx, pushCont, (pf & PF_HAS_MIS) != 0, (pf & PF_HAS_SH) != 0, (pf != 0 && !deferred && !envNee) || (envNee && pushCont), slot, Q_MIS_BIT,
                   vert, pushCont ? 0u : Q_RESOLVE_BIT, envNee || envMiss, envMiss ? Q_ENV_MISS_BIT : 0u);
    PROF(8);
    if (pctx.iters == ipc) {
      const uint32_t g = stage_flush<NQ>(s_push, pctx, q.closestQ, q.nClosest, q.anyQ, q.nAny, q.activeOut, q.nActiveOut, &q.ctr->shade_cont,
                                         q.work, q.envQ, q.nEnv);
      cCur = cNext;
      cNext = g;
    }
    PROF(9);
  }
  stage_finish(s_push, pctx, q.closestQ, q.anyQ, q.activeOut, &q.ctr->shade_cont);
  PROF_FLUSH;
  shade_count(s_push, q.ctr, nIn);
}

// Everything of PathIntegrator.Li that touches the InfiniteAreaLight's map, for plain-triangle scenes (the ENVQ
// instantiations of k_shade_path leave it here): a stage's list q.envQ holds
//   slot                    a vertex whose UniformSampleOneLight picked the infinite light: EstimateDirect's set-up
//                           (integrator.dart:119-185 with infinite_area_light.dart:92-131,190-205) from the inputs
//                           k_shade_path parked in the slot -- wo in misD, the throughput in betaNee, the light sample
//                           in Ld1[0..1], the BSDF sample in Ld2[0..1], the light's number in misLight; the vertex is
//                           ro, its frame and material the hit primitive's shading record.  Writes what setup_nee
//                           writes, finishes the slot's flags and queues its shadow / MIS rays and its entry of the
//                           next stage's list;
//   slot | Q_ENV_MISS_BIT   an escaped camera ray: L = 0 + light.Le(ray) (sampler_renderer.dart:87-92).
// Lane = list entry: every lane runs the map's CDF searches and trigonometry, where the inline form ran them with the
// one lane in nLights that picked the map.  Runs between a stage's shade launch and its traversals.
#ifndef DR_ENV_MARG_LDS_ROWS
#define DR_ENV_MARG_LDS_ROWS 8192  // tallest map whose marginal distribution k_env copies into LDS (64 KiB + the staging lists)
#endif
#ifndef DR_ENV_BLOCK
#define DR_ENV_BLOCK 256
#endif
struct LdsFloats {
  lds_cu32* p;
  DR_DEV float operator[](int i) const { return __uint_as_float(p[i]); }
};
struct GlobalFloats {
  const float* p;
  DR_DEV float operator[](int i) const { return p[i]; }
};
template <class MF>
struct MatsOnlyT {  // k_env's view of the scene tables: the material table and the map's marginal distribution
  const float4* mats;
  MF margFunc, margCdf;
  DR_DEV float4 mat(uint32_t m, int k) const { return mats[4 * (size_t)m + k]; }
};
// MLDS: the marginal distribution in LDS (maps of up to DR_ENV_MARG_LDS_ROWS rows: the usual case); taller maps search
// the global-memory arrays -- the same indices, a dozen cold round trips more per light sample.
template <bool MLDS>
__global__ void __launch_bounds__(DR_ENV_BLOCK) k_env(DScene sc, RenderParams rp, BatchState st, StageQueues q, int bounce) {
  extern __shared__ __align__(16) unsigned char s_dyn[];
  PushStage& s_push = *(PushStage*)s_dyn;
  PushCtx pctx = {{0, 0, 0, 0}, 0, 0};
  using MF = typename std::conditional<MLDS, LdsFloats, GlobalFloats>::type;
  using MatsOnly = MatsOnlyT<MF>;
  MatsOnly lv;
  lv.mats = sc.mats;
  if constexpr (MLDS) {
    // the marginal distribution (h func + h + 1 cdf entries) behind the staging lists: its 10-step search runs on LDS
    uint32_t* const sMarg = (uint32_t*)(s_dyn + push_stage_bytes(DR_ENV_BLOCK));
    const uint32_t h = (uint32_t)sc.env.h;
    for (uint32_t i = threadIdx.x; i < h; i += blockDim.x) sMarg[i] = __float_as_uint(sc.env.margFunc[i]);
    for (uint32_t i = threadIdx.x; i < h + 1u; i += blockDim.x) sMarg[h + i] = __float_as_uint(sc.env.margCdf[i]);
    lv.margFunc = LdsFloats{(lds_cu32*)sMarg};
    lv.margCdf = LdsFloats{(lds_cu32*)(sMarg + sc.env.h)};
  } else {
    lv.margFunc = GlobalFloats{sc.env.margFunc};
    lv.margCdf = GlobalFloats{sc.env.margCdf};
  }
  const uint32_t nIn = *q.nEnv;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nIter = (nIn + stride - 1) / stride;
  stage_init(s_push);
  for (uint32_t it = 0; it < nIter; ++it) {
    const uint32_t idx = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t slot = 0, pf = 0;
    bool active = false, cont = false;
    if (idx < nIn) {
      const uint32_t entry = q.envQ[idx];
      slot = entry & ~Q_ENV_MISS_BIT;
      const SlotRef sr = SlotRef::of(st, slot);
      if (entry & Q_ENV_MISS_BIT) {
        stcf<F_L>(sr, cadd(C3{0.f, 0.f, 0.f}, env_Le(sc.env, ld3f<F_RD>(sr))));
      } else {
        const uint32_t flags = sr.u32<F_FLAGS>();  // the continuation bit(s) k_shade_path stored
        const ShTri tr = load_shtri(sc, (uint32_t)sr.i32<F_HPRIM>());
        const F3 p = ld3f<F_RO>(sr), wo = ld3f<F_MISD>(sr);
        const C3 beta = ldcf<F_BETANEE>(sr);
        const double ls0 = sr.f32<F_LD1>(0), ls1 = sr.f32<F_LD1>(1), bs0 = sr.f32<F_LD2>(0), bs1 = sr.f32<F_LD2>(1);
        const int lightNum = sr.i32<F_MISLIGHT>();
        const Bsdf bsdf = make_bsdf_pre(lv, p, tr.nn, tr.sn, tr.mat);
        pf = flags | setup_nee<true, false, false, true, MatsOnly, true>(sc, lv, sr, lightNum, bsdf, p, bsdf.nn, wo, ls0, ls1, 0.0, bs0, bs1, 0.0,
                                                                         beta, (double)rp.nLights);
        // (a raw estimate reads the throughput from betaNee: it is there already)
        cont = (flags & PF_HAS_CONT) != 0;
        const bool deferred = !cont && (pf & PF_HAS_SH) && !(pf & (PF_RAW_NEE | PF_HAS_MIS));
        if (deferred) pf |= PF_DEFERRED;
        sr.u32<F_FLAGS>() = pf;
        active = pf != 0 && !deferred;
      }
    }
    // (a continuing path's entry of the next stage's list was pushed by k_shade_path, in place: see there)
    stage_push(s_push, pctx, false, (pf & PF_HAS_MIS) != 0, (pf & PF_HAS_SH) != 0, active && !cont, slot, Q_MIS_BIT, false, Q_RESOLVE_BIT);
    if (pctx.iters == DR_PUSH_ITERS || it + 1 == nIter)
      stage_flush(s_push, pctx, q.closestQ, q.nClosest, q.anyQ, q.nAny, q.activeOut, q.nActiveOut, &q.ctr->shade_cont);
  }
  stage_finish(s_push, pctx, q.closestQ, q.anyQ, q.activeOut, &q.ctr->shade_cont);
}

// DirectLightingIntegrator.Li with strategy "all" or "one" (direct_lighting_integrator.dart:30-68;
// integrator.dart:39-117; "one": a single call whose light comes from the sample's lightNum slot).  Stage s sets up EstimateDirect call s of UniformSampleAllLights (rp.dstages[s]:
// sample j of light i) at the camera hit and folds in the result of call s-1; the last stage finishes the
// sum.  st.betaNee() carries the current light's Ld, st.beta() the running L of UniformSampleAllLights.
template <bool QUAD, bool LLDS>
__global__ void __launch_bounds__(SHADE_BLOCK_OF(QUAD), SHADE_WAVES_OF(QUAD)) k_shade_direct(DScene sc, RenderParams rp, BatchState st, StageQueues q, int stage) {
  extern __shared__ __align__(16) unsigned char s_dyn[];
  PushStage& s_push = *(PushStage*)s_dyn;
  PushCtx pctx = {{0, 0, 0, 0}, 0, 0};
  using LV = typename std::conditional<LLDS, LdsLights, GlobalLights>::type;
  LV lv;
  if constexpr (LLDS) lv = stage_lights(sc, s_dyn, push_stage_bytes(SHADE_BLOCK_OF(QUAD)));
  else lv = GlobalLights{sc.lights, sc.ltris, sc.lcdf, sc.mats};
  const uint32_t nIn = q.nActiveIn ? *q.nActiveIn : st.nslots;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nIter = (nIn + stride - 1) / stride;
  const int nCalls = rp.nDirectStages;
  shade_count_init(s_push);
  DirectStage prev{}, cur{};
  if (stage > 0) prev = rp.dstages[stage - 1];
  if (stage < nCalls) cur = rp.dstages[stage];
  for (uint32_t it = 0; it < nIter; ++it) {
    const uint32_t idx = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = idx < nIn;
    uint32_t slot = 0, pf = 0;
    bool again = false;
    if (valid) {
      slot = q.activeIn ? q.activeIn[idx] : idx;
      const SlotRef sr = SlotRef::of(st, slot);
      const uint32_t flags = stage == 0 ? PF_HAS_CONT : sr.u32<F_FLAGS>();  // k_raygen leaves flags / L / beta unwritten
      const int prim = sr.i32<F_HPRIM>();
      if (prim >= 0) {
        Tri tr = load_tri(sc, (uint32_t)prim);
        const F3 d = ld3f<F_RD>(sr);
        const F3 wo = vneg(d);
        C3 L = stage == 0 ? C3{0.f, 0.f, 0.f} : ldcf<F_L>(sr);
        C3 Lall = stage == 0 ? C3{0.f, 0.f, 0.f} : ldcf<F_BETA>(sr);
        C3 Ld = C3{0.f, 0.f, 0.f};
        DGeo dg, dgs;
        const bool isQuad = QUAD && tr.kind != 0;
        const bool hasRec = QUAD && !isQuad && sc.srec && __float_as_uint(sc.srec[7 * (size_t)prim + 6].x) != 0u;
        ShadeRec srec;
        if (hasRec) srec = load_srec(sc, (uint32_t)prim);
        if (stage == 0) {
          const F3 o = ld3f<F_RO>(sr);
          const double t = sr.f64<F_HT>();
          if (isQuad) {
            quadric_dg_at(sc.quads[tr.quad], o, d, t, &dg);
            st3f<F_RO0>(sr, o);  // later stages rebuild the hit from the camera ray
          } else if (hasRec) {
            tri_dg_srec(tr, srec, o, d, t, &dg);
            st3f<F_RO0>(sr, o);
          } else {
            tri_dg(tr.p1, tr.p2, tr.p3, tr.reverse, o, d, t, &dg);
          }
          C3 Le = tr.light >= 0 ? light_L(lv.light(tr.light), dg.nn, wo) : C3{0.f, 0.f, 0.f};
          L = cadd(L, Le);
          Lall = C3{0.f, 0.f, 0.f};
          st3f<F_RO>(sr, dg.p);
          sr.f64<F_RTMIN>() = (isQuad ? 5.0e-4 : 1.0e-3) * t;
        } else {
          if (isQuad) quadric_dg_at(sc.quads[tr.quad], ld3f<F_RO0>(sr), d, sr.f64<F_HT>(), &dg);
          else if (hasRec) tri_dg_srec(tr, srec, ld3f<F_RO0>(sr), d, sr.f64<F_HT>(), &dg);
          else tri_dg(tr.p1, tr.p2, tr.p3, tr.reverse, F3{0, 0, 0}, d, 0.0, &dg);
          dg.p = ld3f<F_RO>(sr);
          Ld = ldcf<F_BETANEE>(sr);
          const C3 Ed = resolve_nee<true, QUAD>(sc, lv, sr, flags, sr.i32<F_SHOCC>(), ldcf<F_LD1>(sr));
          Ld = cadd(Ld, Ed);  // Ld += EstimateDirect
          if (prev.light < 0) {
            Lall = cmulD(Ed, (double)rp.nLights);  // UniformSampleOneLight: EstimateDirect(...) * nLights (integrator.dart:113-116)
            Ld = C3{0.f, 0.f, 0.f};
          } else if (prev.last) {
            Lall = cadd(Lall, cdivD(Ld, (double)prev.n));  // L += Ld / nSamples
            Ld = C3{0.f, 0.f, 0.f};
          }
        }
        if (stage < nCalls) {
          if (hasRec && (srec.flags & (DR_SHADING_N | DR_SHADING_S))) shading_geometry(sc, srec, tr.reverse, dg, &dgs);
          else dgs = dg;
          Bsdf bsdf = make_bsdf<QUAD>(lv, dgs, tr.mat);  // mirror / glass are refused for DirectLighting; Oren-Nayar is not
          bsdf.ng = dg.nn;
          // sample slots of this call (direct_lighting_integrator.dart:70-87)
          float l0, l1, b0, b1;
          sv_pair(rp, st, slot, cur.lp, &l0, &l1);
          sv_pair(rp, st, slot, cur.bd, &b0, &b1);
          double lsc = sv_one(rp, st, slot, cur.lc);
          double ls0 = l0, ls1 = l1, bs0 = b0, bs1 = b1;
          double bsc = QUAD ? (double)sv_one(rp, st, slot, cur.bc) : 0.0;
          int light = cur.light;
          if (light < 0) {  // strategy "one": lightNum = min(floor(sample.oneD[lightNumOffset][0] * nLights), nLights - 1) (integrator.dart:92-99)
            light = (int)floor((double)sv_one(rp, st, slot, cur.ln) * (double)rp.nLights);
            light = min(light, rp.nLights - 1);
          }
          pf |= setup_nee<true, QUAD, false, false>(sc, lv, sr, light, bsdf, bsdf.p, bsdf.nn, wo, ls0, ls1, lsc, bs0, bs1, bsc);
          again = true;
        } else {
          if (rp.nLights > 0) L = cadd(L, Lall);
          if (!rp.dlSpecular && 0 + 1 < rp.maxDepth) {  // SpecularReflect / SpecularTransmit find no specular lobe: += 0
            L = cadd(L, C3{0.f, 0.f, 0.f});             // (scenes with mirror / glass: k_shade_spec adds them)
            L = cadd(L, C3{0.f, 0.f, 0.f});
          }
        }
        stcf<F_L>(sr, L);
        stcf<F_BETA>(sr, Lall);
        stcf<F_BETANEE>(sr, Ld);
      } else if (stage == 0) {
        // escaped camera ray: Li = sum of light.Le(ray) (sampler_renderer.dart:87-92) -- the env map's, else 0
        stcf<F_L>(sr, sc.hasEnv ? env_Le(sc.env, ld3f<F_RD>(sr)) : C3{0.f, 0.f, 0.f});
      }
      sr.u32<F_FLAGS>() = pf;
    }
    stage_push(s_push, pctx, false, (pf & PF_HAS_MIS) != 0, (pf & PF_HAS_SH) != 0, again, slot, Q_MIS_BIT, again);
    if (pctx.iters == DR_PUSH_ITERS || it + 1 == nIter)
      stage_flush(s_push, pctx, q.closestQ, q.nClosest, q.anyQ, q.nAny, q.activeOut, q.nActiveOut, &q.ctr->shade_cont);
  }
  stage_finish(s_push, pctx, q.closestQ, q.anyQ, q.activeOut, &q.ctr->shade_cont);
  shade_count(s_push, q.ctr, nIn);
}

// DirectLightingIntegrator.Li's recursion through Integrator.SpecularReflect / SpecularTransmit
// (direct_lighting_integrator.dart:59-65, integrator.dart:187-290) for scenes with mirror / glass, as an explicit
// depth-first walk: every slot works on ONE vertex of its ray tree per round (closest-hit trace, the all-lights
// stages of k_shade_direct, then this kernel).  A vertex with a specular lobe is suspended in a SpecFrame while its
// child ray is traced -- the reflected one first, then the transmitted one, the reference's order -- and a finished
// vertex hands its radiance to the frame below it, `L += f * Li * (AbsDot(wi, n) / pdf)` with the reference's f32
// rounding, until the stack is empty and st.L holds the camera sample's radiance.  The three RNG draws each call
// burns (BSDFSample.random) only select among SEVERAL matching lobes; a BSDF has at most one specular lobe per
// hemisphere here, so they never reach the image and the keyed per-sample streams need not reproduce them.
// In: q.activeIn = the slots whose vertex was traced this round.  Out: q.activeOut = the slots that launched a child
// ray (ro / rd / rtmin written), the next round's list.
__global__ void __launch_bounds__(SHADE_BLOCK_OF(true), SHADE_WAVES_OF(true)) k_shade_spec(DScene sc, RenderParams rp, BatchState st, StageQueues q) {
  extern __shared__ __align__(16) unsigned char s_dyn[];
  PushStage& s_push = *(PushStage*)s_dyn;
  PushCtx pctx = {{0, 0, 0, 0}, 0, 0};
  const GlobalLights lv{sc.lights, sc.ltris, sc.lcdf, sc.mats};
  const uint32_t nIn = q.nActiveIn ? *q.nActiveIn : st.nslots;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nIter = (nIn + stride - 1) / stride;
  shade_count_init(s_push);
  for (uint32_t it = 0; it < nIter; ++it) {
    const uint32_t idx = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t slot = 0;
    bool launched = false;
    if (idx < nIn) {
      slot = q.activeIn ? q.activeIn[idx] : idx;
      const SlotRef sr = SlotRef::of(st, slot);
      int sp = st.specSp[slot];
      auto frameAt = [&](int level) -> SpecFrame* { return (SpecFrame*)(st.specFrames + ((size_t)level * st.cap + slot) * DR_SPEC_FRAME_WORDS); };
      auto launch = [&](F3 o, F3 dir, double eps) {  // RayDifferential.child(p, wi, ray, isect.rayEpsilon): [eps, inf)
        st3f<F_RO>(sr, o);
        st3f<F_RD>(sr, dir);
        sr.f64<F_RTMIN>() = eps;
        launched = true;
      };
      const int prim = sr.i32<F_HPRIM>();
      C3 V = ldcf<F_L>(sr);  // this vertex's Le + direct lighting (a hit), or the lights' Le along an escaped ray
      if (prim >= 0 && sp + 1 < rp.maxDepth) {  // ray.depth + 1 < maxDepth: ray.depth == the number of suspended ancestors
        // the hit's geometry and BSDF, rebuilt as the later stages of k_shade_direct rebuild them
        Tri tr = load_tri(sc, (uint32_t)prim);
        const F3 d = ld3f<F_RD>(sr);
        const F3 wo = vneg(d);
        DGeo dg, dgs;
        const bool isQuad = tr.kind != 0;
        const bool hasRec = !isQuad && sc.srec && __float_as_uint(sc.srec[7 * (size_t)prim + 6].x) != 0u;
        ShadeRec srec;
        if (hasRec) srec = load_srec(sc, (uint32_t)prim);
        if (isQuad) quadric_dg_at(sc.quads[tr.quad], ld3f<F_RO0>(sr), d, sr.f64<F_HT>(), &dg);
        else if (hasRec) tri_dg_srec(tr, srec, ld3f<F_RO0>(sr), d, sr.f64<F_HT>(), &dg);
        else tri_dg(tr.p1, tr.p2, tr.p3, tr.reverse, F3{0, 0, 0}, d, 0.0, &dg);
        dg.p = ld3f<F_RO>(sr);
        if (hasRec && (srec.flags & (DR_SHADING_N | DR_SHADING_S))) shading_geometry(sc, srec, tr.reverse, dg, &dgs);
        else dgs = dg;
        Bsdf bsdf = make_bsdf<true>(lv, dgs, tr.mat);
        bsdf.ng = dg.nn;
        const F3 p = bsdf.p, n = bsdf.nn;
        const double eps = sr.f64<F_RTMIN>();  // isect.rayEpsilon, stored by stage 0
        F3 wr = F3{0, 0, 0}, wt = F3{0, 0, 0};
        double pr = 0.0, pt = 0.0;
        C3 fr = spec_lobe_sample_f(bsdf, wo, &wr, &pr, true);
        C3 ft = spec_lobe_sample_f(bsdf, wo, &wt, &pt, false);
        const bool hasR = pr > 0.0 && !cblack(fr) && fabs(vdot(wr, n)) != 0.0;
        const bool hasT = pt > 0.0 && !cblack(ft) && fabs(vdot(wt, n)) != 0.0;
        if (hasR || hasT) {
          SpecFrame* f = frameAt(sp);
          f->L[0] = V.r; f->L[1] = V.g; f->L[2] = V.b;
          f->p[0] = p.x; f->p[1] = p.y; f->p[2] = p.z;
          f->eps = eps;
          if (hasT) {
            f->ft[0] = ft.r; f->ft[1] = ft.g; f->ft[2] = ft.b;
            f->wt[0] = wt.x; f->wt[1] = wt.y; f->wt[2] = wt.z;
            f->st = fabs(vdot(wt, n)) / pt;
          }
          if (hasR) {
            f->fr[0] = fr.r; f->fr[1] = fr.g; f->fr[2] = fr.b;
            f->sr = fabs(vdot(wr, n)) / pr;
            f->state = 1u | (hasT ? 256u : 0u);
            launch(p, wr, eps);
          } else {
            f->L[0] = V.r + 0.f; f->L[1] = V.g + 0.f; f->L[2] = V.b + 0.f;  // L += SpecularReflect(...) == Spectrum(0)
            f->state = 2u;
            launch(p, wt, eps);
          }
          ++sp;
        } else {
          V = cadd(cadd(V, C3{0.f, 0.f, 0.f}), C3{0.f, 0.f, 0.f});  // both calls return Spectrum(0)
        }
      }
      // a finished vertex returns to the frames below it
      while (!launched && sp > 0) {
        SpecFrame* f = frameAt(sp - 1);
        C3 L = C3{f->L[0], f->L[1], f->L[2]};
        if ((f->state & 255u) == 1u) {
          L = cadd(L, cmulD(cmul(C3{f->fr[0], f->fr[1], f->fr[2]}, V), f->sr));  // L += f * Li * (AbsDot(wi, n) / pdf)
          if (f->state & 256u) {
            f->L[0] = L.r; f->L[1] = L.g; f->L[2] = L.b;
            f->state = 2u;
            launch(F3{f->p[0], f->p[1], f->p[2]}, F3{f->wt[0], f->wt[1], f->wt[2]}, f->eps);
          } else {
            V = cadd(L, C3{0.f, 0.f, 0.f});  // + SpecularTransmit(...) == Spectrum(0)
            --sp;
          }
        } else {
          V = cadd(L, cmulD(cmul(C3{f->ft[0], f->ft[1], f->ft[2]}, V), f->st));
          --sp;
        }
      }
      st.specSp[slot] = sp;
      if (!launched) stcf<F_L>(sr, V);  // the camera sample's radiance
    }
    stage_push(s_push, pctx, false, false, false, launched, slot, Q_MIS_BIT, false);
    if (pctx.iters == DR_PUSH_ITERS || it + 1 == nIter)
      stage_flush(s_push, pctx, q.closestQ, q.nClosest, q.anyQ, q.nAny, q.activeOut, q.nActiveOut, nullptr);
  }
  stage_finish(s_push, pctx, q.closestQ, q.anyQ, q.activeOut);
}

// ---------------------------------------------------------------------------
// film (image_film.dart:99-185).  First lane = sample (coalesced reads of L and the image sample; XYZ conversion
// and filter weight; the sample's contribution to its OWN pixel parked in LDS, contributions to other pixels -- wide
// filters, or imageX exactly integral -- sent through float atomics), then lane = (pixel, channel) adds the parked
// contributions in the reference's sample order: one serial f32 chain per lane.
// spp >= 64: a block owns 16 pixels and walks their samples 64 at a time (one 64-slot tile per pixel and pass: 1024
// samples in LDS), so that all 64 lanes of the adding wave carry a chain (16 pixels x X, Y, Z, weight) whatever spp is;
// with one 1024-slot chunk per block only 4096 / spp lanes did (4 at 1024 spp: 22 of the kernel's 32 ms).
// spp < 64: a block owns 1024 consecutive slots = 1024 / spp whole pixels, one pass.
// ---------------------------------------------------------------------------
#define DR_FILM_CHUNK 1024
__global__ void __launch_bounds__(256) k_film(RenderParams rp, BatchState st, const float* table, uint32_t npix, float* film) {
  __shared__ float s_X[DR_FILM_CHUNK + 16], s_Y[DR_FILM_CHUNK + 16], s_Z[DR_FILM_CHUNK + 16], s_W[DR_FILM_CHUNK + 16];
  __shared__ uint8_t s_own[DR_FILM_CHUNK + 16];
  __shared__ float s_one[1];
  const uint32_t cap = st.tileStride;  // words per 64-slot tile
  const uint32_t spp = (uint32_t)rp.spp;
  const uint32_t nslots = st.nslots;
  const bool tiled = spp >= 64u;                              // 16 pixels x 64 samples per pass
  const uint32_t seg = tiled ? 64u : spp;                     // samples of one pixel per pass
  const uint32_t segShift = tiled ? 6u : (uint32_t)rp.sppShift;
  const uint32_t str = tiled ? 65u : spp;                     // LDS row pitch (65: the 16 rows start in 16 different banks)
  const uint32_t nPixBlk = DR_FILM_CHUNK >> segShift;         // pixels of this block
  const uint32_t p0 = blockIdx.x * nPixBlk;                   // its first batch pixel
  const uint32_t nPass = tiled ? spp >> 6 : 1u;
  if (threadIdx.x == 0) s_one[0] = 1.0f;
  float acc = 0.f;  // tiled: lane (pixel, channel) of wave 0 keeps its chain across the passes
  auto add = [&](float& a, uint8_t own, float wt, float v) {
    const float t = (float)((double)a + (double)wt * (double)v);  // _Lxyz += wt * xyz; weightSum += wt (wt * 1.0 is exact)
    a = own ? t : a;
  };
  // Eight samples per trip, branch-free, every load unconditional (the weight lane reads its factor 1.0 from LDS with
  // stride 0): a group's LDS reads are in flight together and only the additions are serial.
  auto chain = [&](float& a, uint32_t row, uint32_t c, uint32_t n) {
    const float* src = c == 0 ? s_X : (c == 1 ? s_Y : (c == 2 ? s_Z : s_one));
    const uint32_t step = c == 3u ? 0u : 1u, first = c == 3u ? 0u : row;
    if (n >= 8u) {
      for (uint32_t i = 0; i < n; i += 8u) {
        float v[8], wv[8];
        uint8_t ow[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
          wv[j] = s_W[row + i + j];
          ow[j] = s_own[row + i + j];
          v[j] = src[first + (i + j) * step];
        }
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) add(a, ow[j], wv[j], v[j]);
      }
    } else {
      for (uint32_t i = 0; i < n; ++i) add(a, s_own[row + i], s_W[row + i], src[first + i * step]);
    }
  };
  auto flush = [&](uint32_t pl, uint32_t c, float a) {
    const uint32_t p = p0 + pl;
    if (p < npix) {
      const int2 xy = st.pix[p];
      if (xy.x >= rp.left && xy.x < rp.left + rp.width && xy.y >= rp.top && xy.y < rp.top + rp.height)
        atomicAdd(film + 4 * ((size_t)(xy.y - rp.top) * rp.width + (size_t)(xy.x - rp.left)) + c, a);
    }
  };
  for (uint32_t pass = 0; pass < nPass; ++pass) {
    if (pass) __syncthreads();  // the previous pass's rows have been added
    for (uint32_t e = threadIdx.x; e < DR_FILM_CHUNK; e += 256u) {
      const uint32_t pl = e >> segShift, j = e & (seg - 1u);
      const uint32_t le = pl * str + j;  // where this sample parks
      const uint64_t s64 = (uint64_t)(p0 + pl) * spp + (uint64_t)pass * seg + j;
      uint8_t own = 0;
      if (p0 + pl < npix && s64 < nslots) {
        const uint32_t s = (uint32_t)s64;
        const int2 xy = st.pix[p0 + pl];
        C3 L = ldc(st.L(), cap, s);
        if (rp.deferredNee) {
          // the last light term of a path that ended (PF_DEFERRED, k_shade_path): L += pathThroughput * Ld unless the
          // shadow ray found an occluder (path_integrator.dart:56-68); independent loads, no second round trip
          const SlotRef sr = SlotRef::of(st, s);
          const uint32_t flags = sr.u32<F_FLAGS>();
          const int occ = sr.i32<F_SHOCC>();
          const C3 Ld1 = ldcf<F_LD1>(sr);
          if ((flags & PF_DEFERRED) && occ == 0) L = cadd(L, Ld1);
        }
        // guards of sampler_renderer.dart:181-193
        double lum = clum(L);
        if (L.r != L.r || L.g != L.g || L.b != L.b) L = C3{0.f, 0.f, 0.f};
        else if (lum < -1e-5) L = C3{0.f, 0.f, 0.f};
        else if (isinf(lum)) L = C3{0.f, 0.f, 0.f};
        float sx, sy;
        sv_pair(rp, st, s, 0, &sx, &sy);
        const double dimageX = ((double)xy.x + (double)sx) - 0.5;
        const double dimageY = ((double)xy.y + (double)sy) - 0.5;
        int x0 = (int)ceil(dimageX - rp.fxw), x1 = (int)floor(dimageX + rp.fxw);
        int y0 = (int)ceil(dimageY - rp.fyw), y1 = (int)floor(dimageY + rp.fyw);
        x0 = max(x0, rp.left); x1 = min(x1, rp.left + rp.width - 1);
        y0 = max(y0, rp.top);  y1 = min(y1, rp.top + rp.height - 1);
        if ((x1 - x0) >= 0 && (y1 - y0) >= 0) {
          // L.toXYZ(): Float32List store (xyz_color.dart:39-42; spectrum.dart:294-298)
          const float X = (float)(0.412453 * (double)L.r + 0.357580 * (double)L.g + 0.180423 * (double)L.b);
          const float Y = (float)(0.212671 * (double)L.r + 0.715160 * (double)L.g + 0.072169 * (double)L.b);
          const float Z = (float)(0.019334 * (double)L.r + 0.119193 * (double)L.g + 0.950227 * (double)L.b);
          for (int y = y0; y <= y1; ++y) {
            const double fy = fabs(((double)y - dimageY) * rp.invY * 16);
            const int iy = min((int)floor(fy), 15);
            for (int x = x0; x <= x1; ++x) {
              const double fx = fabs(((double)x - dimageX) * rp.invX * 16);
              const int ix = min((int)floor(fx), 15);
              const float wt = table[iy * 16 + ix];
              if (x == xy.x && y == xy.y) {
                s_X[le] = X; s_Y[le] = Y; s_Z[le] = Z; s_W[le] = wt;
                own = 1;
              } else {
                float* px = film + 4 * ((size_t)(y - rp.top) * rp.width + (size_t)(x - rp.left));
                atomicAdd(px + 0, (float)((double)wt * (double)X));
                atomicAdd(px + 1, (float)((double)wt * (double)Y));
                atomicAdd(px + 2, (float)((double)wt * (double)Z));
                atomicAdd(px + 3, wt);
              }
            }
          }
        }
      }
      s_own[le] = own;
    }
    __syncthreads();
    if (tiled) {
      if (threadIdx.x < 64u) chain(acc, (threadIdx.x >> 2) * str, threadIdx.x & 3u, 64u);  // 16 pixels x 4 channels
    } else {
      for (uint32_t w = threadIdx.x; w < 4u * nPixBlk; w += 256u) {
        float a = 0.f;
        chain(a, (w >> 2) * str, w & 3u, spp);
        flush(w >> 2, w & 3u, a);
      }
    }
  }
  if (tiled && threadIdx.x < 64u) flush(threadIdx.x >> 2, threadIdx.x & 3u, acc);
}

// ImageFilm.writeImage (image_film.dart:268-299), splat == 0.
__global__ void k_film_resolve(const float* film, int64_t npix, float* rgb) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const double X = film[4 * p], Y = film[4 * p + 1], Z = film[4 * p + 2], w = film[4 * p + 3];
  const double c0 = 3.240479 * X - 1.537150 * Y - 0.498535 * Z;  // spectrum.dart:287-291
  const double c1 = -0.969256 * X + 1.875991 * Y + 0.041556 * Z;
  const double c2 = 0.055648 * X - 0.204043 * Y + 1.057311 * Z;
  float r = 0.f, g = 0.f, b = 0.f;
  if (w != 0.0) {
    const double invWt = 1.0 / w;
    r = (float)fmax(0.0, c0 * invWt);
    g = (float)fmax(0.0, c1 * invWt);
    b = (float)fmax(0.0, c2 * invWt);
  }
  rgb[3 * p] = (float)((double)r + 0.0);
  rgb[3 * p + 1] = (float)((double)g + 0.0);
  rgb[3 * p + 2] = (float)((double)b + 0.0);
}

// float4 copy: the measured HBM-bandwidth denominator of the roofline.  The shape that reads fastest on MI355X
// (tools/copy_bw.hip: 5.5 TB/s against 4.6-4.8 for a grid-stride loop): every workgroup owns one contiguous chunk,
// four independent 16-byte non-temporal loads per lane are in flight before the first store.
typedef float dr_v4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ src, float4* __restrict__ dst, uint64_t n4) {
  const dr_v4* s = (const dr_v4*)src;
  dr_v4* d = (dr_v4*)dst;
  const uint64_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const uint64_t b0 = (uint64_t)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
  uint64_t i = b0 + threadIdx.x;
  for (; i + 3 * 256 < b1; i += 4 * 256) {
    const dr_v4 a = __builtin_nontemporal_load(s + i), b = __builtin_nontemporal_load(s + i + 256);
    const dr_v4 c = __builtin_nontemporal_load(s + i + 512), e = __builtin_nontemporal_load(s + i + 768);
    __builtin_nontemporal_store(a, d + i);
    __builtin_nontemporal_store(b, d + i + 256);
    __builtin_nontemporal_store(c, d + i + 512);
    __builtin_nontemporal_store(e, d + i + 768);
  }
  for (; i < b1; i += 256) d[i] = s[i];
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
void launch_gather_tris(const float* verts, const uint32_t* idx, const uint32_t* mat, const int32_t* light,
                        const uint8_t* rev, float4* out, uint64_t ntris, hipStream_t s) {
  if (ntris == 0) return;
  hipLaunchKernelGGL(k_gather_tris, dim3((unsigned)((ntris + 255) / 256)), dim3(256), 0, s, verts, idx, mat, light, rev, out,
                     ntris);
}
void launch_make_shtris(const DScene& sc, float4* out, uint64_t ntris, hipStream_t s) {
  if (ntris == 0) return;
  hipLaunchKernelGGL(k_make_shtris, dim3((unsigned)((ntris + 255) / 256)), dim3(256), 0, s, sc, out, ntris);
}
void launch_gen_samples(const RenderParams& rp, const BatchState& stIn, uint32_t npix, hipStream_t s) {
  // the pre-pass (k_gen_burnin) pays where the shuffle kernels run at one wave per SIMD -- above 256 spp: C5 151 -> 141.5 ms, the 1024-spp
  // image 119 -> 112 -- and costs a launch where nine waves per CU hide the burn-in anyway (C2, 256 spp: 9.1 -> 9.4 ms)
  BatchState st = stIn;
  if (rp.spp <= 256) st.genState = nullptr;
  const int nBlocks = rp.blocks ? rp.nBlocks : 3 + rp.n1D + (rp.nFloats - 5 - rp.n1D) / 2;
  const dim3 grid((npix + 63) / 64, nBlocks);
  if (!st.svFloat && rp.spp >= 64) {
    const int nGen = rp.genMask ? __builtin_popcountll(rp.genMask) : nBlocks;  // compact form (rp.blocks is null), whole index runs per pixel
    // (above 1024 spp a pixel's table is 4 / 8 KB: 32 / 16 pixels per group keep the group's tables within 128 KB; the lazy-generation
    // exit of k_gen_samples_lm tests ONE 64-pixel group per workgroup: the group size must divide 64)
    const int ln = rp.spp <= 1024 ? 64 : (rp.spp <= 2048 ? 32 : 16);
    const dim3 g((npix + ln - 1) / ln, nGen);
    if (st.genState) hipLaunchKernelGGL(k_gen_burnin, dim3((npix + 255) / 256, nGen), dim3(256), 0, s, rp, st, npix);
    const size_t lds = (size_t)rp.spp * ln * (rp.spp <= 256 ? 1 : 2) + ((size_t)rp.spp + 1) * 4;
    if (rp.spp <= 256) {
      hipLaunchKernelGGL(k_gen_samples_lm<uint8_t>, g, dim3(ln), lds, s, rp, st, npix);
    } else {
      static bool attrSet = false;
      if (!attrSet) {
        (void)hipFuncSetAttribute((const void*)k_gen_samples_lm<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_gen_samples_pc, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attrSet = true;
      }
      if (rp.spp <= 1024) {  // two waves per 64 pixels (generator / table); above, a group's tables leave no room for the ring: one wave
        const size_t ldsPc = (size_t)rp.spp * 64 * 2 + (((size_t)rp.spp + 2) & ~(size_t)1) * 4 + (size_t)DR_GEN_RING * 64 * 8 + 16;
        hipLaunchKernelGGL(k_gen_samples_pc, dim3((npix + 63) / 64, nGen), dim3(128), ldsPc, s, rp, st, npix);
      } else {
        hipLaunchKernelGGL(k_gen_samples_lm<uint16_t>, g, dim3(ln), lds, s, rp, st, npix);
      }
    }
    return;
  }
  if (rp.blocks) hipLaunchKernelGGL(k_gen_samples_multi, grid, dim3(64), 0, s, rp, st, npix);
  if (rp.spp <= 256) {
    const size_t lds = (size_t)rp.spp * 68 + 512 + ((size_t)rp.spp + 1) * 4;
    hipLaunchKernelGGL((k_gen_samples<uint8_t, 68>), grid, dim3(64), lds, s, rp, st, npix);
  } else {
    const size_t lds = (size_t)rp.spp * 66 * sizeof(uint16_t) + 512 + ((size_t)rp.spp + 1) * 4;
    static bool attrSet = false;
    if (!attrSet) {
      (void)hipFuncSetAttribute((const void*)k_gen_samples<uint16_t, 66>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attrSet = true;
    }
    hipLaunchKernelGGL((k_gen_samples<uint16_t, 66>), grid, dim3(64), lds, s, rp, st, npix);
  }
}
void launch_mark_alive(const uint32_t* list, const uint32_t* nList, uint32_t shift, uint8_t* alive, hipStream_t s) {
  hipLaunchKernelGGL(k_mark_alive, dim3(2048), dim3(256), 0, s, list, nList, shift, alive);
}
void launch_sum_alive(const uint8_t* alive, uint32_t nGroups, uint32_t npix, const uint32_t nb[3], TraceCounters* ctr, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_alive, dim3(std::max(1u, std::min(256u, (3u * nGroups + 255u) / 256u))), dim3(256), 0, s, alive, nGroups, npix, nb[0], nb[1], nb[2], ctr);
}
void launch_transpose_samples(const float* aos, int stride, const BatchState& st, int nFloats, hipStream_t s) {
  hipLaunchKernelGGL(k_transpose_samples, dim3((st.nslots + 255) / 256), dim3(256), 0, s, aos, stride, st, nFloats);
}
void launch_raygen(const RenderParams& rp, const BatchState& st, hipStream_t s) {
  hipLaunchKernelGGL(k_raygen, dim3((st.nslots + 255) / 256), dim3(256), 0, s, rp, st);
}
// The shade kernels stage their queue entries in dynamic LDS (PushStage, dr_wave.h): ~96 KB of the CU's 160 KB; the
// light tables follow when they are small (LLDS instantiations).
template <auto kernel, int BLOCK, int NQ = 4, class... A>
static void launch_shade(int grid, size_t extraLds, hipStream_t s, A... args) {
  const size_t lds = push_stage_bytes(BLOCK, NQ) + extraLds;
  static size_t attrSet = 0;  // one instance of this template, hence one value, per kernel
  if (attrSet < lds) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attrSet = lds;
  }
  hipLaunchKernelGGL(kernel, dim3(DR_SHADE_GRID(grid)), dim3(BLOCK), lds, s, args...);
}
static bool lightsInLds(const DScene& sc) {
  return sc.nlights > 0 && light_table_bytes(sc) <= DR_LDS_LIGHT_BYTES;
}
void launch_shade_path(const DScene& sc, const RenderParams& rp, const BatchState& st, const StageQueues& q, int bounce,
                       int grid, hipStream_t s) {
  const bool gen = sc.nquads || sc.hasSpec || sc.srec;
  const size_t x = light_table_bytes(sc) + mat_table_bytes(sc);
  // (the tables only go to LDS when the 160 KB hold them next to the staging lists: five per wave in the env-map variant)
  const bool envq = !gen && sc.hasEnv;
  if (lightsInLds(sc) && push_stage_bytes(SHADE_BLOCK_OF(gen), envq ? 5 : 4) + x <= 160 * 1024) {
    if (gen) launch_shade<k_shade_path<true, true, true>, SHADE_BLOCK_OF(true)>(grid, x, s, sc, rp, st, q, bounce);
    else if (sc.hasEnv) launch_shade<k_shade_path<true, false, true>, SHADE_BLOCK_OF(false), 5>(grid, x, s, sc, rp, st, q, bounce);
    else launch_shade<k_shade_path<false, false, true>, SHADE_BLOCK_OF(false)>(grid, x, s, sc, rp, st, q, bounce);
  } else {
    if (gen) launch_shade<k_shade_path<true, true, false>, SHADE_BLOCK_OF(true)>(grid, 0, s, sc, rp, st, q, bounce);
    else if (sc.hasEnv) launch_shade<k_shade_path<true, false, false>, SHADE_BLOCK_OF(false), 5>(grid, 0, s, sc, rp, st, q, bounce);
    else launch_shade<k_shade_path<false, false, false>, SHADE_BLOCK_OF(false)>(grid, 0, s, sc, rp, st, q, bounce);
  }
}
// k_env: DR_ENV_GRID_PER_CU workgroups of 256 threads per CU, a grid-stride loop over the stage's environment-map list
#ifndef DR_ENV_GRID_PER_CU
#define DR_ENV_GRID_PER_CU 4
#endif
void launch_env(const DScene& sc, const RenderParams& rp, const BatchState& st, const StageQueues& q, int bounce, int grid, hipStream_t s) {
  if ((size_t)sc.env.h > DR_ENV_MARG_LDS_ROWS) {  // a map too tall for the LDS copy: the global-memory marginal (any height)
    hipLaunchKernelGGL(k_env<false>, dim3(grid * DR_ENV_GRID_PER_CU), dim3(DR_ENV_BLOCK), push_stage_bytes(DR_ENV_BLOCK), s, sc, rp, st, q, bounce);
    return;
  }
  const size_t lds = push_stage_bytes(DR_ENV_BLOCK) + (2 * (size_t)sc.env.h + 1) * 4;
  static size_t attrSet = 0;
  if (attrSet < lds) {
    (void)hipFuncSetAttribute((const void*)k_env<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attrSet = lds;
  }
  hipLaunchKernelGGL(k_env<true>, dim3(grid * DR_ENV_GRID_PER_CU), dim3(DR_ENV_BLOCK), lds, s, sc, rp, st, q, bounce);
}
void launch_shade_direct(const DScene& sc, const RenderParams& rp, const BatchState& st, const StageQueues& q, int stage,
                         int grid, hipStream_t s) {
  const bool gen = sc.nquads || sc.hasSpec || sc.srec;
  if (lightsInLds(sc) && push_stage_bytes(SHADE_BLOCK_OF(gen)) + light_table_bytes(sc) + mat_table_bytes(sc) <= 160 * 1024) {
    const size_t x = light_table_bytes(sc) + mat_table_bytes(sc);
    if (gen) launch_shade<k_shade_direct<true, true>, SHADE_BLOCK_OF(true)>(grid, x, s, sc, rp, st, q, stage);
    else launch_shade<k_shade_direct<false, true>, SHADE_BLOCK_OF(false)>(grid, x, s, sc, rp, st, q, stage);
  } else {
    if (gen) launch_shade<k_shade_direct<true, false>, SHADE_BLOCK_OF(true)>(grid, 0, s, sc, rp, st, q, stage);
    else launch_shade<k_shade_direct<false, false>, SHADE_BLOCK_OF(false)>(grid, 0, s, sc, rp, st, q, stage);
  }
}
void launch_shade_spec(const DScene& sc, const RenderParams& rp, const BatchState& st, const StageQueues& q, int grid, hipStream_t s) {
  launch_shade<k_shade_spec, SHADE_BLOCK_OF(true)>(grid, 0, s, sc, rp, st, q);
}
void launch_film(const RenderParams& rp, const BatchState& st, const float* filterTable, uint32_t npix, float* film,
                 hipStream_t s) {
  const uint64_t nblk = rp.spp >= 64 ? ((uint64_t)npix + 15) / 16  // 16 pixels per block, 64 samples of each per pass
                                     : ((uint64_t)npix * (uint64_t)rp.spp + DR_FILM_CHUNK - 1) / DR_FILM_CHUNK;
  if (nblk == 0) return;
  hipLaunchKernelGGL(k_film, dim3((unsigned)nblk), dim3(256), 0, s, rp, st, filterTable, npix, film);
}
void launch_film_resolve(const float* film, int64_t npix, float* rgb, hipStream_t s) {
  hipLaunchKernelGGL(k_film_resolve, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, film, npix, rgb);
}
void launch_copy(const float4* src, float4* dst, uint64_t n4, hipStream_t s) {
  hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, s, src, dst, n4);
}
int layout_state_words() { return DR_STATE_WORDS_K; }
int layout_sub() { return DR_SUB; }

#ifdef DR_NS
}  // namespace DR_NS
#endif
