// dr_trace.hip -- BVHAccel.intersect / intersectP (accelerators/bvh_accel.dart:101-226,439-472)
// and Triangle.intersect / intersectP (shapes/triangle.dart:44-240) as persistent gfx950 kernels.
//
// v2 design (k_trace / k_intersect):
//  * persistent threads, ONE RAY PER LANE, per-lane refill: a lane whose ray has finished becomes
//    idle; when >= DR_REFILL_TH lanes of the wave are idle the wave takes that many queue entries
//    with one atomicAdd on the device work counter, so a wave no longer runs as long as its
//    longest ray;
//  * the traversal ORDER of the reference is kept exactly (near child first by dirIsNeg[axis], far
//    child pushed; maxDistance shrinks on every accepted hit), so hit records, equal-t tie breaks
//    and the node/triangle visit counts equal the reference's;
//  * leaf batching: a lane that reaches a leaf WAITS (it may not run ahead: the next box test
//    depends on the shrunk maxDistance) until >= DR_LEAF_TH lanes are at leaves or nobody can
//    traverse; the f64 Moeller-Trumbore code then runs for all of them at once instead of for
//    ~7 % of the lanes every iteration;
//  * slab test: an f32 interval filter decides almost every box; only when the f32 enclosure of
//    the f64 quantities straddles a comparison (or the ray has a zero direction component, where
//    0*inf = NaN matters) is the literal f64 test of bvh_accel.dart:439-472 evaluated.  The
//    decision is therefore always the f64 one;
//  * todo stack: [depth][lane] u32 in LDS (bank == lane => conflict free) + global spill.
//  * what the node loop never reads -- direction, minDistance, the queue handle -- waits in LDS between the refill and
//    the leaf tests / the result store => 72 VGPRs, 7 waves per SIMD.
// Measured (MEASUREMENTS.md, round 2): the loop is bound by the latency of its dependent node fetches, not by VALU
// issue (9 % fewer VALU instructions: no change; 3..6 workgroups per CU: t = 96 + 839 / w ms), so occupancy is what
// pays; k_trace3 (sibling pairs, further down) wins on big incoherent trees and is chosen per scene and ray kind by
// the pilot in dr_render_device.  About 1 flop per byte: no MFMA.
#include "dr_kernels.h"
#include "dr_wave.h"
#include "dr_rng.h"

#ifdef DR_NS  // a second instantiation of this file (another state layout, -DDR_SUB=...): every symbol in its own namespace
namespace DR_NS {
#endif

#ifndef DR_REFILL_TH
#define DR_REFILL_TH 16
#endif
// k_trace's own thresholds per ray kind (closest hit, any hit); the other kernels use the two above
#ifndef DR_REFILL_TH_C
#define DR_REFILL_TH_C DR_REFILL_TH
#endif
#ifndef DR_REFILL_TH_A
#define DR_REFILL_TH_A DR_REFILL_TH
#endif
#ifndef DR_LEAF_TH_C
#define DR_LEAF_TH_C 14  // (end of round 2, queue in 512-entry runs: 14 is 0.3 % better than 12 for closest hits; 16 / 20 / 24 any-hit lanes: no gain)
#endif
#ifndef DR_LEAF_TH_A
#define DR_LEAF_TH_A 12
#endif
#ifndef DR_LEAF_TH
#define DR_LEAF_TH 12
#endif
#ifndef DR_NSHARD
#define DR_NSHARD 1  // work-queue shards: 1 = one shared counter; 8 = one per XCD.  Measured on C2: 8 shards are 12 % SLOWER (each XCD walks its own eighth of the queue, so the chip-wide working set in the shared Infinity Cache is 8 regions instead of 1)
#endif
#ifndef DR_TRACE3_WAVES
#define DR_TRACE3_WAVES 5  // k_trace3: 30 KiB of LDS per workgroup => 5 workgroups per CU anyway
#endif
#ifndef DR_WORK_CHUNK
#define DR_WORK_CHUNK 256  // queue entries a wave reserves per atomic on the work counter (256+ loses cache locality, 64 is atomic bound)
#endif
// A THIN launch (round 5): with fewer than 256 entries per resident wave the first n / 256 waves would each walk four generations of
// rays one after the other while the rest of the chip idles (C5: 262 922 closest-hit rays = 1 027 busy waves of 7 168, 0.43 ms for what is
// one ray chain deep).  The reservation then shrinks to the queue's share per wave, in whole 64-ray generations -- never below 64 (one
// atomic per wave generation is what the work counter sustains) -- and every wave that finds work walks one generation.
#ifndef DR_WORK_CHUNK_DYNAMIC
#define DR_WORK_CHUNK_DYNAMIC 1
#endif
__device__ __forceinline__ uint32_t work_chunk(uint32_t n) {
#if DR_WORK_CHUNK_DYNAMIC
  const uint32_t waves = gridDim.x * (DR_TRACE_BLOCK / 64);
  const uint32_t share = ((n + waves - 1u) / waves + 63u) & ~63u;
  return share < (uint32_t)DR_WORK_CHUNK ? (share < 64u ? 64u : share) : (uint32_t)DR_WORK_CHUNK;
#else
  return (uint32_t)DR_WORK_CHUNK;
#endif
}

DR_DEV void flush_counters(TraceCounters* ctr, int any, uint32_t rays, uint32_t nodes, uint32_t tris) {
  unsigned long long r = wave_sum(rays), n = wave_sum(nodes), t = wave_sum(tris);
  // (a wave that traced nothing reports nothing: the ~7000 waves of a persistent launch otherwise queue 21 000 atomics
  // on three addresses -- ~0.2 ms, the floor of the small launches at the end of a path's stage loop)
  if (lane_id() == 0 && ctr && (r | n | t) != 0ull) {
    if (any) {
      atomicAdd(&ctr->any_rays, r);
      atomicAdd(&ctr->any_nodes, n);
      atomicAdd(&ctr->any_tris, t);
    } else {
      atomicAdd(&ctr->closest_rays, r);
      atomicAdd(&ctr->closest_nodes, n);
      atomicAdd(&ctr->closest_tris, t);
    }
  }
}



// ===========================================================================
// v2
// ===========================================================================
struct TraceRay {
  F3 o, d;
  float ivx, ivy, ivz;         // invDir: a Vector, i.e. rounded to f32 (bvh_accel.dart:109-111)
  double tmin, tmax;           // Ray.minDistance / maxDistance (f64)
  float tminLo, tminHi;        // f32 brackets: tminLo <= tmin <= tminHi
  float tmaxLo, tmaxHi;
  bool needF64;                // a zero direction component: 0*inf = NaN can occur, always take the literal test
};
DR_DEV float f32_below(double v) {  // largest float <= v
  float f = (float)v;
  if ((double)f > v) f = __uint_as_float(f > 0.f ? __float_as_uint(f) - 1u : (f < 0.f ? __float_as_uint(f) + 1u : 0x80000001u));
  return f;
}
DR_DEV float f32_above(double v) {  // smallest float >= v
  float f = (float)v;
  if ((double)f < v) f = __uint_as_float(f > 0.f ? __float_as_uint(f) + 1u : (f < 0.f ? __float_as_uint(f) - 1u : 0x00000001u));
  return f;
}
DR_DEV void ray_set_tmax(TraceRay& r, double tmax) {
  r.tmax = tmax;
  r.tmaxLo = f32_below(tmax);
  r.tmaxHi = f32_above(tmax);
}
DR_DEV void ray_init(TraceRay& r, F3 o, F3 d, double tmin, double tmax) {
  r.o = o;
  r.d = d;
  r.ivx = (float)(1.0 / (double)d.x);
  r.ivy = (float)(1.0 / (double)d.y);
  r.ivz = (float)(1.0 / (double)d.z);
  r.tmin = tmin;
  r.tminLo = f32_below(tmin);
  r.tminHi = f32_above(tmin);
  ray_set_tmax(r, tmax);
  r.needF64 = (d.x == 0.f) || (d.y == 0.f) || (d.z == 0.f);
}

// The literal slab test (bvh_accel.dart:439-472).
DR_DEV bool slab_f64(const TraceRay& r, float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz) {
  const bool n0 = r.ivx < 0.f, n1 = r.ivy < 0.f, n2 = r.ivz < 0.f;
  const double ox = r.o.x, oy = r.o.y, oz = r.o.z;
  double t0 = ((double)(n0 ? bmaxx : bminx) - ox) * (double)r.ivx;
  double t1 = ((double)(n0 ? bminx : bmaxx) - ox) * (double)r.ivx;
  const double ty0 = ((double)(n1 ? bmaxy : bminy) - oy) * (double)r.ivy;
  const double ty1 = ((double)(n1 ? bminy : bmaxy) - oy) * (double)r.ivy;
  if ((t0 > ty1) || (ty0 > t1)) return false;
  if (ty0 > t0) t0 = ty0;
  if (ty1 < t1) t1 = ty1;
  const double tz0 = ((double)(n2 ? bmaxz : bminz) - oz) * (double)r.ivz;
  const double tz1 = ((double)(n2 ? bminz : bmaxz) - oz) * (double)r.ivz;
  if ((t0 > tz1) || (tz0 > t1)) return false;
  if (tz0 > t0) t0 = tz0;
  if (tz1 < t1) t1 = tz1;
  return (t0 < r.tmax) && (t1 > r.tmin);
}

// f32 enclosure of the slab test.  With E = max of the three entry parameters and X = min of the
// three exit parameters, the reference's test is (E <= X) && (E < maxDistance) && (X > minDistance)
// whenever no NaN occurs (same-axis entry <= exit holds by monotonic rounding).  Each f32 product
// (b - o) * inv is within 2^-23 relative of the f64 one; +-(|x| * 2^-21 + 1e-37) encloses it with
// room for the rounding of the enclosure arithmetic itself and for f32 denormals.  Overflow to
// +-inf makes an enclosure bound NaN or +-inf, every comparison below is then false and the box
// is reported ambiguous.  Returns 1 = certain hit, 0 = certain miss, -1 = evaluate the f64 test.
DR_DEV void slab_f32_sure(const TraceRay& r, float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, bool* sureHit, bool* sureMiss) {
  const float ax = (bminx - r.o.x) * r.ivx, bx = (bmaxx - r.o.x) * r.ivx;
  const float ay = (bminy - r.o.y) * r.ivy, by = (bmaxy - r.o.y) * r.ivy;
  const float az = (bminz - r.o.z) * r.ivz, bz = (bmaxz - r.o.z) * r.ivz;
  const float lo = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
  const float hi = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  const float R = 4.76837158203125e-07f;  // 2^-21
  const float A = 1.0e-37f;
  // |x| as a source modifier of the fma (left to itself the compiler packs the two fmas into one v_pk_fma_f32, which
  // takes no modifiers, and pays two v_and and a v_mov for it)
  float eLo, eHi;
  asm("v_fma_f32 %0, |%1|, %2, %3" : "=v"(eLo) : "v"(lo), "v"(R), "v"(A));
  asm("v_fma_f32 %0, |%1|, %2, %3" : "=v"(eHi) : "v"(hi), "v"(R), "v"(A));
  const float loU = lo + eLo, loL = lo - eLo, hiU = hi + eHi, hiL = hi - eHi;
  // One bracket end per bound is enough.  tminLo / tminHi (and tmaxLo / tmaxHi) are equal or adjacent floats, so for the
  // float hiL:  hiL > tminLo  =>  hiL >= tminHi >= minDistance, and the true exit parameter lies strictly above the real
  // hi - eHi (eHi exceeds the rounding error of the products and of the enclosure arithmetic): exit > minDistance.
  // Likewise loU < tmaxHi  =>  loU <= tmaxLo <= maxDistance with the true entry strictly below loU.
  *sureHit = (loU <= hiL) && (loU < r.tmaxHi) && (hiL > r.tminLo);
  *sureMiss = (loL > hiU) || (loL >= r.tmaxHi) || (hiU <= r.tminLo);
}
DR_DEV int slab_f32(const TraceRay& r, float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz) {
  bool sureHit, sureMiss;
  slab_f32_sure(r, bminx, bminy, bminz, bmaxx, bmaxy, bmaxz, &sureHit, &sureMiss);
  return sureHit ? 1 : (sureMiss ? 0 : -1);
}

// The global part of the todo stack, [entry][launch thread]: a 32-bit byte offset from the wave-uniform base (the buffer
// is < 4 GiB: ensureSpill).  The offset is rebuilt at every use and hidden from the optimiser -- left alone it keeps a
// 64-bit per-lane pointer alive across the traversal loop.
DR_DEV uint32_t* spill_at(uint32_t* spillBase, uint32_t spillStride, int entry) {
  uint32_t o = blockIdx.x * DR_TRACE_BLOCK + threadIdx.x;
  asm volatile("" : "+v"(o));
  return (uint32_t*)((char*)spillBase + (size_t)(uint32_t)((o + (uint32_t)entry * spillStride) << 2));
}
// Pop: the LDS read is issued unconditionally (ds_read, not a flat load through a selected pointer);
// the global spill is only touched by lanes deeper than the STACK entries kept in LDS.
template <int STACK>
DR_DEV uint32_t stack_pop(const uint32_t* lds, uint32_t* spillBase, uint32_t spillStride, int sp) {
  // (an explicit LDS-address-space load: left generic, the compiler merges the two loads into ONE flat load
  // through a selected pointer)
  typedef __attribute__((address_space(3))) const uint32_t lds_u32;
  uint32_t v = ((lds_u32*)lds)[(sp < STACK ? sp : STACK - 1) * DR_TRACE_BLOCK];
  if (sp >= STACK) {
    v = *spill_at(spillBase, spillStride, sp - STACK);
    // consume the value HERE: the wait for this (rare) load then sits in this branch.  Left pending, it makes the
    // compiler put `s_waitcnt vmcnt(0)` in front of every node fetch -- vmcnt retires in order, so that wait also
    // covers the result stores and spill pushes issued before it, on every iteration of every lane.
    asm volatile("" : "+v"(v));
  }
  return v;
}

#define M_IDLE 0
#define M_TRAV 1
#define M_LEAF 2
#define M_DONE 3  // finished, result not stored yet

// IO policy of the path-state kernel: queue entries -> rays, results -> slot arrays.
template <int ANY>
struct StateIO {
  BatchState st;
  const uint32_t* queue;
  DR_DEV void load(uint32_t idx, TraceRay& r, uint32_t& handle) const {
    const uint32_t e = queue ? queue[idx] : idx;
    const uint32_t slot = e & ~Q_MIS_BIT;
    const uint32_t cap = st.tileStride;  // words per 64-slot tile
    handle = e;
    const size_t ti = TI(cap, slot), td = TD(cap, slot);
    const F3 o = F3{LDS_STREAM(st.ro() + ti), LDS_STREAM(st.ro() + ti + DR_SUB), LDS_STREAM(st.ro() + ti + 2 * DR_SUB)};
    const float* dir = (ANY ? st.shD() : ((e & Q_MIS_BIT) ? st.misD() : st.rd())) + ti;
    const F3 d = F3{LDS_STREAM(dir), LDS_STREAM(dir + DR_SUB), LDS_STREAM(dir + 2 * DR_SUB)};
    ray_init(r, o, d, LDS_STREAM(st.rtmin() + td), ANY ? LDS_STREAM(st.shTmax() + td) : DR_INF);
  }
  DR_DEV void store(uint32_t handle, const TraceRay& r, int prim, const DScene&) const {
    const uint32_t slot = handle & ~Q_MIS_BIT;
    const size_t ti = TI(st.tileStride, slot);
    if (ANY) {
      STS_STREAM(st.shOcc() + ti, (prim >= 0) ? 1 : 0);
    } else if (handle & Q_MIS_BIT) {
      STS_STREAM(st.misPrim() + ti, prim);
    } else {
      STS_STREAM(st.hprim() + ti, prim);
      STS_STREAM(st.ht() + TD(st.tileStride, slot), r.tmax);
    }
  }
};
// IO policy of dr_intersect: caller-supplied DrRay -> DrHit.
template <int ANY>
struct RayIO {
  const DrRay* rays;
  DrHit* out;
  DR_DEV void load(uint32_t idx, TraceRay& r, uint32_t& handle) const {
    const DrRay q = rays[idx];
    handle = idx;
    ray_init(r, F3{q.o[0], q.o[1], q.o[2]}, F3{q.d[0], q.d[1], q.d[2]}, q.tmin, q.tmax);
  }
  DR_DEV void store(uint32_t handle, const TraceRay& r, int prim, const DScene& sc) const {
    DrHit h;
    h.prim = prim;
    h.pad = 0;
    h.t = h.b1 = h.b2 = 0.0;
    if (!ANY && prim >= 0) {
      Tri tr = load_tri(sc, (uint32_t)prim);
      double tt, b1 = 0.0, b2 = 0.0;
      if (tr.kind == 0) tri_hit(tr.p1, tr.p2, tr.p3, r.o, r.d, r.tmin, DR_INF, &tt, &b1, &b2);  // same arithmetic as the accepting test
      h.t = r.tmax;
      h.b1 = b1;
      h.b2 = b2;
    }
    out[handle] = h;
  }
};

// -DDR_TRACE_PROF: a diagnostic build that stamps s_memtime between the phases of the v2 loop and sums the cycles the
// waves spent in each (printed and cleared by dr_get_stats through trace_prof_dump).  No stamp executes in the product build.
#ifdef DR_TRACE_PROF
__device__ unsigned long long g_traceProf[2][16];  // [0..3] phases, [4..11] counts, [12..14] parts of the node-visit phase (round 5)
DR_DEV unsigned long long tprof_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define TPROF_DECL unsigned long long tpT = tprof_now(), tpAcc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
// (round 5) inside the node-visit phase: wait until the loads issued so far have arrived, then stamp
#define TPROF_ARRIVED(i)                                  \
  do {                                                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      \
    TPROF(i);                                             \
  } while (0)
#define TPROF(i)                              \
  do {                                        \
    const unsigned long long n_ = tprof_now(); \
    tpAcc[i] += n_ - tpT;                     \
    tpT = n_;                                 \
  } while (0)
#define TPROF_COUNT(i, v) tpAcc[i] += (v)
#define TPROF_FLUSH                                                                  \
  do {                                                                               \
    if (lane_id() == 0)                                                              \
      for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_traceProf[ANY][i_], tpAcc[i_]);   \
  } while (0)
void trace_prof_dump() {
  unsigned long long h[2][16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_traceProf), sizeof(h)) != hipSuccess) return;
  static const char* names[12] = {"refill: queue entry + ray state + ray_init", "node visits", "leaf tests", "result stores + bookkeeping",
                                  "(wave iterations)", "(refill events)", "(leaf phases)", "(leaf triangle rounds)",
                                  "(lanes visiting a node, summed over iterations)", "(lanes waiting at a leaf, summed over iterations)",
                                  "(idle lanes, summed over iterations)", "(lanes in the leaf phases that ran)"};
  for (int a = 0; a < 2; ++a) {
    double tot = 0;
    for (int i = 0; i < 4; ++i) tot += (double)h[a][i];
    if (tot == 0) continue;
    for (int i = 0; i < 4; ++i) fprintf(stderr, "trace_prof %s %-44s %6.2f %%  (%.3g wave-cycles)\n", a ? "any    " : "closest", names[i], 100.0 * h[a][i] / tot, (double)h[a][i]);
    for (int i = 4; i < 12; ++i) fprintf(stderr, "trace_prof %s %-44s %.4g\n", a ? "any    " : "closest", names[i], (double)h[a][i]);
    // parts of the node-visit phase, in wave-cycles per wave iteration (they were taken out of names[1]'s sum above)
    static const char* sub[3] = {"node visit: ballots + fetch issue + result stores -> node data arrived", "node visit: slab filter -> decision",
                                 "node visit: push / pop (LDS, rare global spill)"};
    for (int i = 0; i < 3; ++i)
      fprintf(stderr, "trace_prof %s %-72s %8.1f cycles per wave iteration (%5.2f %% of all phases)\n", a ? "any    " : "closest", sub[i],
              (double)h[a][12 + i] / (double)(h[a][4] ? h[a][4] : 1), 100.0 * (double)h[a][12 + i] / (tot + (double)h[a][12] + (double)h[a][13] + (double)h[a][14]));
    fprintf(stderr, "trace_prof %s %-72s %8.1f cycles\n", a ? "any    " : "closest", "a whole wave iteration",
            (tot + (double)h[a][12] + (double)h[a][13] + (double)h[a][14]) / (double)(h[a][4] ? h[a][4] : 1));
  }
  for (int a = 0; a < 2; ++a) for (int i = 0; i < 16; ++i) h[a][i] = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_traceProf), h, sizeof(h));
}
#else
#define TPROF_DECL
#define TPROF(i)
#define TPROF_ARRIVED(i)
#define TPROF_COUNT(i, v)
#define TPROF_FLUSH
#ifndef DR_STACK_PROF
void trace_prof_dump() {}
#endif
#endif
// -DDR_STACK_PROF: a diagnostic build of the sibling-pair kernels that histograms, per ray, the deepest its todo stack got
// (printed and cleared by dr_get_stats): how many LDS rows does a scene's traversal need before entries go to the global rows?
#ifdef DR_STACK_PROF
__device__ unsigned long long g_stackHist[2][40];
#define SPROF_DECL int spMax = 0
#define SPROF_PUSH() (spMax = sp > spMax ? sp : spMax)
#define SPROF_RAY(any)                                                   \
  do {                                                                   \
    atomicAdd(&g_stackHist[any][spMax < 39 ? spMax : 39], 1ull);         \
    spMax = 0;                                                           \
  } while (0)
void trace_prof_dump() {
  unsigned long long h[2][40];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stackHist), sizeof(h)) != hipSuccess) return;
  for (int a = 0; a < 2; ++a) {
    double tot = 0, acc = 0;
    for (int i = 0; i < 40; ++i) tot += (double)h[a][i];
    if (tot == 0) continue;
    fprintf(stderr, "stack_prof %s: rays %.4g; share of rays whose stack never exceeded d entries:", a ? "any-hit" : "closest", tot);
    for (int i = 0; i < 40; ++i) {
      acc += (double)h[a][i];
      if (i >= 4 && i <= 32) fprintf(stderr, " %d:%.4f", i, acc / tot);
    }
    fprintf(stderr, "\n");
  }
  memset(h, 0, sizeof(h));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stackHist), h, sizeof(h));
}
#else
#define SPROF_DECL
#define SPROF_PUSH()
#define SPROF_RAY(any)
#endif

template <int ANY, bool QUAD, int STACK, class IO>
DR_DEV void trace_persistent(const DScene& sc, const IO& io, uint32_t n, uint32_t* lds, uint32_t* spill,
                             uint32_t spillStride, uint32_t* work, TraceCounters* ctr, uint32_t* cold) {
  // Ray state the node loop never reads -- direction, minDistance, the queue handle -- lives in LDS (6 dwords per
  // lane) between the refill and the leaf tests / the result store: 6 VGPRs fewer in the loop that sets the occupancy.
  // (Keeping it in a global scratch row instead -- to make room for an eighth workgroup -- costs more than the extra
  // waves return: every phase slows by 6..8 %, MEASUREMENTS.md.)
  typedef __attribute__((address_space(3))) uint32_t cold_u32;
#define COLD_ST(i, v) (((cold_u32*)cold)[(i) * DR_TRACE_BLOCK] = (v))
#define COLD_LD(i) (((const cold_u32*)cold)[(i) * DR_TRACE_BLOCK])
#define COLD_D() F3{__uint_as_float(COLD_LD(0)), __uint_as_float(COLD_LD(1)), __uint_as_float(COLD_LD(2))}
#define COLD_TMIN() __hiloint2double((int)COLD_LD(4), (int)COLD_LD(3))
  const int lane = lane_id();
  uint32_t nRays = 0, nNodes = 0, nTris = 0;
  TraceRay ray;
  ray.needF64 = false;
  uint32_t handle = 0, node = 0, leafOff = 0, leafN = 0;
  int sp = 0, hit = -1, mode = M_IDLE;
  bool exhausted = false;  // wave-uniform: the queue has no more entries
  uint32_t resNext = 0, resEnd = 0;  // wave-uniform: this wave's reservation [resNext, resEnd) of queue indices
  // XCD-aware work distribution: the queue is cut into 8 contiguous shards, one per XCD (each XCD has its
  // own 4 MiB L2); a wave drains its own XCD's shard first and then steals from the others.  Placement is
  // read from HW_REG_XCC_ID and only affects speed, never results.
  uint32_t shard = ((uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) % (uint32_t)DR_NSHARD;  // XCC_ID[3:0]
  uint32_t tried = 0;
  const uint32_t chunk = work_chunk(n);  // wave-uniform
  TPROF_DECL;
  for (;;) {
    TPROF(3);
    TPROF_COUNT(4, 1);
    // ---- refill idle lanes ----
    const unsigned long long idleMask = __ballot(mode == M_IDLE);
    const int nIdle = __popcll(idleMask);
    if (!exhausted && (nIdle >= (ANY ? DR_REFILL_TH_A : DR_REFILL_TH_C) || nIdle == 64)) {
      if (resNext == resEnd) {
        // Reserve DR_WORK_CHUNK entries per atomic: same-address atomics are a chip-wide serial resource.
        for (;;) {
          if (tried == (uint32_t)DR_NSHARD) {
            exhausted = true;
            break;
          }
          const uint32_t s0 = (uint32_t)(((unsigned long long)n * shard) / DR_NSHARD);
          const uint32_t s1 = (uint32_t)(((unsigned long long)n * (shard + 1u)) / DR_NSHARD);
          uint32_t fresh = 0;
          // (the counter only grows: once a plain load sees it past the shard's end nothing is left to reserve -- a wave
          // that arrives late does not join the queue of same-address atomics)
          if (lane == 0) {
            uint32_t* const wc = work + shard * (uint32_t)DR_WORK_STRIDE;
            // (only before a wave's first reservation: later ones go straight to the atomic)
            fresh = nRays == 0u ? __hip_atomic_load(wc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (fresh < s1 - s0) fresh = atomicAdd(wc, chunk);
          }
          fresh = wave_bcast_first(fresh);
          if (fresh < s1 - s0) {
            resNext = s0 + fresh;
            resEnd = min(s0 + fresh + chunk, s1);
            break;
          }
          shard = (shard + 1u) % (uint32_t)DR_NSHARD;  // this shard is drained (its counter only grows)
          ++tried;
        }
      }
      const uint32_t take = min(resEnd - resNext, (uint32_t)nIdle);
      TPROF_COUNT(5, 1);
      if (mode == M_IDLE) {
        const uint32_t j = __builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0u));  // rank among the idle lanes
        if (j < take) {
          const uint32_t idx = resNext + j;
          io.load(idx, ray, handle);
          COLD_ST(0, __float_as_uint(ray.d.x));
          COLD_ST(1, __float_as_uint(ray.d.y));
          COLD_ST(2, __float_as_uint(ray.d.z));
          COLD_ST(3, (uint32_t)__double2loint(ray.tmin));
          COLD_ST(4, (uint32_t)__double2hiint(ray.tmin));
          COLD_ST(5, handle);
          hit = -1;
          if (sc.nnodes == 0) {
            mode = M_DONE;  // an empty scene: a miss, stored like every other result
          } else {
            mode = M_TRAV;
            node = 0;
            sp = 0;
          }
        }
      }
      nRays += take;
      resNext += take;
#ifdef DR_TRACE_PROF
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TPROF(0);
#endif
    }
    const unsigned long long travMask = __ballot(mode == M_TRAV);
    unsigned long long leafMask = __ballot(mode == M_LEAF);
    const unsigned long long doneMask = __ballot(mode == M_DONE);
    if ((travMask | leafMask | doneMask) == 0ull) {
      if (exhausted) break;
      continue;
    }
    bool finished = false;
    TPROF_COUNT(8, __popcll(travMask));
    TPROF_COUNT(9, __popcll(leafMask));
    TPROF_COUNT(10, 64 - __popcll(travMask | leafMask | doneMask));
    // ---- one node visit (bvh_accel.dart:122-160) ----
    nNodes += (uint32_t)__popcll(travMask);  // wave-uniform: lane 0 carries the wave's count (flush_counters sums lanes)
    uint4 a = uint4{0, 0, 0, 0}, b = uint4{0, 0, 0, 0};
    if (mode == M_TRAV) {
      // 32-bit byte offset from a scalar base (dr_scene_create refuses trees beyond 2^27 nodes): one shift instead of
      // a 64-bit shift + add per visit
      const uint4* np = (const uint4*)((const char*)sc.nodes + (size_t)(uint32_t)(node << 5));
      a = np[0];
      b = np[1];
    }
    // results of the rays that finished in the previous iteration: stored HERE, behind the node fetches.  vmcnt retires
    // in order and stores count in it: issued before the fetches (at the end of the previous iteration) their
    // acknowledgement is waited for before the fetches are even issued; issued behind them it overlaps the fetch.
#define STORE_DONE_RAYS()               \
  if (mode == M_DONE) {                 \
    TraceRay rr = ray;                  \
    rr.d = COLD_D();                    \
    rr.tmin = COLD_TMIN();              \
    io.store(COLD_LD(5), rr, hit, sc);  \
    mode = M_IDLE;                      \
  }
#ifndef DR_TRACE_PROF
    STORE_DONE_RAYS()
#else
    // (the profiling build waits for the node data FIRST and stamps, then issues the stores: stamped behind them, the wait would
    // include the stores' acknowledgements -- vmcnt counts loads and stores together, in order; issued here their acknowledgements
    // still overlap the next iteration's fetches)
    TPROF_ARRIVED(12);
    STORE_DONE_RAYS()
#endif
#undef STORE_DONE_RAYS
    bool ok = false;
#ifdef DR_TRACE_PROF
    const bool trav_ = mode == M_TRAV;  // (the profiling build closes and re-opens the block below around a wave-uniform stamp)
#define TPROF_SPLIT(i) } TPROF(i); if (trav_) {
#else
#define TPROF_SPLIT(i)
#endif
    if (mode == M_TRAV) {
      const float bminx = __uint_as_float(a.x), bminy = __uint_as_float(a.y), bminz = __uint_as_float(a.z);
      const float bmaxx = __uint_as_float(a.w), bmaxy = __uint_as_float(b.x), bmaxz = __uint_as_float(b.y);
      bool amb = true;
      if (!ray.needF64) {  // (two predicates instead of a three-valued int: they stay lane masks in SGPRs)
        bool sureMiss;
        slab_f32_sure(ray, bminx, bminy, bminz, bmaxx, bmaxy, bmaxz, &ok, &sureMiss);
        amb = !ok && !sureMiss;
      }
      if (amb) {
        TraceRay rr = ray;
        rr.tmin = COLD_TMIN();
        ok = slab_f64(rr, bminx, bminy, bminz, bmaxx, bmaxy, bmaxz);
      }
      TPROF_SPLIT(13)
      bool pop = true;
      if (ok) {
        const uint32_t nprims = b.w & 0xffffu;
        if (nprims > 0) {
          mode = M_LEAF;
          leafOff = b.z;
          leafN = nprims;
          pop = false;
        } else {
          const uint32_t axis = (b.w >> 16) & 0xffu;
          // dirIsNeg[axis]; any-hit rays of a far-child-first launch take the other child first (DScene.anyFarFirst: the boolean does not depend on it)
          const bool neg = ((axis == 0 ? ray.ivx : (axis == 1 ? ray.ivy : ray.ivz)) < 0.f) != (ANY && sc.anyFarFirst != 0u);
          const uint32_t far = neg ? node + 1 : b.z;  // bvh_accel.dart:147-153
          node = neg ? b.z : node + 1;
          if (sp < STACK) lds[sp * DR_TRACE_BLOCK] = far;
          else if (sp < DR_MAX_STACK) *spill_at(spill, spillStride, sp - STACK) = far;
          ++sp;
          pop = false;
        }
      }
      if (pop) {
        if (sp == 0) {
          finished = true;
        } else {
          --sp;
          node = stack_pop<STACK>(lds, spill, spillStride, sp);
        }
      }
    }
#ifdef DR_TRACE_PROF
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the popped entry has arrived from LDS)
#endif
    TPROF(14);
    // ---- batched leaf tests (bvh_accel.dart:126-143 / :189-204) ----
    leafMask = __ballot(mode == M_LEAF);
    const unsigned long long stillTrav = __ballot(mode == M_TRAV && !finished);
    if (leafMask != 0ull && (__popcll(leafMask) >= (ANY ? DR_LEAF_TH_A : DR_LEAF_TH_C) || stillTrav == 0ull)) {
      TPROF_COUNT(6, 1);
      TPROF_COUNT(11, __popcll(leafMask));
      if (mode == M_LEAF) {
        bool occluded = false;
        const F3 rayD = COLD_D();
        const double rayTmin = COLD_TMIN();
        for (uint32_t i = 0; i < leafN; ++i) {
          ++nTris;
          const float4* tp = sc.tris + 3 * (size_t)(leafOff + i);
          const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
          if (QUAD && PRIM_KIND(__float_as_uint(q2.w))) {  // GeometricPrimitive over a Sphere / Disk
            double t;
            F3 phit;
            if (quadric_hit(sc.quads[__float_as_uint(q0.x)], ray.o, rayD, rayTmin, ray.tmax, &t, &phit)) {
              if (ANY) {
                occluded = true;
                break;
              }
              ray_set_tmax(ray, t);
              hit = (int)(leafOff + i);
            }
            continue;
          }
          const F3 p1 = F3{q0.x, q0.y, q0.z}, p2 = F3{q0.w, q1.x, q1.y}, p3 = F3{q1.z, q1.w, q2.x};
          if (ANY) {
            if (tri_hitP(p1, p2, p3, ray.o, rayD, rayTmin, ray.tmax)) {  // return true (bvh_accel.dart:193-195)
              occluded = true;
              break;
            }
          } else {
            double t, b1, b2;
            if (tri_hit(p1, p2, p3, ray.o, rayD, rayTmin, ray.tmax, &t, &b1, &b2)) {
              ray_set_tmax(ray, t);  // r.maxDistance = thit (geometric_primitive.dart:59)
              hit = (int)(leafOff + i);
            }
          }
        }
        if (occluded) {
          hit = 0;
          finished = true;
        } else if (sp == 0) {
          finished = true;
        } else {
          --sp;
          node = stack_pop<STACK>(lds, spill, spillStride, sp);
          mode = M_TRAV;
        }
      }
      TPROF(2);
    }
    if (finished) mode = M_DONE;  // stored at the head of the next iteration, behind that iteration's node fetches
  }
  TPROF_FLUSH;
  flush_counters(ctr, ANY, lane == 0 ? nRays : 0u, lane == 0 ? nNodes : 0u, nTris);
}

// Occupancy of k_trace (the loop is latency bound: every resident wave counts): 72 VGPRs => 7 waves per SIMD, 16 + 6
// rows of LDS per lane = 22 KiB per workgroup => 7 workgroups per CU.  An eighth (any-hit rays fit 64 VGPRs) needs
// <= 19 KiB, i.e. 13 stack rows: measured 9 % SLOWER -- the entries pushed past the LDS rows cost more than the waves
// return (14 rows: +2 %, 12 rows: +19 %).
template <int ANY>
__global__ void __launch_bounds__(DR_TRACE_BLOCK, DR_V2_WG_PER_CU) k_trace(DScene sc, BatchState st, const uint32_t* queue,
                                                          const uint32_t* nQueue, uint32_t* spill, uint32_t* work,
                                                          TraceCounters* ctr) {
  __shared__ uint32_t s_stack[DR_V2_LDS_STACK * DR_TRACE_BLOCK];
  __shared__ uint32_t s_cold[6 * DR_TRACE_BLOCK];  // direction, minDistance, queue handle per lane
  StateIO<ANY> io{st, queue};
  const uint32_t n = nQueue ? *nQueue : st.nslots;
  trace_persistent<ANY, false, DR_V2_LDS_STACK>(sc, io, n, s_stack + threadIdx.x, spill, gridDim.x * DR_TRACE_BLOCK, work, ctr,
                                              s_cold + threadIdx.x);
}
// scenes with sphere / disk primitives: the quadric tests cost registers, so they get their own instantiation
template <int ANY>
__global__ void __launch_bounds__(DR_TRACE_BLOCK) k_trace_quad(DScene sc, BatchState st, const uint32_t* queue,
                                                               const uint32_t* nQueue, uint32_t* spill, uint32_t* work,
                                                               TraceCounters* ctr) {
  __shared__ uint32_t s_stack[DR_V2_LDS_STACK * DR_TRACE_BLOCK];
  __shared__ uint32_t s_cold[6 * DR_TRACE_BLOCK];
  StateIO<ANY> io{st, queue};
  const uint32_t n = nQueue ? *nQueue : st.nslots;
  trace_persistent<ANY, true, DR_V2_LDS_STACK>(sc, io, n, s_stack + threadIdx.x, spill, gridDim.x * DR_TRACE_BLOCK, work, ctr,
                                               s_cold + threadIdx.x);
}
template <int ANY>
__global__ void __launch_bounds__(DR_TRACE_BLOCK) k_intersect(DScene sc, const DrRay* rays, uint32_t n, DrHit* out,
                                                              uint32_t* spill, uint32_t* work, TraceCounters* ctr) {
  __shared__ uint32_t s_stack[DR_V2_LDS_STACK * DR_TRACE_BLOCK];
  __shared__ uint32_t s_cold[6 * DR_TRACE_BLOCK];
  RayIO<ANY> io{rays, out};
  trace_persistent<ANY, true, DR_V2_LDS_STACK>(sc, io, n, s_stack + threadIdx.x, spill, gridDim.x * DR_TRACE_BLOCK, work, ctr,
                                               s_cold + threadIdx.x);
}

// ===========================================================================
// v3: sibling-pair traversal (an experiment kept for A/B: DARTRAY_TRACE_IMPL=3; bit-exact like v2).
// Measured on C2: closest 329 ms vs 319 ms (v2), any-hit 206 vs 176 ms -- 30 % fewer memory-wait cycles, but
// 94 VGPRs / 8-byte stack entries leave 20 instead of 24 waves per CU and the step has ~2x the instructions;
// v2's near-child fetches were already cheap (depth-first layout => same or adjacent cache line).
//
// Same visits, same order, same decisions as bvh_accel.dart:101-226 -- but the box of a child is read
// from its PARENT's 64-byte pair record, so one fetch serves two box tests:
//   * expanding an interior node (its own box is already known to be hit) fetches pairs[k] = {left, right};
//   * the near child (by dirIsNeg[axis], :147-153) is tested at once against the current maxDistance --
//     exactly when the reference would visit it (no triangle test can happen in between);
//   * the far child is what the reference pushes.  Its box test happens at POP time in the reference, with
//     the maxDistance of that moment; only `entry < maxDistance` depends on it.  So the maxDistance-free part
//     (entry <= exit, exit > minDistance) is decided now (f32 filter, f64 fallback), a child failing it is
//     pushed as a DEAD entry (it still counts as a visit when popped, like the reference), and for a live
//     child the f32 estimate of its entry parameter E is kept next to its reference on the stack;
//   * at pop time E's enclosure is compared with the current maxDistance: surely beyond => pruned with NO
//     memory access (this is where closest-hit rays spend most pops after their first hit), surely before
//     => expanded; in the ambiguous band (|E - maxDistance| within ~5e-7 relative) the literal f64 test is
//     evaluated on the child's own box, reconstructed exactly from data that is fetched anyway: an interior
//     node's bounds are the union of its children's (initInterior, bvh_accel.dart:518-524), a leaf's the
//     union of its triangles' vertices (:238-241).
// Rays with a zero direction component (0*inf = NaN possible) take the literal test for every box.
// Packed child reference: 0xffffffff dead | leaf: 1<<31 | nprims<<26 | firstPrim | interior: axis<<29 | pair.
// ===========================================================================
#ifndef DR_PSTACK
#define DR_PSTACK 15  // (reference, E) entries per lane kept in LDS: 8 B x 15 x 256 = 30 KiB per workgroup => 5 workgroups per CU (at 16 entries = 32 KiB only FOUR are resident: 5 x 32 KiB is the whole LDS and does not fit; measured on C4: +10 %)
#endif

struct SlabB {
  float lo, hi, loU, loL, hiU, hiL;
};
DR_DEV SlabB slab_bounds(const TraceRay& r, float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz) {
  const float ax = (bminx - r.o.x) * r.ivx, bx = (bmaxx - r.o.x) * r.ivx;
  const float ay = (bminy - r.o.y) * r.ivy, by = (bmaxy - r.o.y) * r.ivy;
  const float az = (bminz - r.o.z) * r.ivz, bz = (bmaxz - r.o.z) * r.ivz;
  SlabB s;
  s.lo = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
  s.hi = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  const float R = 4.76837158203125e-07f, A = 1.0e-37f;  // 2^-21, see slab_f32
  const float eLo = __fmaf_rn(fabsf(s.lo), R, A), eHi = __fmaf_rn(fabsf(s.hi), R, A);
  s.loU = s.lo + eLo;
  s.loL = s.lo - eLo;
  s.hiU = s.hi + eHi;
  s.hiL = s.hi - eHi;
  return s;
}
// The maxDistance-free part of the slab test in f64 (no NaN can occur: no zero direction component):
// returns whether entry <= exit on all axis pairs and exit > minDistance; *E = the entry parameter.
DR_DEV bool slab_geom_f64(const TraceRay& r, float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz,
                          double* E) {
  const bool n0 = r.ivx < 0.f, n1 = r.ivy < 0.f, n2 = r.ivz < 0.f;
  const double ox = r.o.x, oy = r.o.y, oz = r.o.z;
  double t0 = ((double)(n0 ? bmaxx : bminx) - ox) * (double)r.ivx;
  double t1 = ((double)(n0 ? bminx : bmaxx) - ox) * (double)r.ivx;
  const double ty0 = ((double)(n1 ? bmaxy : bminy) - oy) * (double)r.ivy;
  const double ty1 = ((double)(n1 ? bminy : bmaxy) - oy) * (double)r.ivy;
  if ((t0 > ty1) || (ty0 > t1)) return false;
  if (ty0 > t0) t0 = ty0;
  if (ty1 < t1) t1 = ty1;
  const double tz0 = ((double)(n2 ? bmaxz : bminz) - oz) * (double)r.ivz;
  const double tz1 = ((double)(n2 ? bminz : bmaxz) - oz) * (double)r.ivz;
  if ((t0 > tz1) || (tz0 > t1)) return false;
  if (tz0 > t0) t0 = tz0;
  if (tz1 < t1) t1 = tz1;
  *E = t0;
  return t1 > r.tmin;
}
DR_DEV uint32_t pack_ref(uint32_t ref, uint32_t meta) {
  const uint32_t nprims = meta & 0xffffu;
  return nprims ? (PREF_LEAF | (nprims << 26) | ref) : ((((meta >> 16) & 3u) << 29) | ref);
}

#define M_EXPAND 1  // `cur` is an interior node whose box is hit (or must be re-tested): fetch its pair

// COLD (k_trace3c): direction, minDistance, maxDistance and the queue handle wait in 8 LDS rows per lane (`cold`) between the refill,
// the leaf tests, the rare literal slab tests and the result store, as in k_trace / k_trace3a: they are re-read in front of every use
// (reloadCold), so the loop that sets the occupancy does not hold them in registers.
// Closest-hit rays only (BVHAccel.intersect); the any-hit rays have their own 4-byte-entry form, trace_pairs_any below.  (Round 3's
// any-hit instantiation of this function -- k_trace3<1>, DARTRAY_ANY8 -- is experiments/r06_pair_kernel_any8.diff.)
template <class IO, bool COLD = false, int PSTACK = DR_PSTACK>
DR_DEV void trace_pairs(const DScene& sc, const IO& io, uint32_t n, uint32_t* ldsRef, float* ldsE, uint32_t* spill,
                        uint32_t spillStride, uint32_t spillHalf, uint32_t* work, TraceCounters* ctr, uint32_t* cold = nullptr) {
  typedef __attribute__((address_space(3))) uint32_t cold_u32;
#define COLD_TMAX() __hiloint2double((int)COLD_LD(7), (int)COLD_LD(6))
  // (the lane number is recomputed where it is needed -- lane_id(): one mbcnt pair -- instead of living in a register)
  // (rank of this lane among the set lanes of a mask: mbcnt, no 64-bit lane mask held in registers)
  auto rankIn = [](unsigned long long m) -> uint32_t {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  };
  uint32_t nRays = 0, nNodes = 0, nTris = 0;
  TraceRay ray;
  ray.needF64 = false;
  uint32_t handle = 0, cur = 0;
  int sp = 0, hit = -1, mode = M_IDLE;  // mode: M_* in bits 0..1; bit 2 (M_RETEST): cur was popped inside the ambiguous band --
                                        // evaluate the literal test on its own box (one register for both)
#define M_RETEST 4
#define MODE_IS(m) ((mode & 3) == (m))
  bool exhausted = false;
  uint32_t resNext = 0, resEnd = 0;
  auto reloadCold = [&]() {
    if constexpr (COLD) {
      ray.d = COLD_D();
      ray.tmin = COLD_TMIN();
      ray.tmax = COLD_TMAX();
    }
  };
  auto stackGet = [&](int i, uint32_t* ref, float* e) {
    *ref = ldsRef[(i < PSTACK ? i : PSTACK - 1) * DR_TRACE_BLOCK];
    *e = ldsE[(i < PSTACK ? i : PSTACK - 1) * DR_TRACE_BLOCK];
    if (i >= PSTACK) {
      *ref = *spill_at(spill, spillStride, i - PSTACK);
      *e = __uint_as_float(*spill_at(spill + spillHalf, spillStride, i - PSTACK));
    }
  };
  auto stackSet = [&](int i, uint32_t ref, float e) {
    if (i < PSTACK) {
      ldsRef[i * DR_TRACE_BLOCK] = ref;
      ldsE[i * DR_TRACE_BLOCK] = e;
    } else if (i < DR_MAX_STACK) {
      *spill_at(spill, spillStride, i - PSTACK) = ref;
      *spill_at(spill + spillHalf, spillStride, i - PSTACK) = __float_as_uint(e);
    }
  };
  SPROF_DECL;
  auto push = [&](uint32_t ref, float e) {
    stackSet(sp, ref, e);
    ++sp;
    SPROF_PUSH();
  };
  // A far child that can never be hit still is ONE visit when the reference pops it.  Closest-hit rays pop every entry
  // sooner or later, so the visit is counted right away and nothing is pushed.
  auto pushDead = [&]() { ++nNodes; };
  // maxDistance just shrank (closest hit only): drop every entry that is now certainly beyond it and count
  // its visit -- the reference would pop and reject each of them later, with no other effect.  What stays on
  // the stack survives its pop (up to the ambiguous band), so pops never turn into long chains.
  auto pruneStack = [&]() {
    const float R = 4.76837158203125e-07f, A = 1.0e-37f;
    int j = 0;
    for (int i = 0; i < sp; ++i) {
      uint32_t r;
      float e;
      stackGet(i, &r, &e);
      const float eb = __fmaf_rn(fabsf(e), R, A);
      if (e - eb >= ray.tmaxHi) {
        ++nNodes;
        continue;
      }
      if (j != i) stackSet(j, r, e);
      ++j;
    }
    sp = j;
  };
  // Pop until an entry survives (bvh_accel.dart:139-143,156-159); every popped entry is one node visit.
  // Returns false when the stack is empty (the ray is finished).
  auto popNext = [&]() -> bool {
    for (;;) {
      if (sp == 0) return false;
      --sp;
      uint32_t ref;
      float e;
      stackGet(sp, &ref, &e);
      ++nNodes;
      const float R = 4.76837158203125e-07f, A = 1.0e-37f;
      const float eb = __fmaf_rn(fabsf(e), R, A);
      if (e - eb >= ray.tmaxHi) continue;  // entry >= maxDistance for certain: pruned, nothing fetched
      const float tmaxLo = COLD ? __uint_as_float(__float_as_uint(ray.tmaxHi) - (ray.tmaxHi < 0.f ? 0xffffffffu : 1u)) : ray.tmaxLo;
      cur = ref;
      mode = ((ref & PREF_LEAF) ? M_LEAF : M_EXPAND) | (!(e + eb < tmaxLo) ? M_RETEST : 0);  // retest: not certainly before maxDistance (also NaN: needF64 rays)
      return true;
    }
  };

  for (;;) {
    // ---- refill idle lanes (as in v2) ----
    const unsigned long long idleMask = __ballot(MODE_IS(M_IDLE));
    const int nIdle = __popcll(idleMask);
    if (!exhausted && (nIdle >= DR_REFILL_TH || nIdle == 64)) {
      if (resNext == resEnd) {
        uint32_t fresh = 0;
        if (lane_id() == 0) {
          fresh = nRays == 0u ? __hip_atomic_load(work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;  // (see trace_persistent)
          if (fresh < n) fresh = atomicAdd(work, work_chunk(n));
        }
        fresh = wave_bcast_first(fresh);
        if (fresh < n) {
          resNext = fresh;
          resEnd = min(fresh + work_chunk(n), n);
        } else {
          exhausted = true;
        }
      }
      const uint32_t take = min(resEnd - resNext, (uint32_t)nIdle);
      if (MODE_IS(M_IDLE)) {
        const uint32_t j = rankIn(idleMask);
        if (j < take) {
          io.load(resNext + j, ray, handle);
          sp = 0;
          hit = -1;
          // visit node 0 (its box and packed reference live in kernel arguments)
          bool ok = false;
          if (sc.rootRef != PREF_DEAD) {
            ++nNodes;
            int d = ray.needF64 ? -1 : slab_f32(ray, sc.rootBox[0], sc.rootBox[1], sc.rootBox[2], sc.rootBox[3], sc.rootBox[4], sc.rootBox[5]);
            if (d < 0) d = slab_f64(ray, sc.rootBox[0], sc.rootBox[1], sc.rootBox[2], sc.rootBox[3], sc.rootBox[4], sc.rootBox[5]) ? 1 : 0;
            ok = d != 0;
          }
          if (ok) {
            cur = sc.rootRef;
            mode = (cur & PREF_LEAF) ? M_LEAF : M_EXPAND;
            if constexpr (COLD) {
              COLD_ST(0, __float_as_uint(ray.d.x));
              COLD_ST(1, __float_as_uint(ray.d.y));
              COLD_ST(2, __float_as_uint(ray.d.z));
              COLD_ST(3, (uint32_t)__double2loint(ray.tmin));
              COLD_ST(4, (uint32_t)__double2hiint(ray.tmin));
              COLD_ST(5, handle);
              COLD_ST(6, (uint32_t)__double2loint(ray.tmax));
              COLD_ST(7, (uint32_t)__double2hiint(ray.tmax));
            }
          } else {
            io.store(handle, ray, -1, sc);
          }
        }
      }
      nRays += take;  // wave-uniform: lane 0 reports it
      resNext += take;
    }
    const unsigned long long expMask = __ballot(MODE_IS(M_EXPAND));
    unsigned long long leafMask = __ballot(MODE_IS(M_LEAF));
    if ((expMask | leafMask | __ballot(MODE_IS(M_DONE))) == 0ull) {
      if (exhausted) break;
      continue;
    }
    bool finished = false;
    // ---- expand one interior node: ONE 64-byte fetch, two box tests ----
    // (the near child is the second record when the ray runs against the split axis, bvh_accel.dart:147-153: known from the
    // reference's axis bits before the fetch, so the two halves are LOADED as near / far instead of being selected
    // component by component afterwards -- 16 v_cndmask and their registers less per step)
    uint4 n0 = uint4{0, 0, 0, 0}, n1 = n0, f0 = n0, f1 = n0;
    if (MODE_IS(M_EXPAND)) {
      const uint32_t axis = (cur >> 29) & 3u;
      const float iv = axis == 0 ? ray.ivx : (axis == 1 ? ray.ivy : ray.ivz);
      const uint32_t nearHalf = iv < 0.f ? 2u : 0u;
      const uint4* pp = sc.pairs + 4 * (size_t)(cur & 0x1fffffffu);
      n0 = pp[nearHalf], n1 = pp[nearHalf + 1u], f0 = pp[2u - nearHalf], f1 = pp[3u - nearHalf];
    }
    // results of the rays that finished in the previous iteration, stored behind this iteration's fetches (vmcnt
    // retires in order: see trace_persistent)
    if (MODE_IS(M_DONE)) {
      reloadCold();
      if constexpr (COLD) handle = COLD_LD(5);
      io.store(handle, ray, hit, sc);
      SPROF_RAY(0);
      mode = M_IDLE;
    }
    if (MODE_IS(M_EXPAND)) {
      bool alive = true;
      if (mode & M_RETEST) {
        // own box = union of the children's (bvh_accel.dart:521): the literal test the reference does at this pop
        const float ux0 = fminf(__uint_as_float(n0.x), __uint_as_float(f0.x)), uy0 = fminf(__uint_as_float(n0.y), __uint_as_float(f0.y));
        const float uz0 = fminf(__uint_as_float(n0.z), __uint_as_float(f0.z)), ux1 = fmaxf(__uint_as_float(n0.w), __uint_as_float(f0.w));
        const float uy1 = fmaxf(__uint_as_float(n1.x), __uint_as_float(f1.x)), uz1 = fmaxf(__uint_as_float(n1.y), __uint_as_float(f1.y));
        reloadCold();
        alive = slab_f64(ray, ux0, uy0, uz0, ux1, uy1, uz1);
        mode &= 3;
      }
      if (alive) {
        const float nbx0 = __uint_as_float(n0.x), nby0 = __uint_as_float(n0.y), nbz0 = __uint_as_float(n0.z);
        const float nbx1 = __uint_as_float(n0.w), nby1 = __uint_as_float(n1.x), nbz1 = __uint_as_float(n1.y);
        const float fbx0 = __uint_as_float(f0.x), fby0 = __uint_as_float(f0.y), fbz0 = __uint_as_float(f0.z);
        const float fbx1 = __uint_as_float(f0.w), fby1 = __uint_as_float(f1.x), fbz1 = __uint_as_float(f1.y);
        // far child: what the reference pushes; decide everything that does not depend on maxDistance now
        uint32_t farRef = pack_ref(f1.z, f1.w);
        float farE;
        if (ray.needF64) {
          farE = __uint_as_float(0x7fc00000u);  // NaN: always evaluated literally when popped
        } else {
          const SlabB fb = slab_bounds(ray, fbx0, fby0, fbz0, fbx1, fby1, fbz1);
          // (COLD: one bracket end per bound is kept; the other is its neighbour or itself -- taking the neighbour only sends a
          // few more boxes to the literal test).  The neighbour is chosen by the SIGN BIT, not by `< 0`: minDistance == -0.0 has
          // tminLo == -0.0, for which `< 0` is false and bits + 1 would be the negative denormal BELOW it (a bracket end under
          // the value it must bound: sureIn could skip a literal test); by the sign bit -0.0 steps to 0x7fffffff, a NaN, and
          // every comparison with it fails -- the box goes to the literal test.  (pred(tmaxHi) at the pops yields NaN at both
          // zeros already: conservative.)  Render rays have minDistance 0 or an epsilon > 0; dr_intersect takes any ray.
          const float tminHi = COLD ? __uint_as_float(__float_as_uint(ray.tminLo) + ((int32_t)__float_as_uint(ray.tminLo) < 0 ? 0xffffffffu : 1u)) : ray.tminHi;
          const bool sureIn = (fb.loU <= fb.hiL) && (fb.hiL > tminHi);
          const bool sureOut = (fb.loL > fb.hiU) || (fb.hiU <= ray.tminLo);
          farE = fb.lo;
          if (sureOut) {
            farRef = PREF_DEAD;
          } else if (!sureIn) {
            double E;
            reloadCold();
            if (slab_geom_f64(ray, fbx0, fby0, fbz0, fbx1, fby1, fbz1, &E)) farE = (float)E;
            else farRef = PREF_DEAD;
          }
          // already beyond maxDistance for certain => it will be when popped (maxDistance only shrinks)
          if (farRef != PREF_DEAD && fb.loL >= ray.tmaxHi && sureIn) farRef = PREF_DEAD;
        }
        if (farRef == PREF_DEAD) pushDead();
        else push(farRef, farE);
        // near child: visited now
        ++nNodes;
        int d = ray.needF64 ? -1 : slab_f32(ray, nbx0, nby0, nbz0, nbx1, nby1, nbz1);
        if (d < 0) {
          reloadCold();
          d = slab_f64(ray, nbx0, nby0, nbz0, nbx1, nby1, nbz1) ? 1 : 0;
        }
        if (d) {
          cur = pack_ref(n1.z, n1.w);
          mode = (cur & PREF_LEAF) ? M_LEAF : M_EXPAND;
        } else if (!popNext()) {
          finished = true;
        }
      } else if (!popNext()) {
        finished = true;
      }
    }
    // ---- batched leaf tests ----
    leafMask = __ballot(MODE_IS(M_LEAF));
    const unsigned long long stillExp = __ballot(MODE_IS(M_EXPAND) && !finished);
    if (leafMask != 0ull && (__popcll(leafMask) >= DR_LEAF_TH || stillExp == 0ull)) {
      if (MODE_IS(M_LEAF)) {
        const uint32_t leafN = (cur >> 26) & 31u, leafOff = cur & 0x3ffffffu;
        bool alive = true;
        reloadCold();
        if (mode & M_RETEST) {
          // own box = union of the triangles' vertices (Triangle.worldBound, :238-241)
          float x0 = __uint_as_float(0x7f800000u), y0 = x0, z0 = x0, x1 = -x0, y1 = -x0, z1 = -x0;
          for (uint32_t i = 0; i < leafN; ++i) {
            const float4* tp = sc.tris + 3 * (size_t)(leafOff + i);
            const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
            x0 = fminf(x0, fminf(q0.x, fminf(q0.w, q1.z))); x1 = fmaxf(x1, fmaxf(q0.x, fmaxf(q0.w, q1.z)));
            y0 = fminf(y0, fminf(q0.y, fminf(q1.x, q1.w))); y1 = fmaxf(y1, fmaxf(q0.y, fmaxf(q1.x, q1.w)));
            z0 = fminf(z0, fminf(q0.z, fminf(q1.y, q2.x))); z1 = fmaxf(z1, fmaxf(q0.z, fmaxf(q1.y, q2.x)));
          }
          alive = slab_f64(ray, x0, y0, z0, x1, y1, z1);
          mode &= 3;
        }
        bool shrunk = false;
        if (alive) {
          for (uint32_t i = 0; i < leafN; ++i) {
            ++nTris;
            const float4* tp = sc.tris + 3 * (size_t)(leafOff + i);
            const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
            const F3 p1 = F3{q0.x, q0.y, q0.z}, p2 = F3{q0.w, q1.x, q1.y}, p3 = F3{q1.z, q1.w, q2.x};
            double t, b1, b2;
            if (tri_hit(p1, p2, p3, ray.o, ray.d, ray.tmin, ray.tmax, &t, &b1, &b2)) {
              ray_set_tmax(ray, t);
              hit = (int)(leafOff + i);
              shrunk = true;
            }
          }
        }
        if (shrunk) {
          if constexpr (COLD) {
            COLD_ST(6, (uint32_t)__double2loint(ray.tmax));
            COLD_ST(7, (uint32_t)__double2hiint(ray.tmax));
          }
          pruneStack();
        }
        if (!popNext()) finished = true;
      }
    }
    if (finished) mode = M_DONE;  // stored at the head of the next iteration
  }
  flush_counters(ctr, 0, lane_id() == 0 ? nRays : 0u, nNodes, nTris);
#undef MODE_IS
#undef M_RETEST
}

template <int ANY>
__global__ void __launch_bounds__(DR_TRACE_BLOCK, DR_TRACE3_WAVES) k_trace3(DScene sc, BatchState st, const uint32_t* queue,
                                                                            const uint32_t* nQueue, uint32_t* spill,
                                                                            uint32_t* work, TraceCounters* ctr) {
  __shared__ uint32_t s_ref[DR_PSTACK * DR_TRACE_BLOCK];
  __shared__ float s_e[DR_PSTACK * DR_TRACE_BLOCK];
  static_assert(ANY == 0, "closest-hit rays only: the any-hit pair kernel is k_trace3a");
  StateIO<0> io{st, queue};
  const uint32_t n = nQueue ? *nQueue : st.nslots;
  const uint32_t stride = gridDim.x * DR_TRACE_BLOCK;
  trace_pairs(sc, io, n, s_ref + threadIdx.x, s_e + threadIdx.x, spill, stride, stride * (uint32_t)(DR_MAX_STACK - DR_PSTACK), work, ctr);
}
template <int ANY>
__global__ void __launch_bounds__(DR_TRACE_BLOCK, DR_TRACE3_WAVES) k_intersect3(DScene sc, const DrRay* rays, uint32_t n,
                                                                                DrHit* out, uint32_t* spill, uint32_t* work,
                                                                                TraceCounters* ctr) {
  __shared__ uint32_t s_ref[DR_PSTACK * DR_TRACE_BLOCK];
  __shared__ float s_e[DR_PSTACK * DR_TRACE_BLOCK];
  static_assert(ANY == 0, "closest-hit rays only: the any-hit pair kernel is k_intersect3a");
  RayIO<0> io{rays, out};
  const uint32_t stride = gridDim.x * DR_TRACE_BLOCK;
  trace_pairs(sc, io, n, s_ref + threadIdx.x, s_e + threadIdx.x, spill, stride, stride * (uint32_t)(DR_MAX_STACK - DR_PSTACK), work, ctr);
}

// k_trace3c: the closest-hit pair traversal with its cold ray state in LDS.  Stacks are shallow -- on C4 98.8 % of the closest-hit
// rays never hold more than 10 entries (the far children that fail are not pushed, and every hit prunes the stack), 99.96 % not more
// than 13 (-DDR_STACK_PROF) -- so DR_PSTACK_C rows of 8-byte entries + 8 cold rows fit SIX workgroups per CU where k_trace3<0> runs five.
#ifndef DR_PSTACK_C
#define DR_PSTACK_C 9
#endif
#ifndef DR_TRACE3C_WAVES
#define DR_TRACE3C_WAVES 6
#endif
__global__ void __launch_bounds__(DR_TRACE_BLOCK, DR_TRACE3C_WAVES) k_trace3c(DScene sc, BatchState st, const uint32_t* queue,
                                                                              const uint32_t* nQueue, uint32_t* spill, uint32_t* work,
                                                                              TraceCounters* ctr) {
  __shared__ uint32_t s_ref[DR_PSTACK_C * DR_TRACE_BLOCK];
  __shared__ float s_e[DR_PSTACK_C * DR_TRACE_BLOCK];
  __shared__ uint32_t s_cold[8 * DR_TRACE_BLOCK];
  StateIO<0> io{st, queue};
  const uint32_t n = nQueue ? *nQueue : st.nslots;
  const uint32_t stride = gridDim.x * DR_TRACE_BLOCK;
  trace_pairs<StateIO<0>, true, DR_PSTACK_C>(sc, io, n, s_ref + threadIdx.x, s_e + threadIdx.x, spill, stride,
                                               stride * (uint32_t)(DR_MAX_STACK - DR_PSTACK_C), work, ctr, s_cold + threadIdx.x);
}

// ===========================================================================
// v3 for any-hit rays (k_trace3a): 4-byte stack entries.
// intersectP never shrinks maxDistance (bvh_accel.dart:167-226), so everything the reference decides about a far child
// when it POPS it is already decided when it PUSHES it.  The far child's whole slab test -- the near child's: f32
// filter, literal f64 fallback -- is therefore evaluated at push time; a child that fails is a DEAD entry (one node
// visit if the ray ever gets to pop it, like the reference; a run of them is a count in a register / one merged entry),
// a child that passes is pushed as a bare reference and expanded when popped with no further test.  No entry parameter
// next to the reference, no re-test, no pruning pass: 4 bytes per entry instead of 8, and direction / minDistance /
// maxDistance / the queue handle wait in LDS between refill and leaf test as in k_trace -- 14 + 8 rows = 22 KiB per
// workgroup and 70 VGPRs: seven workgroups per CU where k_trace3<1> had five (MEASUREMENTS.md, round 4).
// ===========================================================================
// 14 stack rows + 8 rows of cold ray state = 22 KiB per workgroup and 70 VGPRs => SEVEN workgroups per CU, like k_trace.  Measured:
// C4 any-hit 113.3 (k_trace3<1>, 5 workgroups) -> 103.6 ms with 16 rows / 6 workgroups, 104.6 with 14 / 7; on the cache-resident C2
// 16 / 6 loses to k_trace<1> (114 against 96 ms) and 14 / 7 equals it (96.0): the seventh workgroup matters more than two stack rows.
// 12 rows / 8 workgroups spill (64 VGPRs): C4 117 ms.
#ifndef DR_PSTACK_A
#define DR_PSTACK_A 14
#endif
#ifndef DR_TRACE3A_WAVES
#define DR_TRACE3A_WAVES 7
#endif
#ifndef DR_REFILL_TH_3A
#define DR_REFILL_TH_3A 24  // idle lanes before a refill (C4 any-hit: 8 / 16 / 24 idle lanes 111.2 / 105.1 / 101.8 ms; C5 indifferent)
#endif
#define PREF_DEADN 0x60000000u  // (axis bits == 3: no interior reference carries them) | number of merged dead entries

template <class IO>
DR_DEV void trace_pairs_any(const DScene& sc, const IO& io, uint32_t n, uint32_t* ldsRef, uint32_t* cold, uint32_t* spill,
                            uint32_t spillStride, uint32_t* work, TraceCounters* ctr) {
  typedef __attribute__((address_space(3))) uint32_t cold_u32;
#define COLD_TMAX() __hiloint2double((int)COLD_LD(7), (int)COLD_LD(6))
  const int lane = lane_id();
  uint32_t nRays = 0, nNodes = 0, nTris = 0;
  TraceRay ray;
  ray.needF64 = false;
  uint32_t cur = 0, deadTop = 0;  // deadTop: dead entries on top of the stack that have not been written yet
  int sp = 0, hit = -1, mode = M_IDLE;
  bool exhausted = false;
  uint32_t resNext = 0, resEnd = 0;

  auto stackGet = [&](int i) -> uint32_t { return stack_pop<DR_PSTACK_A>(ldsRef, spill, spillStride, i); };
  auto stackSet = [&](int i, uint32_t ref) {
    if (i < DR_PSTACK_A) ldsRef[i * DR_TRACE_BLOCK] = ref;
    else if (i < DR_MAX_STACK) *spill_at(spill, spillStride, i - DR_PSTACK_A) = ref;
  };
  SPROF_DECL;
  auto push = [&](uint32_t ref) {
    if (deadTop) {
      stackSet(sp, PREF_DEADN | deadTop);
      ++sp;
      deadTop = 0;
    }
    stackSet(sp, ref);
    ++sp;
    SPROF_PUSH();
  };
  // the literal test needs the f64 bounds of the ray: they wait in LDS
  auto boxHit = [&](float x0, float y0, float z0, float x1, float y1, float z1) -> bool {
    bool ok = false, amb = true;
    if (!ray.needF64) {
      bool sureMiss;
      slab_f32_sure(ray, x0, y0, z0, x1, y1, z1, &ok, &sureMiss);
      amb = !ok && !sureMiss;
    }
    if (amb) {
      TraceRay rr = ray;
      rr.tmin = COLD_TMIN();
      rr.tmax = COLD_TMAX();
      ok = slab_f64(rr, x0, y0, z0, x1, y1, z1);
    }
    return ok;
  };
  // Pop the next live entry (bvh_accel.dart:206-210); every popped entry is one node visit.  False: the stack is empty.
  auto popNext = [&]() -> bool {
    for (;;) {
      nNodes += deadTop;
      deadTop = 0;
      if (sp == 0) return false;
      --sp;
      const uint32_t ref = stackGet(sp);
      if ((ref & 0xe0000000u) == PREF_DEADN) {
        nNodes += ref & 0x1fffffffu;
        continue;
      }
      ++nNodes;
      cur = ref;
      mode = (ref & PREF_LEAF) ? M_LEAF : M_EXPAND;
      return true;
    }
  };

  for (;;) {
    // ---- refill idle lanes (as in trace_persistent) ----
    const unsigned long long idleMask = __ballot(mode == M_IDLE);
    const int nIdle = __popcll(idleMask);
    if (!exhausted && (nIdle >= DR_REFILL_TH_3A || nIdle == 64)) {
      if (resNext == resEnd) {
        uint32_t fresh = 0;
        if (lane == 0) {
          fresh = nRays == 0u ? __hip_atomic_load(work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;  // (see trace_persistent)
          if (fresh < n) fresh = atomicAdd(work, work_chunk(n));
        }
        fresh = wave_bcast_first(fresh);
        if (fresh < n) {
          resNext = fresh;
          resEnd = min(fresh + work_chunk(n), n);
        } else {
          exhausted = true;
        }
      }
      const uint32_t take = min(resEnd - resNext, (uint32_t)nIdle);
      if (mode == M_IDLE) {
        const uint32_t j = __builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0u));
        if (j < take) {
          uint32_t handle;
          io.load(resNext + j, ray, handle);
          COLD_ST(0, __float_as_uint(ray.d.x));
          COLD_ST(1, __float_as_uint(ray.d.y));
          COLD_ST(2, __float_as_uint(ray.d.z));
          COLD_ST(3, (uint32_t)__double2loint(ray.tmin));
          COLD_ST(4, (uint32_t)__double2hiint(ray.tmin));
          COLD_ST(5, handle);
          COLD_ST(6, (uint32_t)__double2loint(ray.tmax));
          COLD_ST(7, (uint32_t)__double2hiint(ray.tmax));
          sp = 0;
          deadTop = 0;
          hit = -1;
          mode = M_DONE;  // a ray that misses the root box: a miss, stored like every other result
          // visit node 0 (its box and packed reference live in kernel arguments)
          if (sc.rootRef != PREF_DEAD) {
            ++nNodes;
            if (boxHit(sc.rootBox[0], sc.rootBox[1], sc.rootBox[2], sc.rootBox[3], sc.rootBox[4], sc.rootBox[5])) {
              cur = sc.rootRef;
              mode = (cur & PREF_LEAF) ? M_LEAF : M_EXPAND;
            }
          }
        }
      }
      if (lane == 0) nRays += take;  // (lane 0 carries the wave's count: flush_counters sums the lanes)
      resNext += take;
    }
    const unsigned long long expMask = __ballot(mode == M_EXPAND);
    unsigned long long leafMask = __ballot(mode == M_LEAF);
    if ((expMask | leafMask | __ballot(mode == M_DONE)) == 0ull) {
      if (exhausted) break;
      continue;
    }
    bool finished = false;
    // ---- expand one interior node: ONE 64-byte fetch, two box tests ----
    uint4 n0 = uint4{0, 0, 0, 0}, n1 = n0, f0 = n0, f1 = n0;  // (loaded as near / far: see trace_pairs)
    if (mode == M_EXPAND) {
      const uint32_t axis = (cur >> 29) & 3u;
      const float iv = axis == 0 ? ray.ivx : (axis == 1 ? ray.ivy : ray.ivz);
      const uint32_t nearHalf = ((iv < 0.f) != (sc.anyFarFirst != 0u)) ? 2u : 0u;  // (far child first: DScene.anyFarFirst)
      const uint4* pp = (const uint4*)((const char*)sc.pairs + ((size_t)(cur & 0x1fffffffu) << 6));
      n0 = pp[nearHalf], n1 = pp[nearHalf + 1u], f0 = pp[2u - nearHalf], f1 = pp[3u - nearHalf];
    }
    // results of the rays that finished in the previous iteration, stored behind this iteration's fetches (vmcnt
    // retires in order: see trace_persistent)
    if (mode == M_DONE) {
      io.store(COLD_LD(5), ray, hit, sc);
      SPROF_RAY(1);
      mode = M_IDLE;
    }
    if (mode == M_EXPAND) {
      // far child: what the reference pushes; its test cannot change before it is popped, so it is decided now
      if (boxHit(__uint_as_float(f0.x), __uint_as_float(f0.y), __uint_as_float(f0.z), __uint_as_float(f0.w), __uint_as_float(f1.x),
                 __uint_as_float(f1.y)))
        push(pack_ref(f1.z, f1.w));
      else
        ++deadTop;
      // near child: visited now
      ++nNodes;
      if (boxHit(__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z), __uint_as_float(n0.w), __uint_as_float(n1.x),
                 __uint_as_float(n1.y))) {
        cur = pack_ref(n1.z, n1.w);
        mode = (cur & PREF_LEAF) ? M_LEAF : M_EXPAND;
      } else if (!popNext()) {
        finished = true;
      }
    }
    // ---- batched leaf tests (bvh_accel.dart:189-204) ----
    leafMask = __ballot(mode == M_LEAF);
    const unsigned long long stillExp = __ballot(mode == M_EXPAND && !finished);
    if (leafMask != 0ull && (__popcll(leafMask) >= DR_LEAF_TH || stillExp == 0ull)) {
      if (mode == M_LEAF) {
        const uint32_t leafN = (cur >> 26) & 31u, leafOff = cur & 0x3ffffffu;
        const F3 rayD = COLD_D();
        const double rayTmin = COLD_TMIN(), rayTmax = COLD_TMAX();
        bool occluded = false;
        for (uint32_t i = 0; i < leafN; ++i) {
          ++nTris;
          const float4* tp = sc.tris + 3 * (size_t)(leafOff + i);
          const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
          const F3 p1 = F3{q0.x, q0.y, q0.z}, p2 = F3{q0.w, q1.x, q1.y}, p3 = F3{q1.z, q1.w, q2.x};
          if (tri_hitP(p1, p2, p3, ray.o, rayD, rayTmin, rayTmax)) {  // return true (bvh_accel.dart:193-195)
            occluded = true;
            break;
          }
        }
        if (occluded) {
          hit = 0;
          finished = true;
        } else if (!popNext()) {
          finished = true;
        }
      }
    }
    if (finished) mode = M_DONE;  // stored at the head of the next iteration
  }
  flush_counters(ctr, 1, nRays, nNodes, nTris);
#undef COLD_TMAX
}

__global__ void __launch_bounds__(DR_TRACE_BLOCK, DR_TRACE3A_WAVES) k_trace3a(DScene sc, BatchState st, const uint32_t* queue,
                                                                              const uint32_t* nQueue, uint32_t* spill, uint32_t* work,
                                                                              TraceCounters* ctr) {
  __shared__ uint32_t s_ref[DR_PSTACK_A * DR_TRACE_BLOCK];
  __shared__ uint32_t s_cold[8 * DR_TRACE_BLOCK];  // direction, minDistance, queue handle, maxDistance per lane
  StateIO<1> io{st, queue};
  const uint32_t n = nQueue ? *nQueue : st.nslots;
  trace_pairs_any(sc, io, n, s_ref + threadIdx.x, s_cold + threadIdx.x, spill, gridDim.x * DR_TRACE_BLOCK, work, ctr);
}
__global__ void __launch_bounds__(DR_TRACE_BLOCK, DR_TRACE3A_WAVES) k_intersect3a(DScene sc, const DrRay* rays, uint32_t n, DrHit* out,
                                                                                  uint32_t* spill, uint32_t* work, TraceCounters* ctr) {
  __shared__ uint32_t s_ref[DR_PSTACK_A * DR_TRACE_BLOCK];
  __shared__ uint32_t s_cold[8 * DR_TRACE_BLOCK];
  RayIO<1> io{rays, out};
  trace_pairs_any(sc, io, n, s_ref + threadIdx.x, s_cold + threadIdx.x, spill, gridDim.x * DR_TRACE_BLOCK, work, ctr);
}

// ===========================================================================
// k_trace_pk (round 5): closest-hit traversal of COHERENT waves -- the camera rays.  A tile's 64 slots are samples of ONE pixel
// (spp >= 64; below that, of neighbouring pixels), i.e. 64 nearly identical rays.  In k_trace each lane still fetches its own
// node with two divergent 16-byte loads, and MEASUREMENTS.md 5.2 (f) finds what a ray waits for there: the queue in front of the
// CU's L1 address path, 61 cycles per wave and node visit.  Here the WAVE walks the tree with one stack: the node (and, at a
// leaf, every triangle) is loaded once for the wave from a wave-uniform address, and every lane tests ITS ray against it.
//
// Same visits, same order, same decisions per ray as bvh_accel.dart:101-163:
//  * rays are grouped by the sign pattern of their inverse direction (dirIsNeg, :111-113): within a group every ray takes the
//    same near / far order at every node (:147-153), so one shared order is every ray's own order.  The groups of a tile are
//    walked one after the other (same-pixel rays: almost always one group);
//  * a stack entry is (node, mask of the lanes that pushed it): exactly the lanes whose own traversal holds that far child on
//    its own stack; when the entry is popped those lanes test the node's box with THEIR current maxDistance, as the reference
//    does at its pop (:139-143).  A lane's entries are a sub-sequence of the wave's, so its pops come in its own LIFO order;
//  * a node is visited (and counted) by the lanes of the mask only; lanes whose box test fails simply leave the mask.
// Box and triangle tests are the functions k_trace uses (f32 enclosure, literal f64 fallback; tri_hit), so hits, hit
// parameters and the visit counters equal k_trace's and the oracle's.
// ===========================================================================
#ifndef DR_PK_WG_PER_CU
#define DR_PK_WG_PER_CU 6  // 80 VGPRs (the f64 triangle test at full lane width is the peak); C2's camera rays: 17.2 ms at six, 17.6 at five (94 VGPRs, no scratch), 18.8 at seven
#endif
// A load from a WAVE-UNIFORM address of data that no kernel writes while this one runs (the scene's nodes and triangles), through the
// constant address space: the compiler then uses a scalar load -- one request of the scalar cache for the wave, the data in SGPRs
// (operands of the lanes' VALU instructions) instead of 4 VGPRs per lane and 64 lane requests of the vector L1.
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t pk_u32x4 __attribute__((ext_vector_type(4)));
typedef float pk_f32x4 __attribute__((ext_vector_type(4)));
DR_DEV uint4 ld_uniform(const uint4* p) {
  const pk_u32x4 v = *(__attribute__((address_space(4))) const pk_u32x4*)(uintptr_t)p;
  return uint4{v.x, v.y, v.z, v.w};
}
DR_DEV float4 ld_uniform(const float4* p) {
  const pk_f32x4 v = *(__attribute__((address_space(4))) const pk_f32x4*)(uintptr_t)p;
  return float4{v.x, v.y, v.z, v.w};
}
#else
DR_DEV uint4 ld_uniform(const uint4* p) { return *p; }
DR_DEV float4 ld_uniform(const float4* p) { return *p; }
#endif
// (An any-hit instantiation for the camera vertices' shadow rays was measured in round 5 and lost -- C2 any-hit 95 -> 136 ms, the wave's
// union of visited nodes is several rays' worth: experiments/r06_coherent_shadow_rays.diff.)
__global__ void __launch_bounds__(DR_TRACE_BLOCK, DR_PK_WG_PER_CU) k_trace_pk(DScene sc, BatchState st, const uint32_t* queue, const uint32_t* nQueue,
                                                                             uint32_t* work, TraceCounters* ctr) {
  __shared__ uint32_t s_pk[(DR_TRACE_BLOCK / 64) * DR_MAX_STACK * 3];
  typedef __attribute__((address_space(3))) uint32_t pk_u32;
  pk_u32* const stk = (pk_u32*)(s_pk + (threadIdx.x >> 6) * (DR_MAX_STACK * 3));
  StateIO<0> io{st, queue};
  const uint32_t n = nQueue ? *nQueue : st.nslots;
  const int lane = lane_id();
  uint32_t nRays = 0, nNodes = 0, nTris = 0;  // nRays / nNodes: wave-uniform, lane 0 reports them; nTris: per lane
  const uint32_t nTiles = (n + 63u) >> 6;
  for (;;) {
    uint32_t t0 = 0;
    if (lane == 0) t0 = atomicAdd(work, 4u);  // four tiles (256 rays) per atomic, as k_trace's reservations
    t0 = wave_bcast_first(t0);
    if (t0 >= nTiles) break;
    const uint32_t t1 = min(t0 + 4u, nTiles);
    for (uint32_t tile = t0; tile < t1; ++tile) {
      const uint32_t idx = (tile << 6) + (uint32_t)lane;
      const bool have = idx < n;
      TraceRay ray;
      int hit = -1;
      {
        uint32_t handle;  // (re-read from the queue at the store: not held in a register across the walk)
        io.load(have ? idx : n - 1u, ray, handle);  // (a lane beyond the end loads the last ray and takes no part)
      }
      const unsigned long long haveMask = __ballot(have);
      nRays += (uint32_t)__popcll(haveMask);
      if (sc.nnodes != 0u) {
        // dirIsNeg (bvh_accel.dart:111-113) as three bits, rebuilt where it is read instead of living in a register
        auto octant = [&]() -> uint32_t { return (ray.ivx < 0.f ? 1u : 0u) | (ray.ivy < 0.f ? 2u : 0u) | (ray.ivz < 0.f ? 4u : 0u); };
        unsigned long long remaining = haveMask;
        while (remaining != 0ull) {
          const uint32_t oct = (uint32_t)__builtin_amdgcn_readlane((int)octant(), __ffsll((long long)remaining) - 1);
          const unsigned long long pmask = __ballot(have && octant() == oct) & remaining;
          remaining &= ~pmask;
          int sp = 0;
          uint32_t cur = 0u;
          unsigned long long curMask = pmask;
          for (;;) {
            const uint4* np = (const uint4*)((const char*)sc.nodes + (size_t)(cur << 5));  // wave-uniform address
            const uint4 a = ld_uniform(np), b = ld_uniform(np + 1);
            nNodes += (uint32_t)__popcll(curMask);
            bool ok = false;
            if ((curMask >> lane) & 1ull) {
              const float bminx = __uint_as_float(a.x), bminy = __uint_as_float(a.y), bminz = __uint_as_float(a.z);
              const float bmaxx = __uint_as_float(a.w), bmaxy = __uint_as_float(b.x), bmaxz = __uint_as_float(b.y);
              bool amb = true;
              if (!ray.needF64) {
                bool sureMiss;
                slab_f32_sure(ray, bminx, bminy, bminz, bmaxx, bmaxy, bmaxz, &ok, &sureMiss);
                amb = !ok && !sureMiss;
              }
              if (amb) ok = slab_f64(ray, bminx, bminy, bminz, bmaxx, bmaxy, bmaxz);
            }
            const unsigned long long hitMask = __ballot(ok);
            bool descend = false;
            if (hitMask != 0ull) {
              const uint32_t nprims = (uint32_t)__builtin_amdgcn_readfirstlane((int)(b.w & 0xffffu));
              if (nprims > 0u) {  // a leaf: every lane of hitMask tests every triangle (bvh_accel.dart:126-138)
                const uint32_t leafOff = (uint32_t)__builtin_amdgcn_readfirstlane((int)b.z);
                for (uint32_t i = 0; i < nprims; ++i) {
                  const float4* tp = sc.tris + 3 * (size_t)(leafOff + i);
                  const float4 q0 = ld_uniform(tp), q1 = ld_uniform(tp + 1), q2 = ld_uniform(tp + 2);
                  if (ok) {
                    ++nTris;
                    const F3 p1 = F3{q0.x, q0.y, q0.z}, p2 = F3{q0.w, q1.x, q1.y}, p3 = F3{q1.z, q1.w, q2.x};
                    double t, b1, b2;
                    if (tri_hit(p1, p2, p3, ray.o, ray.d, ray.tmin, ray.tmax, &t, &b1, &b2)) {
                      ray_set_tmax(ray, t);  // r.maxDistance = thit (geometric_primitive.dart:59)
                      hit = (int)(leafOff + i);
                    }
                  }
                }
              } else {  // interior: near child now, far child pushed for the lanes that are here (bvh_accel.dart:145-155)
                const uint32_t axis = (uint32_t)__builtin_amdgcn_readfirstlane((int)((b.w >> 16) & 0xffu));
                const uint32_t second = (uint32_t)__builtin_amdgcn_readfirstlane((int)b.z);
                const bool neg = ((oct >> axis) & 1u) != 0u;
                if (lane == 0 && sp < DR_MAX_STACK) {
                  stk[3 * sp] = neg ? cur + 1u : second;
                  stk[3 * sp + 1] = (uint32_t)hitMask;
                  stk[3 * sp + 2] = (uint32_t)(hitMask >> 32);
                }
                ++sp;
                cur = neg ? second : cur + 1u;
                curMask = hitMask;
                descend = true;
              }
            }
            if (!descend) {
              if (sp == 0) break;
              --sp;
              cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)stk[3 * sp]);
              const uint32_t mlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)stk[3 * sp + 1]);
              const uint32_t mhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)stk[3 * sp + 2]);
              curMask = (unsigned long long)mlo | ((unsigned long long)mhi << 32);
            }
          }
        }
      }
      if (have) io.store(queue ? queue[idx] : idx, ray, hit, sc);
      if (st.markAlive && have && hit >= 0) st.markAlive[idx >> st.markShift] = 1;  // lazy sample generation: this 64-pixel group has a vertex
    }
  }
  {  // (what THIS kernel traced, next to the totals: the bench line prices k_trace<0> and k_trace_pk separately)
    const unsigned long long r = wave_sum(lane == 0 ? nRays : 0u), nn = wave_sum(lane == 0 ? nNodes : 0u), t = wave_sum(nTris);
    if (lane == 0 && ctr && (r | nn | t) != 0ull) {
      atomicAdd(&ctr->pk_rays[0], r);
      atomicAdd(&ctr->pk_nodes[0], nn);
      atomicAdd(&ctr->pk_tris[0], t);
    }
  }
  flush_counters(ctr, 0, lane == 0 ? nRays : 0u, lane == 0 ? nNodes : 0u, nTris);
}

// ---------------------------------------------------------------------------
// launchers.  DARTRAY_TRACE_IMPL selects the kernel for A/B runs: 2 = k_trace, 3 = sibling pairs, 5 = 3 with k_trace3c.
// ---------------------------------------------------------------------------
// Kernel ids (DScene.traceKernel, DARTRAY_TRACE_IMPL, dr_scene_set_trace_kernels): 2 k_trace, 3 sibling pairs (k_trace3<0> /
// k_trace3a), 5 sibling pairs with the closest-hit rays' cold state in LDS (k_trace3c; the any-hit rays: k_trace3a as with 3); any-hit
// rays only: 6 = 2 and 7 = 3 with the far child first (DScene.anyFarFirst; DARTRAY_TRACE_IMPL: a trailing 'f', e.g. 5f).  Returned
// here: 2 / 3, with `*cold` set for id 5 on closest-hit rays and `*farFirst` for ids 6 / 7.
static int traceImpl(const DScene& sc, int anyHit, int force = 0, bool* cold = nullptr, bool* farFirst = nullptr) {
  const DrOpt eo = dr_opt("DARTRAY_TRACE_IMPL");  // (read per launch: dr_set_option may change it between renders)
  const char e = eo.first();
  int env = (e == '2' || e == '3' || e == '5') ? e - '0' : 0;
  if (env && anyHit && eo.value.size() > 1 && eo.value[1] == 'f') env = env == 2 ? 6 : 7;
  // v2 is the fastest on cache-resident trees, v3 (sibling pairs) on big incoherent ones (MEASUREMENTS.md):
  // unless DARTRAY_TRACE_IMPL fixes it, the choice is the one measured for this scene (sc.traceKernel, set by the
  // pilot of dr_render_device)
  int impl = force ? force : (env ? env : (sc.traceKernel[anyHit ? 1 : 0] ? (int)sc.traceKernel[anyHit ? 1 : 0] : 2));
  if (cold) *cold = !anyHit && impl == 5;
  if (farFirst) *farFirst = anyHit && (impl == 6 || impl == 7);
  if (impl == 5 || impl == 7) impl = 3;
  if (impl == 6) impl = 2;
  if (sc.nquads) return 2;                     // only v2 tests quadric primitives
  return (impl == 3 && !sc.pairs) ? 2 : impl;  // scenes the pair layout cannot encode use v2
}
void launch_intersect(const DScene& scIn, const DrRay* rays, int64_t n, DrHit* out, int anyHit, uint32_t* spill,
                      uint32_t* workCounter, TraceCounters* ctr, int grid, hipStream_t s, int forceImpl) {
  const dim3 g(grid), b(DR_TRACE_BLOCK);
  bool farFirst = false;
  const int impl = traceImpl(scIn, anyHit, forceImpl, nullptr, &farFirst);
  DScene sc = scIn;
  sc.anyFarFirst = farFirst ? 1u : 0u;
  if (impl == 3) {
    if (anyHit) hipLaunchKernelGGL(k_intersect3a, g, b, 0, s, sc, rays, (uint32_t)n, out, spill, workCounter, ctr);
    else hipLaunchKernelGGL(k_intersect3<0>, g, b, 0, s, sc, rays, (uint32_t)n, out, spill, workCounter, ctr);
  } else {
    if (anyHit) hipLaunchKernelGGL(k_intersect<1>, g, b, 0, s, sc, rays, (uint32_t)n, out, spill, workCounter, ctr);
    else hipLaunchKernelGGL(k_intersect<0>, g, b, 0, s, sc, rays, (uint32_t)n, out, spill, workCounter, ctr);
  }
}
// The camera rays of a batch (identity queue: tile t = slots 64 t .. 64 t + 63, samples of one pixel at spp >= 64) through the
// wave-coherent kernel.  Returns false when this scene / build cannot use it (quadric primitives: k_trace_quad's tests).
bool launch_trace_coherent(const DScene& sc, const BatchState& st, const uint32_t* queue, const uint32_t* nQueue, int anyHit, uint32_t* workCounter,
                           TraceCounters* ctr, int grid, hipStream_t s) {
  if (sc.nquads || anyHit) return false;
  grid = std::min(grid, traceGridFor(DR_PK_WG_PER_CU));
  hipLaunchKernelGGL(k_trace_pk, dim3(grid), dim3(DR_TRACE_BLOCK), 0, s, sc, st, queue, nQueue, workCounter, ctr);
  return true;
}
int trace_kernel_id(const DScene& sc, int anyHit) {
  bool cold = false, farFirst = false;
  const int impl = traceImpl(sc, anyHit, 0, &cold, &farFirst);
  if (farFirst) return impl == 3 ? 7 : 6;
  return impl == 3 && cold ? 5 : impl;
}
void launch_trace(const DScene& scIn, const BatchState& st, const uint32_t* queue, const uint32_t* nQueue, int anyHit,
                  uint32_t* spill, uint32_t* workCounter, TraceCounters* ctr, int grid, hipStream_t s) {
  bool cold = false, farFirst = false;
  const int impl = traceImpl(scIn, anyHit, 0, &cold, &farFirst);
  DScene sc = scIn;
  sc.anyFarFirst = farFirst ? 1u : 0u;
  if (impl == 3) grid = std::min(grid, traceGridFor(anyHit ? DR_TRACE3A_WAVES : (cold ? DR_TRACE3C_WAVES : DR_TRACE3_WAVES)));  // k_trace3: 30 KiB of LDS, 5 resident; k_trace3a: 22 KiB, 7; k_trace3c: 26 KiB, 6
  else if (!(impl == 2 && !sc.nquads)) grid = std::min(grid, traceGridFor(6));  // only k_trace fits 7 workgroups per CU
  const dim3 g(grid), b(DR_TRACE_BLOCK);
  if (impl == 3) {
    if (anyHit) hipLaunchKernelGGL(k_trace3a, g, b, 0, s, sc, st, queue, nQueue, spill, workCounter, ctr);
    else if (cold) hipLaunchKernelGGL(k_trace3c, g, b, 0, s, sc, st, queue, nQueue, spill, workCounter, ctr);
    else hipLaunchKernelGGL(k_trace3<0>, g, b, 0, s, sc, st, queue, nQueue, spill, workCounter, ctr);
  } else if (sc.nquads) {
    if (anyHit) hipLaunchKernelGGL(k_trace_quad<1>, g, b, 0, s, sc, st, queue, nQueue, spill, workCounter, ctr);
    else hipLaunchKernelGGL(k_trace_quad<0>, g, b, 0, s, sc, st, queue, nQueue, spill, workCounter, ctr);
  } else {
    if (anyHit) hipLaunchKernelGGL(k_trace<1>, g, b, 0, s, sc, st, queue, nQueue, spill, workCounter, ctr);
    else hipLaunchKernelGGL(k_trace<0>, g, b, 0, s, sc, st, queue, nQueue, spill, workCounter, ctr);
  }
}

#ifdef DR_NS
}  // namespace DR_NS
#endif
