// dr_kernels.h -- structures shared between the kernels (dr_kernels.hip) and
// the host driver (dr_api.hip).
#ifndef DR_KERNELS_H
#define DR_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dartray_hip.h"
#include "dr_device.h"

// One LDShuffleScrambled1D/2D block of the camera-sample vector (montecarlo.dart:437-448): `n` entries per
// pixel sample starting at float `dst` (2-D entries interleave x, y).
struct LdBlock {
  int32_t dst, n, is2D, pad;
};
// One EstimateDirect call of UniformSampleAllLights (integrator.dart:39-77): sample j of light `light`;
// lc / lp / bd are the float indices of its light component, light position and BSDF direction samples.
// light < 0: the one call of UniformSampleOneLight (integrator.dart:79-117; DirectLighting strategy "one"): the light is
// floor(u * nLights) of the 1-D slot at float index `ln`, and the estimate is scaled by nLights instead of divided by n.
struct DirectStage {
  int32_t light, n, last, lc, lp, bd, bc, ln;  // bc: float index of the BSDF sample's uComponent
};

// Per-render constants (by-value kernel argument).
struct RenderParams {
  float r2c[16], c2w[16];
  double lensRadius, focalDistance, shutterOpen, shutterClose;  // Dart doubles (projective_camera.dart:31-32)
  int32_t cameraType, padCam;  // DR_CAMERA_*
  // ImageFilm window (image_film.dart:61-65)
  int32_t xres, yres, left, top, width, height;
  double fxw, fyw, invX, invY;
  int32_t integrator, maxDepth, spp, sppShift;
  int32_t nFloats, n1D;  // floats per sample vector, number of 1-D floats
  int32_t samplerMode;
  int32_t nLights;
  uint64_t seed;
  // full sampler extent (ImageFilm.getSampleExtent image_film.dart:247-252): keys of the counter streams
  int32_t extX0, extY0, extW, extH;
  int32_t maxTail;
  int32_t nBlocks;             // LD blocks per pixel: image, lens, time, 1-D slots, 2-D slots
  const LdBlock* blocks;       // null => every slot has one entry and the layout is arithmetic
  const DirectStage* dstages;  // DirectLighting: nDirectStages entries
  int32_t nDirectStages;
  int32_t dlSpecular;  // DirectLighting over mirror / glass: k_shade_spec adds SpecularReflect / SpecularTransmit after the last stage
  int32_t deferredNee; // PathIntegrator: k_film adds the PF_DEFERRED light terms
  int32_t genSlowDraws;  // tests: k_gen_samples_lm redoes every group of draws the slow way (the path a rare generator value takes)
  uint64_t genMask;    // LD blocks the device sampler must produce (bit = block index; 0 = all): blocks no kernel reads are skipped
};

// Flags of a path slot.
#define PF_HAS_SH 1u    // shadow (any-hit) ray of the light-sampling half is pending
#define PF_HAS_MIS 2u   // closest-hit ray of the BSDF-sampling half is pending
#define PF_HAS_CONT 4u  // continuation ray is pending
#define PF_SPECULAR 8u  // the continuation ray was sampled from a specular lobe (specularBounce)
#define PF_RAW_NEE 16u  // path kernel: Ld1 / Ld2 / betaNee hold the raw EstimateDirect terms (a MIS ray is pending, a delta
#define PF_DEFERRED 32u  // path kernel: the path ended here and its last light term (finished, in Ld1) is added by k_film once the
                         // shadow ray has been traced: the slot is in no later stage's list
                        // light, or a non-finite throughput); clear: Ld1 already is pathThroughput * (Ld1 * nLights)

#define Q_MIS_BIT 0x80000000u
#define Q_ENV_MISS_BIT 0x80000000u  // entry of the environment-map list: an escaped camera ray (else: a parked light estimate)
#define Q_RESOLVE_BIT 0x40000000u  // active-list entry of k_shade_path: the slot has no vertex to shade, only a light estimate to fold in

// State of one batch of camera samples.  Slot s belongs to batch pixel s >> sppShift, sample s & (spp-1).
// Layout: array-of-structures-of-arrays in tiles of 64 slots (one wave).  A tile holds, for its 64 slots, every
// field as one 256-byte run -- the f64 fields, the 3-vectors component by component, the i32 fields and the
// sample vectors -- so the ~45 coalesced loads and ~35 stores of one shading step all land in one
// contiguous ~tileStride*4-byte region (a handful of DRAM rows) instead of 80 arrays a gigabyte apart.
// Field pointers below already include the field's offset inside the tile: element (field, slot) is
// ptr[TI(tileStride, slot)]; component c of a 3-vector is 64 words further per component; the f64 arrays use TD.
// Inside a tile the 41 state words of the 64 slots are laid out in SUB-TILES of DR_SUB slots (a power of two <= 64):
// sub-tile j holds, for slots 64 t + DR_SUB j ... + DR_SUB - 1, every field as one run of DR_SUB words -- 164 * DR_SUB
// contiguous bytes per sub-tile.  DR_SUB = 64 is the plain tile (every field one 256-byte run).  A smaller sub-tile
// keeps a wave's dense accesses whole (a 128-byte line still only holds data of the wave's own slots) and lets a
// SPARSE stage -- 3 % of C2's slots are alive at bounce 5, a quarter of C5's at bounce 2 -- touch 164 * DR_SUB / 128
// lines per surviving slot instead of one line per field (~35).  The sample region keeps whole-tile runs (TI64).
#ifndef DR_SUB
#define DR_SUB 64
#endif
#ifndef DR_STATE_WORDS_K
#define DR_STATE_WORDS_K 41  // words of fixed path state per slot (42 pads a slot to 168 B: the experiment with DR_SUB = 1, one slot's words contiguous)
#endif
#define DR_SUB_WORDS (DR_STATE_WORDS_K * DR_SUB)  // words of one sub-tile
static_assert(DR_SUB >= 1 && DR_SUB <= 64 && (DR_SUB & (DR_SUB - 1)) == 0 && (DR_SUB * DR_STATE_WORDS_K) % 2 == 0,
              "DR_SUB: a power of two in 1 .. 64 with an even number of words per sub-tile (the f64 fields)");
#define TI64(ts, s) ((size_t)((s) >> 6) * (size_t)(ts) + (size_t)((s) & 63u))
#define TI(ts, s) ((size_t)((s) >> 6) * (size_t)(ts) + (size_t)(((s) & 63u) / DR_SUB) * DR_SUB_WORDS + (size_t)((s) & (DR_SUB - 1u)))
#define TD(ts, s) ((size_t)((s) >> 6) * (size_t)((ts) >> 1) + (size_t)(((s) & 63u) / DR_SUB) * (DR_SUB_WORDS / 2) + (size_t)((s) & (DR_SUB - 1u)))
// Field offsets inside a tile, in 64-word (256-byte) runs: the three f64 fields first (8-byte aligned), then the
// 3-vectors, the i32 fields and the sample region.  Every field address is ONE base pointer + a constant: the shade
// kernels touch ~25 fields, and 25 separate pointers (50 SGPRs) pushed them into SGPR spilling -- two v_readlane
// per memory access.
#if defined(DR_GROUPED) && DR_GROUPED
// The line-grouped order of the four-slot layout (namespace sp4: -DDR_SUB=4 -DDR_STATE_WORDS_K=48 -DDR_GROUPED=1).  A field's run
// is DR_SUB * 4 = 16 bytes, so eight words are one 128-byte line of the sub-tile, and the words one kernel touches together share a
// line: the ray generator writes line 0 whole, a traversal reads line 0 (and line 2 for a shadow ray), the film kernel reads line 1
// whole.  48 words per slot (768-byte sub-tiles: six whole lines; 7 words of padding).
enum {
  F_RTMIN = 0, F_RO = 2, F_RD = 5,                          // line 0: the ray (written by k_raygen / the shade kernels, read by the traversals)
  F_L = 8, F_LD1 = 11, F_FLAGS = 14, F_SHOCC = 15,          // line 1: what k_film reads
  F_HT = 16, F_SHTMAX = 18, F_SHD = 20, F_HPRIM = 23,       // line 2: hit record + shadow ray
  F_BETA = 24, F_BETANEE = 27, F_MISLIGHT = 30, F_MISPRIM = 31,  // line 3
  F_MISD = 32, F_LD2 = 35,                                  // line 4 (+ 2 words of padding)
  F_RO0 = 40,                                               // line 5 (DirectLighting with quadrics / shading records only)
  F_SAMPLES = DR_STATE_WORDS_K
};
static_assert(DR_SUB == 4 && DR_STATE_WORDS_K == 48, "the line-grouped field order is the four-slot layout's");
#else
enum {
  F_RTMIN = 0, F_HT = 2, F_SHTMAX = 4,
  F_RO = 6, F_RD = 9, F_BETA = 12, F_L = 15, F_BETANEE = 18, F_SHD = 21, F_LD1 = 24, F_MISD = 27, F_LD2 = 30,
  F_HPRIM = 33, F_SHOCC = 34, F_MISLIGHT = 35, F_MISPRIM = 36, F_FLAGS = 37,
  F_RO0 = 38,     // (DirectLighting with quadrics / shading records only: outside the window below)
  F_SAMPLES = DR_STATE_WORDS_K  // == DR_STATE_WORDS: where a tile's sample region starts, in 64-word runs
};
#endif
// The 32 runs F_RO .. F_FLAGS are exactly 8 KiB: the whole signed 13-bit immediate-offset window of a global_load /
// global_store around ONE per-lane base address (SlotRef below), so the shade kernels reach every hot field of a slot
// without any per-access address arithmetic.
#define SLOT_BIAS (DR_SUB == 64 ? F_RO * 256 + 4096 : 0)
struct BatchState {
  uint32_t cap;     // slots allocated (a multiple of 64)
  uint32_t nslots;  // slots used by the current batch
  uint32_t tileStride;  // 4-byte words per 64-slot tile = 64 * 41 + the sample region
  uint32_t idxShift;    // compact samples: 0 = u8 indices (spp <= 256), 1 = u16
  const int2* pix;  // raster pixel of each batch pixel
  float* tiles;     // the tiled state
  const double* tail;  // [cap][maxTail] host-buffer mode, else null
  // packed form of the tail (DrRenderDesc.tail_offsets): slot s owns tail[tailOff[s] - tailBase .. tailOff[s + 1] - tailBase); null: fixed form
  const unsigned long long* tailOff;
  unsigned long long tailBase;
  // The camera-sample vectors (Sample, montecarlo.dart:437-452) live in the sample region of each tile, in one of two forms:
  //  float (svFloat != 0) -- nFloats 64-word runs per tile (host-buffer sampler, multi-entry LD blocks);
  //  compact              -- the on-device LD sampler only stores what cannot be recomputed: per (LD block, slot)
  //           the Fisher-Yates-permuted sample index (one byte; a 64-byte run per block and tile) and per
  //           (LD block, batch pixel) the two scramble words; consumers evaluate VanDerCorput / Sobol2 themselves
  //           (sv_one / sv_pair in dr_kernels.hip).  26 B instead of 148 B per path sample.
  uint32_t svFloat;
  uint32_t pixCap;
  uint32_t* svScr;    // [2 * nBlocks][pixCap]
  // device sampler, compact form: the generator state of every (LD block, batch pixel) stream behind its scramble words and
  // its burn-in draws (k_gen_burnin), [nBlocks][pixCap] (lo, hi); null: the shuffle kernels seed and burn in themselves
  uint2* genState;
  // Lazy sample generation (round 5): the LD blocks of bounce b are only shuffled for the 64-pixel groups that still have a path
  // alive at bounce b.  genAlive: one byte per group of 64 batch pixels, read by the shuffle kernels (null: every group);
  // markAlive: where k_trace_pk marks the groups of the camera rays that hit something (null: nothing is marked);
  // markShift = sppShift + 6: slot -> group.
  const uint8_t* genAlive;
  uint8_t* markAlive;
  uint32_t markShift, padMark;
  // DirectLighting over mirror / glass (Integrator.SpecularReflect / SpecularTransmit, integrator.dart:187-290): the
  // per-slot stack of suspended vertices, [level][cap] SpecFrame records, and its depth per slot; null otherwise
  float* specFrames;
  int32_t* specSp;
  // (device-only: their bodies depend on the layout this translation unit was compiled for -- as host functions the two
  // layouts' copies would be ONE weak symbol and the linker would keep either)
#define DR_FIELD(T, name, F) \
  DR_DEV T* name() const { return (T*)(tiles + ((F) >= F_SAMPLES ? 64 : DR_SUB) * (F)); }
  DR_FIELD(double, rtmin, F_RTMIN)      // Ray.minDistance (isect.rayEpsilon after the first vertex)
  DR_FIELD(double, ht, F_HT)            // closest-hit parameter of the camera / continuation ray
  DR_FIELD(double, shTmax, F_SHTMAX)
  DR_FIELD(float, ro, F_RO)             // ray origin (vertex position p once a vertex has been shaded)
  DR_FIELD(float, ro0, F_RO0)           // DirectLighting with quadrics: the camera ray's origin
  DR_FIELD(float, rd, F_RD)             // continuation / camera ray direction
  DR_FIELD(float, beta, F_BETA)         // pathThroughput (path) / running all-lights sum (direct)
  DR_FIELD(float, L, F_L)               // radiance of the sample
  DR_FIELD(float, betaNee, F_BETANEE)   // pathThroughput at the vertex whose NEE is pending
  DR_FIELD(float, shD, F_SHD)           // shadow-ray direction
  DR_FIELD(float, Ld1, F_LD1)           // light-sampling contribution if unoccluded
  DR_FIELD(float, misD, F_MISD)         // BSDF-sampled direction of the MIS half
  DR_FIELD(float, Ld2, F_LD2)           // its contribution if it reaches the sampled light's front face
  DR_FIELD(int32_t, hprim, F_HPRIM)     // closest-hit result of the camera / continuation ray
  DR_FIELD(int32_t, shOcc, F_SHOCC)     // any-hit result
  DR_FIELD(int32_t, misLight, F_MISLIGHT)
  DR_FIELD(int32_t, misPrim, F_MISPRIM) // closest-hit result of the MIS ray
  DR_FIELD(uint32_t, flags, F_FLAGS)
  DR_FIELD(float, sv, F_SAMPLES)        // float form
  DR_FIELD(uint8_t, svIdx, F_SAMPLES)   // compact form: the index runs
#undef DR_FIELD
};

struct TraceCounters {  // device-side totals (Stats probes of bvh_accel.dart:106-163)
  unsigned long long closest_rays, any_rays, closest_nodes, any_nodes, closest_tris, any_tris;
  // shading stages: active-list entries processed, path vertices set up, rays queued (DrRenderStats.shade_*)
  unsigned long long shade_items, shade_vertices, shade_cont, shade_mis, shade_shadow;
  // the part of the closest_* / any_* totals above that k_trace_pk (coherent waves: the camera rays) traced, [0] closest [1] any hit
  unsigned long long pk_rays[2], pk_nodes[2], pk_tris[2];
  // lazy sample generation: (pixel, LD block) pairs the shuffle kernels generated for bounces 0..2 (k_sum_alive, once per batch)
  unsigned long long gen_pixel_blocks;
};

// A slot's place in the tiled state: `a` addresses its element of the 4-byte fields (biased by SLOT_BIAS so that
// field F sits at the compile-time offset F * 256 - SLOT_BIAS, within +-4 KiB for F_RO .. F_FLAGS), `b` its element
// of the 8-byte fields (F_RTMIN, F_HT, F_SHTMAX at b + F * 256).  Round 1 indexed every field array separately,
// `field()[TI(stride, slot)]`: ~40 scalar base pointers (spilled to VGPR lanes: two v_readlane per access) and a
// 64-bit shift-add per access -- about a sixth of the shade kernel's VALU instructions.
struct SlotRef {
  char* a;
  char* b;
  DR_DEV static SlotRef of(const BatchState& st, uint32_t slot) {
    char* tile = (char*)st.tiles + (size_t)(slot >> 6) * ((size_t)st.tileStride * 4) + ((slot & 63u) / DR_SUB) * (DR_SUB_WORDS * 4);
    SlotRef r;
    r.a = tile + (slot & (DR_SUB - 1u)) * 4 + SLOT_BIAS;
    r.b = tile + (slot & (DR_SUB - 1u)) * 8;
    return r;
  }
  template <int F> DR_DEV float& f32(int comp = 0) const { return *(float*)(a + ((F + comp) * (DR_SUB * 4) - SLOT_BIAS)); }
  template <int F> DR_DEV int32_t& i32() const { return *(int32_t*)(a + (F * (DR_SUB * 4) - SLOT_BIAS)); }
  template <int F> DR_DEV uint32_t& u32() const { return *(uint32_t*)(a + (F * (DR_SUB * 4) - SLOT_BIAS)); }
  template <int F> DR_DEV double& f64() const { return *(double*)(b + F * (DR_SUB * 4)); }
};
template <int F> DR_DEV F3 ld3f(const SlotRef& r) { return F3{LDS_STREAM(&r.f32<F>(0)), LDS_STREAM(&r.f32<F>(1)), LDS_STREAM(&r.f32<F>(2))}; }
template <int F> DR_DEV C3 ldcf(const SlotRef& r) { return C3{LDS_STREAM(&r.f32<F>(0)), LDS_STREAM(&r.f32<F>(1)), LDS_STREAM(&r.f32<F>(2))}; }
template <int F> DR_DEV void st3f(const SlotRef& r, F3 v) {
  STS_STREAM(&r.f32<F>(0), v.x);
  STS_STREAM(&r.f32<F>(1), v.y);
  STS_STREAM(&r.f32<F>(2), v.z);
}
template <int F> DR_DEV void stcf(const SlotRef& r, C3 v) {
  STS_STREAM(&r.f32<F>(0), v.r);
  STS_STREAM(&r.f32<F>(1), v.g);
  STS_STREAM(&r.f32<F>(2), v.b);
}

// A vertex of DirectLightingIntegrator.Li that waits for the radiance of a specular child ray (the recursion of
// direct_lighting_integrator.dart:59-65 unrolled into an explicit stack, one frame per suspended vertex).
struct SpecFrame {   // 96 bytes
  float L[3];        // Le + UniformSampleAllLights (+ the reflected radiance once it has come back)
  float ft[3];       // transmission: f of the sampled lobe,
  float wt[3];       //   its direction,
  float p[3];        //   the vertex (origin of the child rays)
  float fr[3];       // reflection: f of the sampled lobe (only while its child is being traced)
  uint32_t state;    // 1 = the reflected ray is being traced, 2 = the transmitted ray; bit 8: a transmitted ray follows
  double sr, st;     // AbsDot(wi, n) / pdf of the two lobes
  double eps;        // isect.rayEpsilon
  float pad[2];
};
#define DR_SPEC_FRAME_WORDS 24
static_assert(sizeof(SpecFrame) == 4 * DR_SPEC_FRAME_WORDS, "SpecFrame layout");

// Work queues of one stage.  Counts live in device memory so that no host
// round trip is needed between launches.
struct StageQueues {
  const uint32_t* activeIn;   // slots shaded in the previous stage (null => identity)
  const uint32_t* nActiveIn;  // device count (null => nslots)
  uint32_t* activeOut;
  uint32_t* nActiveOut;
  uint32_t* closestQ;  // slot | Q_MIS_BIT
  uint32_t* nClosest;
  uint32_t* anyQ;
  uint32_t* nAny;
  TraceCounters* ctr;  // shade_* totals (one no-return atomic per counter, workgroup and flush)
  uint32_t* work;      // k_shade_path's chunk counter (zero at launch): waves take chunks of 64 * DR_PUSH_ITERS entries
  // plain-triangle scenes under an environment map: the slots whose light estimate picked the infinite light and the
  // escaped camera rays (| Q_ENV_MISS_BIT), written by k_shade_path, worked off by k_env before the stage's traversals
  uint32_t* envQ;
  uint32_t* nEnv;
};

// k_trace's per-XCD work counters of one launch sit this many words apart (same-line atomics serialise)
#ifndef DR_WORK_STRIDE
#define DR_WORK_STRIDE 64
#endif
#ifndef DR_V2_LDS_STACK
#define DR_V2_LDS_STACK 16    // k_trace: 16 KiB of stack + 6 KiB of cold ray state per workgroup => 7 workgroups (28 waves) per CU
#endif
#ifndef DR_V2_WG_PER_CU
#define DR_V2_WG_PER_CU 7
#endif
int traceGridFor(int wgPerCU);  // dr_api.hip: workgroups of a persistent traversal launch
#include "dr_options.h"  // dr_opt(name): a tuning / diagnostic switch of the library (DARTRAY_*), by value
#define DR_MAX_STACK 128      // LDS + spill entries per lane (the reference uses 64, bvh_accel.dart:120)
#define DR_TRACE_BLOCK 256

// ---- launchers (dr_kernels.hip, dr_trace.hip): one set per state layout ----
#include "dr_launchers.inc"
namespace sp4 {
#include "dr_launchers.inc"
}

#endif
