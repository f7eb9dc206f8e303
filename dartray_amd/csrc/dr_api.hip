// dr_api.hip -- the C ABI of include/dartray_hip.h: scene upload, batch
// driver of the wavefront path tracer, film, statistics.
//
// Host logic restated from the reference where it decides WHAT is traced:
//   sampler window       ImageFilm.getSampleExtent (film/image_film.dart:247-252),
//                        GetSubWindow (core/common.dart:52-73), dartray.dart:1009-1023
//   pixel order          LinearPixelSampler (pixel_samplers/linear_pixel_sampler.dart:29-40)
//   light tables         ShapeSet ctor (core/light/shape_set.dart:24-51), Distribution1D (core/montecarlo.dart:25-52)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "dr_kernels.h"
#include "dr_scene_prep.h"
static_assert(DR_PREP_MAX_STACK == DR_MAX_STACK, "dr_scene_prep.h");

namespace {

thread_local std::string g_err;
int g_device = -1;
int g_numCU = 256;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(DR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    if (count <= n && p) return hipSuccess;
    release();
    hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  size_t bytes() const { return p ? std::max<size_t>(n, 1) * sizeof(T) : 0; }
  ~DevBuf() { release(); }
};

struct Workspace {
  uint32_t cap = 0;
  int stateWords = 0;  // words per slot the tiles were sized for
  int svWords = 0, maxTail = 0;  // svWords: 4-byte words of the sample region of one tile
  uint32_t pixCap = 0;
  DevBuf<float> tiles;  // the tiled path state (see BatchState in dr_kernels.h): cap/64 tiles of 64*41+svWords words
  DevBuf<uint32_t> scr;  // compact samples: scramble words [2 * nBlocks][pixCap]
  DevBuf<uint2> genState;  //   and the streams' generator states behind their burn-in draws [nBlocks][pixCap] (k_gen_burnin)
  DevBuf<double> tail;
  DevBuf<unsigned long long> tailOff;  // packed tail (DrRenderDesc.tail_offsets): the batch's nslots + 1 offsets
  DevBuf<uint32_t> activeA, activeB, closestQ, anyQ, counters, spill;
  DevBuf<uint32_t> envQ;  // plain-triangle scenes under an environment map: k_env's list of a stage (cap entries)
  DevBuf<uint8_t> alive;  // lazy sample generation: [3][groups of 64 batch pixels] = a path of the group is alive at bounce 0 / 1 / 2
  size_t spillHalf = 0;
  DevBuf<uint32_t> roundA, roundB;  // DirectLighting over mirror / glass: the slots whose child ray is traced next round
  DevBuf<float> specFrames;         //   [maxDepth][cap] SpecFrame
  DevBuf<int32_t> specSp;           //   [cap]
  DevBuf<int2> pix;
  DevBuf<float> filterTable, aosSamples;
  int spillGrid = 0;
};

#define DR_PAIR_TOP_LEVELS 12  // sibling-pair records: this many levels of the tree breadth-first in front (dr_scene_create)
#define DR_STATE_WORDS F_SAMPLES  // 41 4-byte words of fixed path state per slot: 3 f64 + 10 3-vectors + 5 i32
#define N_COUNTERS_TRACE (1024 + 8 * DR_WORK_STRIDE * 400)
#define N_COUNTERS (N_COUNTERS_TRACE + 64 * 256)  // ... then the counts of k_env's lists, one cache line per stage  // [0,1024): stage queue counts; then 8 per-XCD work counters per trace launch, DR_WORK_STRIDE words apart

}  // namespace

int dr_fail(int code, const std::string& msg) { return fail(code, msg); }  // for dr_comm.cpp

// ---- tuning / diagnostic switches: dr_set_option, else the environment ----
namespace {
std::mutex g_optMutex;
std::map<std::string, std::string> g_options;  // name -> value; an empty value = "unset for this process" (hides the environment's)
const char* const kOptionNames[] = {
    // what a render runs: traversal kernels, state layout, batch size, schedule, the pilot
    "DARTRAY_TRACE_IMPL", "DARTRAY_TRACE_WG_PER_CU", "DARTRAY_STATE_LAYOUT", "DARTRAY_BATCH_BITS", "DARTRAY_OVERLAP_ANY", "DARTRAY_PILOT",
    "DARTRAY_COHERENT_CAMERA",
    // the device sampler (bit-exact variants, tests)
    "DARTRAY_LAZY_GEN", "DARTRAY_GEN_ALL_BLOCKS", "DARTRAY_GEN_SLOW_DRAWS",
    // scene set-up, the collective's library, diagnostics
    "DARTRAY_SCENE_PREP", "DARTRAY_BUILD_THREADS", "DARTRAY_RCCL_LIB", "DARTRAY_STAGE_COUNTS", "DARTRAY_VERBOSE"};
static_assert(sizeof(kOptionNames) / sizeof(kOptionNames[0]) <= 15, "a new switch replaces one: measured negatives go to experiments/");

}  // namespace

DrOpt dr_opt(const char* name) {
  DrOpt o;
  {
    std::lock_guard<std::mutex> lock(g_optMutex);
    auto it = g_options.find(name);
    if (it != g_options.end()) {
      o.set = !it->second.empty();  // "" hides the environment's value
      o.value = it->second;
      return o;
    }
  }
  if (const char* e = getenv(name)) {
    o.set = true;
    o.value = e;
  }
  return o;
}

struct DrScene {
  DScene d;
  DevBuf<uint4> nodes, pairs;
  DevBuf<float4> tris, mats, shtris;
  DevBuf<DLight> lights;
  DevBuf<DLightTri> ltris;
  DevBuf<DQuadric> quads;
  std::vector<DQuadric> hostQuads;
  DevBuf<float4> srec;
  DevBuf<float> xforms;
  DevBuf<float> lcdf;
  DevBuf<float> envTexels, envCondFunc, envCondCdf, envCondInt, envMargFunc, envMargCdf;
  DevBuf<uint16_t> envCondGuide;
  DevBuf<TraceCounters> ctr;
  uint32_t bvhDepth = 0;
  bool traceCalibrated = false;
  uint32_t bigRenders = 0;      // big renders this scene has finished (planBatches: the first one keeps its batches at 2^27 slots)
  int stateLayout = 0;          // path-state layout of this scene's path renders: 0 = not measured yet, 64 / 4 (LayoutOps)
  float layoutDensity = -1.f;   //   what decided it: the share of a pilot batch's slots still alive at the second bounce
  float calibMs[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // pilot of dr_render_device: [closest / any][v2 / v3 / v3c] ms
  float calibFarFirst = 0.f;  // any-hit rays, far child first over the reference order: time per ray of k_trace<1> in the pilot's first two batches (0 = not measured)
  float calibPerGB[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  //   the same as ms per algorithmic GB (what the choice compares; 0 = not measured)
  // what the last dr_render_device call actually ran with (dr_scene_last_render_info): state layout, the traversal kernels of
  // its last batch, a reserved word (-1), calibration batches, workgroups per CU
  int32_t lastInfo[8] = {0, 0, 0, -1, 0, 0, 0, 0};
  std::vector<int32_t> lightNSamples;
  bool hasSpecular = false;  // some material is a mirror / glass
  bool hasDeltaLight = false;
  // DirectLighting sample layout (direct_lighting_integrator.dart:70-87), fixed by the lights' nsamples
  DevBuf<LdBlock> dlBlocks;
  DevBuf<DirectStage> dlStages;
  int dlNBlocks = 0, dlNStages = 0, dlNFloats = 0, dlN1D = 0;
  bool dlMulti = false;
  Workspace ws;
  hipStream_t s3 = nullptr;  // the any-hit launches of a stage, beside the closest-hit ones
  // stats of the last render
  DrRenderStats stats;
  // any: 0 closest, 1 any-hit, 2 shade, 3 sample gen + raygen, 4 film.  after: set for an any-hit launch that ran beside
  // the stage's closest-hit launch (its end): only the time AFTER that counts as any-hit time
  struct TraceEv { hipEvent_t e0, e1; int any; hipEvent_t after = nullptr; };
  std::vector<TraceEv> traceEvents;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> renderEvents;
  std::vector<hipEvent_t> eventPool;
  size_t eventsUsed = 0;
  bool statsPending = false;
  hipEvent_t lastEvent = nullptr;
  // dr_scene_get_coherent_stats: what k_trace_pk traced of the closest-hit totals since the last dr_reset_stats
  double pkMs = 0.0;
  uint64_t pkLaunches = 0;
  unsigned long long pkRays = 0, pkNodes = 0, pkTris = 0;
  // dr_scene_get_sampler_stats: (pixel, LD block) pairs the device sampler shuffled / that the path's reads name (rp.genMask), since the
  // last dr_reset_stats (lazy generation: the first is smaller where paths end early -- sky pixels)
  unsigned long long genDone = 0, genDoneHost = 0, genNamed = 0;
  // Timings of finished launches are folded into `stats` and their events recycled, so a long-lived scene (a frame
  // loop calling dr_render_device) does not grow the pool or the lists without bound.
  void foldEvents() {
    for (auto& ev : traceEvents) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, ev.after ? ev.after : ev.e0, ev.e1) != hipSuccess) continue;
      if (t < 0.f) t = 0.f;  // (an any-hit launch that ended before the closest-hit one beside it)
      if (ev.any == 1) { stats.any_ms += t; stats.any_launches++; }
      else if (ev.any == 0) { stats.closest_ms += t; stats.closest_launches++; }
      else if (ev.any == 6) { stats.closest_ms += t; stats.closest_launches++; pkMs += t; pkLaunches++; }  // k_trace_pk: part of the closest-hit time
      else if (ev.any == 2) stats.shade_ms += t;
      else if (ev.any == 3) stats.gen_ms += t;
      else if (ev.any == 5) stats.pilot_ms += t;
      else stats.film_ms += t;
    }
    for (auto& ev : renderEvents) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess) stats.total_ms += t;
    }
    stats.trace_ms = stats.closest_ms + stats.any_ms;
    stats.trace_launches = stats.closest_launches + stats.any_launches;
    traceEvents.clear();
    renderEvents.clear();
    eventsUsed = 0;
  }
  hipEvent_t getEvent() {
    if (eventsUsed == eventPool.size()) {
      hipEvent_t e;
      (void)hipEventCreate(&e);
      eventPool.push_back(e);
    }
    return eventPool[eventsUsed++];
  }
  ~DrScene() {
    (void)hipDeviceSynchronize();  // nothing of this scene may still be in flight when its buffers and events go away
    for (auto e : eventPool) (void)hipEventDestroy(e);
    if (s3) (void)hipStreamDestroy(s3);
  }
};

namespace {

inline double r32(double x) { return (double)(float)x; }

// Triangle.area (shapes/triangle.dart:265-269): Vector temporaries are f32.
double host_tri_area(const float* a, const float* b, const float* c) {
  double e1[3], e2[3];
  for (int k = 0; k < 3; ++k) {
    e1[k] = r32((double)b[k] - (double)a[k]);
    e2[k] = r32((double)c[k] - (double)a[k]);
  }
  double cx = r32(e1[1] * e2[2] - e1[2] * e2[1]);
  double cy = r32(e1[2] * e2[0] - e1[0] * e2[2]);
  double cz = r32(e1[0] * e2[1] - e1[1] * e2[0]);
  return 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
}

// DifferentialGeometry.nn of a hit on triangle (a,b,c) with the default UVs (triangle.dart:100-132,
// differential_geometry.dart:84-99) and the normal Triangle.sample returns (triangle.dart:376-381);
// the same f64-expression / f32-store arithmetic as tri_dg() in dr_device.h.
void host_tri_normals(const float* a, const float* b, const float* c, bool reverse, float nn[3], float ns[3],
                      const float* uv = nullptr) {
  static const float kDefaultUV[6] = {0.f, 0.f, 1.f, 0.f, 1.f, 1.f};  // triangle.dart:255-262
  if (!uv) uv = kDefaultUV;
  const double du1 = (double)uv[0] - (double)uv[4], du2 = (double)uv[2] - (double)uv[4];
  const double dv1 = (double)uv[1] - (double)uv[5], dv2 = (double)uv[3] - (double)uv[5];
  const double determinant = du1 * dv2 - dv1 * du2;
  double dpdu[3], dpdv[3];
  if (determinant == 0.0) {  // degenerate uv mapping: Vector.CoordinateSystem on the face normal (triangle.dart:108-127)
    double e1[3], e2[3];
    for (int k = 0; k < 3; ++k) {
      e1[k] = (double)b[k] - (double)a[k];
      e2[k] = (double)c[k] - (double)a[k];
    }
    const double e3x = (e2[1] * e1[2]) - (e2[2] * e1[1]), e3y = (e2[2] * e1[0]) - (e2[0] * e1[2]), e3z = (e2[0] * e1[1]) - (e2[1] * e1[0]);
    const double len = std::sqrt(e3x * e3x + e3y * e3y + e3z * e3z);
    const double v1[3] = {r32(e3x / len), r32(e3y / len), r32(e3z / len)};
    if (std::fabs(v1[0]) > std::fabs(v1[1])) {
      const double invLen = 1.0 / std::sqrt(v1[0] * v1[0] + v1[2] * v1[2]);
      dpdu[0] = r32(-v1[2] * invLen); dpdu[1] = 0.0; dpdu[2] = r32(v1[0] * invLen);
    } else {
      const double invLen = 1.0 / std::sqrt(v1[1] * v1[1] + v1[2] * v1[2]);
      dpdu[0] = 0.0; dpdu[1] = r32(v1[2] * invLen); dpdu[2] = r32(-v1[1] * invLen);
    }
    dpdv[0] = r32(v1[1] * dpdu[2] - v1[2] * dpdu[1]);
    dpdv[1] = r32(v1[2] * dpdu[0] - v1[0] * dpdu[2]);
    dpdv[2] = r32(v1[0] * dpdu[1] - v1[1] * dpdu[0]);
  } else {
    const double invdet = 1.0 / determinant;
    for (int k = 0; k < 3; ++k) {
      const double dp1 = r32((double)a[k] - (double)c[k]), dp2 = r32((double)b[k] - (double)c[k]);
      dpdu[k] = r32(r32(r32(dp1 * dv2) - r32(dp2 * dv1)) * invdet);
      dpdv[k] = r32(r32(r32(dp1 * -du2) + r32(dp2 * du1)) * invdet);
    }
  }
  auto crossNorm = [](const double* u, const double* v, double out[3]) {
    const double cx = r32(u[1] * v[2] - u[2] * v[1]), cy = r32(u[2] * v[0] - u[0] * v[2]), cz = r32(u[0] * v[1] - u[1] * v[0]);
    const double len = std::sqrt(cx * cx + cy * cy + cz * cz);
    out[0] = r32(cx / len); out[1] = r32(cy / len); out[2] = r32(cz / len);
  };
  double n[3];
  crossNorm(dpdu, dpdv, n);
  for (int k = 0; k < 3; ++k) nn[k] = (float)(reverse ? r32(n[k] * -1.0) : n[k]);
  double e1[3], e2[3];
  for (int k = 0; k < 3; ++k) {
    e1[k] = r32((double)b[k] - (double)a[k]);
    e2[k] = r32((double)c[k] - (double)a[k]);
  }
  crossNorm(e1, e2, n);
  for (int k = 0; k < 3; ++k) ns[k] = (float)(reverse ? n[k] * -1.0 : n[k]);
}

// MIPMap.texture's resampling of an RGB image to power-of-two resolution (mipmap.dart:71-138; wrap mode TEXTURE_REPEAT, the
// InfiniteAreaLight's): a four-tap Lanczos zoom in s, then in t, every product and every partial sum a new Spectrum (f32 stores),
// the t pass clamped to [0, inf) (_clamp, :358).  Weights: _resampleWeights (:360-384) in doubles.
void resample_to_pow2(const float* img, int xres, int yres, std::vector<float>& out, int* wOut, int* hOut) {
  auto roundUpPow2 = [](int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; };  // common.dart:105-113
  struct Weight { int firstTexel; double w[4]; };
  auto lanczos = [](double x) {  // texture.dart:27-39, tau = 2
    x = std::fabs(x);
    if (x < 1.0e-5) return 1.0;
    if (x > 1.0) return 0.0;
    x *= 3.141592653589793;
    const double s = std::sin(x * 2.0) / (x * 2.0);
    return s * (std::sin(x) / x);
  };
  auto weights = [&](int oldres, int newres) {
    std::vector<Weight> wt(newres);
    const double filterwidth = 2.0;
    for (int i = 0; i < newres; ++i) {
      const double center = (i + 0.5) * oldres / newres;
      wt[i].firstTexel = (int)std::floor((center - filterwidth) + 0.5);
      for (int j = 0; j < 4; ++j) wt[i].w[j] = lanczos(((wt[i].firstTexel + j + 0.5) - center) / filterwidth);
      const double invSum = 1.0 / (wt[i].w[0] + wt[i].w[1] + wt[i].w[2] + wt[i].w[3]);
      for (int j = 0; j < 4; ++j) wt[i].w[j] *= invSum;
    }
    return wt;
  };
  auto mod = [](int a, int n) { const int r = a % n; return r < 0 ? r + n : r; };  // Dart's % is never negative for a positive divisor
  const int sPow2 = roundUpPow2(xres), tPow2 = roundUpPow2(yres);
  out.assign(3 * (size_t)sPow2 * tPow2, 0.f);
  const std::vector<Weight> sW = weights(xres, sPow2);
  for (int t = 0; t < yres; ++t)
    for (int s = 0; s < sPow2; ++s)
      for (int j = 0; j < 4; ++j) {
        const int origS = mod(sW[s].firstTexel + j, xres);
        for (int c = 0; c < 3; ++c) {
          float& dst = out[3 * ((size_t)t * sPow2 + s) + c];
          const float px = (float)((double)img[3 * ((size_t)t * xres + origS) + c] * sW[s].w[j]);
          dst = (float)((double)dst + (double)px);
        }
      }
  const std::vector<Weight> tW = weights(yres, tPow2);
  std::vector<float> work(3 * (size_t)tPow2);
  for (int s = 0; s < sPow2; ++s) {
    for (int t = 0; t < tPow2; ++t)
      for (int c = 0; c < 3; ++c) {
        float acc = 0.f;
        for (int j = 0; j < 4; ++j) {
          const int off = mod(tW[t].firstTexel + j, yres);
          const float px = (float)((double)out[3 * ((size_t)off * sPow2 + s) + c] * tW[t].w[j]);
          acc = (float)((double)acc + (double)px);
        }
        work[3 * (size_t)t + c] = acc;
      }
    for (int t = 0; t < tPow2; ++t)
      for (int c = 0; c < 3; ++c) {
        const float v = work[3 * (size_t)t + c];
        out[3 * ((size_t)t * sPow2 + s) + c] = (v < 0.f || v == 0.f) ? 0.f : v;  // num.clamp(0.0, INFINITY): NaN stays, -0.0 -> 0.0
      }
  }
  *wOut = sPow2;
  *hOut = tPow2;
}

int traceGrid() { return traceGridFor(DR_V2_WG_PER_CU); }  // the largest grid any variant launches (sizes the spill stacks)

int ensureSpill(DrScene* sc, Workspace& w, int grid) {
  // deepest stack == tree depth; the v3 kernel keeps 16 (reference, E) pairs in LDS, v2 24 references
  if (sc->bvhDepth != 0 && sc->bvhDepth <= 8) return DR_OK;
  size_t need = (size_t)grid * DR_TRACE_BLOCK * (DR_MAX_STACK - 8) * 2;  // (room for any LDS stack depth >= 8 of either variant)
  w.spillHalf = need;
  HIP_TRY(w.spill.alloc(2 * need));  // second half: the any-hit launch of a stage when it runs beside the closest-hit one
  w.spillGrid = grid;
  return DR_OK;
}

void rp_film(RenderParams& rp, const DrFilm& f) {
  rp.xres = f.xres;
  rp.yres = f.yres;
  // image_film.dart:61-65
  rp.left = (int)std::ceil(f.xres * (double)f.crop[0]);
  rp.width = std::max(1, (int)std::ceil(f.xres * (double)f.crop[1]) - rp.left);
  rp.top = (int)std::ceil(f.yres * (double)f.crop[2]);
  rp.height = std::max(1, (int)std::ceil(f.yres * (double)f.crop[3]) - rp.top);
  rp.fxw = f.filter_xw;
  rp.fyw = f.filter_yw;
  rp.invX = 1.0 / (double)f.filter_xw;  // filter.dart:33-37
  rp.invY = 1.0 / (double)f.filter_yw;
  // ImageFilm.getSampleExtent (image_film.dart:247-252)
  int e0 = (int)std::floor(rp.left + 0.5 - rp.fxw);
  int e1 = (int)std::ceil(rp.left + 0.5 + rp.width + rp.fxw);
  int e2 = (int)std::floor(rp.top + 0.5 - rp.fyw);
  int e3 = (int)std::ceil(rp.top + 0.5 + rp.height + rp.fyw);
  rp.extX0 = e0;
  rp.extY0 = e2;
  rp.extW = e1 - e0;
  rp.extH = e3 - e2;
}

// GetSubWindow (core/common.dart:52-73)
void getSubWindow(int w, int h, int num, int count, int ext[4]) {
  int nx = count, ny = 1;
  while ((nx & 0x1) == 0 && 2 * w * ny < h * nx) {
    nx >>= 1;
    ny <<= 1;
  }
  int xo = num % nx, yo = num / nx;
  double tx0 = (double)xo / nx, tx1 = (double)(xo + 1) / nx;
  double ty0 = (double)yo / ny, ty1 = (double)(yo + 1) / ny;
  auto lerp = [](double t, double v1, double v2) { return v1 * (1.0 - t) + v2 * t; };
  ext[0] = (int)std::floor(lerp(tx0, 0, w));
  ext[1] = std::min((int)std::floor(lerp(tx1, 0, w)), w);
  ext[2] = (int)std::floor(lerp(ty0, 0, h));
  ext[3] = std::min((int)std::floor(lerp(ty1, 0, h)), h);
}

// How the sample vectors of one render are stored (see BatchState)
struct SampleForm {
  bool compact;
  int nFloats, nBlocks, idxShift;
  int svWords() const { return compact ? ((nBlocks * 64) << idxShift) / 4 : 64 * nFloats; }
};

int allocWorkspace(DrScene* sc, Workspace& w, uint32_t cap, const SampleForm& sf, uint32_t pixCap, int maxTail, bool needTail, int stateWords) {
  cap = (cap + 63u) & ~63u;  // whole tiles
  if (cap > w.cap || sf.svWords() > w.svWords || stateWords > w.stateWords) {
    uint32_t c = std::max(cap, w.cap);
    int sw = std::max(sf.svWords(), w.svWords);
    w.stateWords = std::max(stateWords, w.stateWords);
    HIP_TRY(w.tiles.alloc((size_t)(c / 64) * (64 * (size_t)w.stateWords + (size_t)sw)));
    HIP_TRY(w.activeA.alloc(c));
    HIP_TRY(w.activeB.alloc(c));
    HIP_TRY(w.closestQ.alloc(2 * (size_t)c));
    HIP_TRY(w.anyQ.alloc(c));
    w.cap = c;
    w.svWords = sw;
  }
  if (sf.compact) {
    w.pixCap = std::max(w.pixCap, pixCap);
    HIP_TRY(w.scr.alloc(2 * (size_t)sf.nBlocks * w.pixCap));
    HIP_TRY(w.genState.alloc((size_t)sf.nBlocks * w.pixCap));
  }
  // (sized for THIS render's batches, not for the largest batch the workspace has ever held: a small replay after a big
  // counter-mode render would otherwise allocate cap x maxTail doubles -- 86 GB behind a C2 batch)
  if (needTail && ((size_t)cap * maxTail > w.tail.n)) HIP_TRY(w.tail.alloc((size_t)cap * maxTail));
  w.maxTail = maxTail;
  HIP_TRY(w.counters.alloc(N_COUNTERS));
  HIP_TRY(w.filterTable.alloc(256));
  return DR_OK;
}

BatchState makeState(Workspace& w, const SampleForm& sf, const int2* pix, uint32_t nslots, bool useTail, int stateWords) {
  BatchState st;
  st.cap = w.cap;
  st.nslots = nslots;
  st.tileStride = 64u * (uint32_t)stateWords + (uint32_t)w.svWords;  // (the layout's own words per slot: its sample region starts behind them)
  st.idxShift = (uint32_t)sf.idxShift;
  st.pix = pix;
  st.tail = useTail ? w.tail.p : nullptr;
  st.tailOff = nullptr;  // (the packed form: set per batch by dr_render_device)
  st.tailBase = 0ull;
  st.tiles = w.tiles.p;  // field offsets inside a tile: the F_* constants of dr_kernels.h
  st.svFloat = sf.compact ? 0u : 1u;
  st.svScr = sf.compact ? w.scr.p : nullptr;
  {  // (the pre-pass k_gen_burnin fills genState above 256 spp; at and below, the shuffle kernels seed and burn in their streams themselves)
    st.genState = sf.compact ? w.genState.p : nullptr;
    st.genAlive = nullptr;
    st.markAlive = nullptr;
    st.markShift = 0;
    st.padMark = 0;
  }
  st.pixCap = w.pixCap;
  st.specFrames = w.specFrames.p;
  st.specSp = w.specSp.p;
  return st;
}

// Raster pixels one call traces in counter mode, in trace order.  task_*: the reference's
// GetSubWindow rectangle; tile_*: tile_size^2 tiles dealt round-robin over ranks.  The dead
// border row/column of the sampler window (W+1 x H+1 for the box filter, image_film.dart:247-252)
// is traced as the reference does; a border sample only reaches the film when imageX/Y is integral.
void enumeratePixels(const RenderParams& rp, const DrRenderDesc* rd, std::vector<int2>& pixels) {
  // NB the reference hands GetSubWindow's extents to the sampler as they are (dartray.dart:1009-1022): they are
  // computed from 0 and ignore the sample extent's origin (common.dart:69-72), so a cropped film -- or a filter
  // wider than half a pixel, whose extent starts at -1 -- samples the window [0, w) x [0, h) instead of
  // [x0, x0 + w) x [y0, y0 + h).  Reproduced: the window is an input of the path.
  int ext[4];
  getSubWindow(rp.extW, rp.extH, rd->task_num, std::max(1, rd->task_count), ext);
  const int ts = rd->tile_size > 0 ? rd->tile_size : 32;
  const int ntx = (rp.extW + ts - 1) / ts;
  const bool tiled = rd->tile_count > 1;
  pixels.clear();
  pixels.reserve((size_t)(ext[1] - ext[0]) * (ext[3] - ext[2]) / (tiled ? rd->tile_count : 1) + 1024);
  if (!tiled) {  // LinearPixelSampler order (linear_pixel_sampler.dart:29-40)
    for (int y = ext[2]; y < ext[3]; ++y)
      for (int x = ext[0]; x < ext[1]; ++x) pixels.push_back(make_int2(x, y));
  } else {  // tile-major so that a batch covers whole tiles (coherent camera rays)
    const int nty = (rp.extH + ts - 1) / ts;
    for (int ty = 0; ty < nty; ++ty)
      for (int tx = 0; tx < ntx; ++tx) {
        if ((ty * ntx + tx) % rd->tile_count != rd->tile_rank) continue;
        for (int y = std::max(ty * ts, ext[2]); y < std::min((ty + 1) * ts, ext[3]); ++y)
          for (int x = std::max(tx * ts, ext[0]); x < std::min((tx + 1) * ts, ext[1]); ++x)
            pixels.push_back(make_int2(x, y));
      }
  }
}

}  // namespace

// The kernels that read or write the path state exist twice: the default layout (every field of a tile's 64 slots one
// 256-byte run) and sp4 (sub-tiles of four slots: a slot's 41 words within 656 contiguous bytes; the same sources compiled
// with -DDR_SUB=4 -DDR_NS=sp4).  Dense stage lists are faster in the first; lists that thin out early -- open scenes under an
// environment map, where most bounce rays leave -- in the second (C5: shade 711 -> 536 ms, MEASUREMENTS.md round 3).
// A render picks one (dr_render_device); results do not depend on it.
struct LayoutOps {
  decltype(&launch_trace) trace;
  decltype(&launch_trace_coherent) trace_coherent;
  decltype(&trace_kernel_id) trace_kernel_id;
  decltype(&launch_gen_samples) gen_samples;
  decltype(&launch_mark_alive) mark_alive;
  decltype(&launch_sum_alive) sum_alive;
  decltype(&launch_transpose_samples) transpose_samples;
  decltype(&launch_raygen) raygen;
  decltype(&launch_shade_path) shade_path;
  decltype(&launch_env) env;
  decltype(&launch_shade_direct) shade_direct;
  decltype(&launch_shade_spec) shade_spec;
  decltype(&launch_film) film;
  int stateWords;  // 4-byte words of fixed path state per slot in this layout (a tile is 64 of them + the sample region):
                   // what the kernels' own translation unit was compiled with (layout_state_words), not a constant repeated here
};
static const LayoutOps kLayout64 = {&launch_trace, &launch_trace_coherent, &trace_kernel_id, &launch_gen_samples, &launch_mark_alive, &launch_sum_alive, &launch_transpose_samples, &launch_raygen, &launch_shade_path,
                                    &launch_env, &launch_shade_direct, &launch_shade_spec, &launch_film, layout_state_words()};
static const LayoutOps kLayoutSp4 = {&sp4::launch_trace, &sp4::launch_trace_coherent, &sp4::trace_kernel_id, &sp4::launch_gen_samples, &sp4::launch_mark_alive, &sp4::launch_sum_alive, &sp4::launch_transpose_samples, &sp4::launch_raygen,
                                     &sp4::launch_shade_path, &sp4::launch_env, &sp4::launch_shade_direct, &sp4::launch_shade_spec,
                                     &sp4::launch_film, sp4::layout_state_words()};

int traceGridFor(int wgPerCU) {
  // workgroups of the persistent traversal kernels: as many as are resident at once.  v2 (k_trace): 16 KiB of stack +
  // 6 KiB of cold ray state in LDS and 72 VGPRs => 7 workgroups = 28 waves per CU; the other variants (v3: 32 KiB of
  // LDS, the quadric kernel: more registers) 6, the sixth queueing behind five where only five fit.
  wgPerCU = dr_opt("DARTRAY_TRACE_WG_PER_CU").toInt(wgPerCU);
  return g_numCU * std::max(1, std::min(wgPerCU, 8));
}

extern "C" {

const char* dr_last_error(void) { return g_err.c_str(); }

int dr_set_option(const char* name, const char* value) {
  if (!name || !*name) return fail(DR_ERR_INVALID, "dr_set_option: null name");
  std::string n(name);
  for (char& c : n) c = (char)toupper((unsigned char)c);
  if (n.rfind("DARTRAY_", 0) != 0) n = "DARTRAY_" + n;
  bool known = false;
  for (const char* k : kOptionNames) known = known || n == k;
  if (!known) return fail(DR_ERR_INVALID, "dr_set_option: unknown option " + n);
  std::lock_guard<std::mutex> lock(g_optMutex);
  if (value) g_options[n] = value;  // "" hides the environment's value
  else g_options.erase(n);          // null: back to the environment's value
  return DR_OK;
}
const char* dr_version(void) { return "dartray_amd 0.6 (gfx950, abi 7)"; }
int32_t dr_abi_version(void) { return DR_ABI_VERSION; }

int dr_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return fail(DR_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= n) return fail(DR_ERR_INVALID, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  g_numCU = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // the two sets of state-touching kernels must be the two layouts this file hands out (dr_kernels.hip / dr_trace.hip are
  // compiled twice: as they are, and with -DDR_SUB=4 -DDR_NS=sp4 -DDR_STATE_WORDS_K=48 -DDR_GROUPED=1)
  if (layout_sub() != 64 || layout_state_words() != DR_STATE_WORDS || sp4::layout_sub() != 4 || sp4::layout_state_words() < DR_STATE_WORDS ||
      kLayout64.stateWords != layout_state_words() || kLayoutSp4.stateWords != sp4::layout_state_words())
    return fail(DR_ERR_UNSUPPORTED, "libdartray_hip was linked from kernel objects of unexpected state layouts");
  g_device = device;
  return DR_OK;
}

int32_t dr_sample_floats(int32_t integrator, uint32_t nlights) {
  // SURVEY.md Appendix B.  Path: 3 x (light 1D+2D, lightNum 1D, bsdf 1D+2D, path 1D+2D) + tau + scatter.
  if (integrator == DR_INTEGRATOR_PATH) return 5 + 14 + 18;
  // strategy "one" (direct_lighting_integrator.dart:82-87): light component, lightNum, BSDF component + tau + scatter; light position, BSDF direction
  if (integrator == DR_INTEGRATOR_DIRECT_ONE) return 5 + 5 + 4;
  return 5 + (2 * (int)nlights + 2) + 4 * (int)nlights;
}

int32_t dr_scene_sample_floats(const DrScene* scene, int32_t integrator) {
  if (!scene) return -1;
  return integrator == DR_INTEGRATOR_DIRECT_ALL ? scene->dlNFloats : dr_sample_floats(integrator, scene->d.nlights);  // ("all": the lights' nsamples decide)
}

// dr_scene_create in units (round 6): SceneBuilder carries what the steps share -- the host's description, the scene under construction, the
// primitive tables on the device -- and every step returns DR_OK or the error it has already reported (dr_scene_create then deletes the scene).
}  // extern "C"
namespace {
#define TRY_SC(expr)                                                             \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) return fail(DR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
struct SceneBuilder {
  const DrSceneDesc* desc;
  DrScene* sc;
  bool hostPrep = false;           // DARTRAY_SCENE_PREP=host: the serial host loops (the reference the device code is tested against)
  std::vector<uint8_t> level;      //   their per-node levels
  uint32_t measuredDepth = 0;
  DevBuf<float> dV;                // the primitive tables on the device (the device-side validation and the gather read them)
  DevBuf<uint32_t> dI, dM;
  DevBuf<int32_t> dL;
  DevBuf<uint8_t> dR;

  int validateOnHost();
  int quadrics();
  int uploadTables();
  int pairsOnDevice();
  int pairsOnHost();
  int gatherPrimitives();
  int shadingRecords();
  int materials();
  int lights();
  int envLightTables(int envLight);
  int finish();
  int directLightingLayout();
};

int SceneBuilder::validateOnHost() {
  // Validation of the marshalled tree, independent of which kernels can use it: a foreign host's BVHAccel.nodes are
  // input, and a malformed node must come back as DR_ERR_INVALID, not as an out-of-bounds device read or an endless
  // traversal.  Children always have larger indices than their parent (first child i + 1, second child offset > i + 1:
  // the depth-first numbering of bvh_accel.dart:419-437), so every walk terminates, and one forward pass gives each
  // node's level: the height of the tree bounds the traversal stack (desc->bvh_depth == 0, "unknown", is measured here).
  level.assign(hostPrep ? desc->nnodes : 0, 0);
  if (desc->nnodes && hostPrep) {
    const DrBvhNode* N = desc->nodes;
    for (uint64_t i = 0; i < desc->nnodes; ++i) {
      if (N[i].nprims == 0) {
        if (N[i].offset <= i + 1 || N[i].offset >= desc->nnodes || N[i].axis > 2)
          return fail(DR_ERR_INVALID, "malformed BVH node (interior node: second child must follow the first sub-tree, axis 0..2)");
        const uint32_t l = (uint32_t)level[i] + 1u;
        if (l > DR_MAX_STACK) return fail(DR_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
        level[i + 1] = std::max<uint8_t>(level[i + 1], (uint8_t)l);
        level[N[i].offset] = std::max<uint8_t>(level[N[i].offset], (uint8_t)l);
        measuredDepth = std::max(measuredDepth, l);
      } else if ((uint64_t)N[i].offset + N[i].nprims > desc->ntris) {
        return fail(DR_ERR_INVALID, "leaf primitive range");
      }
    }
  }
  if (desc->bvh_depth > DR_MAX_STACK) return fail(DR_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
  if (desc->bvh_depth != 0 && desc->bvh_depth < measuredDepth)
    return fail(DR_ERR_INVALID, "bvh_depth is smaller than the tree's height (pass 0 to have it measured)");
  for (uint64_t i = 0; hostPrep && i < 3 * desc->ntris; i += 3) {
    if (desc->tri_idx[i] == DR_PRIM_QUADRIC) continue;
    for (int k = 0; k < 3; ++k)
      if (desc->tri_idx[i + k] >= desc->nverts) return fail(DR_ERR_INVALID, "vertex index out of range");
  }
  return DR_OK;
}

int SceneBuilder::quadrics() {
  // quadric shapes (sphere.dart:24-32, disk.dart:24-28): constructor-derived fields in f64
  for (uint32_t i = 0; i < desc->nquadrics; ++i) {
    if (!desc->quadrics) return fail(DR_ERR_INVALID, "quadrics missing");
    const DrQuadric& a = desc->quadrics[i];
    DQuadric q;
    memset(&q, 0, sizeof(q));
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) {
        q.o2w[4 * r + c] = a.object_to_world[4 * r + c];
        q.w2o[4 * r + c] = a.world_to_object[4 * r + c];
      }
    for (int c = 0; c < 4; ++c)
      if (a.object_to_world[12 + c] != (c == 3 ? 1.0f : 0.0f) || a.world_to_object[12 + c] != (c == 3 ? 1.0f : 0.0f))
        return fail(DR_ERR_UNSUPPORTED, "projective object transforms are not on the path");
    auto clampd = [](double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); };
    auto radians = [](double deg) { return (3.141592653589793 / 180.0) * deg; };  // common.dart:87-88
    q.kind = a.kind;
    if (a.kind == DR_QUADRIC_SPHERE) {
      q.radius = a.params[0];
      const double z0 = a.params[1], z1 = a.params[2];
      q.zmin = clampd(std::min(z0, z1), -q.radius, q.radius);
      q.zmax = clampd(std::max(z0, z1), -q.radius, q.radius);
      q.thetaMin = std::acos(clampd(q.zmin / q.radius, -1.0, 1.0));
      q.thetaMax = std::acos(clampd(q.zmax / q.radius, -1.0, 1.0));
      q.phiMax = radians(clampd(a.params[3], 0.0, 360.0));
    } else if (a.kind == DR_QUADRIC_DISK) {
      q.height = a.params[0];
      q.radius = a.params[1];
      q.innerRadius = a.params[2];
      q.phiMax = radians(clampd(a.params[3], 0.0, 360.0));
    } else {
      return fail(DR_ERR_INVALID, "unknown quadric kind");
    }
    sc->hostQuads.push_back(q);
  }

  return DR_OK;
}

int SceneBuilder::uploadTables() {
  // nodes: the 32-byte marshalled node is consumed as two 16-byte loads
  TRY_SC(sc->nodes.alloc(2 * desc->nnodes));
  if (desc->nnodes) TRY_SC(hipMemcpy(sc->nodes.p, desc->nodes, desc->nnodes * sizeof(DrBvhNode), hipMemcpyHostToDevice));
  // the primitive tables (gathered into 48-byte records further down; the device-side validation reads them too)
  if (desc->ntris) {
    TRY_SC(dV.alloc(3 * std::max<uint64_t>(desc->nverts, 1)));
    TRY_SC(dI.alloc(3 * desc->ntris));
    TRY_SC(dM.alloc(desc->ntris));
    TRY_SC(dL.alloc(desc->ntris));
    TRY_SC(dR.alloc(desc->ntris));
    if (desc->nverts) TRY_SC(hipMemcpy(dV.p, desc->verts, 3 * desc->nverts * sizeof(float), hipMemcpyHostToDevice));
    TRY_SC(hipMemcpy(dI.p, desc->tri_idx, 3 * desc->ntris * sizeof(uint32_t), hipMemcpyHostToDevice));
    TRY_SC(hipMemcpy(dM.p, desc->tri_material, desc->ntris * sizeof(uint32_t), hipMemcpyHostToDevice));
    TRY_SC(hipMemcpy(dL.p, desc->tri_light, desc->ntris * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  return DR_OK;
}

int SceneBuilder::pairsOnDevice() {
  // sibling-pair layout for the v3 traversal (see dr_device.h): children of the k-th interior node side by side
  sc->d.pairs = nullptr;
  sc->d.npairs = 0;
  sc->d.topPairs = 0;
  sc->d.rootRef = PREF_DEAD;
  if (!hostPrep) {
    ScenePrepIn pin;
    memset(&pin, 0, sizeof(pin));
    pin.nodes = sc->nodes.p;
    pin.hostNodes = desc->nodes;
    pin.nnodes = desc->nnodes;
    pin.verts = dV.p;
    pin.nverts = desc->nverts;
    pin.triIdx = dI.p;
    pin.triMaterial = dM.p;
    pin.triLight = dL.p;
    pin.ntris = desc->ntris;
    pin.nquadrics = desc->nquadrics;
    pin.nmaterials = desc->nmaterials;
    pin.nlights = desc->nlights;
    pin.wantPairs = desc->nnodes && desc->ntris < (1ull << 26) && desc->nquadrics == 0;  // only the v2 kernel tests quadrics
    pin.topLevels = DR_PAIR_TOP_LEVELS;
    if (pin.wantPairs) {
      pin.pairsCap = desc->nnodes / 2 + 1;  // a binary tree of n nodes has (n - 1) / 2 interior ones
      TRY_SC(sc->pairs.alloc(4 * pin.pairsCap));
      pin.pairsOut = sc->pairs.p;
    }
    ScenePrepOut pout;
    const int prc = scene_prepare_device(pin, &pout);
    if (prc != DR_OK) return fail(prc, pout.message);
    measuredDepth = pout.depth;
    if (desc->bvh_depth != 0 && desc->bvh_depth < measuredDepth)
      return fail(DR_ERR_INVALID, "bvh_depth is smaller than the tree's height (pass 0 to have it measured)");
    if (pout.pairsOk) {
      const DrBvhNode& r = desc->nodes[0];
      sc->d.pairs = sc->pairs.p;
      sc->d.npairs = pout.npairs;
      sc->d.topPairs = pout.topPairs;
      sc->d.rootRef = r.nprims ? (PREF_LEAF | ((uint32_t)r.nprims << 26) | r.offset) : ((uint32_t)r.axis << 29);  // (the root's pair is slot 0 in either order)
      for (int k = 0; k < 3; ++k) {
        sc->d.rootBox[k] = r.bmin[k];
        sc->d.rootBox[3 + k] = r.bmax[k];
      }
    } else {
      sc->pairs.release();
    }
  }
  sc->bvhDepth = std::max(desc->bvh_depth, measuredDepth);  // (a caller may pass a bound larger than the height)
  if (desc->nnodes && sc->bvhDepth == 0) sc->bvhDepth = 1;  // a single leaf: "known, no stack needed"
  return DR_OK;
}

// The serial reference of scene_prepare_device (DARTRAY_SCENE_PREP=host): pair records in the same memory order, the union check.
int SceneBuilder::pairsOnHost() {
  if (desc->nnodes && hostPrep) {
    const DrBvhNode* N = desc->nodes;
    bool ok = desc->ntris < (1ull << 26) && desc->nquadrics == 0;  // only the v2 kernel tests quadrics
    std::vector<uint32_t> pairIndex(desc->nnodes, 0);
    uint32_t np = 0;
    for (uint64_t i = 0; i < desc->nnodes; ++i) {
      if (N[i].nprims == 0) {
        if (i + 1 >= desc->nnodes || N[i].offset >= desc->nnodes || N[i].axis > 2) return fail(DR_ERR_INVALID, "malformed BVH node");
        pairIndex[i] = np++;
      } else if (N[i].nprims > 31) {
        ok = false;  // packed references carry at most 31 primitives per leaf; fall back to the v2 kernel
      }
    }
    // Memory order of the pair records (results never depend on it -- the references are explicit): the top DR_PAIR_TOP_LEVELS levels
    // breadth-first (3 774 records = 236 KiB on C4), then every other interior node in depth-first (= node index) order, so a sub-tree
    // below the top is one contiguous run (round 4: C4 closest-hit -1 %, any-hit -3.5 % against plain depth-first; the other orders
    // that were tried -- sibling lines, padded records, van Emde Boas treelets -- are experiments/r06_runtime_switches.diff).
    {
      std::vector<uint32_t> slotOf(desc->nnodes, 0);
      uint32_t slots = 0;
      std::vector<uint32_t> top;
      for (uint64_t i = 0; i < desc->nnodes; ++i)
        if (N[i].nprims == 0 && level[i] < DR_PAIR_TOP_LEVELS) top.push_back((uint32_t)i);
      std::stable_sort(top.begin(), top.end(), [&](uint32_t a, uint32_t b) { return level[a] < level[b]; });
      for (uint32_t i : top) slotOf[i] = slots++;
      sc->d.topPairs = slots;
      for (uint64_t i = 0; i < desc->nnodes; ++i)
        if (N[i].nprims == 0 && level[i] >= DR_PAIR_TOP_LEVELS) slotOf[i] = slots++;
      if (slots) {
        pairIndex.swap(slotOf);
        np = slots;
      }
    }
    if (np >= (1u << 29)) ok = false;
    // The v3 kernel re-derives a node's own box when it needs the literal test: an interior node's bounds
    // must be the union of its children's (initInterior, bvh_accel.dart:518-524) and a leaf's the union of
    // its triangles' vertices (:238-241).  Trees built otherwise keep the v2 kernel.
    for (uint64_t i = 0; ok && i < desc->nnodes; ++i) {
      float lo[3], hi[3];
      if (N[i].nprims == 0) {
        const DrBvhNode &a = N[i + 1], &b = N[N[i].offset];
        for (int k = 0; k < 3; ++k) {
          lo[k] = std::min(a.bmin[k], b.bmin[k]);
          hi[k] = std::max(a.bmax[k], b.bmax[k]);
        }
      } else {
        if ((uint64_t)N[i].offset + N[i].nprims > desc->ntris) return fail(DR_ERR_INVALID, "leaf primitive range");
        for (int k = 0; k < 3; ++k) {
          lo[k] = std::numeric_limits<float>::infinity();
          hi[k] = -lo[k];
        }
        for (uint32_t t = 0; t < N[i].nprims; ++t)
          for (int v = 0; v < 3; ++v) {
            const uint32_t vi = desc->tri_idx[3 * ((uint64_t)N[i].offset + t) + v];
            if (vi >= desc->nverts) return fail(DR_ERR_INVALID, "vertex index out of range");
            for (int k = 0; k < 3; ++k) {
              lo[k] = std::min(lo[k], desc->verts[3 * (size_t)vi + k]);
              hi[k] = std::max(hi[k], desc->verts[3 * (size_t)vi + k]);
            }
          }
      }
      for (int k = 0; k < 3; ++k)
        if (lo[k] != N[i].bmin[k] || hi[k] != N[i].bmax[k]) ok = false;
    }
    if (ok) {
      auto packRef = [&](uint64_t c) -> uint32_t {
        return N[c].nprims ? (PREF_LEAF | ((uint32_t)N[c].nprims << 26) | N[c].offset) : (((uint32_t)N[c].axis << 29) | pairIndex[c]);
      };
      std::vector<DrBvhNode> P(2 * (size_t)std::max<uint32_t>(np, 1));
      for (uint64_t i = 0; i < desc->nnodes; ++i) {
        if (N[i].nprims != 0) continue;
        const uint64_t c[2] = {i + 1, N[i].offset};
        for (int k = 0; k < 2; ++k) {
          DrBvhNode r = N[c[k]];
          if (r.nprims == 0) r.offset = pairIndex[c[k]];
          P[2 * (size_t)pairIndex[i] + k] = r;
        }
      }
      TRY_SC(sc->pairs.alloc(4 * (size_t)std::max<uint32_t>(np, 1)));
      TRY_SC(hipMemcpy(sc->pairs.p, P.data(), P.size() * sizeof(DrBvhNode), hipMemcpyHostToDevice));
      sc->d.pairs = sc->pairs.p;
      sc->d.npairs = np;
      sc->d.rootRef = packRef(0);
      for (int k = 0; k < 3; ++k) {
        sc->d.rootBox[k] = N[0].bmin[k];
        sc->d.rootBox[3 + k] = N[0].bmax[k];
      }
    }
  }
  return DR_OK;
}

int SceneBuilder::gatherPrimitives() {
  // primitives: gather vertices on the device
  TRY_SC(sc->tris.alloc(3 * desc->ntris));
  if (desc->ntris) {
    // per-primitive flag byte: bit 0 = Shape.reverseOrientation, bits 1.. = the quadric kind (the device-side validation has
    // checked every index when the host loops did not)
    std::vector<uint8_t> flags(desc->ntris);
    for (uint64_t i = 0; i < desc->ntris; ++i) flags[i] = desc->tri_reverse[i] ? 1 : 0;
    for (uint64_t i = 0; (desc->nquadrics || hostPrep) && i < desc->ntris; ++i) {
      if (desc->tri_idx[3 * i] == DR_PRIM_QUADRIC) {
        const uint32_t qi = desc->tri_idx[3 * i + 1];
        if (qi >= desc->nquadrics) return fail(DR_ERR_INVALID, "quadric index out of range");
        flags[i] |= (uint8_t)(sc->hostQuads[qi].kind << 1);
        sc->hostQuads[qi].reverse = desc->tri_reverse[i] ? 1 : 0;  // Shape.reverseOrientation of the primitive's shape
        continue;
      }
      for (int k = 0; k < 3; ++k)
        if (desc->tri_idx[3 * i + k] >= desc->nverts) return fail(DR_ERR_INVALID, "vertex index out of range");
    }
    for (uint64_t i = 0; hostPrep && i < desc->ntris; ++i) {
      if (desc->tri_material[i] >= desc->nmaterials) return fail(DR_ERR_INVALID, "material index out of range");
      if (desc->tri_light[i] >= (int32_t)desc->nlights) return fail(DR_ERR_INVALID, "light index out of range");
    }
    TRY_SC(hipMemcpy(dR.p, flags.data(), desc->ntris, hipMemcpyHostToDevice));
    launch_gather_tris(dV.p, dI.p, dM.p, dL.p, dR.p, sc->tris.p, desc->ntris, 0);
    TRY_SC(hipDeviceSynchronize());
  }
  return DR_OK;
}

int SceneBuilder::shadingRecords() {
  // per-primitive shading records of meshes with N / S / uv (see ShadeRec in dr_device.h)
  sc->d.srec = nullptr;
  sc->d.xforms = nullptr;
  if (desc->tri_shading && desc->ntris) {
    bool any = false;
    for (uint64_t i = 0; i < desc->ntris; ++i)
      if (desc->tri_shading[i] && desc->tri_idx[3 * i] != DR_PRIM_QUADRIC) any = true;
    if (any) {
      std::vector<float> R(28 * (size_t)desc->ntris, 0.f);
      for (uint64_t i = 0; i < desc->ntris; ++i) {
        const uint32_t f = desc->tri_shading[i];
        if (!f || desc->tri_idx[3 * i] == DR_PRIM_QUADRIC) continue;
        if (f > 7u) return fail(DR_ERR_INVALID, "unknown tri_shading bits");
        if (((f & DR_SHADING_N) && !desc->vert_normals) || ((f & DR_SHADING_S) && !desc->vert_tangents) ||
            ((f & DR_SHADING_UV) && !desc->vert_uvs))
          return fail(DR_ERR_INVALID, "tri_shading names an attribute whose vertex array is missing");
        uint32_t xf = 0;
        if (f & (DR_SHADING_N | DR_SHADING_S)) {
          if (!desc->tri_xform || !desc->mesh_xforms || desc->tri_xform[i] >= desc->nmesh_xforms)
            return fail(DR_ERR_INVALID, "per-vertex normals / tangents need their mesh transform");
          xf = desc->tri_xform[i];
        }
        float* r = &R[28 * (size_t)i];
        for (int k = 0; k < 3; ++k) {
          const size_t v = desc->tri_idx[3 * i + k];
          for (int c = 0; c < 3; ++c) {
            if (f & DR_SHADING_N) r[3 * k + c] = desc->vert_normals[3 * v + c];
            if (f & DR_SHADING_S) r[9 + 3 * k + c] = desc->vert_tangents[3 * v + c];
          }
          if (f & DR_SHADING_UV) {
            r[18 + 2 * k] = desc->vert_uvs[2 * v];
            r[18 + 2 * k + 1] = desc->vert_uvs[2 * v + 1];
          }
        }
        memcpy(&r[24], &f, 4);
        memcpy(&r[25], &xf, 4);
      }
      TRY_SC(sc->srec.alloc(7 * (size_t)desc->ntris));
      TRY_SC(hipMemcpy(sc->srec.p, R.data(), R.size() * sizeof(float), hipMemcpyHostToDevice));
      sc->d.srec = sc->srec.p;
      std::vector<float> X(24 * (size_t)std::max<uint32_t>(desc->nmesh_xforms, 1), 0.f);
      for (uint32_t i = 0; i < desc->nmesh_xforms; ++i)
        for (int k = 0; k < 12; ++k) {
          X[24 * (size_t)i + k] = desc->mesh_xforms[i].object_to_world[k];
          X[24 * (size_t)i + 12 + k] = desc->mesh_xforms[i].world_to_object[k];
        }
      TRY_SC(sc->xforms.alloc(X.size()));
      TRY_SC(hipMemcpy(sc->xforms.p, X.data(), X.size() * sizeof(float), hipMemcpyHostToDevice));
      sc->d.xforms = sc->xforms.p;
    }
  }
  return DR_OK;
}

int SceneBuilder::materials() {
  // materials
  {
    // 4 x float4 per material: (Kd, -) (Kr, type) (Kt, -) (index, sigma: each double's low / high word)
    std::vector<float4> m(4 * (size_t)std::max<uint32_t>(desc->nmaterials, 1), make_float4(0.f, 0.f, 0.f, 0.f));
    auto bitsf = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    for (uint32_t i = 0; i < desc->nmaterials; ++i) {
      const DrMaterial& a = desc->materials[i];
      uint64_t ib, sb;
      memcpy(&ib, &a.index, 8);
      memcpy(&sb, &a.sigma, 8);
      m[4 * i] = make_float4(a.kd[0], a.kd[1], a.kd[2], 0.f);
      m[4 * i + 1] = make_float4(a.kr[0], a.kr[1], a.kr[2], bitsf((uint32_t)a.type));
      m[4 * i + 2] = make_float4(a.kt[0], a.kt[1], a.kt[2], 0.f);
      m[4 * i + 3] = make_float4(bitsf((uint32_t)ib), bitsf((uint32_t)(ib >> 32)), bitsf((uint32_t)sb), bitsf((uint32_t)(sb >> 32)));
    }
    TRY_SC(sc->mats.alloc(m.size()));
    TRY_SC(hipMemcpy(sc->mats.p, m.data(), m.size() * sizeof(float4), hipMemcpyHostToDevice));
  }
  return DR_OK;
}

int SceneBuilder::lights() {
  // lights: ShapeSet areas + Distribution1D (shape_set.dart:40-50; montecarlo.dart:25-52)
  {
    std::vector<DLight> L(std::max<uint32_t>(desc->nlights, 1));
    std::vector<DLightTri> LT(std::max<uint32_t>(desc->nlight_tris, 1));
    std::vector<float> cdf;
    int envLight = -1;
    for (uint32_t i = 0; i < desc->nlights; ++i) {
      const DrAreaLight& a = desc->lights[i];
      if (a.kind == DR_LIGHT_INFINITE) {
        if (a.env_index >= desc->nenv_maps || !desc->env_maps) return fail(DR_ERR_INVALID, "infinite light without a radiance map");
        if (envLight >= 0) return fail(DR_ERR_UNSUPPORTED, "more than one infinite light");
        envLight = (int)i;
        DLight& d = L[i];
        d.L[0] = a.L[0]; d.L[1] = a.L[1]; d.L[2] = a.L[2];
        d.nsamples = std::max(1, a.nsamples);
        d.first_tri = d.ntris = d.cdf_off = 0;
        d.kind = DR_LIGHT_INFINITE;
        d.area = 0.0;
        sc->lightNSamples.push_back(d.nsamples);
        continue;
      }
      if (a.kind == DR_LIGHT_POINT || a.kind == DR_LIGHT_SPOT || a.kind == DR_LIGHT_SPOT_COS || a.kind == DR_LIGHT_DISTANT) {
        DLight& d = L[i];
        memset(&d, 0, sizeof(d));
        d.L[0] = a.L[0]; d.L[1] = a.L[1]; d.L[2] = a.L[2];
        d.nsamples = 1;
        d.kind = a.kind == DR_LIGHT_SPOT_COS ? DR_LIGHT_SPOT : a.kind;
        d.pos[0] = a.position[0]; d.pos[1] = a.position[1]; d.pos[2] = a.position[2];
        if (a.kind == DR_LIGHT_SPOT) {  // spot_light.dart:42-48
          for (int k = 0; k < 12; ++k) d.w2l[k] = a.world_to_light[k];
          d.cosTotalWidth = std::cos((3.141592653589793 / 180.0) * a.cone_width);
          d.cosFalloffStart = std::cos((3.141592653589793 / 180.0) * a.cone_falloff_start);
        } else if (a.kind == DR_LIGHT_SPOT_COS) {  // the cosines a constructed SpotLight keeps (spot_light.dart:46-47)
          for (int k = 0; k < 12; ++k) d.w2l[k] = a.world_to_light[k];
          d.cosTotalWidth = a.cone_width;
          d.cosFalloffStart = a.cone_falloff_start;
        }
        sc->lightNSamples.push_back(1);
        sc->hasDeltaLight = true;
        continue;
      }
      if (a.kind != DR_LIGHT_DIFFUSE_AREA) return fail(DR_ERR_INVALID, "unknown light kind");
      if (a.ntris == 0 || (uint64_t)a.first_tri + a.ntris > desc->nlight_tris) return fail(DR_ERR_INVALID, "light triangle range");
      DLight& d = L[i];
      d.L[0] = a.L[0]; d.L[1] = a.L[1]; d.L[2] = a.L[2];
      d.nsamples = std::max(1, a.nsamples);
      d.first_tri = a.first_tri;
      d.ntris = a.ntris;
      d.kind = DR_LIGHT_DIFFUSE_AREA;
      sc->lightNSamples.push_back(d.nsamples);
      double area = 0.0;
      std::vector<double> areas(a.ntris);
      for (uint32_t t = 0; t < a.ntris; ++t) {
        const DrLightTri& lt = desc->light_tris[a.first_tri + t];
        DLightTri& o = LT[a.first_tri + t];
        if (lt.v[0] == DR_PRIM_QUADRIC) {
          if (lt.v[1] >= desc->nquadrics) return fail(DR_ERR_INVALID, "light quadric index out of range");
          const DQuadric& q = sc->hostQuads[lt.v[1]];
          memset(o.p, 0, sizeof(o.p));
          memcpy(&o.p[0], &lt.v[1], sizeof(uint32_t));
          o.reverse = (lt.reverse_orientation ? 1u : 0u) | ((uint32_t)q.kind << 8);
          if (q.kind == DR_QUADRIC_SPHERE) {
            o.area = q.phiMax * q.radius * (q.zmax - q.zmin);  // sphere.dart:251-253
            for (int k = 0; k < 3; ++k) o.ns[k] = o.nn[k] = 0.f;  // Sphere.sample2 computes Ns per sample
          } else {
            o.area = q.phiMax * 0.5 * (q.radius * q.radius - q.innerRadius * q.innerRadius);  // disk.dart:139-142
            // Ns of Disk.sample (disk.dart:149-153): normalize(objectToWorld.transformNormal((0,0,1))), flipped
            // when reverseOrientation; nn (the hit's dg.nn) depends on the hit point and is evaluated on the device
            double n[3] = {(double)(float)q.w2o[8], (double)(float)q.w2o[9], (double)(float)q.w2o[10]};  // mInv^T * (0,0,1), stored f32
            const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            for (int k = 0; k < 3; ++k) {
              float v = (float)(n[k] / len);
              if (lt.reverse_orientation) v = (float)((double)v * -1.0);
              o.ns[k] = v;
              o.nn[k] = v;
            }
          }
          areas[t] = o.area;
          area += o.area;
          continue;
        }
        for (int k = 0; k < 3; ++k) {
          if (lt.v[k] >= desc->nverts) return fail(DR_ERR_INVALID, "light vertex index out of range");
          for (int c = 0; c < 3; ++c) o.p[3 * k + c] = desc->verts[3 * (size_t)lt.v[k] + c];
        }
        o.reverse = lt.reverse_orientation & 1u;
        o.area = host_tri_area(o.p, o.p + 3, o.p + 6);
        float luv[6];
        const bool hasUV = (lt.reverse_orientation & 2u) != 0;
        if (hasUV) {
          if (!desc->vert_uvs) return fail(DR_ERR_INVALID, "light triangle with uvs but no vert_uvs");
          for (int k = 0; k < 3; ++k) {
            luv[2 * k] = desc->vert_uvs[2 * (size_t)lt.v[k]];
            luv[2 * k + 1] = desc->vert_uvs[2 * (size_t)lt.v[k] + 1];
          }
        }
        host_tri_normals(o.p, o.p + 3, o.p + 6, (lt.reverse_orientation & 1u) != 0, o.nn, o.ns, hasUV ? luv : nullptr);
        areas[t] = o.area;
        area += o.area;
      }
      d.area = area;
      // Distribution1D(areas, n)
      int count = (int)a.ntris;
      std::vector<float> func(count), c(count + 1);
      for (int k = 0; k < count; ++k) func[k] = (float)areas[k];
      c[0] = 0.0f;
      for (int k = 1; k < count + 1; ++k) c[k] = (float)((double)c[k - 1] + (double)func[k - 1] / (double)count);
      double funcInt = c[count];
      if (funcInt == 0.0) {
        for (int k = 1; k < count + 1; ++k) c[k] = (float)((double)k / (double)count);
      } else {
        for (int k = 1; k < count + 1; ++k) c[k] = (float)((double)c[k] / funcInt);
      }
      d.cdf_off = (uint32_t)cdf.size();
      cdf.insert(cdf.end(), c.begin(), c.end());
    }
    if (cdf.empty()) cdf.push_back(0.f);
    memset(&sc->d.env, 0, sizeof(sc->d.env));
    sc->d.hasEnv = 0;
    if (envLight >= 0) {
      const int erc = envLightTables(envLight);
      if (erc) return erc;
    }
    TRY_SC(sc->lights.alloc(L.size()));
    TRY_SC(hipMemcpy(sc->lights.p, L.data(), L.size() * sizeof(DLight), hipMemcpyHostToDevice));
    TRY_SC(sc->ltris.alloc(LT.size()));
    TRY_SC(hipMemcpy(sc->ltris.p, LT.data(), LT.size() * sizeof(DLightTri), hipMemcpyHostToDevice));
    TRY_SC(sc->lcdf.alloc(cdf.size()));
    TRY_SC(hipMemcpy(sc->lcdf.p, cdf.data(), cdf.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  return DR_OK;
}

// The InfiniteAreaLight's tables: the radiance map's level 0 (resampled like MIPMap.texture when its size is no power of two) and the
// Distribution2D over luminance x sin(theta) (_setRadianceMap, infinite_area_light.dart:283-307).
int SceneBuilder::envLightTables(int envLight) {
    const DrAreaLight& a = desc->lights[envLight];
    const DrEnvMap& m = desc->env_maps[a.env_index];
    if (!m.texels || m.width <= 0 || m.height <= 0) return fail(DR_ERR_INVALID, "radiance map: null texels or empty size");
    if (m.width > (1 << 14) || m.height > (1 << 14)) return fail(DR_ERR_UNSUPPORTED, "radiance map larger than 16384 texels a side");
    // MIPMap.texture resamples an image whose width or height is no power of two up to the next one (mipmap.dart:71-138) before
    // anything reads it; a host that hands over the decoded image (not a pyramid level) gets the same level 0 here
    std::vector<float> resampled;
    int w = m.width, h = m.height;
    const float* texels = m.texels;
    if ((w & (w - 1)) || (h & (h - 1))) {
      resample_to_pow2(m.texels, w, h, resampled, &w, &h);
      texels = resampled.data();
    }
    // _setRadianceMap (infinite_area_light.dart:283-307): img = luminance(_radiance(u/w, v/h, filter)) * sin(theta),
    // filter = 1/max(w,h).  For a power-of-two map MIPMap.lookup's level = levels-1 + log2(filter) is 0 up to
    // rounding (mipmap.dart:211): either `triangle(0,s,t)` directly or triangle(0)*(1-d) + triangle(1)*d with
    // d ~ 1e-15, which rounds to the same f32 -- so the bilinear level-0 value is used.
    std::vector<float> img((size_t)w * h);
    auto texel = [&](int s, int t, int c) {
      s %= w; if (s < 0) s += w;
      t %= h; if (t < 0) t += h;
      return (double)texels[3 * ((size_t)t * w + s) + c];
    };
    for (int v = 0; v < h; ++v) {
      const double sinTheta = std::sin(3.141592653589793 * (v + 0.5) / h);
      for (int u = 0; u < w; ++u) {
        double s = ((double)u / w) * w - 0.5, t = ((double)v / h) * h - 0.5;
        const int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
        const double ds = s - s0, dt = t - t0;
        double rgbv[3];
        for (int c = 0; c < 3; ++c) {
          double acc = r32(texel(s0, t0, c) * ((1.0 - ds) * (1.0 - dt)));
          acc = r32(acc + r32(texel(s0, t0 + 1, c) * ((1.0 - ds) * dt)));
          acc = r32(acc + r32(texel(s0 + 1, t0, c) * (ds * (1.0 - dt))));
          acc = r32(acc + r32(texel(s0 + 1, t0 + 1, c) * (ds * dt)));
          rgbv[c] = r32(acc * (double)a.L[c]);
        }
        float y = (float)(0.212671 * rgbv[0] + 0.715160 * rgbv[1] + 0.072169 * rgbv[2]);
        img[u + (size_t)v * w] = (float)((double)y * sinTheta);
      }
    }
    // Distribution2D (montecarlo.dart:223-237): one Distribution1D per row + the marginal over their integrals
    auto dist1d = [](const float* f, int count, float* func, float* c, float* funcIntOut) {
      for (int k = 0; k < count; ++k) func[k] = f[k];
      c[0] = 0.0f;
      for (int k = 1; k < count + 1; ++k) c[k] = (float)((double)c[k - 1] + (double)func[k - 1] / (double)count);
      const double funcInt = c[count];
      if (funcInt == 0.0) {
        for (int k = 1; k < count + 1; ++k) c[k] = (float)((double)k / (double)count);
      } else {
        for (int k = 1; k < count + 1; ++k) c[k] = (float)((double)c[k] / funcInt);
      }
      *funcIntOut = (float)funcInt;
    };
    std::vector<float> cf((size_t)w * h), cc((size_t)(w + 1) * h), ci(h), mf(h), mc(h + 1);
    for (int v = 0; v < h; ++v) dist1d(&img[(size_t)v * w], w, &cf[(size_t)v * w], &cc[(size_t)v * (w + 1)], &ci[v]);
    float mi = 0.f;
    dist1d(ci.data(), h, mf.data(), mc.data(), &mi);
    TRY_SC(sc->envTexels.alloc(3 * (size_t)w * h));
    TRY_SC(hipMemcpy(sc->envTexels.p, texels, 3 * (size_t)w * h * sizeof(float), hipMemcpyHostToDevice));
    TRY_SC(sc->envCondFunc.alloc(cf.size()));
    TRY_SC(hipMemcpy(sc->envCondFunc.p, cf.data(), cf.size() * sizeof(float), hipMemcpyHostToDevice));
    TRY_SC(sc->envCondCdf.alloc(cc.size()));
    TRY_SC(hipMemcpy(sc->envCondCdf.p, cc.data(), cc.size() * sizeof(float), hipMemcpyHostToDevice));
    TRY_SC(sc->envCondInt.alloc(ci.size()));
    TRY_SC(hipMemcpy(sc->envCondInt.p, ci.data(), ci.size() * sizeof(float), hipMemcpyHostToDevice));
    TRY_SC(sc->envMargFunc.alloc(mf.size()));
    TRY_SC(hipMemcpy(sc->envMargFunc.p, mf.data(), mf.size() * sizeof(float), hipMemcpyHostToDevice));
    TRY_SC(sc->envMargCdf.alloc(mc.size()));
    TRY_SC(hipMemcpy(sc->envMargCdf.p, mc.data(), mc.size() * sizeof(float), hipMemcpyHostToDevice));
    DEnv& e = sc->d.env;
    // guide rows of the conditional CDFs (DEnv::condGuide): upper_bound at u = k / G, G = w / 4 (a power of two)
    e.condGuide = nullptr;
    e.guideN = 0;
    if (w >= 16 && w + 1 <= 65535) {
      const int G = w / 4;
      std::vector<uint16_t> guide((size_t)h * (G + 1));
      for (int v = 0; v < h; ++v) {
        const float* c = &cc[(size_t)v * (w + 1)];
        int i = 0;  // upper_bound is monotone in u: one sweep per row
        for (int k = 0; k <= G; ++k) {
          const double u = (double)k / (double)G;
          while (i < w + 1 && !(u < (double)c[i])) ++i;
          guide[(size_t)v * (G + 1) + k] = (uint16_t)i;
        }
      }
      TRY_SC(sc->envCondGuide.alloc(guide.size()));
      TRY_SC(hipMemcpy(sc->envCondGuide.p, guide.data(), guide.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      e.condGuide = sc->envCondGuide.p;
      e.guideN = G;
    }
    e.texels = sc->envTexels.p;
    e.condFunc = sc->envCondFunc.p;
    e.condCdf = sc->envCondCdf.p;
    e.condInt = sc->envCondInt.p;
    e.margFunc = sc->envMargFunc.p;
    e.margCdf = sc->envMargCdf.p;
    e.margInt = mi;
    e.w = w;
    e.h = h;
    for (int c = 0; c < 3; ++c) e.L[c] = a.L[c];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        e.l2w[3 * r + c] = m.light_to_world[4 * r + c];
        e.w2l[3 * r + c] = m.world_to_light[4 * r + c];
      }
    sc->d.hasEnv = 1;
  return DR_OK;
}

int SceneBuilder::finish() {
  TRY_SC(sc->ctr.alloc(1));
  TRY_SC(hipMemset(sc->ctr.p, 0, sizeof(TraceCounters)));
  TRY_SC(sc->quads.alloc(std::max<size_t>(sc->hostQuads.size(), 1)));
  if (!sc->hostQuads.empty())
    TRY_SC(hipMemcpy(sc->quads.p, sc->hostQuads.data(), sc->hostQuads.size() * sizeof(DQuadric), hipMemcpyHostToDevice));
  sc->d.quads = sc->quads.p;
  sc->d.nquads = (uint32_t)sc->hostQuads.size();
  sc->d.hasSpec = 0;
  for (uint32_t i = 0; i < desc->nmaterials; ++i)
    if (desc->materials[i].type != DR_MATERIAL_MATTE) sc->d.hasSpec = 1;
    else if (desc->materials[i].sigma != 0.0) sc->d.hasSpec = 1;  // Oren-Nayar: general shading kernels too
  if (sc->hasDeltaLight) sc->d.hasSpec = 1;  // point lights are handled by the general kernels
  sc->hasSpecular = false;
  for (uint32_t i = 0; i < desc->nmaterials; ++i)
    if (desc->materials[i].type == DR_MATERIAL_MIRROR || desc->materials[i].type == DR_MATERIAL_GLASS) sc->hasSpecular = true;
  sc->d.nodes = sc->nodes.p;
  sc->d.tris = sc->tris.p;
  sc->d.mats = sc->mats.p;
  sc->d.lights = sc->lights.p;
  sc->d.ltris = sc->ltris.p;
  sc->d.lcdf = sc->lcdf.p;
  sc->d.nnodes = (uint32_t)desc->nnodes;
  sc->d.ntris = (uint32_t)desc->ntris;
  sc->d.nlights = desc->nlights;
  sc->d.nmats = desc->nmaterials;
  sc->d.nltris = desc->nlight_tris;
  sc->d.ncdf = (uint32_t)sc->lcdf.n;
  // plain triangles + matte materials (the !QUAD shade kernels): the 32-byte shading records (ShTri in dr_device.h)
  sc->d.shtris = nullptr;
  if (!(sc->d.nquads || sc->d.hasSpec || sc->d.srec) && desc->ntris) {
    TRY_SC(sc->shtris.alloc(2 * desc->ntris));
    launch_make_shtris(sc->d, sc->shtris.p, desc->ntris, 0);
    TRY_SC(hipDeviceSynchronize());
    sc->d.shtris = sc->shtris.p;
  }
  return DR_OK;
}

int SceneBuilder::directLightingLayout() {
  {  // DirectLighting: one 1-D + one 2-D slot pair per light for the light sample and one for the BSDF sample, each
     // with roundSize(nSamples) entries (low_discrepancy_sampler.dart:43-49), then the two 1-D volume slots
    auto rp2 = [](int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; };
    const size_t nl = sc->lightNSamples.size();
    std::vector<int> ns(nl);
    int n1D = 2;
    for (size_t i = 0; i < nl; ++i) {
      ns[i] = rp2(std::max(1, sc->lightNSamples[i]));
      n1D += 2 * ns[i];
      if (ns[i] != 1) sc->dlMulti = true;
    }
    std::vector<LdBlock> blocks;
    blocks.push_back({0, 1, 1, 0});
    blocks.push_back({2, 1, 1, 0});
    blocks.push_back({4, 1, 0, 0});
    std::vector<DirectStage> stages;
    int o1 = 5, o2 = 5 + n1D;
    for (size_t i = 0; i < nl; ++i) {
      blocks.push_back({o1, ns[i], 0, 0});
      blocks.push_back({o1 + ns[i], ns[i], 0, 0});
      for (int j = 0; j < ns[i]; ++j)
        stages.push_back({(int)i, ns[i], j == ns[i] - 1 ? 1 : 0, o1 + j, o2 + 2 * j, o2 + 2 * ns[i] + 2 * j, o1 + ns[i] + j, 0});
      o1 += 2 * ns[i];
      o2 += 4 * ns[i];
    }
    blocks.push_back({o1, 1, 0, 0});
    blocks.push_back({o1 + 1, 1, 0, 0});
    {  // the 2-D blocks follow all 1-D blocks (montecarlo.dart:441-448)
      int p2 = 5 + n1D;
      for (size_t i = 0; i < nl; ++i) {
        blocks.push_back({p2, ns[i], 1, 0});
        blocks.push_back({p2 + 2 * ns[i], ns[i], 1, 0});
        p2 += 4 * ns[i];
      }
    }
    // strategy "one": ONE EstimateDirect call; light < 0 = "the light floor(u * nLights) of the 1-D slot at float index pad1"; its slots
    // are requested in the order light (1-D, 2-D), lightNum (1-D), BSDF (1-D, 2-D) (direct_lighting_integrator.dart:82-87), then tau / scatter
    stages.push_back({-1, 1, 1, 5, 10, 12, 7, 6});
    sc->dlNBlocks = (int)blocks.size();
    sc->dlNStages = (int)stages.size() - 1;
    sc->dlNFloats = o2;
    sc->dlN1D = n1D;
    TRY_SC(sc->dlBlocks.alloc(blocks.size()));
    TRY_SC(sc->dlStages.alloc(std::max<size_t>(stages.size(), 1)));
    TRY_SC(hipMemcpy(sc->dlBlocks.p, blocks.data(), blocks.size() * sizeof(LdBlock), hipMemcpyHostToDevice));
    if (!stages.empty())
      TRY_SC(hipMemcpy(sc->dlStages.p, stages.data(), stages.size() * sizeof(DirectStage), hipMemcpyHostToDevice));
  }
  return DR_OK;
}
#undef TRY_SC
}  // namespace
extern "C" {

int dr_scene_create(const DrSceneDesc* desc, DrScene** out) {
  if (g_device < 0) return fail(DR_ERR_NO_DEVICE, "dr_init has not been called");
  if (!desc || !out) return fail(DR_ERR_INVALID, "null argument");
  if (desc->ntris > 0 && (!desc->nodes || !desc->verts || !desc->tri_idx || !desc->tri_material || !desc->tri_light ||
                          !desc->tri_reverse || !desc->materials))
    return fail(DR_ERR_INVALID, "scene arrays missing");
  if (desc->ntris >= (1ull << 31) || desc->nnodes >= (1ull << 31)) return fail(DR_ERR_INVALID, "scene too large");
  for (uint32_t i = 0; i < desc->nmaterials; ++i) {
    if (desc->materials[i].type < DR_MATERIAL_MATTE || desc->materials[i].type > DR_MATERIAL_PLASTIC)
      return fail(DR_ERR_INVALID, "unknown material type");
  }
  // k_trace addresses node i at byte offset i * 32 from a scalar base, in 32 bits (dr_trace.hip)
  if (desc->nnodes > (1ull << 27)) return fail(DR_ERR_UNSUPPORTED, "more than 2^27 BVH nodes");
  SceneBuilder B;
  B.desc = desc;
  B.sc = new DrScene();
  memset(&B.sc->stats, 0, sizeof(B.sc->stats));
  // Round 4: validation, height, pair records and the union check run on the device (dr_scene_prep.hip: C4 0.6 s -> 0.1 s).  The
  // serial host loops remain as the reference the device results are tested against (DARTRAY_SCENE_PREP=host).
  B.hostPrep = dr_opt("DARTRAY_SCENE_PREP").is("host");
  int (SceneBuilder::*const steps[])() = {&SceneBuilder::validateOnHost, &SceneBuilder::quadrics, &SceneBuilder::uploadTables, &SceneBuilder::pairsOnDevice,
                                          &SceneBuilder::pairsOnHost, &SceneBuilder::gatherPrimitives, &SceneBuilder::shadingRecords, &SceneBuilder::materials,
                                          &SceneBuilder::lights, &SceneBuilder::finish, &SceneBuilder::directLightingLayout};
  for (auto step : steps) {
    const int rc = (B.*step)();
    if (rc != DR_OK) {
      delete B.sc;
      return rc;
    }
  }
  B.sc->d.traceKernel[0] = B.sc->d.traceKernel[1] = 0;
  B.sc->d.anyFarFirst = 0;
  *out = B.sc;
  return DR_OK;
}

void dr_scene_destroy(DrScene* scene) { delete scene; }

int dr_scene_get_trace_kernels(const DrScene* sc, uint32_t out[2]) {
  if (!sc || !out) return fail(DR_ERR_INVALID, "null argument");
  out[0] = sc->traceCalibrated ? sc->d.traceKernel[0] : 0u;
  out[1] = sc->traceCalibrated ? sc->d.traceKernel[1] : 0u;
  return DR_OK;
}

int dr_scene_last_render_info(const DrScene* sc, int32_t out[8]) {
  if (!sc || !out) return fail(DR_ERR_INVALID, "null argument");
  for (int i = 0; i < 8; ++i) out[i] = sc->lastInfo[i];
  return DR_OK;
}

int dr_scene_get_pilot(const DrScene* sc, float out[6]) {
  if (!sc || !out) return fail(DR_ERR_INVALID, "null argument");
  for (int c = 0; c < 3; ++c) out[c] = sc->calibPerGB[0][c];
  out[3] = sc->calibPerGB[1][0];
  out[4] = sc->calibPerGB[1][1];
  out[5] = sc->calibFarFirst;
  return DR_OK;
}

int dr_scene_set_trace_kernels(DrScene* sc, const uint32_t in[2]) {
  if (!sc || !in) return fail(DR_ERR_INVALID, "null argument");
  if (in[0] == 0u && in[1] == 0u) {
    sc->traceCalibrated = false;
    sc->d.traceKernel[0] = sc->d.traceKernel[1] = 0u;
    return DR_OK;
  }
  for (int k = 0; k < 2; ++k) {
    if (in[k] != 2u && in[k] != 3u && !(k == 0 && in[k] == 5u) && !(k == 1 && (in[k] == 6u || in[k] == 7u)))
      return fail(DR_ERR_INVALID, "trace kernel must be 2 or 3 (closest-hit rays also 5, any-hit rays also 6 / 7 = 2 / 3 far child first; or 0, 0 to measure again)");
    if (in[k] != 2u && in[k] != 6u && (!sc->d.pairs || sc->d.nquads)) return fail(DR_ERR_UNSUPPORTED, "this scene cannot use the sibling-pair kernels");
  }
  sc->d.traceKernel[0] = in[0];
  sc->d.traceKernel[1] = in[1];
  sc->traceCalibrated = true;
  return DR_OK;
}

int dr_scene_get_pairs(const DrScene* sc, void* out, uint64_t cap_bytes, uint64_t* npairs_out, uint32_t* top_pairs_out, uint32_t* depth_out) {
  if (!sc || !npairs_out) return fail(DR_ERR_INVALID, "null argument");
  *npairs_out = sc->d.pairs ? sc->d.npairs : 0;
  if (top_pairs_out) *top_pairs_out = sc->d.topPairs;
  if (depth_out) *depth_out = sc->bvhDepth;
  if (out && sc->d.pairs) {
    if (cap_bytes < (uint64_t)sc->d.npairs * 64) return fail(DR_ERR_INVALID, "pair buffer too small");
    HIP_TRY(hipMemcpy(out, sc->d.pairs, (size_t)sc->d.npairs * 64, hipMemcpyDeviceToHost));
  }
  return DR_OK;
}

int dr_scene_workspace_bytes(const DrScene* sc, uint64_t* bytes_out) {
  if (!sc || !bytes_out) return fail(DR_ERR_INVALID, "null argument");
  const Workspace& w = sc->ws;
  *bytes_out = w.tiles.bytes() + w.scr.bytes() + w.genState.bytes() + w.tail.bytes() + w.tailOff.bytes() + w.activeA.bytes() + w.activeB.bytes() +
               w.closestQ.bytes() + w.anyQ.bytes() + w.counters.bytes() + w.spill.bytes() + w.envQ.bytes() + w.alive.bytes() + w.roundA.bytes() +
               w.roundB.bytes() + w.specFrames.bytes() + w.specSp.bytes() + w.pix.bytes() + w.filterTable.bytes() + w.aosSamples.bytes();
  return DR_OK;
}

int dr_scene_get_state_layout(const DrScene* sc, int32_t* layout_out, float* density_out) {
  if (!sc || !layout_out) return fail(DR_ERR_INVALID, "null argument");
  *layout_out = sc->stateLayout;
  if (density_out) *density_out = sc->layoutDensity;
  return DR_OK;
}

int dr_scene_set_state_layout(DrScene* sc, int32_t layout) {
  if (!sc) return fail(DR_ERR_INVALID, "null argument");
  if (layout != 0 && layout != 4 && layout != 64) return fail(DR_ERR_INVALID, "state layout must be 64 or 4 (or 0 to measure again)");
  sc->stateLayout = layout;
  if (layout == 0) sc->layoutDensity = -1.f;
  return DR_OK;
}

int dr_intersect(DrScene* sc, const DrRay* rays, int64_t n, DrHit* out, int32_t any_hit) {
  if (!sc || (n > 0 && (!rays || !out))) return fail(DR_ERR_INVALID, "null argument");
  if (n <= 0) return DR_OK;
  if (n >= (1ll << 31)) return fail(DR_ERR_INVALID, "too many rays in one call");
  int grid = std::min<int64_t>(traceGrid(), (n + DR_TRACE_BLOCK - 1) / DR_TRACE_BLOCK);
  int rc = ensureSpill(sc, sc->ws, traceGrid());
  if (rc) return rc;
  DevBuf<DrRay> dR;
  DevBuf<DrHit> dH;
  DevBuf<uint32_t> work;
  HIP_TRY(dR.alloc(n));
  HIP_TRY(dH.alloc(n));
  HIP_TRY(work.alloc(8 * DR_WORK_STRIDE));
  HIP_TRY(hipMemcpy(dR.p, rays, n * sizeof(DrRay), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(work.p, 0, 8 * DR_WORK_STRIDE * sizeof(uint32_t)));
  HIP_TRY(hipMemset(sc->ctr.p, 0, sizeof(TraceCounters)));
  launch_intersect(sc->d, dR.p, n, dH.p, any_hit, sc->ws.spill.p, work.p, sc->ctr.p, grid, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, dH.p, n * sizeof(DrHit), hipMemcpyDeviceToHost));
  TraceCounters c;
  HIP_TRY(hipMemcpy(&c, sc->ctr.p, sizeof(c), hipMemcpyDeviceToHost));
  memset(&sc->stats, 0, sizeof(sc->stats));
  sc->traceEvents.clear();
  sc->renderEvents.clear();
  sc->eventsUsed = 0;
  sc->stats.closest_rays = c.closest_rays; sc->stats.any_rays = c.any_rays;
  sc->stats.closest_nodes = c.closest_nodes; sc->stats.any_nodes = c.any_nodes;
  sc->stats.closest_tris = c.closest_tris; sc->stats.any_tris = c.any_tris;
  sc->statsPending = false;
  HIP_TRY(hipMemset(sc->ctr.p, 0, sizeof(TraceCounters)));
  return DR_OK;
}

}  // extern "C"

// ===========================================================================
// dr_render_device, in units: RenderPlan (what this call runs: planRender, planBatches), prepareRender (workspace, pilot
// decision, uploads), BatchRunner (one batch through the stage loop), runPilot (the calibration batches) and the entry
// point, which strings them together and records what ran.
// ===========================================================================
namespace {

struct RenderPlan {
  DrScene* sc = nullptr;
  const DrRenderDesc* rd = nullptr;
  float* film = nullptr;
  hipStream_t s = nullptr;
  RenderParams rp;
  SampleForm sf;
  const LayoutOps* L = nullptr;  // state layout of the NEXT batch (the layout pilot decides it after the first calibration batch)
  int spp = 0;
  bool direct = false, dlSpec = false, envStage = false, hostBuf = false, packedTail = false;
  int needTail = 0;       // RNG draws a path can make beyond the sample vector (host-buffer mode: the recorded tail)
  bool layoutKnown = false;
  int maxStateWords = 0;  // words per slot the workspace is sized for (both layouts while the layout is not known)
  bool coherentCamera = false, lazyGen = false, overlapAny = false;
  bool calibrateTrace = false, measureLayout = false;
  int pilotSets = 0;      // calibration batches: warm-up, k_trace timed, k_trace3 timed, k_trace3c timed; the layout alone: one
  size_t calibPix = 0;    // pixels per calibration batch: the first pilotSets * calibPix entries of `pixels`
  std::vector<int2> pixels;
  size_t npixTotal = 0;
  uint64_t filmSamples = 0;
  uint32_t pixPerBatch = 0, cap = 0;
  uint64_t nBatches = 0;
  int tgrid = 0, sgrid = 0, nStages = 0;
  bool calibrate() const { return calibrateTrace || measureLayout; }
};

// What the call asks for, checked, as RenderParams + the flags every later unit reads; which pixels it traces.
int planRender(RenderPlan& P) {
  DrScene* sc = P.sc;
  const DrRenderDesc* rd = P.rd;
  const int spp = P.spp = rd->spp;
  if (spp <= 0 || (spp & (spp - 1)) != 0) return fail(DR_ERR_INVALID, "spp must be a power of two (low_discrepancy_sampler.dart:43-49)");
  if (spp > 4096) return fail(DR_ERR_UNSUPPORTED, "spp > 4096 (one pixel's shuffle table of a 16-pixel sampler group would not fit the LDS)");
  if (rd->integrator != DR_INTEGRATOR_PATH && rd->integrator != DR_INTEGRATOR_DIRECT_ALL && rd->integrator != DR_INTEGRATOR_DIRECT_ONE)
    return fail(DR_ERR_INVALID, "unknown integrator");
  if (rd->max_depth < 0 || rd->max_depth > 64) return fail(DR_ERR_INVALID, "max_depth out of range");
  P.direct = rd->integrator != DR_INTEGRATOR_PATH;
  const bool directOne = rd->integrator == DR_INTEGRATOR_DIRECT_ONE;  // strategy "one": the last entry of the scene's stage table, an arithmetic slot layout
  // DirectLighting over mirror / glass recurses through SpecularReflect / SpecularTransmit (integrator.dart:187-290):
  // an explicit per-slot stack and one round of the stage loop per vertex of the ray tree (k_shade_spec)
  P.dlSpec = P.direct && sc->hasSpecular;
  // k_env (dr_kernels.hip): the environment-map work of a plain-triangle scene's path stages runs in its own kernel
  P.envStage = rd->integrator == DR_INTEGRATOR_PATH && sc->d.hasEnv && !(sc->d.nquads || sc->d.hasSpec || sc->d.srec);
  // State layout (see LayoutOps): the four-slot sub-tiles for the renders whose lists thin out early.  Which one is MEASURED on
  // the render's own work: the first calibration batch -- rendered into the film like any other -- runs in the 64-slot layout
  // and its stage lists say how fast the paths die; when less than half of the slots are still alive at the second bounce the
  // rest of the render, and every later render of the scene, uses the four-slot sub-tiles (C5: 0.38 -> sp4; C2 0.80, C4: 64-slot).
  // DARTRAY_STATE_LAYOUT=64|4 forces one, dr_scene_set_state_layout stores one; renders too small for a pilot keep round 3's
  // rule (plain-triangle scenes under an environment map: sp4).
  const DrOpt layoutEnv = dr_opt("DARTRAY_STATE_LAYOUT");
  P.layoutKnown = layoutEnv || sc->stateLayout != 0 || rd->integrator != DR_INTEGRATOR_PATH;
  const bool sparseLayout = layoutEnv ? layoutEnv.toInt(0) == 4 : (sc->stateLayout ? sc->stateLayout == 4 : P.envStage);
  P.L = sparseLayout ? &kLayoutSp4 : &kLayout64;
  P.maxStateWords = P.layoutKnown ? P.L->stateWords : std::max(kLayout64.stateWords, kLayoutSp4.stateWords);

  RenderParams& rp = P.rp;
  memset(&rp, 0, sizeof(rp));
  memcpy(rp.r2c, rd->camera.raster_to_camera, sizeof(rp.r2c));
  memcpy(rp.c2w, rd->camera.camera_to_world, sizeof(rp.c2w));
  rp.lensRadius = rd->camera.lens_radius;
  rp.focalDistance = rd->camera.focal_distance;
  rp.shutterOpen = rd->camera.shutter_open;
  rp.shutterClose = rd->camera.shutter_close;
  rp.cameraType = rd->camera.type;
  if (rp.cameraType < DR_CAMERA_PERSPECTIVE || rp.cameraType > DR_CAMERA_ENVIRONMENT) return fail(DR_ERR_INVALID, "unknown camera type");
  rp_film(rp, rd->film);
  rp.integrator = rd->integrator;
  rp.maxDepth = rd->max_depth;
  rp.spp = spp;
  rp.sppShift = 0;
  while ((1 << rp.sppShift) < spp) ++rp.sppShift;
  rp.nLights = (int)sc->d.nlights;
  rp.nFloats = P.direct && !directOne ? sc->dlNFloats : dr_sample_floats(rd->integrator, sc->d.nlights);
  rp.n1D = directOne ? 5 : (P.direct ? sc->dlN1D : 14);
  rp.blocks = P.direct && !directOne && sc->dlMulti ? sc->dlBlocks.p : nullptr;
  rp.nBlocks = sc->dlNBlocks;
  rp.dstages = directOne ? sc->dlStages.p + sc->dlNStages : sc->dlStages.p;
  rp.nDirectStages = directOne ? (sc->d.nlights ? 1 : 0) : (P.direct ? sc->dlNStages : 0);
  rp.dlSpecular = P.dlSpec ? 1 : 0;
  rp.deferredNee = rd->integrator == DR_INTEGRATOR_PATH ? 1 : 0;
  rp.genMask = 0ull;
  rp.genSlowDraws = dr_opt("DARTRAY_GEN_SLOW_DRAWS").set ? 1 : 0;
  if (rd->integrator == DR_INTEGRATOR_PATH && !rp.blocks && !dr_opt("DARTRAY_GEN_ALL_BLOCKS").set) {
    // What the path kernels read of a pixel sample (dr_kernels.hip: k_raygen, load_shade_in, k_film): the image sample,
    // the lens sample of a thin-lens camera, and per SAMPLE_DEPTH level b <= maxDepth the light number, the light
    // sample (component + position), the BSDF and path directions; the two uComponent slots only where a material has
    // more than one lobe.  Never read: the time sample, the volume integrator's two slots, levels beyond maxDepth.
    const bool general = sc->d.nquads || sc->d.hasSpec || sc->d.srec;
    uint64_t m = 1ull | (rd->camera.lens_radius > 0.0 ? 2ull : 0ull);
    for (int b = 0; b < 3 && b <= rd->max_depth; ++b) {
      m |= 3ull << (3 + 4 * b);
      if (general) m |= 12ull << (3 + 4 * b);
      m |= 7ull << (3 + rp.n1D + 3 * b);
    }
    rp.genMask = m;
  }
  rp.samplerMode = rd->sampler_mode;
  rp.seed = (uint64_t)rd->seed;
  const int perNee = rp.nLights > 0 ? 7 : 0;
  P.needTail = rd->integrator == DR_INTEGRATOR_PATH && rd->max_depth >= 3 ? (rd->max_depth - 2) * (perNee + 3) + std::max(0, rd->max_depth - 3) : 0;
  rp.maxTail = rd->max_tail;

  // ---- which pixels does this call trace? ----
  P.hostBuf = rd->sampler_mode == DR_SAMPLER_HOST_BUFFER;
  if (P.hostBuf) {
    if (rd->nsamples <= 0 || rd->nsamples % spp != 0 || !rd->pixel_xy || !rd->sample_vec)
      return fail(DR_ERR_INVALID, "host-buffer sampler: nsamples must be a positive multiple of spp with pixel_xy and sample_vec set");
    if (rd->sample_stride < rp.nFloats) return fail(DR_ERR_INVALID, "sample_stride smaller than the sample vector");
    if (P.needTail > 0 && (!rd->tail || rd->max_tail < P.needTail))
      return fail(DR_ERR_INVALID, "host-buffer sampler: tail buffer missing or max_tail too small for max_depth");
    P.packedTail = P.needTail > 0 && rd->tail_offsets != nullptr;  // (its device buffers are sized per batch: BatchRunner::loadSamples)
    if (P.packedTail && rd->tail_offsets[rd->nsamples] < rd->tail_offsets[0])
      return fail(DR_ERR_INVALID, "host-buffer sampler: tail_offsets must be non-decreasing");
    const int64_t np = rd->nsamples / spp;
    P.pixels.resize(np);
    for (int64_t i = 0; i < np; ++i) P.pixels[i] = make_int2(rd->pixel_xy[2 * i], rd->pixel_xy[2 * i + 1]);
  } else if (rd->sampler_mode == DR_SAMPLER_COUNTER) {
    enumeratePixels(rp, rd, P.pixels);
  } else {
    return fail(DR_ERR_INVALID, "unknown sampler mode");
  }
  P.npixTotal = P.pixels.size();
  P.filmSamples = 0;
  for (const int2& p : P.pixels)
    if (p.x >= rp.left && p.x < rp.left + rp.width && p.y >= rp.top && p.y < rp.top + rp.height) P.filmSamples += spp;
  P.sgrid = g_numCU;  // the shade launchers size their grid per CU (DR_SHADE_GRID), grid-stride over the active list
  P.nStages = rd->integrator == DR_INTEGRATOR_PATH ? rd->max_depth + 2 : rp.nDirectStages + 1;
  if (P.nStages > 248 || 8 * DR_WORK_STRIDE * (1 + 2 * P.nStages) > N_COUNTERS_TRACE - 1024) return fail(DR_ERR_UNSUPPORTED, "too many stages");
  return DR_OK;
}

// The sample form and the batches: how many camera samples are in flight at once.
int planBatches(RenderPlan& P) {
  DrScene* sc = P.sc;
  const DrRenderDesc* rd = P.rd;
  const RenderParams& rp = P.rp;
  const int spp = P.spp;
  // Sample vectors: the on-device LD sampler stores permuted indices + scrambles (compact form) whenever every LD block
  // has one entry per pixel sample; host buffers and multi-entry blocks (DirectLighting with nsamples > 1) use floats.
  SampleForm& sf = P.sf;
  sf.compact = !P.hostBuf && rp.blocks == nullptr;
  if (!sf.compact && !P.hostBuf && spp > 1024)
    return fail(DR_ERR_UNSUPPORTED, "spp > 1024 with LD blocks of several entries per sample (DirectLighting with nsamples > 1): the float-form sampler's table exceeds the LDS");
  sf.nFloats = rp.nFloats;
  sf.nBlocks = 3 + rp.n1D + (rp.nFloats - 5 - rp.n1D) / 2;
  sf.idxShift = spp > 256 ? 1 : 0;
  // Camera samples in flight per batch.  The throughput end is one batch per image (2^28 slots: C2's whole sampler window, 64 GB of a
  // 288 GB MI355X); a scene's FIRST big render -- all a one-shot host ever does (Renderer.render once per task, dartray.dart:574) --
  // stays at 2^27 (C2: three batches, 21 GB, whose hipMalloc does not wait for the driver to scrub 64 GB: profiles/r05_alloc_probe.txt)
  // and the workspace grows to the image when the same scene is rendered again (a frame loop, bench.py's steps: + 3 % steady state).
  const DrOpt bitsOpt = dr_opt("DARTRAY_BATCH_BITS");
  const int slotBits = std::min(28, std::max(16, bitsOpt ? bitsOpt.toInt(28) : (sc->bigRenders == 0 ? 27 : 28)));
  uint64_t maxSlots = 1ull << slotBits;
  {
    // path state per camera sample: 164 B of ray / hit / NEE state, 20 B of queues and the sample vector (24 B of permuted
    // indices in the compact form, 4 B per float otherwise; + the RNG tail in host-buffer mode).  On a device with less free
    // memory the batch shrinks instead of failing (results do not depend on the batch size).
    const uint64_t tailPerSlot = !(P.hostBuf && P.needTail > 0) ? 0ull
                                 : (P.packedTail ? 16ull + 8ull * ((rd->tail_offsets[rd->nsamples] - rd->tail_offsets[0]) / (uint64_t)rd->nsamples + 1ull)
                                                 : (uint64_t)rd->max_tail * 8);
    const uint64_t perSlot = (uint64_t)P.maxStateWords * 4 + (uint64_t)(sf.svWords() + 15) / 16 + 20 + tailPerSlot + (P.hostBuf ? (uint64_t)rd->sample_stride * 4 : 0) +
                             (sf.compact ? (uint64_t)(16 * sf.nBlocks + spp - 1) / spp : 0) +  // scramble words + generator states, per (block, pixel)
                             (P.dlSpec ? (uint64_t)std::max(1, rd->max_depth) * sizeof(SpecFrame) + 12 : 0);
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
      const uint64_t have = (uint64_t)sc->ws.cap * ((uint64_t)sc->ws.stateWords * 4 + sc->ws.svWords / 16 + 20);
      const uint64_t budget = (uint64_t)(0.9 * (double)freeB) + have;
      // (+ 1/4: the slack that lets a slightly larger window still go as one batch, below)
      while (maxSlots > (1ull << 16) && std::min<uint64_t>(maxSlots + maxSlots / 4, (uint64_t)P.npixTotal * spp) * perSlot > budget) maxSlots >>= 1;
    }
  }
  // Equal batches, and no tiny tail batch: every stage launch costs ~0.4 ms of ramp-up and tail however small it is
  // (the sampler window of a 1024 x 1024 film is 1025 x 1025 pixels -- 2^20 + 2049).
  const uint64_t pixCapBatch = std::max<uint64_t>(1, maxSlots / spp);
  P.nBatches = (P.npixTotal + pixCapBatch - 1) / pixCapBatch;
  if (P.nBatches > 1 && P.npixTotal <= pixCapBatch + pixCapBatch / 4) P.nBatches = 1;
  P.pixPerBatch = (uint32_t)((P.npixTotal + P.nBatches - 1) / P.nBatches);
  P.cap = P.pixPerBatch * (uint32_t)spp;
  return DR_OK;
}

// Workspace, streams, the pilot decision (which reorders the pixels), and the uploads every batch reads.
int prepareRender(RenderPlan& P) {
  DrScene* sc = P.sc;
  const DrRenderDesc* rd = P.rd;
  const int spp = P.spp;
  const auto tAlloc0 = std::chrono::steady_clock::now();
  const uint32_t capBefore = sc->ws.cap;
  int rc = allocWorkspace(sc, sc->ws, P.cap, P.sf, P.pixPerBatch, rd->max_tail, P.hostBuf && P.needTail > 0 && !P.packedTail, P.maxStateWords);
  if (rc) return rc;
  if (dr_opt("DARTRAY_VERBOSE") && sc->ws.cap != capBefore) {
    (void)hipDeviceSynchronize();
    fprintf(stderr, "dartray_hip: path-state workspace for %u slots (%.1f GB) allocated in %.1f ms\n", sc->ws.cap,
            (double)sc->ws.tiles.n * 4.0e-9 + (double)sc->ws.cap * 20.0e-9,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tAlloc0).count());
  }
  if (P.dlSpec) {
    HIP_TRY(sc->ws.specFrames.alloc((size_t)sc->ws.cap * std::max(1, rd->max_depth) * DR_SPEC_FRAME_WORDS));
    HIP_TRY(sc->ws.specSp.alloc(sc->ws.cap));
    HIP_TRY(sc->ws.roundA.alloc(sc->ws.cap));
    HIP_TRY(sc->ws.roundB.alloc(sc->ws.cap));
  }
  if (P.envStage) HIP_TRY(sc->ws.envQ.alloc(sc->ws.cap));
  P.tgrid = traceGrid();
  rc = ensureSpill(sc, sc->ws, P.tgrid);
  if (rc) return rc;
  // the camera rays (a tile = 64 samples of one pixel) through the wave-coherent kernel k_trace_pk (DARTRAY_COHERENT_CAMERA=0: k_trace & co.)
  P.coherentCamera = !dr_opt("DARTRAY_COHERENT_CAMERA").isZero() && !P.dlSpec;
  // lazy sample generation (DARTRAY_LAZY_GEN=0: every block for every pixel up front): needs the device sampler's compact form, the keyed
  // per-(pixel, block) streams (a block that is left out disturbs no other) and k_trace_pk's marks of the camera rays that hit
  P.lazyGen = !P.hostBuf && P.sf.compact && P.rp.genMask != 0ull && rd->integrator == DR_INTEGRATOR_PATH && P.coherentCamera && !sc->d.nquads && spp >= 64 &&
              !dr_opt("DARTRAY_LAZY_GEN").isZero();
  // a stage's any-hit launch beside its closest-hit launch, on a second stream (DARTRAY_OVERLAP_ANY=0: one after the other)
  P.overlapAny = !dr_opt("DARTRAY_OVERLAP_ANY").isZero() && !P.dlSpec;
  if (P.overlapAny && !sc->s3) HIP_TRY(hipStreamCreateWithFlags(&sc->s3, hipStreamNonBlocking));
  // Which traversal kernel?  k_trace (one node per step, f32 filter) is issue bound and wins while the hot part of the tree stays in
  // cache; the pair kernels (half the dependent fetches) win on big incoherent trees (C4 hairball +26 %) and lose on others of the same
  // size; random probe rays mispredict both.  So the first big render of a big scene measures it ON ITS OWN WORK: small calibration
  // batches -- 64-pixel groups spread over the image -- are rendered first, into the film like every other batch: one with k_trace to
  // warm the caches, then one per candidate, timed; each ray kind keeps the kernel with the better time per algorithmic byte (the
  // device's own node / triangle counters of that batch).  Nothing is traced twice.  All kernels are bit-exact, so results do not depend
  // on the choice; dr_scene_set_trace_kernels / DARTRAY_TRACE_IMPL fix it (an N-rank host measures on rank 0 and hands the choice on).
  const DrOpt pilotOpt = dr_opt("DARTRAY_PILOT");  // 0: never; force: also on renders too small to need one (tests)
  const bool bigJob = (sc->d.nnodes >= (1u << 20) && (uint64_t)P.npixTotal * spp >= (1ull << 25)) || pilotOpt.is("force");
  const bool pilotOk = !pilotOpt.is("0") && !P.hostBuf && !P.dlSpec && bigJob && P.npixTotal >= 3 * 64 * 4;
  P.calibrateTrace = !sc->traceCalibrated && pilotOk && !dr_opt("DARTRAY_TRACE_IMPL") && sc->d.pairs && !sc->d.nquads;
  P.measureLayout = !P.layoutKnown && pilotOk;
  if (P.measureLayout) P.L = &kLayout64;  // the batch whose stage lists are measured runs in the 64-slot layout
  P.pilotSets = P.calibrateTrace ? 4 : 1;
  P.calibPix = 0;
  if (P.calibrate()) {
    // (at least 2^24 samples per calibration batch: with 2^22 the launches are so short that their tails decide -- the pair kernel,
    // whose rays are half as many fetches long, looked 10 % faster than k_trace<0> on C2 and is 12 % slower at full size)
    uint64_t pilotSamples = std::min<uint64_t>(1ull << 25, std::max<uint64_t>(1ull << 24, (uint64_t)P.npixTotal * spp / 16));
    pilotSamples = std::min<uint64_t>(pilotSamples, (uint64_t)(P.pixPerBatch / 64 * 64) * spp);
    const size_t totalGroups = P.npixTotal / 64;
    const size_t groups = std::min<size_t>(std::max<size_t>(1, (size_t)(pilotSamples / spp) / 64), totalGroups / 4);
    P.calibPix = groups * 64;
    std::vector<int2> ordered;
    ordered.reserve(P.npixTotal);
    std::vector<uint8_t> taken(totalGroups, 0);
    for (int set = 0; set < P.pilotSets; ++set)
      for (size_t g = 0; g < groups; ++g) {
        const size_t grp = (((size_t)P.pilotSets * g + set) * totalGroups) / ((size_t)P.pilotSets * groups);  // interleaved: the sets see the same regions
        taken[grp] = 1;
        ordered.insert(ordered.end(), P.pixels.begin() + grp * 64, P.pixels.begin() + grp * 64 + 64);
      }
    for (size_t grp = 0; grp < totalGroups; ++grp)
      if (!taken[grp]) ordered.insert(ordered.end(), P.pixels.begin() + grp * 64, P.pixels.begin() + grp * 64 + 64);
    ordered.insert(ordered.end(), P.pixels.begin() + totalGroups * 64, P.pixels.end());
    P.pixels.swap(ordered);
  }
  HIP_TRY(sc->ws.pix.alloc(P.npixTotal));
  HIP_TRY(hipMemcpyAsync(sc->ws.pix.p, P.pixels.data(), P.npixTotal * sizeof(int2), hipMemcpyHostToDevice, P.s));
  HIP_TRY(hipMemcpyAsync(sc->ws.filterTable.p, rd->film.filter_table, 256 * sizeof(float), hipMemcpyHostToDevice, P.s));
  HIP_TRY(hipStreamSynchronize(P.s));  // (the copies read host memory the caller and this plan own)
  return DR_OK;
}

// The traversal launches of one calibration batch, per ray kind (the coherent camera launch is every candidate's and is left out).
struct PilotTimes {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[2];
};

// One batch through the stage loop: gen_samples -> raygen -> trace(camera) -> [shade(b) -> (env) -> trace_closest || trace_any] -> film,
// no host round trips (DirectLighting over mirror / glass: one round of the loop per vertex of a slot's ray tree, one count read back
// per round).  pilot != null: a calibration batch -- a normal batch whose per-lane traversal launches are also collected in pilot->ev.
class BatchRunner {
 public:
  BatchRunner(RenderPlan& plan, Workspace& ws, const int2* pixDev, size_t firstPixel, uint32_t npixels, PilotTimes* pilotTimes)
      : P(plan), sc(plan.sc), rd(plan.rd), rp(plan.rp), L(*plan.L), w(ws), s(plan.s), p0(firstPixel), np(npixels), nslots(npixels * (uint32_t)plan.spp),
        pilot(pilotTimes), C(ws.counters.p) {
    st = makeState(w, P.sf, pixDev, nslots, P.hostBuf && P.needTail > 0, L.stateWords);
    const DrOpt scOpt = dr_opt("DARTRAY_STAGE_COUNTS");
    stageCounts = scOpt.toInt(0) > 0 ? scOpt.toInt(0) : (scOpt.set ? 1 : 0);
    slog.resize(stageCounts ? (size_t)P.nStages + 1 : 0);  // [0] = the camera rays' traversal, [b + 1] = stage b
    // A stage's two traversals are independent (closest hit of the continuation / MIS rays, occlusion of the shadow rays).  Side by
    // side on two streams the any-hit workgroups take the CU slots the closest-hit launch frees as its queue runs dry (a persistent
    // launch ends with its longest rays).  Calibration batches time each launch alone.
    sideBySide = P.overlapAny && !pilot;
  }
  int run();

 private:
  // DARTRAY_STAGE_COUNTS=1 (diagnostics): per stage the list lengths, the kernel times (this batch's own events) and -- with
  // DARTRAY_STAGE_COUNTS=2, which waits for the device after every stage -- the node visits / triangle tests of its traversals
  struct StageLog {
    hipEvent_t s0 = nullptr, sMid = nullptr, s1 = nullptr, c0 = nullptr, c1 = nullptr, a0 = nullptr, a1 = nullptr;
    TraceCounters ctr;
  };
  int loadSamples();
  int loadHostSamples();
  void genBounce(int b);
  hipEvent_t timed(int kind, hipEvent_t e0);
  hipEvent_t trace(const uint32_t* queue, const uint32_t* nQ, int any, hipStream_t ts, uint32_t* spill, hipEvent_t after = nullptr, bool coherent = false);
  void logTrace(StageLog& g, int any);
  void readCtrNow(TraceCounters* c);
  StageQueues stageQueues(int b, const uint32_t* roundQ, const uint32_t* nRound);
  int stage(int b, int round, const uint32_t* roundQ, const uint32_t* nRound);
  int specRound(int round, const uint32_t*& roundQ, const uint32_t*& nRound, bool& done);
  int finish();
  int printStageLog();

  RenderPlan& P;
  DrScene* sc;
  const DrRenderDesc* rd;
  const RenderParams& rp;
  const LayoutOps& L;
  Workspace& w;
  hipStream_t s;
  size_t p0;
  uint32_t np, nslots;
  PilotTimes* pilot;
  uint32_t* C;  // [0, 1024): stage queue counts; then 8 per-XCD work counters per trace launch; then k_env's counts
  BatchState st;
  uint32_t nGroups = 0;  // lazy generation: 64-pixel groups of this batch
  int wc = 0;            // work counters live at C[1024..], 8 per launch
  int stageCounts = 0;
  bool sideBySide = false;
  std::vector<StageLog> slog;
  TraceCounters ctrBase = {};
};

hipEvent_t BatchRunner::timed(int kind, hipEvent_t e0) {
  hipEvent_t e1 = sc->getEvent();
  (void)hipEventRecord(e1, s);
  sc->traceEvents.push_back({e0, e1, kind});
  return e1;
}

void BatchRunner::readCtrNow(TraceCounters* c) {
  if (stageCounts < 2) return;
  (void)hipStreamSynchronize(s);
  if (sc->s3) (void)hipStreamSynchronize(sc->s3);
  (void)hipMemcpy(c, sc->ctr.p, sizeof(TraceCounters), hipMemcpyDeviceToHost);
}

// lazy sample generation: the LD blocks of bounce b (light number, light component, light position, BSDF direction, path direction:
// the bits genMask gives the level) for the 64-pixel groups marked in alive[b]
void BatchRunner::genBounce(int b) {
  uint64_t m = (15ull << (3 + 4 * b)) | (7ull << (3 + rp.n1D + 3 * b));
  m &= rp.genMask;
  if (!m) return;
  hipEvent_t e0 = sc->getEvent();
  (void)hipEventRecord(e0, s);
  RenderParams rpB = rp;
  rpB.genMask = m;
  BatchState stB = st;
  stB.genAlive = w.alive.p + (size_t)b * nGroups;
  stB.markAlive = nullptr;
  L.gen_samples(rpB, stB, np, s);
  timed(3, e0);
}

// Host-buffer sampler: this batch's sample vectors (and the RNG tail) from the caller's memory.
int BatchRunner::loadHostSamples() {
  const int spp = P.spp;
  HIP_TRY(w.aosSamples.alloc((size_t)((P.cap + 63u) & ~63u) * rd->sample_stride));
  HIP_TRY(hipMemcpyAsync(w.aosSamples.p, rd->sample_vec + (size_t)p0 * spp * rd->sample_stride, (size_t)nslots * rd->sample_stride * sizeof(float),
                         hipMemcpyHostToDevice, s));
  L.transpose_samples(w.aosSamples.p, rd->sample_stride, st, rp.nFloats, s);
  if (P.needTail > 0 && P.packedTail) {
    // the batch's runs are one contiguous piece of the packed array: [off[first], off[first + nslots]).  The header promises
    // non-decreasing offsets and runs of at most max_tail values; a host that breaks the promise gets DR_ERR_INVALID here, not a
    // device read outside the piece that is copied (tailOff[slot + 1] - tailOff[slot] as a huge unsigned run).
    const uint64_t* off = rd->tail_offsets + (size_t)p0 * spp;
    for (uint32_t i = 0; i < nslots; ++i)
      if (off[i + 1] < off[i] || off[i + 1] - off[i] > (uint64_t)rd->max_tail)
        return fail(DR_ERR_INVALID, "host-buffer sampler: tail_offsets must be non-decreasing with runs of at most max_tail values");
    const uint64_t o0 = off[0], o1 = off[nslots];
    HIP_TRY(w.tail.alloc((size_t)(o1 - o0) + (size_t)rd->max_tail + 1));
    HIP_TRY(w.tailOff.alloc((size_t)nslots + 1));
    if (o1 > o0) HIP_TRY(hipMemcpyAsync(w.tail.p, rd->tail + o0, (size_t)(o1 - o0) * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.tailOff.p, off, ((size_t)nslots + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    st.tail = w.tail.p;
    st.tailOff = w.tailOff.p;
    st.tailBase = o0;
  } else if (P.needTail > 0) {
    HIP_TRY(hipMemcpyAsync(w.tail.p, rd->tail + (size_t)p0 * spp * rd->max_tail, (size_t)nslots * rd->max_tail * sizeof(double), hipMemcpyHostToDevice, s));
  }
  return DR_OK;
}

// The batch's pixel samples (host buffers, or the device LD sampler: everything now, or lazily) and its camera rays.
int BatchRunner::loadSamples() {
  hipEvent_t evGen = sc->getEvent();
  (void)hipEventRecord(evGen, s);
  if (P.hostBuf) {
    const int rc = loadHostSamples();
    if (rc) return rc;
  } else if (P.lazyGen) {
    // the image (+ lens) blocks for every pixel now; the blocks of bounce b once it is known which 64-pixel groups still have a path there
    RenderParams rpA = rp;
    rpA.genMask = rp.genMask & 3ull;
    L.gen_samples(rpA, st, np, s);
    nGroups = (np + 63u) / 64u;
    HIP_TRY(w.alive.alloc(3 * (size_t)nGroups));
    HIP_TRY(hipMemsetAsync(w.alive.p, 0, 3 * (size_t)nGroups, s));
    sc->genDoneHost += (unsigned long long)np * (unsigned)__builtin_popcountll(rpA.genMask);
    sc->genNamed += (unsigned long long)np * (unsigned)__builtin_popcountll(rp.genMask);
    st.markAlive = w.alive.p;  // k_trace_pk: the groups whose camera rays hit something
    st.markShift = (uint32_t)rp.sppShift + 6u;
  } else {
    L.gen_samples(rp, st, np, s);
    if (P.sf.compact && rp.genMask) {
      sc->genDoneHost += (unsigned long long)np * (unsigned)__builtin_popcountll(rp.genMask);
      sc->genNamed += (unsigned long long)np * (unsigned)__builtin_popcountll(rp.genMask);
    }
  }
  L.raygen(rp, st, s);
  timed(3, evGen);
  return DR_OK;
}

// One traversal launch over a queue (null: the batch's slots in order = the camera rays).  Returns its end event.
hipEvent_t BatchRunner::trace(const uint32_t* queue, const uint32_t* nQ, int any, hipStream_t ts, uint32_t* spill, hipEvent_t after, bool coherent) {
  hipEvent_t e0 = sc->getEvent(), e1 = sc->getEvent();
  (void)hipEventRecord(e0, ts);
  bool tookCoherent = false;
  if (!(coherent && L.trace_coherent(sc->d, st, queue, nQ, any, C + 1024 + 8 * DR_WORK_STRIDE * wc, sc->ctr.p, P.tgrid, ts))) {
    L.trace(sc->d, st, queue, nQ, any, spill, C + 1024 + 8 * DR_WORK_STRIDE * (wc++), sc->ctr.p, P.tgrid, ts);
    // (lazy sample generation counts on k_trace_pk's marks: should the coherent kernel ever decline a launch that was to leave
    // them, every group counts as alive -- all blocks are generated, nothing is skipped)
    if (coherent && !any && st.markAlive) (void)hipMemsetAsync(st.markAlive, 1, nGroups, ts);
  } else {
    ++wc;  // (k_trace_pk took this queue: the camera rays)
    tookCoherent = true;
  }
  (void)hipEventRecord(e1, ts);
  // (the pilot compares the per-lane kernels: the coherent camera launch is the same kernel for every candidate and would only
  // compress the ratios its thresholds look at)
  if (pilot && !tookCoherent) pilot->ev[any].push_back({e0, e1});
  sc->traceEvents.push_back({e0, e1, tookCoherent && !any ? 6 : any, after});
  return e1;
}

void BatchRunner::logTrace(StageLog& g, int any) {
  (any ? g.a0 : g.c0) = sc->traceEvents.back().e0;
  (any ? g.a1 : g.c1) = sc->traceEvents.back().e1;
}

StageQueues BatchRunner::stageQueues(int b, const uint32_t* roundQ, const uint32_t* nRound) {
  StageQueues q;
  // a stage's four counters sit ~1 KB apart: every wave adds to all four in one round trip (stage_flush), and same-line atomics serialise
  auto cnt = [&](int j, int stg) { return C + 248 * j + stg; };
  q.activeIn = b == 0 ? roundQ : ((b - 1) & 1 ? w.activeB.p : w.activeA.p);
  q.nActiveIn = b == 0 ? nRound : cnt(0, b - 1);
  q.activeOut = (b & 1) ? w.activeB.p : w.activeA.p;
  q.nActiveOut = cnt(0, b);
  q.closestQ = w.closestQ.p;
  q.nClosest = cnt(1, b);
  q.anyQ = w.anyQ.p;
  q.nAny = cnt(2, b);
  q.work = cnt(3, b);
  q.ctr = sc->ctr.p;
  q.envQ = P.envStage ? w.envQ.p : nullptr;
  q.nEnv = C + N_COUNTERS_TRACE + 64 * b;
  return q;
}

// Stage b: shade the active list (+ the environment-map kernel), generate the next bounce's sample blocks where paths are alive,
// trace the continuation / MIS rays and the shadow rays the stage queued.
int BatchRunner::stage(int b, int round, const uint32_t* roundQ, const uint32_t* nRound) {
  const StageQueues q = stageQueues(b, roundQ, nRound);
  const bool log = stageCounts && round == 0;
  hipEvent_t evS = sc->getEvent();
  (void)hipEventRecord(evS, s);
  if (rd->integrator == DR_INTEGRATOR_PATH) L.shade_path(sc->d, rp, st, q, b, P.sgrid, s);
  else L.shade_direct(sc->d, rp, st, q, b, P.sgrid, s);
  hipEvent_t evMid = nullptr;
  if (stageCounts && P.envStage) {
    evMid = sc->getEvent();
    (void)hipEventRecord(evMid, s);
  }
  if (P.envStage) L.env(sc->d, rp, st, q, b, P.sgrid, s);
  hipEvent_t evS1 = timed(2, evS);
  if (P.lazyGen && round == 0 && b < 2 && b + 1 <= rd->max_depth) {  // bounce b + 1's blocks for the groups in this stage's output list
    L.mark_alive(q.activeOut, q.nActiveOut, (uint32_t)rp.sppShift + 6u, w.alive.p + (size_t)(b + 1) * nGroups, s);
    genBounce(b + 1);
  }
  if (log) {
    slog[b + 1].s0 = evS;
    slog[b + 1].sMid = evMid;
    slog[b + 1].s1 = evS1;
  }
  if (b + 1 >= P.nStages) return DR_OK;
  if (sideBySide) {
    hipEvent_t eS = sc->getEvent(), eA = sc->getEvent();
    (void)hipEventRecord(eS, s);
    (void)hipStreamWaitEvent(sc->s3, eS, 0);
    hipEvent_t closestEnd = trace(q.closestQ, q.nClosest, 0, s, w.spill.p);
    if (log) logTrace(slog[b + 1], 0);
    trace(q.anyQ, q.nAny, 1, sc->s3, w.spill.p + w.spillHalf, closestEnd);
    if (log) logTrace(slog[b + 1], 1);
    (void)hipEventRecord(eA, sc->s3);
    (void)hipStreamWaitEvent(s, eA, 0);
  } else {
    trace(q.closestQ, q.nClosest, 0, s, w.spill.p);
    if (log) logTrace(slog[b + 1], 0);
    trace(q.anyQ, q.nAny, 1, s, w.spill.p);
    if (log) logTrace(slog[b + 1], 1);
  }
  if (log) readCtrNow(&slog[b + 1].ctr);
  return DR_OK;
}

// DirectLighting over mirror / glass, the end of a round: k_shade_spec pops / pushes every slot's frame stack and lists the slots
// whose child ray the next round traces.  done: no slot launched a child.
int BatchRunner::specRound(int round, const uint32_t*& roundQ, const uint32_t*& nRound, bool& done) {
  StageQueues q;
  memset(&q, 0, sizeof(q));
  q.activeIn = roundQ;
  q.nActiveIn = nRound;
  uint32_t* nextQ = (round & 1) ? w.roundB.p : w.roundA.p;
  uint32_t* nNext = C + 1008 + (round & 1);
  HIP_TRY(hipMemsetAsync(nNext, 0, sizeof(uint32_t), s));
  q.activeOut = nextQ;
  q.nActiveOut = nNext;
  q.closestQ = w.closestQ.p;  // unused: the child rays are the next round's list
  q.nClosest = C + 1010;
  q.anyQ = w.anyQ.p;
  q.nAny = C + 1011;
  q.ctr = sc->ctr.p;
  hipEvent_t evS = sc->getEvent();
  (void)hipEventRecord(evS, s);
  L.shade_spec(sc->d, rp, st, q, P.sgrid, s);
  timed(2, evS);
  uint32_t live = 0;  // (a synchronous read-back per round: this is not the throughput path)
  HIP_TRY(hipMemcpyAsync(&live, nNext, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  done = live == 0;
  roundQ = nextQ;
  nRound = nNext;
  return DR_OK;
}

int BatchRunner::printStageLog() {
  std::vector<uint32_t> hc(N_COUNTERS);
  HIP_TRY(hipStreamSynchronize(s));
  if (sc->s3) HIP_TRY(hipStreamSynchronize(sc->s3));
  HIP_TRY(hipMemcpy(hc.data(), C, N_COUNTERS * sizeof(uint32_t), hipMemcpyDeviceToHost));
  const size_t batch = (size_t)sc->stats.batches;
  for (int b = 0; b < P.nStages; ++b)
    fprintf(stderr, "stage_counts batch %zu stage %d: in %u active_out %u closest %u any %u env %u\n", batch, b, b == 0 ? nslots : hc[248 * 0 + b - 1],
            hc[248 * 0 + b], hc[248 * 1 + b], hc[248 * 2 + b], hc[N_COUNTERS_TRACE + 64 * b]);
  // the kernel times of the same stages (side-by-side any-hit launches overlap the closest-hit ones: DARTRAY_OVERLAP_ANY=0 gives each its
  // own time) and, with DARTRAY_STAGE_COUNTS=2, the traversal work of each stage
  auto ms = [](hipEvent_t a, hipEvent_t b) {
    float t = 0.f;
    return a && b && hipEventElapsedTime(&t, a, b) == hipSuccess ? (double)t : 0.0;
  };
  for (int i = 0; i <= P.nStages; ++i) {
    const StageLog& g = slog[i];
    const double shade = g.sMid ? ms(g.s0, g.sMid) : ms(g.s0, g.s1), env = g.sMid ? ms(g.sMid, g.s1) : 0.0;
    fprintf(stderr, "stage_times batch %zu stage %d: shade %.4f env %.4f closest %.4f any %.4f ms", batch, i - 1, shade, env, ms(g.c0, g.c1), ms(g.a0, g.a1));
    if (stageCounts >= 2 && (i == 0 || g.c0 || g.a0)) {
      const TraceCounters& p = i ? slog[i - 1].ctr : ctrBase;
      fprintf(stderr, "; closest rays %llu nodes %llu tris %llu any rays %llu nodes %llu tris %llu", g.ctr.closest_rays - p.closest_rays,
              g.ctr.closest_nodes - p.closest_nodes, g.ctr.closest_tris - p.closest_tris, g.ctr.any_rays - p.any_rays, g.ctr.any_nodes - p.any_nodes,
              g.ctr.any_tris - p.any_tris);
    }
    fprintf(stderr, "\n");
  }
  return DR_OK;
}

// After the last stage: the sampler statistics, the film, the diagnostics.
int BatchRunner::finish() {
  if (P.lazyGen) {  // statistics: the (pixel, block) pairs the three genBounce calls came to
    uint32_t nb[3];
    for (int b = 0; b < 3; ++b) nb[b] = (uint32_t)__builtin_popcountll(((15ull << (3 + 4 * b)) | (7ull << (3 + rp.n1D + 3 * b))) & rp.genMask);
    L.sum_alive(w.alive.p, nGroups, np, nb, sc->ctr.p, s);
  }
  hipEvent_t evF = sc->getEvent();
  (void)hipEventRecord(evF, s);
  L.film(rp, st, sc->ws.filterTable.p, np, P.film, s);
  timed(4, evF);
  sc->stats.batches++;
  if (stageCounts) {
    const int rc = printStageLog();
    if (rc) return rc;
  }
  HIP_TRY(hipGetLastError());
  if (P.hostBuf) HIP_TRY(hipStreamSynchronize(s));  // host buffers of the next batch reuse the staging area
  return DR_OK;
}

int BatchRunner::run() {
  HIP_TRY(hipMemsetAsync(C, 0, N_COUNTERS * sizeof(uint32_t), s));
  int rc = loadSamples();
  if (rc) return rc;
  // DirectLighting over mirror / glass: one round of the stage loop per vertex of a slot's ray tree (at most 2^maxDepth rounds, like the
  // recursion itself); `roundQ` lists the slots whose (camera or child) ray this round traces.  Everything else: one round.
  const uint32_t* roundQ = nullptr;
  const uint32_t* nRound = nullptr;
  if (P.dlSpec) HIP_TRY(hipMemsetAsync(w.specSp.p, 0, (size_t)w.cap * sizeof(int32_t), s));
  for (int round = 0;; ++round) {
    if (round > 0) {  // the stage counters are reused every round; the round lists' counts live behind them
      HIP_TRY(hipMemsetAsync(C, 0, 1000 * sizeof(uint32_t), s));
      HIP_TRY(hipMemsetAsync(C + 1024, 0, (N_COUNTERS - 1024) * sizeof(uint32_t), s));  // (work counters and k_env's counts)
      wc = 0;
    }
    if (stageCounts && round == 0) readCtrNow(&ctrBase);
    trace(roundQ, nRound, 0, s, w.spill.p, nullptr, P.coherentCamera && roundQ == nullptr);  // camera rays (or this round's child rays)
    if (stageCounts && round == 0) {  // (before genBounce pushes its own event)
      logTrace(slog[0], 0);
      readCtrNow(&slog[0].ctr);
    }
    if (P.lazyGen && round == 0) {
      st.markAlive = nullptr;
      genBounce(0);
    }
    for (int b = 0; b < P.nStages; ++b) {
      rc = stage(b, round, roundQ, nRound);
      if (rc) return rc;
    }
    if (!P.dlSpec) break;
    bool done = false;
    rc = specRound(round, roundQ, nRound, done);
    if (rc) return rc;
    if (done) break;
  }
  return finish();
}

// The calibration batches of a scene's first big render (prepareRender decided that there are some): part of the render -- nothing is
// traced twice -- and the measurement that picks the state layout and, per ray kind, the traversal kernel.
struct PilotResult {
  int setsRun = 0;
  double perByte[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};  // [closest / any][k_trace / k_trace3 / k_trace3c]: ms per algorithmic GB
  float ms[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  // any-hit rays of the first two batches, both through k_trace<1>: batch 0 far child first, batch 1 in the reference order
  double anyMsPerRayFar = 0.0, anyMsPerRayRef = 0.0;
};

// The choice, from the calibration batches' times per algorithmic byte.
void pickTraceKernels(DrScene* sc, const PilotResult& R) {
  const double(&perByte)[2][3] = R.perByte;
  for (int kind = 0; kind < 2; ++kind)
    for (int c = 0; c < 3; ++c) {
      sc->calibMs[kind][c] = R.ms[kind][c];
      sc->calibPerGB[kind][c] = (float)perByte[kind][c];
    }
  // closest-hit rays: a pair kernel needs 5 % on k_trace<0>
  sc->d.traceKernel[0] = perByte[0][1] < 0.95 * perByte[0][0] ? 3u : 2u;
  // ... and has a second form (round 4): the cold ray state in LDS, six workgroups per CU -- at full size 2.5 - 3 % ahead of k_trace3<0>
  // on C5 (728 against 762 - 786 ms) and level on C4 (123.2 / 123.5), while the calibration batches put it anywhere from 2 % behind to
  // 1 % ahead: it keeps the pair family's place unless k_trace3<0> beats it by 5 % there
  if (perByte[0][2] > 0.0) {
    const double best3 = std::min(perByte[0][1], perByte[0][2]);
    if (best3 < 0.95 * perByte[0][0]) sc->d.traceKernel[0] = perByte[0][2] < 1.05 * perByte[0][1] ? 5u : 3u;
  }
  // The any-hit rays.  Their calibration launches are the least reliable of the pilot -- shadow rays are short, a small launch is
  // mostly ramp-up and tail, and the two families come out within a few per cent of each other on the cache-resident scenes (C2:
  // k_trace3a 6 - 12 % ahead in the calibration batches of five boxes, level at full size) while small launches understate the pair
  // kernel on the big incoherent tree (C4: -2 ... +6 % in a calibration batch, +25 % at full size).  So they stay in the FAMILY the
  // closest-hit rays chose -- k_trace<1> beside k_trace<0>, k_trace3a beside k_trace3<0> / k_trace3c -- and cross over only when
  // their own calibration batch says so by more than 15 %.
  const bool pairFamily = sc->d.traceKernel[0] != 2u;
  const double own = pairFamily ? perByte[1][1] : perByte[1][0], other = pairFamily ? perByte[1][0] : perByte[1][1];
  const bool cross = other > 0.0 && own > 0.0 && other < 0.85 * own;
  sc->d.traceKernel[1] = (pairFamily != cross) ? 3u : 2u;
  // ... and their visit ORDER (round 6): intersectP's boolean does not depend on it (bvh_accel.dart:167-226 never touches the ray), the work
  // of a ray that finds an occluder does.  The pilot's first batch -- the cache warm-up -- ran its any-hit rays far child first, the second
  // in the reference order, both through k_trace<1>: where the far child first is cheaper per ray even in the cold batch (ratio below
  // 0.97), the scene's any-hit rays take it -- in whichever kernel family they run (the order is a property of the rays and the tree).
  // Measured at full size, kernels forced (profiles/r06_far_first_ab.txt): C5 (the courtyard under the sky: 42 % of the shadow rays are
  // occluded and visit 35 % fewer nodes) any-hit 505.8 -> 398.9 ms, 1180 -> 1254 Msamples/s; C2 96.8 -> 93.1 ms and C4 101.0 -> 98.2 ms
  // although their occluded rays visit 11 - 13 % MORE nodes that way -- they test 3 - 4 % fewer triangles, and an f64 triangle test costs
  // several node visits.  Pilot ratios of the same boxes: C5 0.62, C2 0.90, C4 0.94.
  sc->calibFarFirst = R.anyMsPerRayRef > 0.0 ? (float)(R.anyMsPerRayFar / R.anyMsPerRayRef) : 0.f;
  if (R.anyMsPerRayRef > 0.0 && R.anyMsPerRayFar > 0.0 && R.anyMsPerRayFar < 0.97 * R.anyMsPerRayRef) sc->d.traceKernel[1] = sc->d.traceKernel[1] == 3u ? 7u : 6u;
  sc->traceCalibrated = true;
}

int runPilot(RenderPlan& P, PilotResult& R) {
  DrScene* sc = P.sc;
  hipStream_t s = P.s;
  hipEvent_t evP0 = sc->getEvent(), evP1 = sc->getEvent();
  HIP_TRY(hipEventRecord(evP0, s));
  auto readCtr = [&](TraceCounters* c) -> int {
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(c, sc->ctr.p, sizeof(TraceCounters), hipMemcpyDeviceToHost));
    return DR_OK;
  };
  const uint32_t keepKernel[2] = {sc->d.traceKernel[0], sc->d.traceKernel[1]};
  auto abandon = [&](int code) {  // an error in the middle: the scene keeps the choice it had, not a forced one
    sc->d.traceKernel[0] = keepKernel[0];
    sc->d.traceKernel[1] = keepKernel[1];
    return code;
  };
  for (int set = 0; set < P.pilotSets; ++set) {  // warm-up (k_trace), k_trace timed, k_trace3 timed, k_trace3c timed (its any-hit rays: k_trace3a again)
    // where the pair kernel has just lost clearly to k_trace<0> (C2: 8 - 10 % behind) its cold-state sibling is not timed: k_trace3c is
    // never more than a few per cent from k_trace3<0>, and the batch is a quarter of the pilot's cost.  Its pixels stay in the ordinary batches.
    if (set == 3 && P.calibrateTrace && R.perByte[0][1] > 1.05 * R.perByte[0][0]) break;
    ++R.setsRun;
    const int impl = set == 2 ? 3 : (set == 3 ? 5 : 2);
    const int col = set == 2 ? 1 : (set == 3 ? 2 : 0);
    if (P.calibrateTrace) {
      sc->d.traceKernel[0] = (uint32_t)impl;
      sc->d.traceKernel[1] = impl == 5 ? 3u : (set == 0 ? 6u : (uint32_t)impl);  // (the warm-up batch: k_trace<1> far child first, see pickTraceKernels)
    }
    TraceCounters c0, c1;
    int prc = readCtr(&c0);
    if (prc) return abandon(prc);
    PilotTimes pt;
    prc = BatchRunner(P, sc->ws, sc->ws.pix.p + set * P.calibPix, set * P.calibPix, (uint32_t)P.calibPix, &pt).run();
    if (prc) return abandon(prc);
    prc = readCtr(&c1);
    if (prc) return abandon(prc);
    if (set == 0 && P.measureLayout) {
      // the batch's stage lists (still in the counters): how many of its slots are alive at the second bounce?
      uint32_t alive2 = 0;
      if (hipMemcpy(&alive2, sc->ws.counters.p + 1, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)  // entries of stage 1's output list
        return abandon(fail(DR_ERR_HIP, "layout pilot: counter read-back failed"));
      sc->layoutDensity = (float)((double)alive2 / ((double)P.calibPix * P.spp));
      sc->stateLayout = sc->layoutDensity < 0.5f ? 4 : 64;
      P.L = sc->stateLayout == 4 ? &kLayoutSp4 : &kLayout64;
      if (dr_opt("DARTRAY_VERBOSE"))
        fprintf(stderr, "dartray_hip: state-layout pilot: %.3f of a batch's slots alive at the second bounce -> %s\n", sc->layoutDensity,
                sc->stateLayout == 4 ? "four-slot line-grouped sub-tiles (sp4)" : "64-slot runs");
    }
    if (P.calibrateTrace && set <= 1) {  // any-hit time per ray, far child first (batch 0) against the reference order (batch 1)
      float sum = 0.f;
      for (auto& e : pt.ev[1]) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e.first, e.second);
        sum += t;
      }
      const double rays = (double)(c1.any_rays - c0.any_rays);
      (set == 0 ? R.anyMsPerRayFar : R.anyMsPerRayRef) = rays > 0.0 ? (double)sum / rays : 0.0;
    }
    if (set == 0 || !P.calibrateTrace) continue;
    // the per-lane kernels' own work: the batch's totals without what k_trace_pk traced of them (the camera rays)
    const double bytes[2] = {32.0 * (double)((c1.closest_nodes - c0.closest_nodes) - (c1.pk_nodes[0] - c0.pk_nodes[0])) +
                                 48.0 * (double)((c1.closest_tris - c0.closest_tris) - (c1.pk_tris[0] - c0.pk_tris[0])),
                             32.0 * (double)(c1.any_nodes - c0.any_nodes) + 48.0 * (double)(c1.any_tris - c0.any_tris)};
    for (int kind = 0; kind < 2; ++kind) {
      float sum = 0.f;
      for (auto& e : pt.ev[kind]) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e.first, e.second);
        sum += t;
      }
      R.ms[kind][col] = sum;
      R.perByte[kind][col] = bytes[kind] > 0.0 ? (double)sum / (bytes[kind] * 1.0e-9) : 0.0;
    }
  }
  if (P.calibrateTrace) pickTraceKernels(sc, R);
  HIP_TRY(hipEventRecord(evP1, s));
  sc->traceEvents.push_back({evP0, evP1, 5});  // DrRenderStats.pilot_ms: the time of the calibration batches
  if (P.calibrateTrace && dr_opt("DARTRAY_VERBOSE"))
    fprintf(stderr, "dartray_hip: traversal pilot (%d x %zu samples, rendered into the film), ms per algorithmic GB of the per-lane kernels: closest v2 %.4f / v3 %.4f / v3c %.4f -> v%u; "
            "any hit v2 %.4f / v3 %.4f, far child first / reference order per ray %.3f -> v%u\n", R.setsRun, P.calibPix * (size_t)P.spp, R.perByte[0][0], R.perByte[0][1], R.perByte[0][2], sc->d.traceKernel[0],
            R.perByte[1][0], R.perByte[1][1], sc->calibFarFirst, sc->d.traceKernel[1]);
  return DR_OK;
}

}  // namespace

extern "C" {

int dr_render_device(DrScene* sc, const DrRenderDesc* rd, void* film_dev, void* hip_stream) {
  if (!sc || !rd || !film_dev) return fail(DR_ERR_INVALID, "null argument");
  RenderPlan P;
  P.sc = sc;
  P.rd = rd;
  P.film = (float*)film_dev;
  P.s = (hipStream_t)hip_stream;
  int rc = planRender(P);
  if (rc) return rc;
  // recycle the events of earlier renders once they have completed (or when too many are pending)
  if (sc->lastEvent && !sc->traceEvents.empty()) {
    hipError_t q = hipEventQuery(sc->lastEvent);
    if (q != hipSuccess && sc->eventsUsed > 8192) q = hipEventSynchronize(sc->lastEvent);
    if (q == hipSuccess) sc->foldEvents();
    (void)hipGetLastError();  // hipErrorNotReady is not an error
  }
  sc->statsPending = true;
  hipEvent_t evStart = sc->getEvent(), evStop = sc->getEvent();
  sc->renderEvents.push_back({evStart, evStop});
  sc->lastEvent = evStop;
  HIP_TRY(hipEventRecord(evStart, P.s));
  if (P.npixTotal == 0) {
    HIP_TRY(hipEventRecord(evStop, P.s));
    return DR_OK;
  }
  rc = planBatches(P);
  if (rc) return rc;
  rc = prepareRender(P);
  if (rc) return rc;
  PilotResult pilot;
  if (P.calibrate()) {
    rc = runPilot(P, pilot);
    if (rc) return rc;
  }
  // (a calibration set that was skipped left its pixels to the ordinary batches)
  for (size_t p0 = (size_t)pilot.setsRun * P.calibPix; p0 < P.npixTotal; p0 += P.pixPerBatch) {
    const uint32_t np = (uint32_t)std::min<size_t>(P.pixPerBatch, P.npixTotal - p0);
    rc = BatchRunner(P, sc->ws, sc->ws.pix.p + p0, p0, np, nullptr).run();
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(evStop, P.s));
  sc->stats.camera_samples += (uint64_t)P.npixTotal * P.spp;
  sc->stats.film_samples += P.filmSamples;
  if ((uint64_t)P.npixTotal * P.spp >= (1ull << 25)) sc->bigRenders++;  // (planBatches: the next render of this scene may take the whole image as one batch)
  sc->lastInfo[0] = P.L == &kLayoutSp4 ? 4 : 64;
  sc->lastInfo[1] = P.L->trace_kernel_id(sc->d, 0);
  sc->lastInfo[2] = P.L->trace_kernel_id(sc->d, 1);
  sc->lastInfo[3] = -1;  // (reserved: rounds 4-5 reported the treelet-parked traversal's parking rounds here)
  sc->lastInfo[4] = pilot.setsRun;
  sc->lastInfo[5] = (int32_t)std::min<uint64_t>(0x7fffffff, P.nBatches);
  sc->lastInfo[6] = P.tgrid / std::max(1, g_numCU);
  sc->lastInfo[7] = (P.overlapAny ? 1 : 0) | (P.coherentCamera && !sc->d.nquads ? 2 : 0) | (P.lazyGen ? 8 : 0);
  return DR_OK;
}

int dr_enumerate_pixels(const DrRenderDesc* rd, int32_t* out_xy, uint64_t cap, uint64_t* n_out) {
  if (!rd || !n_out) return fail(DR_ERR_INVALID, "null argument");
  RenderParams rp;
  memset(&rp, 0, sizeof(rp));
  rp_film(rp, rd->film);
  std::vector<int2> pixels;
  enumeratePixels(rp, rd, pixels);
  *n_out = pixels.size();
  if (out_xy) {
    if (cap < pixels.size()) return fail(DR_ERR_INVALID, "pixel buffer too small");
    for (size_t i = 0; i < pixels.size(); ++i) {
      out_xy[2 * i] = pixels[i].x;
      out_xy[2 * i + 1] = pixels[i].y;
    }
  }
  return DR_OK;
}

int dr_get_stats(DrScene* sc, DrRenderStats* out) {
  if (!sc || !out) return fail(DR_ERR_INVALID, "null argument");
  if (sc->statsPending) {
    if (sc->lastEvent) HIP_TRY(hipEventSynchronize(sc->lastEvent));
    TraceCounters c;
    HIP_TRY(hipMemcpy(&c, sc->ctr.p, sizeof(c), hipMemcpyDeviceToHost));
    sc->stats.closest_rays = c.closest_rays; sc->stats.any_rays = c.any_rays;
    sc->stats.closest_nodes = c.closest_nodes; sc->stats.any_nodes = c.any_nodes;
    sc->stats.closest_tris = c.closest_tris; sc->stats.any_tris = c.any_tris;
    sc->stats.shade_items = c.shade_items; sc->stats.shade_vertices = c.shade_vertices;
    sc->stats.shade_cont = c.shade_cont; sc->stats.shade_mis = c.shade_mis; sc->stats.shade_shadow = c.shade_shadow;
    sc->pkRays = c.pk_rays[0];
    sc->pkNodes = c.pk_nodes[0];
    sc->pkTris = c.pk_tris[0];
    sc->genDone = sc->genDoneHost + c.gen_pixel_blocks;
    sc->foldEvents();
    sc->statsPending = false;
    shade_prof_dump();
    trace_prof_dump();
    sp4::shade_prof_dump();
    sp4::trace_prof_dump();
  }
  *out = sc->stats;
  return DR_OK;
}

int dr_reset_stats(DrScene* sc) {
  if (!sc) return fail(DR_ERR_INVALID, "null argument");
  if (sc->lastEvent) HIP_TRY(hipEventSynchronize(sc->lastEvent));
  memset(&sc->stats, 0, sizeof(sc->stats));
  sc->traceEvents.clear();
  sc->renderEvents.clear();
  sc->eventsUsed = 0;
  sc->lastEvent = nullptr;
  sc->statsPending = false;
  sc->pkMs = 0.0;
  sc->pkLaunches = 0;
  sc->pkRays = sc->pkNodes = sc->pkTris = 0;
  sc->genDone = sc->genDoneHost = sc->genNamed = 0;
  HIP_TRY(hipMemset(sc->ctr.p, 0, sizeof(TraceCounters)));
  return DR_OK;
}

int dr_scene_get_sampler_stats(DrScene* sc, double out[2]) {
  if (!sc || !out) return fail(DR_ERR_INVALID, "null argument");
  DrRenderStats st;
  const int rc = dr_get_stats(sc, &st);
  if (rc) return rc;
  out[0] = (double)sc->genDone;
  out[1] = (double)sc->genNamed;
  return DR_OK;
}

int dr_scene_get_coherent_stats(DrScene* sc, double out[5]) {
  if (!sc || !out) return fail(DR_ERR_INVALID, "null argument");
  DrRenderStats st;
  const int rc = dr_get_stats(sc, &st);  // (waits for the renders in flight and folds their events, like dr_get_stats)
  if (rc) return rc;
  out[0] = (double)sc->pkRays;
  out[1] = (double)sc->pkNodes;
  out[2] = (double)sc->pkTris;
  out[3] = (double)sc->pkLaunches;
  out[4] = sc->pkMs;
  return DR_OK;
}

int dr_film_resolve_device(const void* film_dev, int64_t npixels, void* rgb_dev, void* hip_stream) {
  if (!film_dev || !rgb_dev || npixels < 0) return fail(DR_ERR_INVALID, "null argument");
  if (npixels == 0) return DR_OK;
  launch_film_resolve((const float*)film_dev, npixels, (float*)rgb_dev, (hipStream_t)hip_stream);
  HIP_TRY(hipGetLastError());
  return DR_OK;
}

int dr_render(DrScene* sc, const DrRenderDesc* rd, float* film_out, float* rgb_out) {
  if (!sc || !rd || !film_out) return fail(DR_ERR_INVALID, "null argument");
  RenderParams rp;
  memset(&rp, 0, sizeof(rp));
  rp_film(rp, rd->film);
  const int64_t npix = (int64_t)rp.width * rp.height;
  DevBuf<float> film, rgb;
  HIP_TRY(film.alloc(4 * npix));
  HIP_TRY(hipMemset(film.p, 0, 4 * npix * sizeof(float)));
  int rc = dr_render_device(sc, rd, film.p, nullptr);
  if (rc) return rc;
  if (rgb_out) {
    HIP_TRY(rgb.alloc(3 * npix));
    rc = dr_film_resolve_device(film.p, npix, rgb.p, nullptr);
    if (rc) return rc;
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(film_out, film.p, 4 * npix * sizeof(float), hipMemcpyDeviceToHost));
  if (rgb_out) HIP_TRY(hipMemcpy(rgb_out, rgb.p, 3 * npix * sizeof(float), hipMemcpyDeviceToHost));
  return DR_OK;
}

int dr_render_sharded(DrScene* sc, const DrRenderDesc* rd, int32_t root, float* film_out, float* rgb_out) {
  if (!sc || !rd) return fail(DR_ERR_INVALID, "null argument");
  const int world = dr_comm_world(), rank = dr_comm_rank();
  if (world > 1 ? (root < 0 || root >= world) : root != 0) return fail(DR_ERR_INVALID, "dr_render_sharded: root out of range");
  RenderParams rp;
  memset(&rp, 0, sizeof(rp));
  rp_film(rp, rd->film);
  const int64_t npix = (int64_t)rp.width * rp.height;
  DevBuf<float> film, rgb;
  HIP_TRY(film.alloc(4 * npix));
  HIP_TRY(hipMemset(film.p, 0, 4 * npix * sizeof(float)));
  int rc = dr_render_device(sc, rd, film.p, nullptr);
  if (rc) return rc;
  if (world > 1) {
    rc = dr_film_reduce(film.p, npix, root, nullptr);
    if (rc) return rc;
  }
  const bool isRoot = world > 1 ? rank == root : true;
  if (isRoot && rgb_out) {
    HIP_TRY(rgb.alloc(3 * npix));
    rc = dr_film_resolve_device(film.p, npix, rgb.p, nullptr);
    if (rc) return rc;
  }
  HIP_TRY(hipDeviceSynchronize());
  if (isRoot && film_out) HIP_TRY(hipMemcpy(film_out, film.p, 4 * npix * sizeof(float), hipMemcpyDeviceToHost));
  if (isRoot && rgb_out) HIP_TRY(hipMemcpy(rgb_out, rgb.p, 3 * npix * sizeof(float), hipMemcpyDeviceToHost));
  return DR_OK;
}

int dr_copy_bandwidth(uint64_t bytes, int32_t iters, double* gbps_out) {
  if (g_device < 0) return fail(DR_ERR_NO_DEVICE, "dr_init has not been called");
  if (!gbps_out || bytes < 16 || iters <= 0) return fail(DR_ERR_INVALID, "bad argument");
  uint64_t n4 = bytes / 16;
  DevBuf<float4> a, b;
  HIP_TRY(a.alloc(n4));
  HIP_TRY(b.alloc(n4));
  HIP_TRY(hipMemset(a.p, 1, n4 * 16));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  launch_copy(a.p, b.p, n4, 0);
  HIP_TRY(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch_copy(a.p, b.p, n4, 0);
  HIP_TRY(hipEventRecord(e1, 0));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *gbps_out = (2.0 * (double)n4 * 16.0 * iters) / ((double)ms * 1.0e-3) / 1.0e9;
  return DR_OK;
}

}  // extern "C"
