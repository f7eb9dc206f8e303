// oracle/dartray_oracle.cpp
//
// TEST INFRASTRUCTURE ONLY -- NOT PRODUCT CODE.
//
// CPU restatement of the DartRay hot path
//   SamplerRenderer.render -> PathIntegrator.Li -> BVHAccel.intersect / Triangle.intersect
// used as (a) the parity oracle for the HIP kernels in dartray_amd/csrc and
// (b) the "port" CPU baseline timed by bench.py.  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may load this library.
//
// PARITY UNPINNED: the reference ships no golden vectors, no known-answer
// tests and cannot be executed here (no Dart SDK).  What
// pins this file instead: (1) line-by-line fidelity to the cited reference
// source, including its numerics contract (every arithmetic expression is
// evaluated in f64; Vector/Point/Normal/Spectrum/Float32List stores round to
// f32), (2) hand-derivable KATs in tests/, (3) BVH == brute force self checks,
// (4) a second restatement written independently from the Dart text in plain
// Python (tests/golden/dart_restatement.py) that must agree with this file bit
// for bit on per-sample Li, films, hit records, sample vectors, RNG draws, BVH
// node arrays and pixel orders over eight scenes (matte, mirror, glass, env
// map, thin lens, both integrators; tests/test_restatement.py): it removes the
// single-reader risk, it does not pin parity to the Dart VM.
//
// Numerics contract: compile with
//   g++ -O2 -ffp-contract=off  (no fast-math)  -- Dart never fuses mul+add.
//
// All citations are relative to /root/reference/lib/.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <vector>

namespace {

typedef double D;

// A store into a Float32List (vector.dart:27, spectrum.dart:27): round to f32.
static inline D r32(D x) { return (D)(float)x; }

static const D kInf = std::numeric_limits<double>::infinity();  // common.dart:26 (1.0e500)
static const D INV_PI = 0.31830988618379067154;                  // common.dart:23
static const D ONE_MINUS_EPSILON = 0.9999999403953552;           // montecarlo.dart:23
static const D kPi = 3.141592653589793;                          // dart:math pi

// ---------------------------------------------------------------------------
// Vector / Point / Normal (core/vector.dart, point.dart, normal.dart).
// Invariant: the three components are always f32-representable values.
// ---------------------------------------------------------------------------
struct V {
  D x, y, z;
  D operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V vec(D x, D y, D z) { return V{r32(x), r32(y), r32(z)}; }           // vector.dart:29-34
static inline V vadd(const V& a, const V& b) { return vec(a.x + b.x, a.y + b.y, a.z + b.z); }  // :57-60
static inline V vsub(const V& a, const V& b) { return vec(a.x - b.x, a.y - b.y, a.z - b.z); }  // :62-65
static inline V vmul(const V& a, D f) { return vec(a.x * f, a.y * f, a.z * f); }   // :67-68
static inline V vdiv(const V& a, D f) { return vec(a.x / f, a.y / f, a.z / f); }   // :70-71
static inline V vneg(const V& a) { return vec(-a.x, -a.y, -a.z); }                  // :73-74
static inline D vdot(const V& a, const V& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // :153-155
static inline D vabsdot(const V& a, const V& b) { return std::fabs(vdot(a, b)); }   // :157-159
static inline D vlen2(const V& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }    // :80-81
static inline D vlen(const V& a) { return std::sqrt(vlen2(a)); }                    // :83
static inline V vcross(const V& a, const V& b) {                                    // :161-171
  return vec((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
static inline V vnormalize(const V& v) { return vdiv(v, vlen(v)); }                 // :173

// ---------------------------------------------------------------------------
// Spectrum == RGBColor (spectrum.dart:40,53-58; rgb_color.dart:136-176).
// ---------------------------------------------------------------------------
struct S {
  D r, g, b;
};
static inline S rgb(D r, D g, D b) { return S{r32(r), r32(g), r32(b)}; }
static inline S sadd(const S& a, const S& b) { return rgb(a.r + b.r, a.g + b.g, a.b + b.b); }
static inline S smul(const S& a, const S& b) { return rgb(a.r * b.r, a.g * b.g, a.b * b.b); }
static inline S smulD(const S& a, D s) { return rgb(a.r * s, a.g * s, a.b * s); }
static inline S sdivD(const S& a, D s) { return rgb(a.r / s, a.g / s, a.b / s); }
static inline bool sblack(const S& a) { return !(a.r != 0.0 || a.g != 0.0 || a.b != 0.0); }  // rgb_color.dart:171-176
static inline D slum(const S& a) { return 0.212671 * a.r + 0.715160 * a.g + 0.072169 * a.b; }  // :167-169
static inline bool snan(const S& a) { return std::isnan(a.r) || std::isnan(a.g) || std::isnan(a.b); }
static inline D clampD(D v, D lo, D hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------
// dart:math Random on the Dart VM (NOT in /root/reference: un-vendored SDK
// dependency, version unpinned; restated from the published algorithm --
// SURVEY.md Appendix E).  Parity never depends on it: sample values and
// in-Li draws are explicit inputs of the GPU path.
// ---------------------------------------------------------------------------
static inline uint64_t mix64(uint64_t n) {  // Thomas Wang 64-bit mix
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
  explicit DartRandom(int64_t seed = 5489) { reseed(seed); }  // rng.dart:29-30
  void reseed(int64_t seed) {
    uint64_t hash = mix64((uint64_t)seed);
    if (hash == 0) hash = 0x5A17;
    lo = (uint32_t)(hash & 0xffffffffu);
    hi = (uint32_t)(hash >> 32);
    step(); step(); step(); step();  // "crank a couple of times"
  }
  void step() {
    uint64_t s = 0xffffda61ULL * (uint64_t)lo + (uint64_t)hi;
    lo = (uint32_t)(s & 0xffffffffu);
    hi = (uint32_t)(s >> 32);
  }
  uint32_t nextInt(uint64_t max) {
    if ((max & (~max + 1)) == max) {  // power of two
      step();
      return (uint32_t)(lo & (max - 1));
    }
    uint64_t rnd32, result;
    do {
      step();
      rnd32 = lo;
      result = rnd32 % max;
    } while ((rnd32 - result + max) > (1ULL << 32));
    return (uint32_t)result;
  }
  D nextDouble() {
    D a = (D)nextInt(1u << 26);
    D b = (D)nextInt(1u << 27);
    return (a * 134217728.0 + b) / 9007199254740992.0;
  }
  D randomFloat() { return nextDouble(); }               // rng.dart:36-38
  uint32_t randomUint() { return nextInt(0xffffffffULL); }  // rng.dart:40-42
};

// Key derivation of the build's own "counter" sampler mode (not in the
// reference: the reference's single serial stream cannot be parallelised,
// SURVEY.md section 7.2).  Each (pixel, LD block) and each (pixel, sample) owns an
// independent DartRandom stream seeded with this hash.  The HIP path uses the
// identical derivation (dartray_amd/csrc/dr_rng.h).
static inline int64_t counter_key(uint64_t seed, uint64_t a, uint64_t b, uint64_t kind) {
  uint64_t h = mix64(seed ^ 0x9E3779B97F4A7C15ULL);
  h = mix64(h ^ (a * 0xD1B54A32D192ED03ULL + kind));
  h = mix64(h ^ (b * 0x8CB92BA72F3D8DD7ULL + 0x5851F42D4C957F2DULL));
  return (int64_t)(h & 0x7fffffffffffffffULL);
}

// ---------------------------------------------------------------------------
// montecarlo.dart
// ---------------------------------------------------------------------------
static inline D VanDerCorput(uint64_t n, uint64_t scramble) {  // montecarlo.dart:495-504
  n = (n << 16) | (n >> 16);
  n = ((n & 0x00ff00ffULL) << 8) | ((n & 0xff00ff00ULL) >> 8);
  n = ((n & 0x0f0f0f0fULL) << 4) | ((n & 0xf0f0f0f0ULL) >> 4);
  n = ((n & 0x33333333ULL) << 2) | ((n & 0xccccccccULL) >> 2);
  n = ((n & 0x55555555ULL) << 1) | ((n & 0xaaaaaaaaULL) >> 1);
  n ^= scramble;
  return std::min((D)((n >> 8) & 0xffffff) / (D)(1 << 24), ONE_MINUS_EPSILON);
}
static inline D Sobol2(uint64_t n, uint64_t scramble) {  // montecarlo.dart:486-493
  for (uint64_t v = 1ULL << 31; n != 0; n >>= 1, v ^= v >> 1) {
    if ((n & 0x1) != 0) scramble ^= v;
  }
  return std::min((D)((scramble >> 8) & 0xffffff) / (D)(1 << 24), ONE_MINUS_EPSILON);
}

// Shuffle on a Float32List (montecarlo.dart:294-303).
template <class RNG>
static void Shuffle(float* samples, int offset, int count, int dims, RNG& rng) {
  for (int i = 0; i < count; ++i) {
    int other = i + (int)(rng.randomUint() % (uint32_t)(count - i));
    for (int j = 0; j < dims; ++j) {
      float s = samples[offset + dims * i + j];
      samples[offset + dims * i + j] = samples[offset + dims * other + j];
      samples[offset + dims * other + j] = s;
    }
  }
}
template <class RNG>
static void LDShuffleScrambled1D(int nSamples, int nPixel, float* samples, RNG& rng) {  // :524-536
  uint64_t scramble = rng.randomUint();
  for (int i = 0; i < nSamples * nPixel; ++i) samples[i] = (float)VanDerCorput((uint64_t)i, scramble);
  for (int i = 0; i < nPixel; ++i) Shuffle(samples, i * nSamples, nSamples, 1, rng);
  Shuffle(samples, 0, nPixel, nSamples, rng);
}
template <class RNG>
static void LDShuffleScrambled2D(int nSamples, int nPixel, float* samples, RNG& rng) {  // :539-551
  uint64_t s0 = rng.randomUint();
  uint64_t s1 = rng.randomUint();
  for (int i = 0; i < nSamples * nPixel; ++i) {  // Sample02 :507-511
    samples[2 * i + 0] = (float)VanDerCorput((uint64_t)i, s0);
    samples[2 * i + 1] = (float)Sobol2((uint64_t)i, s1);
  }
  for (int i = 0; i < nPixel; ++i) Shuffle(samples, 2 * i * nSamples, nSamples, 2, rng);
  Shuffle(samples, 0, nPixel, 2 * nSamples, rng);
}

static void ConcentricSampleDisk(D u1, D u2, D* dx, D* dy) {  // montecarlo.dart:155-201
  D r, theta;
  D sx = 2 * u1 - 1;
  D sy = 2 * u2 - 1;
  if (sx == 0.0 && sy == 0.0) {
    *dx = 0.0;
    *dy = 0.0;
    return;
  }
  if (sx >= -sy) {
    if (sx > sy) {
      r = sx;
      if (sy > 0.0) theta = sy / r; else theta = 8.0 + sy / r;
    } else {
      r = sy;
      theta = 2.0 - sx / r;
    }
  } else {
    if (sx <= sy) {
      r = -sx;
      theta = 4.0 - sy / r;
    } else {
      r = -sy;
      theta = 6.0 + sx / r;
    }
  }
  theta *= kPi / 4.0;
  *dx = r * std::cos(theta);
  *dy = r * std::sin(theta);
}
static V CosineSampleHemisphere(D u1, D u2) {  // montecarlo.dart:203-209
  D dx, dy;
  ConcentricSampleDisk(u1, u2, &dx, &dy);
  D z = std::sqrt(std::max(0.0, 1.0 - dx * dx - dy * dy));
  return vec(dx, dy, z);
}
static inline D PowerHeuristic(int nf, D fPdf, int ng, D gPdf) {  // montecarlo.dart:480-484
  D f = nf * fPdf;
  D g = ng * gPdf;
  return (f * f) / (f * f + g * g);
}

// Distribution1D (montecarlo.dart:25-98); upper_bound (common.dart:304-333).
struct Distribution1D {
  std::vector<float> func, cdf;
  D funcInt = 0;
  int count = 0;
  void init(const std::vector<D>& f) {
    count = (int)f.size();
    func.resize(count);
    for (int i = 0; i < count; ++i) func[i] = (float)f[i];
    cdf.assign(count + 1, 0.0f);
    cdf[0] = 0.0f;
    for (int i = 1; i < count + 1; ++i) cdf[i] = (float)((D)cdf[i - 1] + (D)func[i - 1] / (D)count);
    funcInt = (D)cdf[count];
    if (funcInt == 0.0) {
      for (int i = 1; i < count + 1; ++i) cdf[i] = (float)((D)i / (D)count);
    } else {
      for (int i = 1; i < count + 1; ++i) cdf[i] = (float)((D)cdf[i] / funcInt);
    }
  }
  int upper_bound(D value, int last) const {
    if (cdf.empty()) return -1;
    if (cdf.size() == 1) return 0;
    int first = 0;
    int cnt = last;
    while (cnt > 0) {
      int index = first;
      int step = cnt >> 1;
      index += step;
      if (!(value < (D)cdf[index])) {  // !compare(value, list[index]) with less_than
        first = ++index;
        cnt -= step + 1;
      } else {
        cnt = step;
      }
    }
    return first;
  }
  int sampleDiscrete(D u) const {  // montecarlo.dart:82-92
    int ptr = upper_bound(u, count + 1);
    return std::max(0, ptr - 1);
  }
  D sampleContinuous(D u, D* pdf, int* off) const {  // montecarlo.dart:50-80
    int ptr = upper_bound(u, count + 1);
    int offset = std::max(0, ptr - 1);
    if (offset == count) offset = count - 1;
    if (off) *off = offset;
    D dc = ((D)cdf[offset + 1] - (D)cdf[offset]);
    D du = 0.0;
    if (dc != 0.0) du = (u - (D)cdf[offset]) / dc;
    if (pdf) *pdf = (D)func[offset] / funcInt;
    return (offset + du) / count;
  }
};

// ---------------------------------------------------------------------------
// Ray (core/ray.dart:27-71)
// ---------------------------------------------------------------------------
struct Ray {
  V o, d;
  D mint, maxt;
  D time;
  int depth;
};
static inline V pointAt(const Ray& r, D t) { return vadd(r.o, vmul(r.d, t)); }  // ray.dart:66-67

// DifferentialGeometry subset that matters with constant textures
// (differential_geometry.dart:77-102).
struct DG {
  V p, dpdu, dpdv, nn;
  D u = 0, v = 0;  // parametric coordinates (Triangle.getShadingGeometry reads them back, triangle.dart:291)
};

// ---------------------------------------------------------------------------
// Scene storage
// ---------------------------------------------------------------------------
struct Mesh {
  S Kd;
  D sigma;
  bool reverse;
  int light;  // index into lights or -1
  // per-vertex shading data of the TriangleMesh (triangle_mesh.dart:195-203): N / S stay in OBJECT space and are
  // transformed by objectToWorld at shading time (triangle.dart:303-317); uvs replace the default (0,0),(1,0),(1,1)
  bool hasN = false, hasS = false, hasUV = false;
  float o2w[16], w2o[16];
  // material: 0 matte (matte_material.dart), 1 mirror (mirror_material.dart), 2 glass (glass_material.dart),
  // 3 plastic (plastic_material.dart: Kd, Ks in Kr, roughness in ior)
  int matType = 0;
  S Kr{0, 0, 0}, Kt{0, 0, 0};
  D ior = 1.5;
};
struct Prim {  // one refined Triangle, or a quadric Shape, wrapped in a GeometricPrimitive
  uint32_t v[3];
  int mesh;
  int src_tri;  // triangle index inside its mesh (before the refine reversal)
  int quadric = -1;  // index into Scene::quadrics (Shape.canIntersect() == true: not refined)
};
// Sphere (shapes/sphere.dart:23-38) / Disk (shapes/disk.dart:23-29): constructor-derived fields are Dart doubles.
struct Quadric {
  int kind = 0;  // 1 = sphere, 2 = disk
  float o2w[16], w2o[16];  // objectToWorld.m and its mInv (== worldToObject.m, dartray.dart:383-384)
  D radius = 0, zmin = 0, zmax = 0, thetaMin = 0, thetaMax = 0, phiMax = 0, height = 0, innerRadius = 0;
  bool reverse = false;
};
struct LinearNode {  // bvh_accel.dart:533-538
  V bmin, bmax;
  uint32_t offset;
  int nPrimitives;
  int axis;
};
struct Counters {
  uint64_t closest_rays = 0, any_rays = 0;
  uint64_t closest_nodes = 0, any_nodes = 0;
  uint64_t closest_tris = 0, any_tris = 0;
  uint64_t light_tris = 0;
  uint64_t camera_samples = 0;
  uint64_t max_stack = 0;  // deepest todo stack seen (the reference allocates 64 entries, bvh_accel.dart:120)
  void add(const Counters& o) {
    closest_rays += o.closest_rays; any_rays += o.any_rays;
    closest_nodes += o.closest_nodes; any_nodes += o.any_nodes;
    closest_tris += o.closest_tris; any_tris += o.any_tris;
    light_tris += o.light_tris; camera_samples += o.camera_samples;
    if (o.max_stack > max_stack) max_stack = o.max_stack;
  }
};
// The hot loops count into a per-thread tally (several threads may render one scene: bench.py's threaded CPU baseline;
// a shared counter would bounce its cache line between the cores on every node visit); CounterScope folds it into the
// scene's totals when an entry point returns.
static thread_local Counters t_ctr;

struct Light {  // DiffuseAreaLight (diffuse_area_light.dart:36-70) + ShapeSet (shape_set.dart:24-51), or the InfiniteAreaLight
  int kind = 0;  // 0 = diffuse area light, 1 = infinite area light (Scene::env), 2 = point light (point_light.dart),
                 // 3 = spot light (spot_light.dart), 4 = distant light (distant_light.dart)
  V lightPos{0, 0, 0};  // Point / Spot: lightToWorld(0,0,0); Distant: lightDir.  Lemit holds the intensity / radiance
  float w2l[16];        // Spot: worldToLight
  D cosTotalWidth = 0, cosFalloffStart = 0;
  S Lemit;
  int nSamples;
  std::vector<int> shapes;  // indices into Scene::lightTris
  std::vector<D> areas;
  D area;
  Distribution1D areaDistribution;
};
struct LightTri {  // one Shape of a ShapeSet: a refined triangle or a quadric
  uint32_t v[3];
  bool reverse;
  int quadric = -1;
  int mesh = -1;  // its TriangleMesh (for the uvs)
};

// InfiniteAreaLight (lights/infinite_area_light.dart) with its MIPMap radiance map (core/mipmap.dart,
// core/spectrum_image.dart) and Distribution2D (core/montecarlo.dart:222-268).
struct EnvLight {
  int levels = 0;
  std::vector<int> lw, lh;
  std::vector<std::vector<float>> pyramid;  // RGB f32 texels per level (SpectrumImage.data)
  S L{1, 1, 1};
  float l2w[16], w2l[16];
  std::vector<Distribution1D> cond;  // pConditionalV
  Distribution1D marginal;           // pMarginal
  int nSamples = 1;

  static int emod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }  // Dart % is Euclidean
  S texel(int level, int s, int t) const {  // mipmap.dart:184-207, TEXTURE_REPEAT
    s = emod(s, lw[level]);
    t = emod(t, lh[level]);
    const float* p = &pyramid[level][3 * ((size_t)t * lw[level] + s)];
    return S{p[0], p[1], p[2]};
  }
  S triangle(int level, D s, D t) const {  // mipmap.dart:342-355
    level = std::min(std::max(level, 0), levels - 1);
    s = s * lw[level] - 0.5;
    t = t * lh[level] - 0.5;
    int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
    D ds = s - s0, dt = t - t0;
    return sadd(sadd(sadd(smulD(texel(level, s0, t0), ((1.0 - ds) * (1.0 - dt))),
                          smulD(texel(level, s0, t0 + 1), ((1.0 - ds) * dt))),
                     smulD(texel(level, s0 + 1, t0), (ds * (1.0 - dt)))),
                smulD(texel(level, s0 + 1, t0 + 1), (ds * dt)));
  }
  static D Log2(D x) { static const D invLog2 = 1.0 / std::log(2.0); return std::log(x) * invLog2; }  // common.dart:98-103
  S lookup(D s, D t, D width) const {  // mipmap.dart:209-224
    D level = levels - 1 + Log2(std::max(width, 1.0e-8));
    if (level < 0) return triangle(0, s, t);
    else if (level >= levels - 1) return texel(levels - 1, 0, 0);
    int iLevel = (int)std::floor(level);
    D delta = level - iLevel;
    return sadd(smulD(triangle(iLevel, s, t), (1.0 - delta)), smulD(triangle(iLevel + 1, s, t), delta));
  }
  S radiance(D u, D v, D width = 0.0) const { return smul(lookup(u, v, width), L); }  // infinite_area_light.dart:180-182

  // _resampleWeights (mipmap.dart:360-384): four Lanczos taps (texture.dart:27-39, tau = 2) per new texel, normalised
  struct ResampleWeight { int firstTexel; D weight[4]; };
  static D Lanczos(D x, D tau = 2.0) {
    x = std::fabs(x);
    if (x < 1.0e-5) return 1.0;
    if (x > 1.0) return 0.0;
    x *= kPi;
    D s = std::sin(x * tau) / (x * tau);
    D lanczos = std::sin(x) / x;
    return s * lanczos;
  }
  static std::vector<ResampleWeight> resampleWeights(int oldres, int newres) {
    std::vector<ResampleWeight> wt(newres);
    D filterwidth = 2.0;
    for (int i = 0; i < newres; ++i) {
      D center = (i + 0.5) * oldres / newres;
      wt[i].firstTexel = (int)std::floor((center - filterwidth) + 0.5);
      for (int j = 0; j < 4; ++j) {
        D pos = wt[i].firstTexel + j + 0.5;
        wt[i].weight[j] = Lanczos((pos - center) / filterwidth);
      }
      D invSumWts = 1.0 / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
      for (int j = 0; j < 4; ++j) wt[i].weight[j] *= invSumWts;
    }
    return wt;
  }
  // The resampling branch of MIPMap.texture (mipmap.dart:71-138), wrapMode TEXTURE_REPEAT: zoom in s into the first yres rows of the
  // new image, then in t through workData, clamped to [0, inf).  Spectrum arithmetic: every `*` and `+` a Float32List store.
  static std::vector<float> resampleToPow2(const float* img, int xres, int yres, int* sOut, int* tOut) {
    auto RoundUpPow2 = [](int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; };
    auto dartMod = [](int a, int n) { int r = a % n; return r < 0 ? r + n : r; };
    int sPow2 = RoundUpPow2(xres), tPow2 = RoundUpPow2(yres);
    std::vector<ResampleWeight> sWeights = resampleWeights(xres, sPow2);
    std::vector<float> res(3 * (size_t)sPow2 * tPow2, 0.f);
    auto at = [&](std::vector<float>& v, size_t i) { return S{v[3 * i], v[3 * i + 1], v[3 * i + 2]}; };
    auto put = [&](std::vector<float>& v, size_t i, const S& x) { v[3 * i] = (float)x.r; v[3 * i + 1] = (float)x.g; v[3 * i + 2] = (float)x.b; };
    for (int t = 0; t < yres; ++t)
      for (int s = 0; s < sPow2; ++s) {
        put(res, (size_t)t * sPow2 + s, S{0, 0, 0});
        for (int j = 0; j < 4; ++j) {
          int origS = dartMod(sWeights[s].firstTexel + j, xres);
          if (origS >= 0 && origS < xres) {
            size_t k = (size_t)t * xres + origS;
            S px = smulD(S{img[3 * k], img[3 * k + 1], img[3 * k + 2]}, sWeights[s].weight[j]);
            put(res, (size_t)t * sPow2 + s, sadd(at(res, (size_t)t * sPow2 + s), px));
          }
        }
      }
    std::vector<ResampleWeight> tWeights = resampleWeights(yres, tPow2);
    std::vector<S> workData(tPow2);
    for (int s = 0; s < sPow2; ++s) {
      for (int t = 0; t < tPow2; ++t) {
        workData[t] = S{0, 0, 0};
        for (int j = 0; j < 4; ++j) {
          int offset = dartMod(tWeights[t].firstTexel + j, yres);
          if (offset >= 0 && offset < yres) workData[t] = sadd(workData[t], smulD(at(res, (size_t)offset * sPow2 + s), tWeights[t].weight[j]));
        }
      }
      auto clamp0 = [](D v) { return (v < 0.0 || v == 0.0) ? 0.0 : v; };  // num.clamp(0.0, INFINITY): compareTo puts -0.0 below 0.0; NaN stays
      for (int t = 0; t < tPow2; ++t) put(res, (size_t)t * sPow2 + s, S{clamp0(workData[t].r), clamp0(workData[t].g), clamp0(workData[t].b)});
    }
    *sOut = sPow2;
    *tOut = tPow2;
    return res;
  }

  // MIPMap.texture (mipmap.dart:61-170) + _setRadianceMap (infinite_area_light.dart:283-307)
  bool init(const float* texelsIn, int w, int h) {
    if (w <= 0 || h <= 0) return false;
    std::vector<float> resampled;
    const float* texels = texelsIn;
    if ((w & (w - 1)) || (h & (h - 1))) {  // !IsPowerOf2(xres) || !IsPowerOf2(yres)
      resampled = resampleToPow2(texelsIn, w, h, &w, &h);
      texels = resampled.data();
    }
    levels = 1 + (int)Log2((D)std::max(w, h));
    lw.assign(levels, 0); lh.assign(levels, 0);
    pyramid.resize(levels);
    lw[0] = w; lh[0] = h;
    pyramid[0].assign(texels, texels + 3 * (size_t)w * h);
    for (int i = 1; i < levels; ++i) {
      int sRes = std::max(1, lw[i - 1] / 2), tRes = std::max(1, lh[i - 1] / 2);
      lw[i] = sRes; lh[i] = tRes;
      pyramid[i].assign(3 * (size_t)sRes * tRes, 0.f);
      for (int t = 0, p = 0; t < tRes; ++t)
        for (int s = 0; s < sRes; ++s, ++p) {
          // (texel(a) + texel(b) + texel(c) + texel(d)) * 0.25 -- but SpectrumImage.operator[] returns ONE shared
          // static RGBColor (spectrum_image.dart:104-113,131), so when `texel(a) + texel(b)` is evaluated the
          // receiver already holds b's values: the sum is 2b + c + d.  Only levels >= 1 are affected.
          S b = texel(i - 1, 2 * s + 1, 2 * t), c = texel(i - 1, 2 * s, 2 * t + 1), d = texel(i - 1, 2 * s + 1, 2 * t + 1);
          S v = smulD(sadd(sadd(sadd(b, b), c), d), 0.25);
          pyramid[i][3 * p] = (float)v.r; pyramid[i][3 * p + 1] = (float)v.g; pyramid[i][3 * p + 2] = (float)v.b;
        }
    }
    D filter = 1.0 / std::max(w, h);
    std::vector<float> img((size_t)w * h);
    for (int v = 0; v < h; ++v) {
      D vp = (D)v / h;
      D sinTheta = std::sin(kPi * (v + 0.5) / h);
      for (int u = 0; u < w; ++u) {
        D up = (D)u / w;
        img[u + (size_t)v * w] = (float)slum(radiance(up, vp, filter));
        img[u + (size_t)v * w] = (float)((D)img[u + (size_t)v * w] * sinTheta);
      }
    }
    // Distribution2D(img, width, height) (montecarlo.dart:223-237)
    cond.resize(h);
    std::vector<D> marg(h);
    for (int v = 0; v < h; ++v) {
      std::vector<D> row(w);
      for (int u = 0; u < w; ++u) row[u] = img[u + (size_t)v * w];
      cond[v].init(row);
      marg[v] = (D)(float)cond[v].funcInt;
    }
    marginal.init(marg);
    return true;
  }
  V xf(const float* m, const V& p) const {  // Transform.transformVector
    D x = p.x, y = p.y, z = p.z;
    return vec(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z);
  }
  static D SphericalTheta(const V& v) { return std::acos(clampD(v.z, -1.0, 1.0)); }  // vector.dart:195-197
  static D SphericalPhi(const V& v) {                                                 // vector.dart:199-202
    D p = std::atan2(v.y, v.x);
    return (p < 0.0) ? p + 2.0 * kPi : p;
  }
  S Le(const V& dir) const {  // infinite_area_light.dart:84-90
    V wh = vnormalize(xf(w2l, dir));
    D s = SphericalPhi(wh) * 0.15915494309189533577;  // INV_TWOPI
    D t = SphericalTheta(wh) * INV_PI;
    return radiance(s, t);
  }
  // sampleLAtPoint (infinite_area_light.dart:92-131): returns Ls, sets wi and pdf
  S sampleL(D u0, D u1, V* wi, D* pdf) const {
    D pdfs1, pdfs0;
    int voff;
    D v = marginal.sampleContinuous(u1, &pdfs1, &voff);
    D u = cond[voff].sampleContinuous(u0, &pdfs0, nullptr);
    D mapPdf = pdfs0 * pdfs1;
    if (mapPdf == 0.0) { *pdf = 0.0; return S{0, 0, 0}; }  // pdf[0] is left at its initial 0.0
    D theta = v * kPi, phi = u * 2.0 * kPi;
    D costheta = std::cos(theta), sintheta = std::sin(theta);
    D sinphi = std::sin(phi), cosphi = std::cos(phi);
    *wi = xf(l2w, vec(sintheta * cosphi, sintheta * sinphi, costheta));
    if (sintheta == 0.0) *pdf = 0.0;
    else *pdf = mapPdf / (2.0 * kPi * kPi * sintheta);
    return radiance(u, v);
  }
  D pdfW(const V& w) const {  // infinite_area_light.dart:190-205
    V wi = xf(w2l, w);
    D theta = SphericalTheta(wi), phi = SphericalPhi(wi);
    D sintheta = std::sin(theta);
    if (sintheta == 0.0) return 0.0;
    // Distribution2D.pdf (montecarlo.dart:250-263)
    D uu = phi * 0.15915494309189533577, vv = theta * INV_PI;
    int nu = cond[0].count, nv = marginal.count;
    int iu = std::min(std::max((int)(uu * nu), 0), nu - 1);
    int iv = std::min(std::max((int)(vv * nv), 0), nv - 1);
    D p2;
    if (cond[iv].funcInt * marginal.funcInt == 0.0) p2 = 0.0;
    else p2 = ((D)cond[iv].func[iu] * (D)marginal.func[iv]) / (cond[iv].funcInt * marginal.funcInt);
    return p2 / (2.0 * kPi * kPi * sintheta);
  }
};

struct Scene {
  std::vector<float> P;  // world-space f32 vertices (triangle_mesh.dart:29-36)
  std::vector<float> N, S, UV;  // per-vertex normals / tangents (object space) / uvs; zero where a mesh has none
  std::vector<Mesh> meshes;
  std::vector<Prim> prims;  // after the build: BVH 'primitives' order
  std::vector<LinearNode> nodes;
  std::vector<Light> lights;
  std::vector<LightTri> lightTris;
  std::vector<Quadric> quadrics;
  EnvLight env;
  bool hasEnv = false;
  int maxPrimsInNode = 4;
  int bvhDepth = 0;
  mutable Counters ctr;
  mutable std::mutex ctrMutex;

  V vert(uint32_t i) const { return V{(D)P[3 * i], (D)P[3 * i + 1], (D)P[3 * i + 2]}; }
};
struct CounterScope {  // see t_ctr
  const Scene& sc;
  explicit CounterScope(const Scene& s) : sc(s) { t_ctr = Counters(); }
  ~CounterScope() {
    std::lock_guard<std::mutex> g(sc.ctrMutex);
    sc.ctr.add(t_ctr);
    t_ctr = Counters();
  }
};

// ---------------------------------------------------------------------------
// Triangle (shapes/triangle.dart)
// ---------------------------------------------------------------------------
static const D kUVs[6] = {0.0, 0.0, 1.0, 0.0, 1.0, 1.0};  // triangle.dart:255-262 (mesh.uvs == null)
// Triangle.getUVs (triangle.dart:247-263)
static void tri_uvs(const Scene& sc, int mesh, const uint32_t v[3], D uv[6]) {
  if (mesh >= 0 && sc.meshes[mesh].hasUV) {
    for (int k = 0; k < 3; ++k) {
      uv[2 * k] = sc.UV[2 * (size_t)v[k]];
      uv[2 * k + 1] = sc.UV[2 * (size_t)v[k] + 1];
    }
  } else {
    for (int k = 0; k < 6; ++k) uv[k] = kUVs[k];
  }
}

// Triangle.intersect (triangle.dart:44-160): all scalars f64 on f32 inputs.
static bool tri_intersect(const V& p1, const V& p2, const V& p3, bool reverse, const Ray& ray, D* tHit,
                          D* rayEpsilon, DG* dg, D* ob1 = nullptr, D* ob2 = nullptr, const D* uvs = kUVs) {
  D e1x = p2.x - p1.x, e1y = p2.y - p1.y, e1z = p2.z - p1.z;
  D e2x = p3.x - p1.x, e2y = p3.y - p1.y, e2z = p3.z - p1.z;
  D s1x = (ray.d.y * e2z) - (ray.d.z * e2y);
  D s1y = (ray.d.z * e2x) - (ray.d.x * e2z);
  D s1z = (ray.d.x * e2y) - (ray.d.y * e2x);
  D divisor = (s1x * e1x) + (s1y * e1y) + (s1z * e1z);
  if (divisor == 0.0) return false;
  D invDivisor = 1.0 / divisor;
  D sx = ray.o.x - p1.x, sy = ray.o.y - p1.y, sz = ray.o.z - p1.z;
  D b1 = (sx * s1x + sy * s1y + sz * s1z) * invDivisor;
  if (b1 < 0.0 || b1 > 1.0) return false;
  D s2x = (sy * e1z) - (sz * e1y);
  D s2y = (sz * e1x) - (sx * e1z);
  D s2z = (sx * e1y) - (sy * e1x);
  D b2 = ((ray.d.x * s2x) + (ray.d.y * s2y) + (ray.d.z * s2z)) * invDivisor;
  if (b2 < 0.0 || b1 + b2 > 1.0) return false;
  D t = (e2x * s2x + e2y * s2y + e2z * s2z) * invDivisor;
  if (t < ray.mint || t > ray.maxt) return false;

  // Partial derivatives (triangle.dart:100-132).
  V dpdu, dpdv;
  D du1 = uvs[0] - uvs[4];
  D du2 = uvs[2] - uvs[4];
  D dv1 = uvs[1] - uvs[5];
  D dv2 = uvs[3] - uvs[5];
  V dp1 = vsub(p1, p3);
  V dp2 = vsub(p2, p3);
  D determinant = du1 * dv2 - dv1 * du2;
  if (determinant == 0.0) {  // degenerate uv mapping
    D e3x = (e2y * e1z) - (e2z * e1y);
    D e3y = (e2z * e1x) - (e2x * e1z);
    D e3z = (e2x * e1y) - (e2y * e1x);
    D len = std::sqrt(e3x * e3x + e3y * e3y + e3z * e3z);
    V v1 = vec(e3x / len, e3y / len, e3z / len);
    if (std::fabs(v1.x) > std::fabs(v1.y)) {  // Vector.CoordinateSystem vector.dart:207-224
      D invLen = 1.0 / std::sqrt(v1.x * v1.x + v1.z * v1.z);
      dpdu = vec(-v1.z * invLen, 0.0, v1.x * invLen);
    } else {
      D invLen = 1.0 / std::sqrt(v1.y * v1.y + v1.z * v1.z);
      dpdu = vec(0.0, v1.z * invLen, -v1.y * invLen);
    }
    dpdv = vcross(v1, dpdu);
  } else {
    D invdet = 1.0 / determinant;
    dpdu = vmul(vsub(vmul(dp1, dv2), vmul(dp2, dv1)), invdet);
    dpdv = vmul(vadd(vmul(dp1, -du2), vmul(dp2, du1)), invdet);
  }
  // No alpha texture (triangle.dart:139-150).
  // dg.set (differential_geometry.dart:77-102).
  dg->p = pointAt(ray, t);
  dg->dpdu = dpdu;
  dg->dpdv = dpdv;
  dg->nn = vnormalize(vcross(dpdu, dpdv));  // Normal.Normalize normal.dart:53-55
  if (reverse) dg->nn = vmul(dg->nn, -1.0);  // transformSwapsHandedness is never set (shape.dart:30)
  {  // interpolated (u, v) (triangle.dart:134-137)
    D b0 = 1.0 - b1 - b2;
    dg->u = b0 * uvs[0] + b1 * uvs[2] + b2 * uvs[4];
    dg->v = b0 * uvs[1] + b1 * uvs[3] + b2 * uvs[5];
  }
  *tHit = t;
  *rayEpsilon = 1.0e-3 * t;
  if (ob1) *ob1 = b1;
  if (ob2) *ob2 = b2;
  return true;
}

// Triangle.intersectP (triangle.dart:162-240): Vector temporaries rounded to f32.
static bool tri_intersectP(const V& p1, const V& p2, const V& p3, const Ray& ray) {
  V e1 = vsub(p2, p1);
  V e2 = vsub(p3, p1);
  V s1 = vcross(ray.d, e2);
  D divisor = vdot(s1, e1);
  if (divisor == 0.0) return false;
  D invDivisor = 1.0 / divisor;
  V s = vsub(ray.o, p1);
  D b1 = vdot(s, s1) * invDivisor;
  if (b1 < 0.0 || b1 > 1.0) return false;
  V s2 = vcross(s, e1);
  D b2 = vdot(ray.d, s2) * invDivisor;
  if (b2 < 0.0 || b1 + b2 > 1.0) return false;
  D t = vdot(e2, s2) * invDivisor;
  if (t < ray.mint || t > ray.maxt) return false;
  return true;
}
static D tri_area(const V& p1, const V& p2, const V& p3) {  // triangle.dart:265-269
  return 0.5 * vlen(vcross(vsub(p2, p1), vsub(p3, p1)));
}
static V tri_sample(const V& p1, const V& p2, const V& p3, bool reverse, D u1, D u2, V* Ns) {  // :366-383
  D su1 = std::sqrt(u1);  // UniformSampleTriangle montecarlo.dart:215-220
  D b1 = 1.0 - su1;
  D b2 = u2 * su1;
  V p = vadd(vadd(vmul(p1, b1), vmul(p2, b2)), vmul(p3, (1.0 - b1 - b2)));
  V n = vcross(vsub(p2, p1), vsub(p3, p1));
  *Ns = vnormalize(n);
  if (reverse) *Ns = V{Ns->x * -1.0, Ns->y * -1.0, Ns->z * -1.0};
  return p;
}

static V xfPoint(const float* m, const V& p) {  // transform.dart:110-128
  D x = p.x, y = p.y, z = p.z;
  V o = vec(m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7],
            m[8] * x + m[9] * y + m[10] * z + m[11]);
  D w = (D)m[12] * x + (D)m[13] * y + (D)m[14] * z + (D)m[15];
  if (w != 1.0) o = vec(o.x / w, o.y / w, o.z / w);  // invScale
  return o;
}
static V xfVector(const float* m, const V& p) {  // transform.dart:130-144
  D x = p.x, y = p.y, z = p.z;
  return vec(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z);
}
static V xfNormal(const float* mInv, const V& n) {  // transform.dart:147-161 (transpose of the inverse)
  D x = n.x, y = n.y, z = n.z;
  return vec(mInv[0] * x + mInv[4] * y + mInv[8] * z, mInv[1] * x + mInv[5] * y + mInv[9] * z,
             mInv[2] * x + mInv[6] * y + mInv[10] * z);
}

// ---------------------------------------------------------------------------
// Quadrics: Sphere (shapes/sphere.dart) and Disk (shapes/disk.dart).  They transform the RAY into object
// space per test (transform.dart:180-196) instead of pre-transforming geometry.  dndu / dndv are not
// restated: nothing on the path (constant textures, Lambertian BSDF) reads them.
// ---------------------------------------------------------------------------
static const D INV_TWOPI = 0.15915494309189533577;  // common.dart:24
static bool Quadratic(D A, D B, D C, D* t0, D* t1) {  // common.dart:140-167
  D discrim = B * B - 4.0 * A * C;
  if (discrim < 0.0) return false;
  D rootDiscrim = std::sqrt(discrim);
  D q;
  if (B < 0.0) q = -0.5 * (B - rootDiscrim);
  else q = -0.5 * (B + rootDiscrim);
  *t0 = q / A;
  *t1 = C / q;
  if (*t0 > *t1) std::swap(*t0, *t1);
  return true;
}
static inline Ray xfRay(const float* m, const Ray& r) {  // transform.dart:180-196
  Ray t = r;
  t.o = xfPoint(m, r.o);
  t.d = xfVector(m, r.d);
  return t;
}
// DifferentialGeometry.set (differential_geometry.dart:77-102)
static void dg_set(DG* dg, const V& p, const V& dpdu, const V& dpdv, bool reverse) {
  dg->p = p;
  dg->dpdu = dpdu;
  dg->dpdv = dpdv;
  dg->nn = vnormalize(vcross(dpdu, dpdv));
  if (reverse) dg->nn = vmul(dg->nn, -1.0);
}
// phi of an object-space hit point; shared by sphere and disk
static inline D hit_phi(const V& phit) {
  D phi = std::atan2(phit.y, phit.x);
  if (phi < 0.0) phi += 2.0 * M_PI;
  return phi;
}
// Sphere.intersect (sphere.dart:40-172) / intersectP (:174-249); dg == nullptr => the predicate
static bool sphere_intersect(const Quadric& q, const Ray& r, D* tHit, D* rayEpsilon, DG* dg) {
  Ray ray = xfRay(q.w2o, r);
  D A = ray.d.x * ray.d.x + ray.d.y * ray.d.y + ray.d.z * ray.d.z;
  D B = 2 * (ray.d.x * ray.o.x + ray.d.y * ray.o.y + ray.d.z * ray.o.z);
  D C = ray.o.x * ray.o.x + ray.o.y * ray.o.y + ray.o.z * ray.o.z - q.radius * q.radius;
  D t0, t1;
  if (!Quadratic(A, B, C, &t0, &t1)) return false;
  if (t0 > ray.maxt || t1 < ray.mint) return false;
  D thit = t0;
  if (thit < ray.mint) {
    thit = t1;
    if (thit > ray.maxt) return false;
  }
  V phit = pointAt(ray, thit);
  if (phit.x == 0.0 && phit.y == 0.0) phit.x = r32(1.0e-5 * q.radius);
  D phi = hit_phi(phit);
  auto clipped = [&](const V& ph, D ph_phi) {
    return (q.zmin > -q.radius && ph.z < q.zmin) || (q.zmax < q.radius && ph.z > q.zmax) || ph_phi > q.phiMax;
  };
  if (clipped(phit, phi)) {
    // intersectP compares thit with the LIST t1 (sphere.dart:222: always unequal) and so re-tests the same
    // point, which fails the same clip: both variants return false here.
    if (thit == t1) return false;
    if (t1 > ray.maxt) return false;
    thit = t1;
    phit = pointAt(ray, thit);
    if (phit.x == 0.0 && phit.y == 0.0) phit.x = r32(1.0e-5 * q.radius);
    phi = hit_phi(phit);
    if (clipped(phit, phi)) return false;
  }
  if (!dg) return true;
  D theta = std::acos(clampD(phit.z / q.radius, -1.0, 1.0));
  D zradius = std::sqrt(phit.x * phit.x + phit.y * phit.y);
  D invzradius = 1.0 / zradius;
  D cosphi = phit.x * invzradius;
  D sinphi = phit.y * invzradius;
  V dpdu = vec(-q.phiMax * phit.y, q.phiMax * phit.x, 0.0);
  V dpdv = vmul(vec(phit.z * cosphi, phit.z * sinphi, -q.radius * std::sin(theta)), q.thetaMax - q.thetaMin);
  dg_set(dg, xfPoint(q.o2w, phit), xfVector(q.o2w, dpdu), xfVector(q.o2w, dpdv), q.reverse);
  *tHit = thit;
  *rayEpsilon = 5.0e-4 * thit;
  return true;
}
// Disk.intersect (disk.dart:37-101) / intersectP (:103-137)
static bool disk_intersect(const Quadric& q, const Ray& r, D* tHit, D* rayEpsilon, DG* dg) {
  Ray ray = xfRay(q.w2o, r);
  if (std::fabs(ray.d.z) < 1.0e-7) return false;
  D thit = (q.height - ray.o.z) / ray.d.z;
  if (thit < ray.mint || thit > ray.maxt) return false;
  V phit = pointAt(ray, thit);
  D dist2 = phit.x * phit.x + phit.y * phit.y;
  if (dist2 > q.radius * q.radius || dist2 < q.innerRadius * q.innerRadius) return false;
  D phi = hit_phi(phit);
  if (phi > q.phiMax) return false;
  if (!dg) return true;
  D oneMinusV = (std::sqrt(dist2) - q.innerRadius) / (q.radius - q.innerRadius);
  D invOneMinusV = (oneMinusV > 0.0) ? (1.0 / oneMinusV) : 0.0;
  V dpdu = vec(-q.phiMax * phit.y, q.phiMax * phit.x, 0.0);
  V dpdv = vec(-phit.x * invOneMinusV, -phit.y * invOneMinusV, 0.0);
  dpdu = vmul(dpdu, q.phiMax * INV_TWOPI);
  dpdv = vmul(dpdv, (q.radius - q.innerRadius) / q.radius);
  dg_set(dg, xfPoint(q.o2w, phit), xfVector(q.o2w, dpdu), xfVector(q.o2w, dpdv), q.reverse);
  *tHit = thit;
  *rayEpsilon = 5.0e-4 * thit;
  return true;
}
static bool quadric_intersect(const Quadric& q, const Ray& r, D* tHit, D* rayEpsilon, DG* dg) {
  return q.kind == 1 ? sphere_intersect(q, r, tHit, rayEpsilon, dg) : disk_intersect(q, r, tHit, rayEpsilon, dg);
}
static D quadric_area(const Quadric& q) {  // sphere.dart:251-253, disk.dart:139-142
  if (q.kind == 1) return q.phiMax * q.radius * (q.zmax - q.zmin);
  return q.phiMax * 0.5 * (q.radius * q.radius - q.innerRadius * q.innerRadius);
}
// Disk.sample (disk.dart:144-155); Shape.sample2 defaults to it (shape.dart:96-98)
static V disk_sample(const Quadric& q, D u1, D u2, V* Ns) {
  D t0, t1;
  ConcentricSampleDisk(u1, u2, &t0, &t1);
  V p = vec(t0 * q.radius, t1 * q.radius, q.height);
  V n = vnormalize(xfNormal(q.w2o, vec(0.0, 0.0, 1.0)));
  if (q.reverse) n = vmul(n, -1.0);
  *Ns = n;
  return xfPoint(q.o2w, p);
}
// Sphere.sample (sphere.dart:255-267)
static V sphere_sample(const Quadric& q, D u1, D u2, V* ns) {
  D z = 1.0 - 2.0 * u1;  // UniformSampleSphere (montecarlo.dart:113-120)
  D r = std::sqrt(std::max(0.0, 1.0 - z * z));
  D phi = 2.0 * M_PI * u2;
  V us = vec(r * std::cos(phi), r * std::sin(phi), z);
  V p = vadd(V{0, 0, 0}, vmul(us, q.radius));
  V n = vnormalize(xfNormal(q.w2o, p));
  if (q.reverse) n = V{-n.x, -n.y, -n.z};
  *ns = n;
  return xfPoint(q.o2w, p);
}
static inline D sphere_cos_theta_max(const Quadric& q, const V& p, const V& Pcenter) {
  D sinThetaMax2 = q.radius * q.radius / vlen2(vsub(Pcenter, p));
  return std::sqrt(std::max(0.0, 1.0 - sinThetaMax2));
}
// Sphere.sample2 (sphere.dart:269-311): uniform inside the cone the sphere subtends from p
static V sphere_sample2(const Quadric& q, const V& p, D u1, D u2, V* ns) {
  V Pcenter = xfPoint(q.o2w, V{0, 0, 0});
  V wc = vnormalize(vsub(Pcenter, p));
  V wcX, wcY;  // Vector.CoordinateSystem (vector.dart:198-214)
  if (std::fabs(wc.x) > std::fabs(wc.y)) {
    D invLen = 1.0 / std::sqrt(wc.x * wc.x + wc.z * wc.z);
    wcX = vec(-wc.z * invLen, 0.0, wc.x * invLen);
  } else {
    D invLen = 1.0 / std::sqrt(wc.y * wc.y + wc.z * wc.z);
    wcX = vec(0.0, wc.z * invLen, -wc.y * invLen);
  }
  wcY = vcross(wc, wcX);
  if (vlen2(vsub(Pcenter, p)) - q.radius * q.radius < 1.0e-4) return sphere_sample(q, u1, u2, ns);
  D cosThetaMax = sphere_cos_theta_max(q, p, Pcenter);
  // UniformSampleCone2 (montecarlo.dart:135-142)
  D costheta = cosThetaMax * (1.0 - u1) + 1.0 * u1;  // Lerp(u1, costhetamax, 1.0) (common.dart:80-81)
  D sintheta = std::sqrt(1.0 - costheta * costheta);
  D phi = u2 * 2.0 * M_PI;
  V d = vadd(vadd(vmul(wcX, std::cos(phi) * sintheta), vmul(wcY, std::sin(phi) * sintheta)), vmul(wc, costheta));
  Ray r{p, d, 1.0e-3, kInf, 0.0, 0};
  D thit = 0.0, eps = 0.0;
  DG dg;
  if (!sphere_intersect(q, r, &thit, &eps, &dg)) thit = vdot(vsub(Pcenter, p), vnormalize(r.d));
  V ps = pointAt(r, thit);
  V n = vnormalize(vsub(ps, Pcenter));
  if (q.reverse) n = V{-n.x, -n.y, -n.z};
  *ns = n;
  return ps;
}
// ---------------------------------------------------------------------------
// BVHAccel build (accelerators/bvh_accel.dart:41-91,228-437)
// ---------------------------------------------------------------------------
struct BBox {
  V pMin{kInf, kInf, kInf}, pMax{-kInf, -kInf, -kInf};  // bbox.dart:31-34
};
// dart:math min / max on doubles (sdk/lib/math/math.dart, restated from memory of the SDK source like the generator in
// dr_rng.h): the lesser / greater argument, and for two zeros of different sign min is -0.0 ((a + b) * a * b) and max is
// +0.0 (a + b) -- std::min / std::max keep whichever zero came first.  Only a box whose extreme coordinate is a zero
// of both signs can tell the difference (the union of vertices at -0.0 and +0.0).
static inline D dartMin(D a, D b) {
  if (a > b) return b;
  if (a < b) return a;
  if (a == 0.0) return (a + b) * a * b;
  if (b != b) return b;
  return a;
}
static inline D dartMax(D a, D b) {
  if (a > b) return a;
  if (a < b) return b;
  if (a == 0.0) return a + b;
  if (b != b) return b;
  return a;
}
static inline BBox bunion(const BBox& a, const BBox& b) {  // bbox.dart:152-161,203-205
  BBox r;
  r.pMin = V{dartMin(a.pMin.x, b.pMin.x), dartMin(a.pMin.y, b.pMin.y), dartMin(a.pMin.z, b.pMin.z)};
  r.pMax = V{dartMax(a.pMax.x, b.pMax.x), dartMax(a.pMax.y, b.pMax.y), dartMax(a.pMax.z, b.pMax.z)};
  return r;
}
static inline BBox bunionP(const BBox& a, const V& p) {  // bbox.dart:141-150,199-201
  BBox r;
  r.pMin = V{dartMin(a.pMin.x, p.x), dartMin(a.pMin.y, p.y), dartMin(a.pMin.z, p.z)};
  r.pMax = V{dartMax(a.pMax.x, p.x), dartMax(a.pMax.y, p.y), dartMax(a.pMax.z, p.z)};
  return r;
}
static inline D bsurfaceArea(const BBox& b) {  // bbox.dart:163-166
  V d = vsub(b.pMax, b.pMin);
  return 2.0 * (d.x * d.y + d.x * d.z + d.y * d.z);
}
static inline int bmaximumExtent(const BBox& b) {  // bbox.dart:173-182
  V diag = vsub(b.pMax, b.pMin);
  if (diag.x > diag.y && diag.x > diag.z) return 0;
  else if (diag.y > diag.z) return 1;
  else return 2;
}
struct PrimInfo {  // _BVHPrimitiveInfo bvh_accel.dart:490-501
  int primitiveNumber;
  V centroid;
  BBox bounds;
};
struct BuildNode {  // _BVHBuildNode bvh_accel.dart:508-531
  BBox bounds;
  int children[2] = {-1, -1};
  int splitAxis = 0, firstPrimOffset = 0, nPrimitives = 0;
};

struct Builder {
  Scene* sc;
  std::vector<PrimInfo> buildData;
  std::vector<BuildNode> bnodes;
  std::vector<Prim> orderedPrims;
  int maxPrimsInNode;
  int maxDepth = 0;

  // common.dart:256-287
  template <class Pred>
  int partition(Pred pred, int first, int last) {
    while (first < last) {
      while (pred(buildData[first])) {
        ++first;
        if (first == last) return first;
      }
      do {
        --last;
        if (first == last) return first;
      } while (!pred(buildData[last]));
      std::swap(buildData[first], buildData[last]);
      ++first;
    }
    return first;
  }
  // common.dart:289-297: a full List.sort with comparator pred(a,b) ? -1 : 1.
  // Dart's List.sort uses insertion sort below 32 elements (SDK sort.dart,
  // restated from the published algorithm; the SDK is not vendored).  In SAH
  // mode nth_element is only reached with <= 4 elements (bvh_accel.dart:313).
  void nth_element(int first, int /*nth*/, int last, int dim) {
    std::vector<PrimInfo> l(buildData.begin() + first, buildData.begin() + last);
    int n = (int)l.size();
    if (n <= 32) {
      for (int i = 1; i < n; ++i) {
        PrimInfo el = l[i];
        int j = i;
        // compare(a[j-1], el) > 0  <=>  !(a[j-1].c < el.c)
        while (j > 0 && !(l[j - 1].centroid[dim] < el.centroid[dim])) {
          l[j] = l[j - 1];
          j--;
        }
        l[j] = el;
      }
    } else {
      std::stable_sort(l.begin(), l.end(),
                       [dim](const PrimInfo& a, const PrimInfo& b) { return a.centroid[dim] < b.centroid[dim]; });
    }
    for (int i = first, j = 0; i < last; ++i, ++j) buildData[i] = l[j];
  }

  int recursiveBuild(int start, int end, int depth) {  // bvh_accel.dart:228-417
    maxDepth = std::max(maxDepth, depth);
    int me = (int)bnodes.size();
    bnodes.push_back(BuildNode());
    BBox bbox;
    for (int i = start; i < end; ++i) bbox = bunion(bbox, buildData[i].bounds);
    int nPrimitives = end - start;
    auto makeLeaf = [&]() {
      int firstPrimOffset = (int)orderedPrims.size();
      for (int i = start; i < end; ++i) orderedPrims.push_back(sc->prims[buildData[i].primitiveNumber]);
      bnodes[me].firstPrimOffset = firstPrimOffset;
      bnodes[me].nPrimitives = nPrimitives;
      bnodes[me].bounds = bbox;
    };
    if (nPrimitives == 1) {
      makeLeaf();
      return me;
    }
    BBox centroidBounds;
    for (int i = start; i < end; ++i) centroidBounds = bunionP(centroidBounds, buildData[i].centroid);
    int dim = bmaximumExtent(centroidBounds);
    int mid = (start + end) / 2;
    if (centroidBounds.pMax[dim] == centroidBounds.pMin[dim]) {
      makeLeaf();
      return me;
    }
    // SPLIT_SAH (the default, bvh_accel.dart:310-404)
    if (nPrimitives <= 4) {
      mid = (start + end) / 2;
      nth_element(start, mid, end, dim);
    } else {
      const int nBuckets = 12;
      int count[nBuckets];
      BBox bounds[nBuckets];
      for (int i = 0; i < nBuckets; ++i) count[i] = 0;
      D cmin = centroidBounds.pMin[dim], cmax = centroidBounds.pMax[dim];
      for (int i = start; i < end; ++i) {
        int b = (int)(nBuckets * ((buildData[i].centroid[dim] - cmin) / (cmax - cmin)));  // toInt()
        if (b == nBuckets) b = nBuckets - 1;
        count[b]++;
        bounds[b] = bunion(bounds[b], buildData[i].bounds);
      }
      float cost[nBuckets - 1];  // Float32List bvh_accel.dart:345
      for (int i = 0; i < nBuckets - 1; ++i) {
        BBox b0, b1;
        int count0 = 0, count1 = 0;
        for (int j = 0; j <= i; ++j) {
          b0 = bunion(b0, bounds[j]);
          count0 += count[j];
        }
        for (int j = i + 1; j < nBuckets; ++j) {
          b1 = bunion(b1, bounds[j]);
          count1 += count[j];
        }
        cost[i] = (float)(0.125 + (count0 * bsurfaceArea(b0) + count1 * bsurfaceArea(b1)) / bsurfaceArea(bbox));
      }
      D minCost = cost[0];
      int minCostSplit = 0;
      for (int i = 1; i < nBuckets - 1; ++i) {
        if ((D)cost[i] < minCost) {
          minCost = cost[i];
          minCostSplit = i;
        }
      }
      if (nPrimitives > maxPrimsInNode || minCost < nPrimitives) {
        auto pred = [&](const PrimInfo& p) {
          int b = (int)std::floor(nBuckets * ((p.centroid[dim] - cmin) / (cmax - cmin)));
          if (b == nBuckets) b = nBuckets - 1;
          return b <= minCostSplit;
        };
        mid = partition(pred, start, end);
      } else {
        makeLeaf();
        return me;
      }
    }
    // Right child first (bvh_accel.dart:407-411).
    int c2 = recursiveBuild(mid, end, depth + 1);
    int c1 = recursiveBuild(start, mid, depth + 1);
    bnodes[me].children[0] = c1;
    bnodes[me].children[1] = c2;
    bnodes[me].bounds = bunion(bnodes[c1].bounds, bnodes[c2].bounds);
    bnodes[me].splitAxis = dim;
    bnodes[me].nPrimitives = 0;
    return me;
  }

  int flatten(int bn, int* offset) {  // bvh_accel.dart:419-437
    LinearNode& ln = sc->nodes[*offset];
    ln.bmin = bnodes[bn].bounds.pMin;
    ln.bmax = bnodes[bn].bounds.pMax;
    int myOffset = (*offset)++;
    if (bnodes[bn].nPrimitives > 0) {
      ln.offset = (uint32_t)bnodes[bn].firstPrimOffset;
      ln.nPrimitives = bnodes[bn].nPrimitives;
      ln.axis = 0;
    } else {
      ln.axis = bnodes[bn].splitAxis;
      ln.nPrimitives = 0;
      flatten(bnodes[bn].children[0], offset);
      int second = flatten(bnodes[bn].children[1], offset);
      sc->nodes[myOffset].offset = (uint32_t)second;
    }
    return myOffset;
  }

  void build() {  // bvh_accel.dart:41-91
    int n = (int)sc->prims.size();
    if (n == 0) return;
    buildData.resize(n);
    for (int i = 0; i < n; ++i) {
      const Prim& pr = sc->prims[i];
      BBox bb;
      if (pr.quadric >= 0) {
        // Shape.worldBound = objectToWorld.transformBBox(objectBound()) (shape.dart:37-39, transform.dart:163-178)
        const Quadric& q = sc->quadrics[pr.quadric];
        V lo = q.kind == 1 ? vec(-q.radius, -q.radius, q.zmin) : vec(-q.radius, -q.radius, q.height);
        V hi = q.kind == 1 ? vec(q.radius, q.radius, q.zmax) : vec(q.radius, q.radius, q.height);
        BBox ob;  // BBox(p1, p2) bbox.dart:36-40
        ob.pMin = V{dartMin(lo.x, hi.x), dartMin(lo.y, hi.y), dartMin(lo.z, hi.z)};
        ob.pMax = V{dartMax(lo.x, hi.x), dartMax(lo.y, hi.y), dartMax(lo.z, hi.z)};
        const V c8[8] = {ob.pMin, V{ob.pMax.x, ob.pMin.y, ob.pMin.z}, V{ob.pMin.x, ob.pMax.y, ob.pMin.z},
                         V{ob.pMin.x, ob.pMin.y, ob.pMax.z}, V{ob.pMin.x, ob.pMax.y, ob.pMax.z},
                         V{ob.pMax.x, ob.pMax.y, ob.pMin.z}, V{ob.pMax.x, ob.pMin.y, ob.pMax.z}, ob.pMax};
        for (int k = 0; k < 8; ++k) bb = bunionP(bb, xfPoint(q.o2w, c8[k]));
      } else {
        V a = sc->vert(pr.v[0]), b = sc->vert(pr.v[1]), c = sc->vert(pr.v[2]);
        // Triangle.worldBound triangle.dart:39-42
        bb.pMin = V{dartMin(a.x, b.x), dartMin(a.y, b.y), dartMin(a.z, b.z)};
        bb.pMax = V{dartMax(a.x, b.x), dartMax(a.y, b.y), dartMax(a.z, b.z)};
        bb = bunionP(bb, c);
      }
      buildData[i].primitiveNumber = i;
      buildData[i].bounds = bb;
      buildData[i].centroid = vadd(vmul(bb.pMin, 0.5), vmul(bb.pMax, 0.5));  // bbox.dart:66
    }
    bnodes.reserve(2 * n);
    orderedPrims.reserve(n);
    recursiveBuild(0, n, 0);
    sc->prims = orderedPrims;
    sc->nodes.resize(bnodes.size());
    int offset = 0;
    flatten(0, &offset);
    sc->bvhDepth = maxDepth;
  }
};

// ---------------------------------------------------------------------------
// BVHAccel traversal (bvh_accel.dart:101-226,439-472)
// ---------------------------------------------------------------------------
static inline bool slab(const LinearNode& n, const Ray& ray, const V& invDir, const int dirIsNeg[3]) {
  const V* b[2] = {&n.bmin, &n.bmax};
  D tmin = (b[dirIsNeg[0]]->x - ray.o.x) * invDir.x;
  D tmax = (b[1 - dirIsNeg[0]]->x - ray.o.x) * invDir.x;
  D tymin = (b[dirIsNeg[1]]->y - ray.o.y) * invDir.y;
  D tymax = (b[1 - dirIsNeg[1]]->y - ray.o.y) * invDir.y;
  if ((tmin > tymax) || (tymin > tmax)) return false;
  if (tymin > tmin) tmin = tymin;
  if (tymax < tmax) tmax = tymax;
  D tzmin = (b[dirIsNeg[2]]->z - ray.o.z) * invDir.z;
  D tzmax = (b[1 - dirIsNeg[2]]->z - ray.o.z) * invDir.z;
  if ((tmin > tzmax) || (tzmin > tmax)) return false;
  if (tzmin > tmin) tmin = tzmin;
  if (tzmax < tmax) tmax = tzmax;
  return (tmin < ray.maxt) && (tmax > ray.mint);
}

struct Isect {  // Intersection (intersection.dart) + GeometricPrimitive.intersect (geometric_primitive.dart:47-61)
  DG dg;
  int prim = -1;
  D rayEpsilon = 0;
  D t = 0, b1 = 0, b2 = 0;
};

static bool bvh_intersect(const Scene& sc, Ray& ray, Isect* isect) {  // bvh_accel.dart:101-165
  t_ctr.closest_rays++;
  if (sc.nodes.empty()) return false;
  bool hit = false;
  V invDir = vec(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);  // f32-rounded (:109-111)
  int dirIsNeg[3] = {invDir.x < 0 ? 1 : 0, invDir.y < 0 ? 1 : 0, invDir.z < 0 ? 1 : 0};
  int todoOffset = 0, nodeNum = 0;
  std::vector<uint32_t> todoHeap;
  uint32_t todoSmall[64];
  uint32_t* todo = todoSmall;
  size_t todoCap = 64;
  auto push = [&](uint32_t v) {
    if ((size_t)todoOffset == todoCap) {  // the reference would throw a RangeError at 64
      todoHeap.assign(todo, todo + todoCap);
      todoCap *= 2;
      todoHeap.resize(todoCap);
      todo = todoHeap.data();
    }
    todo[todoOffset++] = v;
    if ((uint64_t)todoOffset > t_ctr.max_stack) t_ctr.max_stack = (uint64_t)todoOffset;
  };
  while (true) {
    const LinearNode& node = sc.nodes[nodeNum];
    t_ctr.closest_nodes++;
    if (slab(node, ray, invDir, dirIsNeg)) {
      if (node.nPrimitives > 0) {
        for (int i = 0; i < node.nPrimitives; ++i) {
          t_ctr.closest_tris++;
          const Prim& pr = sc.prims[node.offset + i];
          D thit, eps, b1 = 0.0, b2 = 0.0, uvs[6];
          DG dg;
          const bool h = pr.quadric >= 0
                             ? quadric_intersect(sc.quadrics[pr.quadric], ray, &thit, &eps, &dg)
                             : (tri_uvs(sc, pr.mesh, pr.v, uvs),
                                tri_intersect(sc.vert(pr.v[0]), sc.vert(pr.v[1]), sc.vert(pr.v[2]),
                                              sc.meshes[pr.mesh].reverse, ray, &thit, &eps, &dg, &b1, &b2, uvs));
          if (h) {
            isect->dg = dg;
            isect->prim = (int)(node.offset + i);
            isect->rayEpsilon = eps;
            isect->t = thit;
            isect->b1 = b1;
            isect->b2 = b2;
            ray.maxt = thit;  // geometric_primitive.dart:59
            hit = true;
          }
        }
        if (todoOffset == 0) break;
        nodeNum = (int)todo[--todoOffset];
      } else {
        if (dirIsNeg[node.axis] != 0) {
          push((uint32_t)(nodeNum + 1));
          nodeNum = (int)node.offset;
        } else {
          push(node.offset);
          nodeNum = nodeNum + 1;
        }
      }
    } else {
      if (todoOffset == 0) break;
      nodeNum = (int)todo[--todoOffset];
    }
  }
  return hit;
}

// ---------------------------------------------------------------------------
// Measurement only (orc_order_study; MEASUREMENTS.md round 6): BVHAccel.intersectP never modifies the ray (bvh_accel.dart:167-226), so a
// leaf is reached iff its ancestors' slab tests pass, whatever order the children are taken in -- the BOOLEAN does not depend on the
// order, only the work of a ray that finds an occluder does (it returns at the first one).  For every any-hit ray of a render this counts
// the node visits / triangle tests of (0) the reference order (near child by the split axis' direction sign), (1) the far child first,
// (2) the child with the larger surface area first, and a lower bound no order can beat: the shallowest occluding leaf's depth + 1.
// It touches none of the counters the parity tests compare.
// ---------------------------------------------------------------------------
struct OrderStudy {
  std::atomic<uint64_t> rays{0}, occluded{0}, nodesAll{0}, trisAll{0};
  std::atomic<uint64_t> nodesOcc[6], trisOcc[6];
  std::atomic<uint64_t> idealNodesOcc{0};
  void reset() {
    rays = occluded = nodesAll = trisAll = idealNodesOcc = 0;
    for (int k = 0; k < 6; ++k) nodesOcc[k] = trisOcc[k] = 0;
  }
};
static OrderStudy g_study;
static std::atomic<int> g_studyOn{0};
static inline bool slab(const LinearNode& n, const Ray& ray, const V& invDir, const int dirIsNeg[3]);
static bool tri_intersectP(const V& p1, const V& p2, const V& p3, const Ray& ray);
static bool study_leaf(const Scene& sc, const LinearNode& node, const Ray& ray, uint64_t* tris) {
  for (int i = 0; i < node.nPrimitives; ++i) {
    ++*tris;
    const Prim& pr = sc.prims[node.offset + i];
    if (pr.quadric >= 0) {
      if (quadric_intersect(sc.quadrics[pr.quadric], ray, nullptr, nullptr, nullptr)) return true;
    } else if (tri_intersectP(sc.vert(pr.v[0]), sc.vert(pr.v[1]), sc.vert(pr.v[2]), ray)) {
      return true;
    }
  }
  return false;
}
static bool study_walk(const Scene& sc, const Ray& ray, int mode, uint64_t* nodes, uint64_t* tris) {
  V invDir = vec(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
  int dirIsNeg[3] = {invDir.x < 0 ? 1 : 0, invDir.y < 0 ? 1 : 0, invDir.z < 0 ? 1 : 0};
  auto area = [](const LinearNode& n) {
    D dx = n.bmax.x - n.bmin.x, dy = n.bmax.y - n.bmin.y, dz = n.bmax.z - n.bmin.z;
    return 2.0 * (dx * dy + dy * dz + dz * dx);
  };
  std::vector<uint32_t> todo;
  uint32_t nodeNum = 0;
  for (;;) {
    const LinearNode& node = sc.nodes[nodeNum];
    ++*nodes;
    bool pop = true;
    if (slab(node, ray, invDir, dirIsNeg)) {
      if (node.nPrimitives > 0) {
        if (study_leaf(sc, node, ray, tris)) return true;
      } else {
        uint32_t first = nodeNum + 1, second = node.offset;  // the reference: second child first when the ray runs against the axis
        bool swap = dirIsNeg[node.axis] != 0;
        if (mode == 1) swap = !swap;
        else if (mode == 2) swap = area(sc.nodes[second]) > area(sc.nodes[first]);
        else if (mode >= 3) {
          // experiments: the child the ray spends the LONGER parameter interval in first (3); a leaf child first, else far first (4);
          // the smaller-area child first (5)
          auto span = [&](const LinearNode& c) {
            D t0 = ray.mint, t1 = ray.maxt;
            const D o[3] = {ray.o.x, ray.o.y, ray.o.z}, iv[3] = {invDir.x, invDir.y, invDir.z};
            const D lo[3] = {c.bmin.x, c.bmin.y, c.bmin.z}, hi[3] = {c.bmax.x, c.bmax.y, c.bmax.z};
            for (int a = 0; a < 3; ++a) {
              D ta = (lo[a] - o[a]) * iv[a], tb = (hi[a] - o[a]) * iv[a];
              if (ta > tb) std::swap(ta, tb);
              if (ta > t0) t0 = ta;
              if (tb < t1) t1 = tb;
            }
            return t1 - t0;  // negative: missed
          };
          const bool farSwap = !(dirIsNeg[node.axis] != 0);
          if (mode == 3) { D a = span(sc.nodes[first]), b = span(sc.nodes[second]); swap = (a == b) ? farSwap : (b > a); }
          else if (mode == 4) { bool l1 = sc.nodes[first].nPrimitives > 0, l2 = sc.nodes[second].nPrimitives > 0; swap = (l1 != l2) ? l2 : farSwap; }
          else swap = area(sc.nodes[second]) < area(sc.nodes[first]);
        }
        if (swap) std::swap(first, second);
        todo.push_back(second);
        nodeNum = first;
        pop = false;
      }
    }
    if (pop) {
      if (todo.empty()) return false;
      nodeNum = todo.back();
      todo.pop_back();
    }
  }
}
static uint64_t study_ideal(const Scene& sc, const Ray& ray) {  // depth + 1 of the shallowest leaf that holds an occluder (0: none)
  V invDir = vec(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
  int dirIsNeg[3] = {invDir.x < 0 ? 1 : 0, invDir.y < 0 ? 1 : 0, invDir.z < 0 ? 1 : 0};
  std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 1u}};
  uint64_t best = 0, dummy = 0;
  while (!todo.empty()) {
    auto [n, depth] = todo.back();
    todo.pop_back();
    if (best && depth >= best) continue;
    const LinearNode& node = sc.nodes[n];
    if (!slab(node, ray, invDir, dirIsNeg)) continue;
    if (node.nPrimitives > 0) {
      if (study_leaf(sc, node, ray, &dummy)) best = depth;
    } else {
      todo.push_back({n + 1, depth + 1});
      todo.push_back({node.offset, depth + 1});
    }
  }
  return best;
}
static void study_ray(const Scene& sc, const Ray& ray) {
  uint64_t n[6] = {0, 0, 0, 0, 0, 0}, t[6] = {0, 0, 0, 0, 0, 0};
  const bool occ = study_walk(sc, ray, 0, &n[0], &t[0]);
  g_study.rays++;
  g_study.nodesAll += n[0];
  g_study.trisAll += t[0];
  if (!occ) return;  // a ray that finds nothing visits the same nodes in every order
  g_study.occluded++;
  for (int m = 1; m < 6; ++m) (void)study_walk(sc, ray, m, &n[m], &t[m]);
  for (int m = 0; m < 6; ++m) {
    g_study.nodesOcc[m] += n[m];
    g_study.trisOcc[m] += t[m];
  }
  g_study.idealNodesOcc += study_ideal(sc, ray);
}

static bool bvh_intersectP(const Scene& sc, const Ray& ray) {  // bvh_accel.dart:167-226
  if (g_studyOn.load(std::memory_order_relaxed) && !sc.nodes.empty()) study_ray(sc, ray);
  t_ctr.any_rays++;
  if (sc.nodes.empty()) return false;
  V invDir = vec(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
  int dirIsNeg[3] = {invDir.x < 0 ? 1 : 0, invDir.y < 0 ? 1 : 0, invDir.z < 0 ? 1 : 0};
  std::vector<uint32_t> todo(64);
  int todoOffset = 0, nodeNum = 0;
  auto push = [&](uint32_t v) {
    if ((size_t)todoOffset == todo.size()) todo.resize(todo.size() * 2);
    todo[todoOffset++] = v;
  };
  while (true) {
    const LinearNode& node = sc.nodes[nodeNum];
    t_ctr.any_nodes++;
    if (slab(node, ray, invDir, dirIsNeg)) {
      if (node.nPrimitives > 0) {
        for (int i = 0; i < node.nPrimitives; ++i) {
          t_ctr.any_tris++;
          const Prim& pr = sc.prims[node.offset + i];
          if (pr.quadric >= 0) {
            if (quadric_intersect(sc.quadrics[pr.quadric], ray, nullptr, nullptr, nullptr)) return true;
          } else if (tri_intersectP(sc.vert(pr.v[0]), sc.vert(pr.v[1]), sc.vert(pr.v[2]), ray)) {
            return true;
          }
        }
        if (todoOffset == 0) break;
        nodeNum = (int)todo[--todoOffset];
      } else {
        if (dirIsNeg[node.axis] != 0) {
          push((uint32_t)(nodeNum + 1));
          nodeNum = (int)node.offset;
        } else {
          push(node.offset);
          nodeNum = nodeNum + 1;
        }
      }
    } else {
      if (todoOffset == 0) break;
      nodeNum = (int)todo[--todoOffset];
    }
  }
  return false;
}

// ---------------------------------------------------------------------------
// ShapeSet / DiffuseAreaLight (shape_set.dart:53-96, shape.dart:100-121,
// diffuse_area_light.dart:44-70)
// ---------------------------------------------------------------------------
static void lt_verts(const Scene& sc, const LightTri& lt, V* a, V* b, V* c) {
  *a = sc.vert(lt.v[0]);
  *b = sc.vert(lt.v[1]);
  *c = sc.vert(lt.v[2]);
}
static V shapeset_sample(const Scene& sc, const Light& L, D uPos0, D uPos1, D uComponent, V* Ns, const V& p) {
  int sn = L.areaDistribution.sampleDiscrete(uComponent) % (int)L.shapes.size();
  V a, b, c;
  const LightTri& lt = sc.lightTris[L.shapes[sn]];
  V pt;
  if (lt.quadric >= 0) {
    const Quadric& q = sc.quadrics[lt.quadric];
    pt = q.kind == 1 ? sphere_sample2(q, p, uPos0, uPos1, Ns) : disk_sample(q, uPos0, uPos1, Ns);
  } else {
    lt_verts(sc, lt, &a, &b, &c);
    pt = tri_sample(a, b, c, lt.reverse, uPos0, uPos1, Ns);  // Shape.sample2 -> sample (shape.dart:96-98)
  }
  Ray r{p, vsub(pt, p), 1.0e-3, kInf, 0.0, 0};
  D rayEps = 0.0, thit = 1.0;
  bool anyHit = false;
  DG dg;
  for (size_t i = 0; i < L.shapes.size(); ++i) {
    const LightTri& t = sc.lightTris[L.shapes[i]];
    t_ctr.light_tris++;
    if (t.quadric >= 0) {
      anyHit = quadric_intersect(sc.quadrics[t.quadric], r, &thit, &rayEps, &dg) || anyHit;
      continue;
    }
    lt_verts(sc, t, &a, &b, &c);
    D uvs[6];
    tri_uvs(sc, t.mesh, t.v, uvs);
    anyHit = tri_intersect(a, b, c, t.reverse, r, &thit, &rayEps, &dg, nullptr, nullptr, uvs) || anyHit;
  }
  if (anyHit) *Ns = dg.nn;
  return pointAt(r, thit);
}
static D shapeset_pdf(const Scene& sc, const Light& L, const V& p, const V& wi) {  // shape_set.dart:82-89
  D pdf = 0.0;
  for (size_t i = 0; i < L.shapes.size(); ++i) {
    const LightTri& t = sc.lightTris[L.shapes[i]];
    V a, b, c;
    if (t.quadric < 0) lt_verts(sc, t, &a, &b, &c);
    // Shape.pdf2 (shape.dart:100-121)
    D pdf2;
    DG dgLight;
    Ray ray{p, wi, 1.0e-3, kInf, 0.0, -1};
    D thit = 0.0, rayEpsilon = 0.0;
    t_ctr.light_tris++;
    if (t.quadric >= 0 && sc.quadrics[t.quadric].kind == 1) {
      // Sphere.pdf2 (sphere.dart:313-326): the cone's solid angle unless p is inside the sphere
      const Quadric& q = sc.quadrics[t.quadric];
      V Pcenter = xfPoint(q.o2w, V{0, 0, 0});
      if (!(vlen2(vsub(Pcenter, p)) - q.radius * q.radius < 1.0e-4)) {
        pdf += L.areas[i] * (1.0 / (2.0 * M_PI * (1.0 - sphere_cos_theta_max(q, p, Pcenter))));  // UniformConePdf
        continue;
      }
    }
    D uvs[6];
    if (t.quadric < 0) tri_uvs(sc, t.mesh, t.v, uvs);
    const bool h = t.quadric >= 0 ? quadric_intersect(sc.quadrics[t.quadric], ray, &thit, &rayEpsilon, &dgLight)
                                  : tri_intersect(a, b, c, t.reverse, ray, &thit, &rayEpsilon, &dgLight, nullptr, nullptr, uvs);
    if (!h) {
      pdf2 = 0.0;
    } else {
      V q = pointAt(ray, thit);
      const D shapeArea = t.quadric >= 0 ? quadric_area(sc.quadrics[t.quadric]) : tri_area(a, b, c);
      pdf2 = vlen2(vsub(q, p)) / (vabsdot(dgLight.nn, vneg(wi)) * shapeArea);
      if (std::isinf(pdf2)) pdf2 = 0.0;
    }
    pdf += L.areas[i] * pdf2;
  }
  return pdf / L.area;
}
static inline S light_L(const Light& L, const V& n, const V& w) {  // diffuse_area_light.dart:44-46
  return vdot(n, w) > 0.0 ? L.Lemit : S{0, 0, 0};
}

// ---------------------------------------------------------------------------
// BSDF with a single Lambertian lobe (matte_material.dart:41-65, bsdf.dart,
// bxdf.dart, lambertian.dart)
// ---------------------------------------------------------------------------
enum {
  BSDF_REFLECTION = 1 << 0,
  BSDF_TRANSMISSION = 1 << 1,
  BSDF_DIFFUSE = 1 << 2,
  BSDF_GLOSSY = 1 << 3,
  BSDF_SPECULAR = 1 << 4,
  BSDF_ALL_TYPES = BSDF_DIFFUSE | BSDF_GLOSSY | BSDF_SPECULAR,
  BSDF_ALL = BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_ALL_TYPES
};
// Fresnel terms (fresnel_dielectric.dart:30-64, fresnel_no_op.dart): returned as a Spectrum (f32).
static S fresnel_dielectric(D cosi, D eta_i, D eta_t) {
  cosi = clampD(cosi, -1.0, 1.0);
  bool entering = cosi > 0.0;
  D ei = eta_i, et = eta_t;
  if (!entering) std::swap(ei, et);
  D sint = ei / et * std::sqrt(std::max(0.0, 1.0 - cosi * cosi));
  if (sint >= 1.0) return S{1, 1, 1};  // total internal reflection
  D cost = std::sqrt(std::max(0.0, 1.0 - sint * sint));
  cosi = std::fabs(cosi);
  D Rparl = ((et * cosi) - (ei * cost)) / ((et * cosi) + (ei * cost));
  D Rperp = ((ei * cosi) - (et * cost)) / ((ei * cosi) + (et * cost));
  D v = (Rparl * Rparl + Rperp * Rperp) / 2.0;
  return rgb(v, v, v);
}
struct BxDF {
  int kind = 0;  // 0 Lambertian (lambertian.dart), 1 SpecularReflection, 2 SpecularTransmission, 3 OrenNayar,
                 // 4 Microfacet(R, FresnelDielectric(ei, et), Blinn(exponent)) (microfacet.dart, blinn.dart)
  D A = 0.0, B = 0.0;  // OrenNayar (oren_nayar.dart:24-32)
  D exponent = 0.0;    // Blinn (blinn.dart:24-28)
  int type = 0;  // BxDFType flags
  S R{0, 0, 0};  // reflectance / transmittance
  bool dielectric = false;  // SpecularReflection: FresnelDielectric(ei, et) instead of FresnelNoOp
  D ei = 1.0, et = 1.0;
  bool matches(int flags) const { return (type & flags) == type; }  // bxdf.dart:31-33
  static D sinTheta(const V& v) { return std::sqrt(std::max(0.0, 1.0 - v.z * v.z)); }  // vector.dart:121-124
  static D cosPhi(const V& v) {                                                          // vector.dart:126-132
    D st = sinTheta(v);
    return st == 0.0 ? 1.0 : clampD(v.x / st, -1.0, 1.0);
  }
  static D sinPhi(const V& v) {                                                          // vector.dart:134-140
    D st = sinTheta(v);
    return st == 0.0 ? 0.0 : clampD(v.y / st, -1.0, 1.0);
  }
  // Blinn.pdf (blinn.dart:62-73) given the half vector's cos(theta) and dot(wo, wh)
  D blinn_pdf(D costheta, D woDotWh) const {
    D p = ((exponent + 1.0) * std::pow(costheta, exponent)) / (2.0 * M_PI * 4.0 * woDotWh);
    if (woDotWh <= 0.0) p = 0.0;
    return p;
  }
  S microfacet_f(const V& wo, const V& wi) const {  // microfacet.dart:27-56
    D cosThetaO = std::fabs(wo.z), cosThetaI = std::fabs(wi.z);
    if (cosThetaI == 0.0 || cosThetaO == 0.0) return S{0, 0, 0};
    V wh = vadd(wi, wo);
    if (wh.x == 0.0 && wh.y == 0.0 && wh.z == 0.0) return S{0, 0, 0};
    wh = vnormalize(wh);
    D cosThetaH = vdot(wi, wh);
    S F = fresnel_dielectric(cosThetaH, ei, et);
    D d = (exponent + 2.0) * INV_TWOPI * std::pow(std::fabs(wh.z), exponent);  // Blinn.d (blinn.dart:30-33)
    D NdotWh = std::fabs(wh.z), NdotWo = std::fabs(wo.z), NdotWi = std::fabs(wi.z), WOdotWh = vabsdot(wo, wh);
    D g = std::min(1.0, std::min((2.0 * NdotWh * NdotWo / WOdotWh), (2.0 * NdotWh * NdotWi / WOdotWh)));
    return sdivD(smul(smulD(R, d * g), F), 4.0 * cosThetaI * cosThetaO);  // R * (D G) * F / (4 cos cos)
  }
  S f(const V& wo, const V& wi) const {
    if (kind == 0) return smulD(R, INV_PI);  // lambertian.dart:35-37
    if (kind == 4) return microfacet_f(wo, wi);
    if (kind != 3) return S{0, 0, 0};        // specular_*.dart: f() == 0
    // oren_nayar.dart:34-60
    D sinthetai = sinTheta(wi), sinthetao = sinTheta(wo);
    D maxcos = 0.0;
    if (sinthetai > 1e-4 && sinthetao > 1e-4) {
      D dcos = cosPhi(wi) * cosPhi(wo) + sinPhi(wi) * sinPhi(wo);
      maxcos = std::max(0.0, dcos);
    }
    D sinalpha, tanbeta;
    if (std::fabs(wi.z) > std::fabs(wo.z)) {
      sinalpha = sinthetao;
      tanbeta = sinthetai / std::fabs(wi.z);
    } else {
      sinalpha = sinthetai;
      tanbeta = sinthetao / std::fabs(wo.z);
    }
    return smulD(R, INV_PI * (A + B * maxcos * sinalpha * tanbeta));
  }
  D pdf(const V& wo, const V& wi) const {  // bxdf.dart:84-88; specular_*.dart pdf() == 0
    if (kind == 1 || kind == 2) return 0.0;
    if (kind == 4) {  // microfacet.dart:75-80, blinn.dart:62-73
      if (!(wo.z * wi.z > 0.0)) return 0.0;
      V wh = vnormalize(vadd(wo, wi));
      return blinn_pdf(std::fabs(wh.z), vdot(wo, wh));
    }
    return (wo.z * wi.z > 0.0) ? std::fabs(wi.z) * INV_PI : 0.0;
  }
  S sample_f(const V& wo, V* wi, D u1, D u2, D* pdf) const {
    if (kind == 4) {  // microfacet.dart:66-73, blinn.dart:35-60
      D costheta = std::pow(u1, 1.0 / (exponent + 1.0));
      D sintheta = std::sqrt(std::max(0.0, 1.0 - costheta * costheta));
      D phi = u2 * 2.0 * M_PI;
      V wh = vec(sintheta * std::cos(phi), sintheta * std::sin(phi), costheta);  // Vector.SphericalDirection
      if (!(wo.z * wh.z > 0.0)) wh = vneg(wh);
      // wi = -wo + wh * 2.0 * Dot(wo, wh): ((wh * 2.0) * dot) added to -wo, each a Vector (f32)
      *wi = vadd(vneg(wo), vmul(vmul(wh, 2.0), vdot(wo, wh)));
      *pdf = blinn_pdf(costheta, vdot(wo, wh));
      if (!(wo.z * wi->z > 0.0)) return S{0, 0, 0};
      return microfacet_f(wo, *wi);
    }
    if (kind == 0 || kind == 3) {  // BxDF.sample_f (bxdf.dart:37-48)
      *wi = CosineSampleHemisphere(u1, u2);
      if (wo.z < 0.0) wi->z *= -1.0;
      *pdf = this->pdf(wo, *wi);
      return f(wo, *wi);
    }
    if (kind == 1) {  // specular_reflection.dart:33-41
      *wi = vec(-wo.x, -wo.y, wo.z);
      *pdf = 1.0;
      S F = dielectric ? fresnel_dielectric(wo.z, ei, et) : S{1, 1, 1};
      return sdivD(smul(F, R), std::fabs(wi->z));
    }
    // specular_transmission.dart:37-71
    bool entering = wo.z > 0.0;
    D e_i = ei, e_t = et;
    if (!entering) std::swap(e_i, e_t);
    D sini2 = std::max(0.0, 1.0 - wo.z * wo.z);  // Vector.SinTheta2 (vector.dart)
    D eta = e_i / e_t;
    D sint2 = eta * eta * sini2;
    if (sint2 >= 1.0) return S{0, 0, 0};  // total internal reflection: pdf stays 0
    D cost = std::sqrt(std::max(0.0, 1.0 - sint2));
    if (entering) cost = -cost;
    D sintOverSini = eta;
    *wi = vec(sintOverSini * -wo.x, sintOverSini * -wo.y, cost);
    *pdf = 1.0;
    S F = fresnel_dielectric(wo.z, ei, et);
    S oneMinusF = rgb(1.0 - F.r, 1.0 - F.g, 1.0 - F.b);
    return sdivD(smul(oneMinusF, R), std::fabs(wi->z));
  }
};
struct BSDF {
  V p, nn, ng, sn, tn;
  int nBxDFs = 0;
  BxDF bx[2];
  void add(const BxDF& b) { bx[nBxDFs++] = b; }
  int numComponents(int flags) const {  // bsdf.dart:162-175
    int n = 0;
    for (int i = 0; i < nBxDFs; ++i) n += bx[i].matches(flags) ? 1 : 0;
    return n;
  }
  V worldToLocal(const V& v) const { return vec(vdot(v, sn), vdot(v, tn), vdot(v, nn)); }  // bsdf.dart:177-179
  V localToWorld(const V& v) const {                                                        // bsdf.dart:181-185
    return vec(sn.x * v.x + tn.x * v.y + nn.x * v.z, sn.y * v.x + tn.y * v.y + nn.y * v.z,
               sn.z * v.x + tn.z * v.y + nn.z * v.z);
  }
  S f(const V& woW, const V& wiW, int flags) const {  // bsdf.dart:187-211
    if (vdot(wiW, ng) * vdot(woW, ng) > 0) flags = flags & ~BSDF_TRANSMISSION;
    else flags = flags & ~BSDF_REFLECTION;
    V wo = worldToLocal(woW), wi = worldToLocal(wiW);
    S f{0, 0, 0};
    for (int i = 0; i < nBxDFs; ++i)
      if (bx[i].matches(flags)) f = sadd(f, bx[i].f(wo, wi));
    return f;
  }
  D pdf(const V& woW, const V& wiW, int flags) const {  // bsdf.dart:135-156
    if (nBxDFs == 0) return 0.0;
    V wo = worldToLocal(woW);
    V wi = worldToLocal(wiW);
    D pdf = 0.0;
    int matchingComps = 0;
    for (int i = 0; i < nBxDFs; ++i)
      if (bx[i].matches(flags)) {
        ++matchingComps;
        pdf += bx[i].pdf(wo, wi);
      }
    return matchingComps > 0 ? pdf / matchingComps : 0.0;
  }
  S sample_f(const V& woW, V* wiW, D uDir0, D uDir1, D uComponent, D* pdf, int flags, int* sampledType) const {
    // bsdf.dart:53-133
    int matchingComps = numComponents(flags);
    if (matchingComps == 0) {
      *pdf = 0.0;
      if (sampledType) *sampledType = 0;
      return S{0, 0, 0};
    }
    int which = std::min((int)std::floor(uComponent * matchingComps), matchingComps - 1);
    const BxDF* bxdf = nullptr;
    int count = which;
    for (int i = 0; i < nBxDFs; ++i)
      if (bx[i].matches(flags) && count-- == 0) {
        bxdf = &bx[i];
        break;
      }
    V wo = worldToLocal(woW);
    V wi{0, 0, 0};
    *pdf = 0.0;
    S f = bxdf->sample_f(wo, &wi, uDir0, uDir1, pdf);
    if (*pdf == 0.0) {
      if (sampledType) *sampledType = 0;
      return S{0, 0, 0};
    }
    if (sampledType) *sampledType = bxdf->type;
    *wiW = localToWorld(wi);
    if (!(bxdf->type & BSDF_SPECULAR) && matchingComps > 1)
      for (int i = 0; i < nBxDFs; ++i)
        if (&bx[i] != bxdf && bx[i].matches(flags)) *pdf += bx[i].pdf(wo, wi);
    if (matchingComps > 1) *pdf /= matchingComps;
    if ((bxdf->type & BSDF_SPECULAR) == 0) {
      f = S{0, 0, 0};
      if (vdot(*wiW, ng) * vdot(woW, ng) > 0) flags = flags & ~BSDF_TRANSMISSION;
      else flags = flags & ~BSDF_REFLECTION;
      for (int i = 0; i < nBxDFs; ++i)
        if (bx[i].matches(flags)) f = sadd(f, bx[i].f(wo, wi));
    }
    return f;
  }
};

// Intersection.getBSDF -> GeometricPrimitive.getBSDF -> Triangle.getShadingGeometry
// (copy, no per-vertex N/S) -> MatteMaterial.getBSDF.
// Triangle.getShadingGeometry (triangle.dart:271-364) for meshes with per-vertex N and / or S; dndu / dndv and the
// ray differentials only feed texture filtering and are not restated.
static DG tri_shading_geometry(const Scene& sc, const Prim& pr, const DG& dg) {
  const Mesh& m = sc.meshes[pr.mesh];
  if (pr.quadric >= 0 || (!m.hasN && !m.hasS)) return dg;  // dgShading.copy(dg) (:273-276, shape.dart:78-82)
  D uv[6];
  tri_uvs(sc, pr.mesh, pr.v, uv);
  D A[4] = {uv[2] - uv[0], uv[4] - uv[0], uv[3] - uv[1], uv[5] - uv[1]};
  D C[2] = {dg.u - uv[0], dg.v - uv[1]};
  D bx, by, bz;
  {  // SolveLinearSystem2x2 (common.dart:170-185)
    D det = A[0] * A[3] - A[1] * A[2];
    bool ok = !(std::fabs(det) < 1.0e-10);
    if (ok) {
      by = (A[3] * C[0] - A[1] * C[1]) / det;
      bz = (A[0] * C[1] - A[2] * C[0]) / det;
      if (std::isnan(by) || std::isnan(bz)) ok = false;
    }
    if (!ok) bx = by = bz = 1.0 / 3.0;  // degenerate parametric mapping
    else bx = 1.0 - by - bz;
  }
  auto vtx = [&](const std::vector<float>& a, int k) {
    return V{(D)a[3 * (size_t)pr.v[k]], (D)a[3 * (size_t)pr.v[k] + 1], (D)a[3 * (size_t)pr.v[k] + 2]};
  };
  V ns, ss, ts;
  if (m.hasN) ns = vnormalize(xfNormal(m.w2o, vadd(vadd(vmul(vtx(sc.N, 0), bx), vmul(vtx(sc.N, 1), by)), vmul(vtx(sc.N, 2), bz))));
  else ns = dg.nn;
  if (m.hasS) ss = vnormalize(xfVector(m.o2w, vadd(vadd(vmul(vtx(sc.S, 0), bx), vmul(vtx(sc.S, 1), by)), vmul(vtx(sc.S, 2), bz))));
  else ss = vnormalize(dg.dpdu);
  ts = vcross(ss, ns);
  if (vlen2(ts) > 0.0) {
    ts = vnormalize(ts);
    ss = vcross(ts, ns);
  } else {  // Vector.CoordinateSystem(ns, ss, ts) (vector.dart:198-214)
    if (std::fabs(ns.x) > std::fabs(ns.y)) {
      D invLen = 1.0 / std::sqrt(ns.x * ns.x + ns.z * ns.z);
      ss = vec(-ns.z * invLen, 0.0, ns.x * invLen);
    } else {
      D invLen = 1.0 / std::sqrt(ns.y * ns.y + ns.z * ns.z);
      ss = vec(0.0, ns.z * invLen, -ns.y * invLen);
    }
    ts = vcross(ns, ss);
  }
  DG out;
  dg_set(&out, dg.p, ss, ts, m.reverse);  // dgShading.set(dg.p, ss, ts, ...): nn = normalize(ss x ts), flipped like dg's
  out.u = dg.u;
  out.v = dg.v;
  return out;
}
static BSDF make_bsdf(const Scene& sc, const Isect& is) {
  const Mesh& m = sc.meshes[sc.prims[is.prim].mesh];
  const DG dgs = tri_shading_geometry(sc, sc.prims[is.prim], is.dg);  // GeometricPrimitive.getBSDF (geometric_primitive.dart:67-71)
  BSDF b;
  b.p = dgs.p;
  b.ng = is.dg.nn;                 // BSDF(dgs, dgGeom.nn)
  b.nn = dgs.nn;
  b.sn = vnormalize(dgs.dpdu);     // bsdf.dart:45-51
  b.tn = vcross(b.nn, b.sn);
  auto clampS = [](const S& c) { return rgb(clampD(c.r, 0.0, kInf), clampD(c.g, 0.0, kInf), clampD(c.b, 0.0, kInf)); };
  if (m.matType == 0) {  // matte_material.dart:41-65
    S r = clampS(m.Kd);
    D sig = clampD(m.sigma, 0.0, 90.0);
    if (!sblack(r)) {
      BxDF x;
      x.kind = 0;
      x.type = BSDF_REFLECTION | BSDF_DIFFUSE;
      x.R = r;
      if (sig != 0.0) {  // OrenNayar(r, sig) (oren_nayar.dart:24-32)
        x.kind = 3;
        D sigma = (M_PI / 180.0) * sig;
        D sigma2 = sigma * sigma;
        x.A = 1.0 - (sigma2 / (2.0 * (sigma2 + 0.33)));
        x.B = 0.45 * sigma2 / (sigma2 + 0.09);
      }
      b.add(x);
    }
  } else if (m.matType == 3) {  // plastic_material.dart:43-70: Lambertian(kd) + Microfacet(ks, FresnelDielectric(1.5, 1), Blinn(1/roughness))
    S kd = clampS(m.Kd), ks = clampS(m.Kr);
    if (!sblack(kd)) {
      BxDF x;
      x.kind = 0;
      x.type = BSDF_REFLECTION | BSDF_DIFFUSE;
      x.R = kd;
      b.add(x);
    }
    if (!sblack(ks)) {
      BxDF x;
      x.kind = 4;
      x.type = BSDF_REFLECTION | BSDF_GLOSSY;
      x.R = ks;
      x.ei = 1.5;
      x.et = 1.0;
      x.exponent = 1.0 / m.ior;  // roughness travels in the `ior` field
      if (x.exponent > 10000.0 || std::isnan(x.exponent)) x.exponent = 10000.0;
      b.add(x);
    }
  } else if (m.matType == 1) {  // mirror_material.dart:38-55
    S r = clampS(m.Kr);
    if (!sblack(r)) {
      BxDF x;
      x.kind = 1;
      x.type = BSDF_REFLECTION | BSDF_SPECULAR;
      x.R = r;
      b.add(x);
    }
  } else {  // glass_material.dart:44-69
    S r = clampS(m.Kr), t = clampS(m.Kt);
    if (!sblack(r)) {
      BxDF x;
      x.kind = 1;
      x.type = BSDF_REFLECTION | BSDF_SPECULAR;
      x.R = r;
      x.dielectric = true;
      x.ei = 1.0;
      x.et = m.ior;
      b.add(x);
    }
    if (!sblack(t)) {
      BxDF x;
      x.kind = 2;
      x.type = BSDF_TRANSMISSION | BSDF_SPECULAR;
      x.R = t;
      x.ei = 1.0;
      x.et = m.ior;
      b.add(x);
    }
  }
  return b;
}
static S isect_Le(const Scene& sc, const Isect& is, const V& wo) {  // intersection.dart:60-63
  int li = sc.meshes[sc.prims[is.prim].mesh].light;
  return li >= 0 ? light_L(sc.lights[li], is.dg.nn, wo) : S{0, 0, 0};
}

// ---------------------------------------------------------------------------
// Random-number source seen by the integrators.  Serial mode: the task's
// shared RNG (sampler_renderer.dart:137).  Replay mode: recorded values.
// ---------------------------------------------------------------------------
struct LiRng {
  DartRandom* rng = nullptr;
  std::vector<D>* record = nullptr;  // every randomFloat drawn inside Li
  const D* replay = nullptr;
  int replayN = 0, replayPos = 0;
  bool underflow = false;
  D randomFloat() {
    D v;
    if (replay) {
      if (replayPos < replayN) v = replay[replayPos++];
      else { underflow = true; v = 0.0; }
    } else {
      v = rng->randomFloat();
    }
    if (record) record->push_back(v);
    return v;
  }
};

struct SampleView {  // Sample (sample.dart:23-79) flattened: [imageU, imageV, lensU, lensV, time, oneD..., twoD...]
  const float* v;
  int n1D;
  D oneD(int i) const { return (D)v[5 + i]; }
  D twoD(int i, int k) const { return (D)v[5 + n1D + 2 * i + k]; }
};

// EstimateDirect (integrator.dart:119-185)
static S EstimateDirect(const Scene& sc, int lightIdx, const V& p, const V& n, const V& wo, D rayEpsilon,
                        const BSDF& bsdf, D lsU0, D lsU1, D lsComp, D bsU0, D bsU1, D bsComp, int flags) {
  const Light& light = sc.lights[lightIdx];
  S Ld{0, 0, 0};
  V wi{0, 0, 0};
  D lightPdf = 0.0, bsdfPdf = 0.0;
  Ray vr;
  S Li;
  if (light.kind >= 2) {
    // delta lights: no MIS, no BSDF-sampling half (integrator.dart:146-150,153)
    lightPdf = 1.0;
    if (light.kind == 4) {
      // DistantLight.sampleLAtPoint (distant_light.dart:54-61): visibility.setRay(p, eps, wi)
      wi = light.lightPos;
      vr = Ray{p, wi, rayEpsilon, kInf, 0.0, 0};
      Li = light.Lemit;
    } else {
      // PointLight.sampleLAtPoint (point_light.dart:41-47) / SpotLight (spot_light.dart:78-85)
      wi = vnormalize(vsub(light.lightPos, p));
      D dist = vlen(vsub(light.lightPos, p));  // VisibilityTester.setSegment(p, eps, lightPos, 0)
      vr = Ray{p, vdiv(vsub(light.lightPos, p), dist), rayEpsilon, dist * (1.0 - 0.0), 0.0, 0};
      if (light.kind == 3) {  // intensity * falloff(-wi) / DistanceSquared (spot_light.dart:54-70)
        V wl = vnormalize(xfVector(light.w2l, vneg(wi)));
        D costheta = wl.z, fo;
        if (costheta < light.cosTotalWidth) fo = 0.0;
        else if (costheta > light.cosFalloffStart) fo = 1.0;
        else {
          D delta = (costheta - light.cosTotalWidth) / (light.cosFalloffStart - light.cosTotalWidth);
          fo = delta * delta * delta * delta;
        }
        Li = sdivD(smulD(light.Lemit, fo), vlen2(vsub(light.lightPos, p)));
      } else {
        Li = sdivD(light.Lemit, vlen2(vsub(light.lightPos, p)));
      }
    }
    if (lightPdf > 0.0 && !sblack(Li)) {
      S f = bsdf.f(wo, wi, flags);
      if (!sblack(f) && !bvh_intersectP(sc, vr)) {
        Li = smul(Li, S{1, 1, 1});
        Ld = sadd(Ld, smulD(smul(f, Li), (vabsdot(wi, n) / lightPdf)));
      }
    }
    return Ld;
  }
  if (light.kind == 0) {
    // light.sampleLAtPoint (diffuse_area_light.dart:60-70)
    V ns{0, 0, 0};
    V ps = shapeset_sample(sc, light, lsU0, lsU1, lsComp, &ns, p);
    wi = vnormalize(vsub(ps, p));
    lightPdf = shapeset_pdf(sc, light, p, wi);
    // VisibilityTester.setSegment (visibility_tester.dart:26-29)
    D dist = vlen(vsub(ps, p));
    vr = Ray{p, vdiv(vsub(ps, p), dist), rayEpsilon, dist * (1.0 - 1.0e-3), 0.0, 0};
    Li = light_L(light, ns, vneg(wi));
  } else {
    // InfiniteAreaLight.sampleLAtPoint (infinite_area_light.dart:92-131); visibility.setRay (visibility_tester.dart:31-33)
    Li = sc.env.sampleL(lsU0, lsU1, &wi, &lightPdf);
    vr = Ray{p, wi, rayEpsilon, kInf, 0.0, 0};
  }
  if (lightPdf > 0.0 && !sblack(Li)) {
    S f = bsdf.f(wo, wi, flags);
    if (!sblack(f) && !bvh_intersectP(sc, vr)) {
      Li = smul(Li, S{1, 1, 1});  // transmittance == 1 (emission_integrator.dart:85-87)
      bsdfPdf = bsdf.pdf(wo, wi, flags);
      D weight = PowerHeuristic(1, lightPdf, 1, bsdfPdf);
      Ld = sadd(Ld, smulD(smul(f, Li), (vabsdot(wi, n) * weight / lightPdf)));
    }
  }
  // BSDF sampling half (not a delta light)
  {
    int sampledType = 0;
    S f = bsdf.sample_f(wo, &wi, bsU0, bsU1, bsComp, &bsdfPdf, flags, &sampledType);
    if (!sblack(f) && bsdfPdf > 0.0) {
      D weight = 1.0;
      if ((sampledType & BSDF_SPECULAR) == 0) {
        lightPdf = light.kind == 0 ? shapeset_pdf(sc, light, p, wi) : sc.env.pdfW(wi);
        if (lightPdf == 0.0) return Ld;
        weight = PowerHeuristic(1, bsdfPdf, 1, lightPdf);
      }
      Isect lightIsect;
      S Li2{0, 0, 0};
      Ray ray{p, wi, rayEpsilon, kInf, 0.0, 0};
      if (bvh_intersect(sc, ray, &lightIsect)) {
        if (sc.meshes[sc.prims[lightIsect.prim].mesh].light == lightIdx) Li2 = isect_Le(sc, lightIsect, vneg(wi));
      } else {
        // light.Le(ray): 0 for area lights (light.dart:70-72), the map for the infinite light
        Li2 = light.kind == 1 ? sc.env.Le(wi) : S{0, 0, 0};
      }
      if (!sblack(Li2)) {
        Li2 = smul(Li2, S{1, 1, 1});
        Ld = sadd(Ld, smulD(smul(f, Li2), (vabsdot(wi, n) * weight / bsdfPdf)));
      }
    }
  }
  return Ld;
}

struct IntegratorCfg {
  int kind;      // 0 = DirectLighting(all), 1 = Path, 2 = DirectLighting(one)
  int maxDepth;
};
static const int SAMPLE_DEPTH = 3;  // path_integrator.dart:139

// Sample-slot layout.
static void sample_layout(const Scene& sc, const IntegratorCfg& cfg, std::vector<int>* n1D, std::vector<int>* n2D,
                          bool roundPow2) {
  n1D->clear();
  n2D->clear();
  auto rp2 = [](int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; };
  if (cfg.kind == 1) {  // path_integrator.dart:124-131
    for (int i = 0; i < SAMPLE_DEPTH; ++i) {
      n1D->push_back(1); n2D->push_back(1);  // LightSampleOffsets
      n1D->push_back(1);                     // lightNumOffset
      n1D->push_back(1); n2D->push_back(1);  // bsdfSampleOffsets
      n1D->push_back(1); n2D->push_back(1);  // pathSampleOffsets
    }
  } else if (cfg.kind == 2) {  // strategy "one" (direct_lighting_integrator.dart:82-87)
    n1D->push_back(1); n2D->push_back(1);  // lightSampleOffsets[0] = LightSampleOffsets(1, sample)
    n1D->push_back(1);                     // lightNumOffset = sample.add1D(1)
    n1D->push_back(1); n2D->push_back(1);  // bsdfSampleOffsets[0] = BSDFSampleOffsets(1, sample)
  } else {  // direct_lighting_integrator.dart:70-81
    for (size_t i = 0; i < sc.lights.size(); ++i) {
      int ns = sc.lights[i].nSamples;
      if (roundPow2) ns = rp2(ns);
      n1D->push_back(ns); n2D->push_back(ns);
      n1D->push_back(ns); n2D->push_back(ns);
    }
  }
  n1D->push_back(1);  // emission_integrator.dart:26-29 tau
  n1D->push_back(1);  // scatter
}

// PathIntegrator.Li (path_integrator.dart:29-122)
static S PathLi(const Scene& sc, const IntegratorCfg& cfg, const Ray& r, const Isect& isect0, const SampleView& sv,
                LiRng& rng) {
  S pathThroughput{1, 1, 1};
  S L{0, 0, 0};
  Ray ray = r;
  bool specularBounce = false;
  Isect isectP = isect0;
  int nLights = (int)sc.lights.size();
  for (int bounces = 0;; ++bounces) {
    if (bounces == 0 || specularBounce) L = sadd(L, smul(pathThroughput, isect_Le(sc, isectP, vneg(ray.d))));
    BSDF bsdf = make_bsdf(sc, isectP);
    V p = bsdf.p;
    V n = bsdf.nn;
    V wo = vneg(ray.d);
    // UniformSampleOneLight (integrator.dart:79-117)
    S Ld{0, 0, 0};
    if (nLights > 0) {
      int lightNum;
      D ls0, ls1, lsc, bs0, bs1, bsc;
      if (bounces < SAMPLE_DEPTH) {
        lightNum = (int)std::floor(sv.oneD(4 * bounces + 1) * nLights);
        ls0 = sv.twoD(3 * bounces + 0, 0); ls1 = sv.twoD(3 * bounces + 0, 1); lsc = sv.oneD(4 * bounces + 0);
        bs0 = sv.twoD(3 * bounces + 1, 0); bs1 = sv.twoD(3 * bounces + 1, 1); bsc = sv.oneD(4 * bounces + 2);
      } else {
        lightNum = (int)std::floor(rng.randomFloat() * nLights);
        ls0 = r32(rng.randomFloat()); ls1 = r32(rng.randomFloat()); lsc = rng.randomFloat();  // light_sample.dart:46-51
        bs0 = r32(rng.randomFloat()); bs1 = r32(rng.randomFloat()); bsc = rng.randomFloat();  // bsdf_sample.dart:37-42
      }
      lightNum = std::min(lightNum, nLights - 1);
      Ld = smulD(EstimateDirect(sc, lightNum, p, n, wo, isectP.rayEpsilon, bsdf, ls0, ls1, lsc, bs0, bs1, bsc,
                                BSDF_ALL & ~BSDF_SPECULAR),
                 (D)nLights);
    }
    L = sadd(L, smul(pathThroughput, Ld));
    // Outgoing BSDF sample
    D o0, o1, oc;
    if (bounces < SAMPLE_DEPTH) {
      o0 = sv.twoD(3 * bounces + 2, 0); o1 = sv.twoD(3 * bounces + 2, 1); oc = sv.oneD(4 * bounces + 3);
    } else {
      o0 = r32(rng.randomFloat()); o1 = r32(rng.randomFloat()); oc = rng.randomFloat();
    }
    V wi{0, 0, 0};
    D pdf = 0.0;
    int flags = 0;
    S f = bsdf.sample_f(wo, &wi, o0, o1, oc, &pdf, BSDF_ALL, &flags);
    if (sblack(f) || pdf == 0.0) break;
    specularBounce = (flags & BSDF_SPECULAR) != 0;
    pathThroughput = smul(pathThroughput, sdivD(smulD(f, vabsdot(wi, n)), pdf));
    ray = Ray{p, wi, isectP.rayEpsilon, kInf, ray.time, ray.depth + 1};  // RayDifferential.child
    if (bounces > 3) {
      D continueProbability = std::min(0.5, slum(pathThroughput));
      if (rng.randomFloat() > continueProbability) break;
      pathThroughput = sdivD(pathThroughput, continueProbability);
    }
    if (bounces == cfg.maxDepth) break;
    Isect localIsect;
    if (!bvh_intersect(sc, ray, &localIsect)) {
      if (specularBounce)  // path_integrator.dart:107-111: area lights' Le(ray) is 0 (light.dart:70-72)
        for (size_t i = 0; i < sc.lights.size(); ++i)
          if (sc.lights[i].kind == 1) L = sadd(L, smul(pathThroughput, sc.env.Le(ray.d)));
      break;
    }
    // transmittance == 1
    isectP = localIsect;
  }
  return L;
}

// SamplerRenderer.Li (sampler_renderer.dart:67-98) as the integrators call it for a spawned ray: intersect, the
// surface integrator on a hit, the sum of light.Le(ray) on a miss; T = 1, Lvi = 0 without a VolumeRegion
// (emission_integrator.dart:39-42).  Only DirectLighting recurses through it (PathIntegrator iterates).
static S DirectLi(const Scene& sc, const IntegratorCfg& cfg, const Ray& ray, const Isect& isect, const SampleView& sv,
                  const std::vector<int>& n1D, const std::vector<int>& n2D, LiRng& rng);
static S RendererLi(const Scene& sc, const IntegratorCfg& cfg, Ray ray, const SampleView& sv, const std::vector<int>& n1D,
                    const std::vector<int>& n2D, LiRng& rng) {
  Isect isect;
  S Li{0, 0, 0};
  if (bvh_intersect(sc, ray, &isect)) {
    Li = DirectLi(sc, cfg, ray, isect, sv, n1D, n2D, rng);
  } else {
    for (const Light& l : sc.lights) Li = sadd(Li, l.kind == 1 ? sc.env.Le(ray.d) : S{0, 0, 0});
  }
  return sadd(smul(S{1, 1, 1}, Li), S{0, 0, 0});  // T * Li + Lvi
}
// Integrator.SpecularReflect / SpecularTransmit (integrator.dart:187-233 / :235-290).  The ray differentials only
// feed texture filtering (constant textures here) and are not restated.
static S SpecularBounce(const Scene& sc, const IntegratorCfg& cfg, const Ray& ray, const BSDF& bsdf, const Isect& isect,
                        int lobeFlags, const SampleView& sv, const std::vector<int>& n1D, const std::vector<int>& n2D,
                        LiRng& rng) {
  V wo = vneg(ray.d);
  V wi{0, 0, 0};
  D pdf = 0.0;
  V p = bsdf.p, n = bsdf.nn;
  // new BSDFSample.random(rng): uDir[0], uDir[1] (Float32List), uComponent (bsdf_sample.dart:37-42)
  D u0 = r32(rng.randomFloat()), u1 = r32(rng.randomFloat()), uc = rng.randomFloat();
  S f = bsdf.sample_f(wo, &wi, u0, u1, uc, &pdf, lobeFlags | BSDF_SPECULAR, nullptr);
  S L{0, 0, 0};
  if (pdf > 0.0 && !sblack(f) && vabsdot(wi, n) != 0.0) {
    Ray rd;  // RayDifferential.child(p, wi, ray, isect.rayEpsilon) (ray.dart: depth = parent.depth + 1, same time)
    rd.o = p;
    rd.d = wi;
    rd.mint = isect.rayEpsilon;
    rd.maxt = kInf;
    rd.time = ray.time;
    rd.depth = ray.depth + 1;
    S Li = RendererLi(sc, cfg, rd, sv, n1D, n2D, rng);
    L = smulD(smul(f, Li), vabsdot(wi, n) / pdf);  // f * Li * (AbsDot(wi, n) / pdf)
  }
  return L;
}

// DirectLightingIntegrator.Li, strategies "all" and "one" (direct_lighting_integrator.dart:30-68)
static S DirectLi(const Scene& sc, const IntegratorCfg& cfg, const Ray& ray, const Isect& isect, const SampleView& sv,
                  const std::vector<int>& n1D, const std::vector<int>& n2D, LiRng& rng) {
  S L{0, 0, 0};
  BSDF bsdf = make_bsdf(sc, isect);
  V wo = vneg(ray.d);
  V p = bsdf.p;
  V n = bsdf.nn;
  L = sadd(L, isect_Le(sc, isect, wo));
  if (!sc.lights.empty() && cfg.kind == 2) {
    // SAMPLE_ONE_UNIFORM (direct_lighting_integrator.dart:51-55): UniformSampleOneLight (integrator.dart:79-117) with the
    // integrator's own slots -- oneD: [light component, lightNum, bsdf component], twoD: [light position, bsdf direction]
    const int nLights = (int)sc.lights.size();
    int lightNum = (int)std::floor(sv.oneD(1) * nLights);
    lightNum = std::min(lightNum, nLights - 1);
    L = sadd(L, smulD(EstimateDirect(sc, lightNum, p, n, wo, isect.rayEpsilon, bsdf, sv.twoD(0, 0), sv.twoD(0, 1), sv.oneD(0), sv.twoD(1, 0),
                                     sv.twoD(1, 1), sv.oneD(2), BSDF_ALL & ~BSDF_SPECULAR),
                      (D)nLights));
  } else if (!sc.lights.empty()) {
    // UniformSampleAllLights (integrator.dart:39-77)
    S Lall{0, 0, 0};
    int off1 = 0, off2 = 0;  // running float offsets of the slots
    for (size_t i = 0; i < sc.lights.size(); ++i) {
      int nSamples = n1D[2 * i];
      int lc = off1;                 // light component slot
      int bc = off1 + n1D[2 * i];    // bsdf component slot
      int lp = off2;                 // light pos slot (floats)
      int bd = off2 + 2 * n2D[2 * i];
      off1 += n1D[2 * i] + n1D[2 * i + 1];
      off2 += 2 * (n2D[2 * i] + n2D[2 * i + 1]);
      int base2 = 5;
      for (size_t k = 0; k < n1D.size(); ++k) base2 += n1D[k];
      S Ld{0, 0, 0};
      for (int j = 0; j < nSamples; ++j) {
        D ls0 = sv.v[base2 + lp + 2 * j], ls1 = sv.v[base2 + lp + 2 * j + 1], lsc = sv.v[5 + lc + j];
        D bs0 = sv.v[base2 + bd + 2 * j], bs1 = sv.v[base2 + bd + 2 * j + 1], bsc = sv.v[5 + bc + j];
        Ld = sadd(Ld, EstimateDirect(sc, (int)i, p, n, wo, isect.rayEpsilon, bsdf, ls0, ls1, lsc, bs0, bs1, bsc,
                                     BSDF_ALL & ~BSDF_SPECULAR));
      }
      Lall = sadd(Lall, sdivD(Ld, (D)nSamples));
    }
    L = sadd(L, Lall);
  }
  if (ray.depth + 1 < cfg.maxDepth) {
    // Trace rays for specular reflection and refraction (direct_lighting_integrator.dart:59-65): each call draws a
    // BSDFSample.random(rng) (3 floats) whether or not the BSDF has such a lobe, and recurses through Renderer.Li
    L = sadd(L, SpecularBounce(sc, cfg, ray, bsdf, isect, BSDF_REFLECTION, sv, n1D, n2D, rng));
    L = sadd(L, SpecularBounce(sc, cfg, ray, bsdf, isect, BSDF_TRANSMISSION, sv, n1D, n2D, rng));
  }
  return L;
}

// ---------------------------------------------------------------------------
// Camera (perspective_camera.dart:93-132) and film (image_film.dart)
// ---------------------------------------------------------------------------
struct Camera {
  float r2c[16], c2w[16];
  D lensRadius, focalDistance;
  int type = 0;      // 0 perspective, 1 orthographic (orthographic_camera.dart:52-80), 2 environment (environment_camera.dart:42-52)
  int xres = 0, yres = 0;  // film resolution (environment camera)
};
static Ray generateRay(const Camera& cam, D imageX, D imageY, D lensU, D lensV, D time) {
  Ray ray;
  ray.mint = 0.0;
  ray.maxt = kInf;
  if (cam.type == 2) {
    // EnvironmentCamera.generateRay (environment_camera.dart:42-52): a lat-long direction from the camera origin
    D theta = kPi * imageY / (D)cam.yres;
    D phi = 2 * kPi * imageX / (D)cam.xres;
    ray.o = vec(0, 0, 0);
    ray.d = vec(std::sin(theta) * std::cos(phi), std::cos(theta), std::sin(theta) * std::sin(phi));
    ray.time = time;
    ray.depth = 0;
    ray.o = xfPoint(cam.c2w, ray.o);
    ray.d = xfVector(cam.c2w, ray.d);
    return ray;
  }
  V Pras = vec(imageX, imageY, 0.0);
  V Pcamera = xfPoint(cam.r2c, Pras);
  if (cam.type == 1) {  // OrthographicCamera.generateRay (orthographic_camera.dart:52-80)
    ray.o = Pcamera;
    ray.d = vec(0.0, 0.0, 1.0);
  } else {
    ray.o = vec(0, 0, 0);
    ray.d = vnormalize(Pcamera);
  }
  if (cam.lensRadius > 0.0) {
    D lu, lv;
    ConcentricSampleDisk(lensU, lensV, &lu, &lv);
    lu *= cam.lensRadius;
    lv *= cam.lensRadius;
    D ft = cam.focalDistance / ray.d.z;
    V Pfocus = pointAt(ray, ft);
    ray.o = vec(lu, lv, 0.0);
    ray.d = vnormalize(vsub(Pfocus, ray.o));
  }
  ray.time = time;
  ray.depth = 0;
  ray.o = xfPoint(cam.c2w, ray.o);   // AnimatedTransform static path (animated_transform.dart:158-169)
  ray.d = xfVector(cam.c2w, ray.d);
  return ray;
}

struct Film {  // ImageFilm (image_film.dart:51-97)
  int xres, yres, left, top, width, height;
  D fxw, fyw, invX, invY;
  float table[256];
  std::vector<float> Lxyz, weightSum;
  void init(int xr, int yr, const double crop[4], D xw, D yw, const float* tbl) {
    xres = xr; yres = yr;
    left = (int)std::ceil(xres * (D)crop[0]);
    width = std::max(1, (int)std::ceil(xres * (D)crop[1]) - left);
    top = (int)std::ceil(yres * (D)crop[2]);
    height = std::max(1, (int)std::ceil(yres * (D)crop[3]) - top);
    fxw = xw; fyw = yw; invX = 1.0 / xw; invY = 1.0 / yw;
    memcpy(table, tbl, sizeof(table));
    Lxyz.assign((size_t)width * height * 3, 0.0f);
    weightSum.assign((size_t)width * height, 0.0f);
  }
  void getSampleExtent(int e[4]) const {  // image_film.dart:247-252
    e[0] = (int)std::floor(left + 0.5 - fxw);
    e[1] = (int)std::ceil(left + 0.5 + width + fxw);
    e[2] = (int)std::floor(top + 0.5 - fyw);
    e[3] = (int)std::ceil(top + 0.5 + height + fyw);
  }
  void addSample(D imageX, D imageY, const S& L) {  // image_film.dart:99-185 (preview repaint omitted)
    D dimageX = imageX - 0.5, dimageY = imageY - 0.5;
    int x0 = (int)std::ceil(dimageX - fxw), x1 = (int)std::floor(dimageX + fxw);
    int y0 = (int)std::ceil(dimageY - fyw), y1 = (int)std::floor(dimageY + fyw);
    x0 = std::max(x0, left); x1 = std::min(x1, left + width - 1);
    y0 = std::max(y0, top);  y1 = std::min(y1, top + height - 1);
    if ((x1 - x0) < 0 || (y1 - y0) < 0) return;
    D xyz[3];  // XYZColor.from(RGBColor): Float32List store (xyz_color.dart:39-42; spectrum.dart:294-298)
    xyz[0] = r32(0.412453 * L.r + 0.357580 * L.g + 0.180423 * L.b);
    xyz[1] = r32(0.212671 * L.r + 0.715160 * L.g + 0.072169 * L.b);
    xyz[2] = r32(0.019334 * L.r + 0.119193 * L.g + 0.950227 * L.b);
    for (int y = y0; y <= y1; ++y) {
      D fy = std::fabs((y - dimageY) * invY * 16);
      int iy = std::min((int)std::floor(fy), 15);
      for (int x = x0; x <= x1; ++x) {
        D fx = std::fabs((x - dimageX) * invX * 16);
        int ix = std::min((int)std::floor(fx), 15);
        D filterWt = table[iy * 16 + ix];
        size_t pi = (size_t)(y - top) * width + (x - left);
        Lxyz[3 * pi] = (float)((D)Lxyz[3 * pi] + filterWt * xyz[0]);
        Lxyz[3 * pi + 1] = (float)((D)Lxyz[3 * pi + 1] + filterWt * xyz[1]);
        Lxyz[3 * pi + 2] = (float)((D)Lxyz[3 * pi + 2] + filterWt * xyz[2]);
        weightSum[pi] = (float)((D)weightSum[pi] + filterWt);
      }
    }
  }
};
// ImageFilm.writeImage for one pixel (image_film.dart:268-299), splat == 0.
static void write_pixel(D X, D Y, D Z, D weightSum, float out[3]) {
  D c0 = 3.240479 * X - 1.537150 * Y - 0.498535 * Z;   // spectrum.dart:287-291
  D c1 = -0.969256 * X + 1.875991 * Y + 0.041556 * Z;
  D c2 = 0.055648 * X - 0.204043 * Y + 1.057311 * Z;
  out[0] = out[1] = out[2] = 0.0f;
  if (weightSum != 0.0) {
    D invWt = 1.0 / weightSum;
    out[0] = (float)std::max(0.0, c0 * invWt);
    out[1] = (float)std::max(0.0, c1 * invWt);
    out[2] = (float)std::max(0.0, c2 * invWt);
  }
  // += splatScale * splatRGB (== +0.0)
  out[0] = (float)((D)out[0] + 0.0);
  out[1] = (float)((D)out[1] + 0.0);
  out[2] = (float)((D)out[2] + 0.0);
}

// GetSubWindow (common.dart:52-73)
static void GetSubWindow(int w, int h, int num, int count, int extents[4]) {
  int nx = count, ny = 1;
  while ((nx & 0x1) == 0 && 2 * w * ny < h * nx) {
    nx >>= 1;
    ny <<= 1;
  }
  int xo = num % nx, yo = num / nx;
  D tx0 = (D)xo / nx, tx1 = (D)(xo + 1) / nx;
  D ty0 = (D)yo / ny, ty1 = (D)(yo + 1) / ny;
  auto lerp = [](D t, D v1, D v2) { return v1 * (1.0 - t) + v2 * t; };
  extents[0] = (int)std::floor(lerp(tx0, 0, w));
  extents[1] = std::min((int)std::floor(lerp(tx1, 0, w)), w);
  extents[2] = (int)std::floor(lerp(ty0, 0, h));
  extents[3] = std::min((int)std::floor(lerp(ty1, 0, h)), h);
}

// LDPixelSample (montecarlo.dart:407-473) -> out[spp][nFloats] in the flat layout.
template <class RNG>
static void LDPixelSample(int nPixelSamples, const std::vector<int>& n1D, const std::vector<int>& n2D,
                          std::vector<float>& buffer, RNG& rng, std::vector<float>& out, int nFloats) {
  size_t need = 5 * (size_t)nPixelSamples;
  for (int c : n1D) need += (size_t)c * nPixelSamples;
  for (int c : n2D) need += 2 * (size_t)c * nPixelSamples;
  buffer.resize(need);
  float* imageSamples = buffer.data();
  float* lensSamples = imageSamples + 2 * nPixelSamples;
  float* timeSamples = lensSamples + 2 * nPixelSamples;
  float* cur = timeSamples + nPixelSamples;
  std::vector<float*> oneD(n1D.size()), twoD(n2D.size());
  for (size_t i = 0; i < n1D.size(); ++i) { oneD[i] = cur; cur += (size_t)n1D[i] * nPixelSamples; }
  for (size_t i = 0; i < n2D.size(); ++i) { twoD[i] = cur; cur += 2 * (size_t)n2D[i] * nPixelSamples; }
  LDShuffleScrambled2D(1, nPixelSamples, imageSamples, rng);
  LDShuffleScrambled2D(1, nPixelSamples, lensSamples, rng);
  LDShuffleScrambled1D(1, nPixelSamples, timeSamples, rng);
  for (size_t i = 0; i < n1D.size(); ++i) LDShuffleScrambled1D(n1D[i], nPixelSamples, oneD[i], rng);
  for (size_t i = 0; i < n2D.size(); ++i) LDShuffleScrambled2D(n2D[i], nPixelSamples, twoD[i], rng);
  out.resize((size_t)nPixelSamples * nFloats);
  for (int i = 0; i < nPixelSamples; ++i) {
    float* s = &out[(size_t)i * nFloats];
    s[0] = imageSamples[2 * i];
    s[1] = imageSamples[2 * i + 1];
    s[2] = lensSamples[2 * i];
    s[3] = lensSamples[2 * i + 1];
    s[4] = timeSamples[i];
    int o = 5;
    for (size_t j = 0; j < n1D.size(); ++j)
      for (int k = 0; k < n1D[j]; ++k) s[o++] = oneD[j][n1D[j] * i + k];
    for (size_t j = 0; j < n2D.size(); ++j)
      for (int k = 0; k < 2 * n2D[j]; ++k) s[o++] = twoD[j][2 * n2D[j] * i + k];
  }
}

// Counter-mode LD pixel sample: every block draws from its own keyed stream;
// otherwise the same LDShuffleScrambled* code.
static void LDPixelSampleCounter(int nPixelSamples, const std::vector<int>& n1D, const std::vector<int>& n2D,
                                 uint64_t seed, uint64_t pixelIndex, std::vector<float>& out, int nFloats) {
  std::vector<float> tmp;
  out.resize((size_t)nPixelSamples * nFloats);
  int block = 0;
  auto run2D = [&](int ns, int dst) {
    DartRandom rng(counter_key(seed, pixelIndex, (uint64_t)block++, 1));
    tmp.resize(2 * (size_t)ns * nPixelSamples);
    LDShuffleScrambled2D(ns, nPixelSamples, tmp.data(), rng);
    for (int i = 0; i < nPixelSamples; ++i)
      for (int k = 0; k < 2 * ns; ++k) out[(size_t)i * nFloats + dst + k] = tmp[2 * (size_t)ns * i + k];
  };
  auto run1D = [&](int ns, int dst) {
    DartRandom rng(counter_key(seed, pixelIndex, (uint64_t)block++, 1));
    tmp.resize((size_t)ns * nPixelSamples);
    LDShuffleScrambled1D(ns, nPixelSamples, tmp.data(), rng);
    for (int i = 0; i < nPixelSamples; ++i)
      for (int k = 0; k < ns; ++k) out[(size_t)i * nFloats + dst + k] = tmp[(size_t)ns * i + k];
  };
  run2D(1, 0);
  run2D(1, 2);
  run1D(1, 4);
  int o = 5;
  for (size_t j = 0; j < n1D.size(); ++j) { run1D(n1D[j], o); o += n1D[j]; }
  for (size_t j = 0; j < n2D.size(); ++j) { run2D(n2D[j], o); o += 2 * n2D[j]; }
}

}  // namespace

// ===========================================================================
// C API (ctypes)
// ===========================================================================
extern "C" {

struct OrcMesh {
  const float* P;        // nverts*3 world-space f32
  const uint32_t* idx;   // ntris*3
  int32_t nverts, ntris;
  float Kd[3];
  float sigma;
  int32_t reverse_orientation;
  int32_t has_light;
  float L[3];
  int32_t light_nsamples;
  // kind 0: triangle mesh (the fields above); 1: Sphere(radius, z0, z1, phimax [deg]) (sphere.dart:313-321);
  // 2: Disk(height, radius, innerradius, phimax [deg]) (disk.dart:157-165).  Quadrics keep objectToWorld
  // (o2w = .m, w2o = .mInv) and ignore P / idx.
  int32_t kind;
  float o2w[16], w2o[16];
  double params[4];
  // material: 0 matte (Kd, sigma), 1 mirror (Kr), 2 glass (Kr, Kt, index)
  int32_t mat_type;
  float Kr[3], Kt[3];
  double ior;
  double sigma_d;  // matte 'sigma' as the Dart double it is (the f32 field above is ignored)
  // optional per-vertex shading data of a triangle mesh (nverts entries each; null = absent): normals and
  // tangents in OBJECT space (then o2w / w2o above must be set), uvs
  const float* N;
  const float* S;
  const float* uv;
};
struct OrcSceneDesc {
  int32_t nmeshes;
  const OrcMesh* meshes;
  int32_t max_prims_in_node;  // "maxnodeprims", default 4
  // optional InfiniteAreaLight, appended to Scene.lights after the area lights
  int32_t has_env;
  const float* env_texels;  // [h][w][3] level-0 texels of the radiance MIPMap (power-of-two size)
  int32_t env_w, env_h;
  float env_L[3];
  float env_l2w[16], env_w2l[16];
  int32_t env_nsamples;
  // position in Scene.lights: the InfiniteAreaLight precedes the area light of mesh `env_before_mesh`
  // (LightSource and Shape directives append in file order, dartray.dart:368-375,461-466); < 0 => after all
  int32_t env_before_mesh;
  // point lights (point_light.dart): position = lightToWorld(0,0,0) and intensity; point light i precedes the area
  // light of mesh point_before_mesh[i] in Scene.lights (< 0 or >= nmeshes: after all meshes, in array order)
  int32_t npoint_lights;
  const float* point_pos;        // [n][3]
  const float* point_intensity;  // [n][3]
  const int32_t* point_before_mesh;
  // kind per entry (null: all point lights): 2 point, 3 spot, 4 distant (point_pos then holds lightDir);
  // spots: worldToLight [n][16] and (width, falloff start) in degrees [n][2]
  const int32_t* point_kind;
  const float* spot_w2l;
  const double* spot_angles;
};
struct OrcNode {  // the 32-byte marshalled node of SURVEY.md Appendix F
  float bmin[3], bmax[3];
  uint32_t offset;
  uint16_t nprims;
  uint8_t axis, pad;
};
struct OrcRay {
  float o[3], d[3];
  double tmin, tmax;
};
struct OrcHit {
  int32_t prim;  // index in BVH primitive order, -1 = miss
  int32_t pad;
  double t, b1, b2;
};
struct OrcRenderDesc {
  int32_t xres, yres;
  double crop[4];                // cropWindow and the filter widths are Dart doubles
  double filter_xw, filter_yw;
  float filter_table[256];
  float raster_to_camera[16], camera_to_world[16];
  double lens_radius, focal_distance, shutter_open, shutter_close;
  int32_t camera_type;  // 0 perspective, 1 orthographic, 2 environment
  int32_t integrator;  // 0 direct(all), 1 path
  int32_t max_depth;
  int32_t spp;
  int32_t sampler_mode;  // 0 = serial (reference: one DartRandom(taskNum) per task), 1 = counter (keyed streams)
  int64_t seed;          // counter mode
  int32_t task_num, task_count;
  // Optional explicit pixel list (counter mode only); when npixels == 0 the whole sampler window is rendered in
  // linear pixel order (linear_pixel_sampler.dart:29-40).
  int32_t npixels;
  const int32_t* pixels;  // npixels * 2 raster pixels (x, y)
  // Pixel order of the whole-window render (npixels == 0): 0 = Pixels "linear" (linear_pixel_sampler.dart:29-40),
  // 1 = "tile" (tile_pixel_sampler.dart:33-100; the reference's default: 32 x 32 tiles, shuffled by their own
  // RNG(5489)), 2 = "random" (random_pixel_sampler.dart:27-58).  Only the serial mode's images depend on it.
  int32_t pixel_order, tile_size, tile_random, pad_po;
};
struct OrcRecord {  // optional per-sample recording (all host arrays sized by the caller)
  int64_t capacity;       // max samples
  int64_t count;          // out
  int32_t nfloats;        // out: floats per sample vector
  int32_t max_tail;       // in: tail doubles reserved per sample
  int32_t* pixel_xy;      // [capacity*2]
  float* sample_vec;      // [capacity*nfloats_cap]
  int32_t nfloats_cap;    // in
  double* tail;           // [capacity*max_tail]
  int32_t* tail_count;    // [capacity]
  float* Ls;              // [capacity*3] radiance after the NaN/neg/inf guards
};
struct OrcCounters {
  uint64_t closest_rays, any_rays, closest_nodes, any_nodes, closest_tris, any_tris, light_tris, camera_samples;
  uint64_t max_stack;
};

const char* orc_version() { return "dartray-oracle 1 (CPU restatement; parity unpinned)"; }

void* orc_scene_create(const OrcSceneDesc* d) {
  Scene* sc = new Scene();
  sc->maxPrimsInNode = std::min(255, d->max_prims_in_node > 0 ? d->max_prims_in_node : 4);
  // DartRay.shape: one GeometricPrimitive per mesh, one DiffuseAreaLight per emissive shape (dartray.dart:380-401).
  // Primitive.fullyRefine / ShapeSet pop a LIFO stack => triangle order within a mesh is reversed
  // (primitive.dart:71-84; shape_set.dart:25-35).
  bool envFailed = false;
  auto addEnv = [&]() {
    if (!d->has_env || sc->hasEnv || envFailed) return;
    sc->env.L = rgb(d->env_L[0], d->env_L[1], d->env_L[2]);
    memcpy(sc->env.l2w, d->env_l2w, sizeof(sc->env.l2w));
    memcpy(sc->env.w2l, d->env_w2l, sizeof(sc->env.w2l));
    sc->env.nSamples = std::max(1, d->env_nsamples);
    if (!sc->env.init(d->env_texels, d->env_w, d->env_h)) {
      envFailed = true;
      return;
    }
    sc->hasEnv = true;
    Light L;
    L.kind = 1;
    L.Lemit = sc->env.L;
    L.nSamples = sc->env.nSamples;
    L.area = 0.0;
    sc->lights.push_back(L);
  };
  auto addPoints = [&](int m) {
    for (int i = 0; i < d->npoint_lights; ++i) {
      int before = d->point_before_mesh ? d->point_before_mesh[i] : -1;
      if (before < 0 || before >= d->nmeshes) before = d->nmeshes;
      if (before != m) continue;
      Light L;
      L.kind = d->point_kind ? d->point_kind[i] : 2;
      if (L.kind == 3) {  // spot_light.dart:42-48
        memcpy(L.w2l, d->spot_w2l + 16 * (size_t)i, sizeof(L.w2l));
        L.cosTotalWidth = std::cos((M_PI / 180.0) * d->spot_angles[2 * i]);
        L.cosFalloffStart = std::cos((M_PI / 180.0) * d->spot_angles[2 * i + 1]);
      }
      L.Lemit = rgb(d->point_intensity[3 * i], d->point_intensity[3 * i + 1], d->point_intensity[3 * i + 2]);
      L.lightPos = V{(D)d->point_pos[3 * i], (D)d->point_pos[3 * i + 1], (D)d->point_pos[3 * i + 2]};
      L.nSamples = 1;
      L.area = 0.0;
      sc->lights.push_back(L);
    }
  };
  for (int m = 0; m < d->nmeshes; ++m) {
    addPoints(m);
    if (d->env_before_mesh >= 0 && m == d->env_before_mesh) addEnv();
    const OrcMesh& om = d->meshes[m];
    uint32_t base = (uint32_t)(sc->P.size() / 3);
    Mesh me;
    me.Kd = rgb(om.Kd[0], om.Kd[1], om.Kd[2]);
    me.sigma = om.sigma_d;
    me.reverse = om.reverse_orientation != 0;
    me.light = -1;
    me.matType = om.mat_type;
    me.Kr = rgb(om.Kr[0], om.Kr[1], om.Kr[2]);
    me.Kt = rgb(om.Kt[0], om.Kt[1], om.Kt[2]);
    me.ior = om.ior;
    if (om.kind != 0) {
      Quadric q;
      q.kind = om.kind;
      memcpy(q.o2w, om.o2w, sizeof(q.o2w));
      memcpy(q.w2o, om.w2o, sizeof(q.w2o));
      q.reverse = me.reverse;
      auto radians = [](D deg) { return (M_PI / 180.0) * deg; };  // common.dart:87-88
      if (om.kind == 1) {  // sphere.dart:24-32
        q.radius = om.params[0];
        D z0 = om.params[1], z1 = om.params[2];
        q.zmin = clampD(std::min(z0, z1), -q.radius, q.radius);
        q.zmax = clampD(std::max(z0, z1), -q.radius, q.radius);
        q.thetaMin = std::acos(clampD(q.zmin / q.radius, -1.0, 1.0));
        q.thetaMax = std::acos(clampD(q.zmax / q.radius, -1.0, 1.0));
        q.phiMax = radians(clampD(om.params[3], 0.0, 360.0));
      } else if (om.kind == 2) {  // disk.dart:24-28
        q.height = om.params[0];
        q.radius = om.params[1];
        q.innerRadius = om.params[2];
        q.phiMax = radians(clampD(om.params[3], 0.0, 360.0));
      } else {
        delete sc;
        return nullptr;
      }
      int qi = (int)sc->quadrics.size();
      sc->quadrics.push_back(q);
      if (om.has_light) {
        Light L;
        L.Lemit = rgb(om.L[0], om.L[1], om.L[2]);
        L.nSamples = std::max(1, om.light_nsamples);
        LightTri lt;
        lt.v[0] = lt.v[1] = lt.v[2] = 0;
        lt.reverse = me.reverse;
        lt.quadric = qi;
        L.shapes.push_back((int)sc->lightTris.size());
        sc->lightTris.push_back(lt);
        L.area = quadric_area(q);  // one shape: sumArea = area (shape_set.dart:37-44)
        L.areas.push_back(L.area);
        L.areaDistribution.init(L.areas);
        me.light = (int)sc->lights.size();
        sc->lights.push_back(L);
      }
      sc->meshes.push_back(me);
      Prim p;
      p.v[0] = p.v[1] = p.v[2] = 0;
      p.mesh = m;
      p.src_tri = 0;
      p.quadric = qi;
      sc->prims.push_back(p);
      continue;
    }
    sc->P.insert(sc->P.end(), om.P, om.P + 3 * (size_t)om.nverts);
    me.hasN = om.N != nullptr;
    me.hasS = om.S != nullptr;
    me.hasUV = om.uv != nullptr;
    memcpy(me.o2w, om.o2w, sizeof(me.o2w));
    memcpy(me.w2o, om.w2o, sizeof(me.w2o));
    {
      const size_t nv = (size_t)om.nverts;
      sc->N.resize(sc->P.size(), 0.f);
      sc->S.resize(sc->P.size(), 0.f);
      sc->UV.resize(sc->P.size() / 3 * 2, 0.f);
      if (om.N) memcpy(&sc->N[3 * (size_t)base], om.N, 3 * nv * sizeof(float));
      if (om.S) memcpy(&sc->S[3 * (size_t)base], om.S, 3 * nv * sizeof(float));
      if (om.uv) memcpy(&sc->UV[2 * (size_t)base], om.uv, 2 * nv * sizeof(float));
    }
    if (om.has_light) {
      Light L;
      L.Lemit = rgb(om.L[0], om.L[1], om.L[2]);
      L.nSamples = std::max(1, om.light_nsamples);
      L.area = 0.0;
      for (int t = om.ntris - 1; t >= 0; --t) {
        LightTri lt;
        for (int k = 0; k < 3; ++k) lt.v[k] = base + om.idx[3 * t + k];
        lt.reverse = me.reverse;
        lt.mesh = m;
        L.shapes.push_back((int)sc->lightTris.size());
        sc->lightTris.push_back(lt);
      }
      // Areas need vertex data which is complete for this mesh now.
      for (size_t i = 0; i < L.shapes.size(); ++i) {
        const LightTri& lt = sc->lightTris[L.shapes[i]];
        D a = tri_area(sc->vert(lt.v[0]), sc->vert(lt.v[1]), sc->vert(lt.v[2]));
        L.areas.push_back(a);
        L.area += a;
      }
      L.areaDistribution.init(L.areas);
      me.light = (int)sc->lights.size();
      sc->lights.push_back(L);
    }
    sc->meshes.push_back(me);
    for (int t = om.ntris - 1; t >= 0; --t) {
      Prim p;
      for (int k = 0; k < 3; ++k) p.v[k] = base + om.idx[3 * t + k];
      p.mesh = m;
      p.src_tri = t;
      sc->prims.push_back(p);
    }
  }
  addPoints(d->nmeshes);
  addEnv();
  if (envFailed) {
    delete sc;
    return nullptr;
  }
  Builder b;
  b.sc = sc;
  b.maxPrimsInNode = sc->maxPrimsInNode;
  b.build();
  return sc;
}
void orc_scene_destroy(void* h) { delete (Scene*)h; }
void orc_scene_info(void* h, int64_t out[6]) {
  Scene* sc = (Scene*)h;
  out[0] = (int64_t)sc->nodes.size();
  out[1] = (int64_t)sc->prims.size();
  out[2] = sc->bvhDepth;
  out[3] = (int64_t)sc->lights.size();
  out[4] = (int64_t)sc->P.size() / 3;
  out[5] = (int64_t)sc->lightTris.size();
}
// Flattened scene exactly as a Dart-side shim would marshal it.
void orc_scene_get_bvh(void* h, OrcNode* nodes, uint32_t* tri_idx, int32_t* prim_mesh, int32_t* prim_src_tri) {
  Scene* sc = (Scene*)h;
  for (size_t i = 0; i < sc->nodes.size(); ++i) {
    const LinearNode& n = sc->nodes[i];
    OrcNode& o = nodes[i];
    o.bmin[0] = (float)n.bmin.x; o.bmin[1] = (float)n.bmin.y; o.bmin[2] = (float)n.bmin.z;
    o.bmax[0] = (float)n.bmax.x; o.bmax[1] = (float)n.bmax.y; o.bmax[2] = (float)n.bmax.z;
    o.offset = n.offset;
    o.nprims = (uint16_t)n.nPrimitives;
    o.axis = (uint8_t)n.axis;
    o.pad = 0;
  }
  for (size_t i = 0; i < sc->prims.size(); ++i) {
    if (tri_idx) for (int k = 0; k < 3; ++k) tri_idx[3 * i + k] = sc->prims[i].v[k];
    if (prim_mesh) prim_mesh[i] = sc->prims[i].mesh;
    if (prim_src_tri) prim_src_tri[i] = sc->prims[i].src_tri;
  }
}
void orc_scene_get_verts(void* h, float* P) {
  Scene* sc = (Scene*)h;
  memcpy(P, sc->P.data(), sc->P.size() * sizeof(float));
}
void orc_counters(void* h, OrcCounters* out, int reset) {
  Scene* sc = (Scene*)h;
  if (out) {
    out->closest_rays = sc->ctr.closest_rays; out->any_rays = sc->ctr.any_rays;
    out->closest_nodes = sc->ctr.closest_nodes; out->any_nodes = sc->ctr.any_nodes;
    out->closest_tris = sc->ctr.closest_tris; out->any_tris = sc->ctr.any_tris;
    out->light_tris = sc->ctr.light_tris; out->camera_samples = sc->ctr.camera_samples;
    out->max_stack = sc->ctr.max_stack;
  }
  if (reset) sc->ctr = Counters();
}

static Ray to_ray(const OrcRay& r) {
  Ray q;
  q.o = V{(D)r.o[0], (D)r.o[1], (D)r.o[2]};
  q.d = V{(D)r.d[0], (D)r.d[1], (D)r.d[2]};
  q.mint = r.tmin; q.maxt = r.tmax; q.time = 0.0; q.depth = 0;
  return q;
}
// BVHAccel.intersect / intersectP on a batch of rays.
void orc_intersect(void* h, const OrcRay* rays, int64_t n, OrcHit* out, int any_hit) {
  Scene* sc = (Scene*)h;
  CounterScope counterScope(*sc);
  for (int64_t i = 0; i < n; ++i) {
    Ray r = to_ray(rays[i]);
    OrcHit& o = out[i];
    o.pad = 0;
    if (any_hit) {
      o.prim = bvh_intersectP(*sc, r) ? 0 : -1;
      o.t = o.b1 = o.b2 = 0.0;
    } else {
      Isect is;
      if (bvh_intersect(*sc, r, &is)) { o.prim = is.prim; o.t = is.t; o.b1 = is.b1; o.b2 = is.b2; }
      else { o.prim = -1; o.t = o.b1 = o.b2 = 0.0; }
    }
  }
}
// Exhaustive per-primitive testing in primitive order (the AggregateTestRenderer recipe,
// renderers/aggregate_test_renderer.dart:42-118): same tie rule as the BVH (later equal-t hit overwrites).
void orc_intersect_brute(void* h, const OrcRay* rays, int64_t n, OrcHit* out, int any_hit) {
  Scene* sc = (Scene*)h;
  CounterScope counterScope(*sc);
  for (int64_t i = 0; i < n; ++i) {
    Ray r = to_ray(rays[i]);
    OrcHit& o = out[i];
    o.prim = -1; o.pad = 0; o.t = o.b1 = o.b2 = 0.0;
    for (size_t p = 0; p < sc->prims.size(); ++p) {
      const Prim& pr = sc->prims[p];
      if (pr.quadric >= 0) {
        D t, e;
        DG dg;
        if (any_hit) {
          if (quadric_intersect(sc->quadrics[pr.quadric], r, nullptr, nullptr, nullptr)) { o.prim = 0; break; }
        } else if (quadric_intersect(sc->quadrics[pr.quadric], r, &t, &e, &dg)) {
          o.prim = (int)p; o.t = t; o.b1 = o.b2 = 0.0;
          r.maxt = t;
        }
        continue;
      }
      V a = sc->vert(pr.v[0]), b = sc->vert(pr.v[1]), c = sc->vert(pr.v[2]);
      if (any_hit) {
        if (tri_intersectP(a, b, c, r)) { o.prim = 0; break; }
      } else {
        D t, e, b1, b2;
        DG dg;
        D uvs[6];
        tri_uvs(*sc, pr.mesh, pr.v, uvs);
        if (tri_intersect(a, b, c, sc->meshes[pr.mesh].reverse, r, &t, &e, &dg, &b1, &b2, uvs)) {
          o.prim = (int)p; o.t = t; o.b1 = b1; o.b2 = b2;
          r.maxt = t;
        }
      }
    }
  }
}

// Measurement only: see OrderStudy.  enable != 0 resets and starts counting; out (may be null): rays, occluded, nodes of all rays (reference
// order), triangle tests of all rays, then for the occluded rays nodes / tris under orders 0 (reference), 1 (far first), 2 (larger area
// first), the lower bound (sum of depth + 1 of the shallowest occluding leaf), then orders 3 (longer ray interval first), 4 (leaf child
// first, else far first), 5 (smaller area first).
int orc_order_study(int enable, unsigned long long out[17]) {
  if (out) {
    out[0] = g_study.rays; out[1] = g_study.occluded; out[2] = g_study.nodesAll; out[3] = g_study.trisAll;
    for (int m = 0; m < 3; ++m) { out[4 + 2 * m] = g_study.nodesOcc[m]; out[5 + 2 * m] = g_study.trisOcc[m]; }
    out[10] = g_study.idealNodesOcc;
    for (int m = 3; m < 6; ++m) { out[11 + 2 * (m - 3)] = g_study.nodesOcc[m]; out[12 + 2 * (m - 3)] = g_study.trisOcc[m]; }
  }
  if (enable >= 0) {
    if (enable) g_study.reset();
    g_studyOn = enable ? 1 : 0;
  }
  return 0;
}

// MIPMap.texture's resampling branch alone (mipmap.dart:71-138): out must hold RoundUpPow2(w) * RoundUpPow2(h) * 3 floats
int orc_resample_pow2(const float* texels, int w, int h, float* out, int* w_out, int* h_out) {
  if (!texels || !out || w <= 0 || h <= 0) return -1;
  int sw = 0, sh = 0;
  std::vector<float> r = EnvLight::resampleToPow2(texels, w, h, &sw, &sh);
  memcpy(out, r.data(), r.size() * sizeof(float));
  *w_out = sw;
  *h_out = sh;
  return 0;
}

int orc_sample_floats(void* h, int integrator, int max_depth) {
  Scene* sc = (Scene*)h;
  IntegratorCfg cfg{integrator, max_depth};
  std::vector<int> n1D, n2D;
  sample_layout(*sc, cfg, &n1D, &n2D, true);
  int n = 5;
  for (int c : n1D) n += c;
  for (int c : n2D) n += 2 * c;
  return n;
}

// SamplerRenderer.Li for one camera sample (sampler_renderer.dart:67-98 + :165-193).
static S li_one(const Scene& sc, const IntegratorCfg& cfg, const Camera& cam, const std::vector<int>& n1D,
                const std::vector<int>& n2D, int px, int py, const float* sv, LiRng& rng, D shutterOpen,
                D shutterClose, D* imageX, D* imageY) {
  t_ctr.camera_samples++;
  *imageX = (D)px + (D)sv[0];   // montecarlo.dart:451-452
  *imageY = (D)py + (D)sv[1];
  D time = shutterOpen * (1.0 - (D)sv[4]) + shutterClose * (D)sv[4];  // Lerp common.dart:80-81
  Ray ray = generateRay(cam, *imageX, *imageY, sv[2], sv[3], time);
  D rayWeight = 1.0;
  Isect isect;
  S Li;
  SampleView view{sv, (int)0};
  int c1 = 0;
  for (int c : n1D) c1 += c;
  view.n1D = c1;
  if (bvh_intersect(sc, ray, &isect)) {
    if (cfg.kind == 1) Li = PathLi(sc, cfg, ray, isect, view, rng);
    else Li = DirectLi(sc, cfg, ray, isect, view, n1D, n2D, rng);
  } else {
    Li = S{0, 0, 0};  // Li += light.Le(ray) for every light (sampler_renderer.dart:87-92): 0 for area lights
    for (const Light& l : sc.lights) Li = sadd(Li, l.kind == 1 ? sc.env.Le(ray.d) : S{0, 0, 0});
  }
  // T * Li + Lvi with T = 1, Lvi = 0 (emission_integrator.dart:39-42), then * rayWeight.
  S Ls = smulD(sadd(smul(S{1, 1, 1}, Li), S{0, 0, 0}), rayWeight);
  if (snan(Ls)) Ls = S{0, 0, 0};                       // sampler_renderer.dart:181-193
  else if (slum(Ls) < -1e-5) Ls = S{0, 0, 0};
  else if (std::isinf(slum(Ls))) Ls = S{0, 0, 0};
  return Ls;
}

static void setup_render(const Scene& sc, const OrcRenderDesc* rd, IntegratorCfg* cfg, Camera* cam, Film* film,
                         std::vector<int>* n1D, std::vector<int>* n2D, int* nFloats, int win[4], int full[4] = nullptr) {
  cfg->kind = rd->integrator;
  cfg->maxDepth = rd->max_depth;
  memcpy(cam->r2c, rd->raster_to_camera, sizeof(cam->r2c));
  memcpy(cam->c2w, rd->camera_to_world, sizeof(cam->c2w));
  cam->lensRadius = rd->lens_radius;
  cam->focalDistance = rd->focal_distance;
  cam->type = rd->camera_type;
  cam->xres = rd->xres;
  cam->yres = rd->yres;
  film->init(rd->xres, rd->yres, rd->crop, rd->filter_xw, rd->filter_yw, rd->filter_table);
  sample_layout(sc, *cfg, n1D, n2D, true);
  *nFloats = 5;
  for (int c : *n1D) *nFloats += c;
  for (int c : *n2D) *nFloats += 2 * c;
  // DartRay._makeSampler (dartray.dart:1009-1023)
  int extent[4];
  film->getSampleExtent(extent);
  int w = extent[1] - extent[0], h = extent[3] - extent[2];
  if (full) { full[0] = extent[0]; full[1] = extent[2]; full[2] = w; full[3] = h; }
  // NB GetSubWindow computes from 0 and ignores the window origin (common.dart:69-72): only right for
  // un-cropped films, which is what every parity run uses.
  GetSubWindow(w, h, rd->task_num, std::max(1, rd->task_count), extent);
  win[0] = extent[0]; win[1] = extent[2]; win[2] = extent[1] - extent[0]; win[3] = extent[3] - extent[2];
}

// SamplerRenderer.render (sampler_renderer.dart:36-65,118-218).
// out_rgb: [height*width*3] (OutputImage.rgb); out_film: [height*width*4] (X,Y,Z,weight) or null.
// PixelSampler.setup of the three pixel samplers over the window (x, y, width, height) -> x0 y0 x1 y1 ...
static void pixel_order(int kind, int left, int top, int width, int height, int tileSize, bool randomize, std::vector<int>& out) {
  const int right = left + width - 1, bottom = top + height - 1;  // pixel_sampler.dart:36-38
  out.clear();
  out.reserve((size_t)width * height * 2);
  if (kind == 1) {  // tile_pixel_sampler.dart:38-95
    const int numXTiles = width / tileSize + ((width % tileSize == 0) ? 0 : 1);
    const int numYTiles = height / tileSize + ((height % tileSize == 0) ? 0 : 1);
    std::vector<int> tiles;
    for (int yi = 0; yi < numYTiles; ++yi)
      for (int xi = 0; xi < numXTiles; ++xi) { tiles.push_back(xi); tiles.push_back(yi); }
    const int numTiles = (int)tiles.size() / 2;
    if (randomize) {
      DartRandom rng(5489);  // RNG() (rng.dart:27-29)
      for (int ti = 1; ti < numTiles; ++ti) {  // NB starts at tile 1 (`ti = 01`, :59)
        const int lx = ti * 2, ly = lx + 1;
        const int rx = (int)(rng.randomUint() % (uint32_t)numTiles) * 2, ry = rx + 1;
        std::swap(tiles[lx], tiles[rx]);
        std::swap(tiles[ly], tiles[ry]);
      }
    }
    for (int i = 0, ti = 0; i < numTiles; ++i) {
      const int tx = tiles[ti++], ty = tiles[ti++];
      const int sx = left + tx * tileSize, sy = top + ty * tileSize;
      for (int yi = 0; yi < tileSize; ++yi) {
        const int y = sy + yi;
        if (y > bottom) break;
        for (int xi = 0; xi < tileSize; ++xi) {
          const int x = sx + xi;
          if (x > right) break;
          out.push_back(x); out.push_back(y);
        }
      }
    }
    return;
  }
  for (int y = top; y <= bottom; ++y)
    for (int x = left; x <= right; ++x) { out.push_back(x); out.push_back(y); }
  if (kind == 2) {  // random_pixel_sampler.dart:40-56
    DartRandom rng(5489);
    const int n = (int)out.size() / 2;
    for (int i = 0, r = 0; i < n; ++i, r += 2) {
      const int l = (int)(rng.randomUint() % (uint32_t)n) * 2;
      std::swap(out[r], out[l]);
      std::swap(out[r + 1], out[l + 1]);
    }
  }
}
void orc_pixel_order(int kind, int left, int top, int width, int height, int tile_size, int randomize, int32_t* out_xy) {
  std::vector<int> v;
  pixel_order(kind, left, top, width, height, tile_size, randomize != 0, v);
  for (size_t i = 0; i < v.size(); ++i) out_xy[i] = v[i];
}

int orc_render(void* h, const OrcRenderDesc* rd, float* out_rgb, float* out_film, OrcRecord* rec) {
  Scene* sc = (Scene*)h;
  CounterScope counterScope(*sc);
  IntegratorCfg cfg;
  Camera cam;
  Film film;
  std::vector<int> n1D, n2D;
  int nFloats, win[4], full[4];
  setup_render(*sc, rd, &cfg, &cam, &film, &n1D, &n2D, &nFloats, win, full);
  int spp = rd->spp;
  if ((spp & (spp - 1)) != 0) return -2;  // LowDiscrepancySampler rounds up; callers pass powers of two
  if (rec) { rec->count = 0; rec->nfloats = nFloats; if (rec->nfloats_cap < nFloats) return -3; }
  DartRandom rng((int64_t)rd->task_num);  // sampler_renderer.dart:137
  std::vector<float> buffer, samples;
  std::vector<D> tailRec;
  int64_t npix = rd->npixels > 0 ? rd->npixels : (int64_t)win[2] * win[3];
  if (rd->npixels > 0 && rd->sampler_mode == 0) return -4;
  std::vector<S> Ls(spp);
  std::vector<D> ix(spp), iy(spp);
  std::vector<int> order;  // the window's pixels in the PixelSampler's order
  if (rd->npixels == 0 && rd->pixel_order != 0)
    pixel_order(rd->pixel_order, win[0], win[1], win[2], win[3], rd->tile_size > 0 ? rd->tile_size : 32, rd->tile_random != 0, order);
  for (int64_t pi = 0; pi < npix; ++pi) {
    int px, py;  // raster pixel
    if (rd->npixels > 0) { px = rd->pixels[2 * pi]; py = rd->pixels[2 * pi + 1]; }
    else if (!order.empty()) { px = order[2 * pi]; py = order[2 * pi + 1]; }
    else { px = win[0] + (int)(pi % win[2]); py = win[1] + (int)(pi / win[2]); }
    // counter streams are keyed by the pixel's position in the FULL sampler extent, so that any task /
    // tile split traces identical samples
    uint64_t pixelIndex = (uint64_t)(py - full[1]) * (uint64_t)full[2] + (uint64_t)(px - full[0]);
    if (rd->sampler_mode == 0) LDPixelSample(spp, n1D, n2D, buffer, rng, samples, nFloats);
    else LDPixelSampleCounter(spp, n1D, n2D, (uint64_t)rd->seed, pixelIndex, samples, nFloats);
    for (int i = 0; i < spp; ++i) {
      LiRng lr;
      DartRandom crng(0);
      if (rd->sampler_mode == 0) lr.rng = &rng;
      else { crng.reseed(counter_key((uint64_t)rd->seed, pixelIndex, (uint64_t)i, 2)); lr.rng = &crng; }
      tailRec.clear();
      if (rec) lr.record = &tailRec;
      const float* sv = &samples[(size_t)i * nFloats];
      Ls[i] = li_one(*sc, cfg, cam, n1D, n2D, px, py, sv, lr, rd->shutter_open, rd->shutter_close, &ix[i], &iy[i]);
      if (rec && rec->count < rec->capacity) {
        int64_t k = rec->count++;
        if (rec->pixel_xy) { rec->pixel_xy[2 * k] = px; rec->pixel_xy[2 * k + 1] = py; }
        if (rec->sample_vec) memcpy(rec->sample_vec + k * rec->nfloats_cap, sv, sizeof(float) * nFloats);
        int nt = std::min((int)tailRec.size(), rec->max_tail);
        if (rec->tail) for (int t = 0; t < nt; ++t) rec->tail[k * rec->max_tail + t] = tailRec[t];
        if (rec->tail_count) rec->tail_count[k] = (int)tailRec.size();
        if (rec->Ls) { rec->Ls[3 * k] = (float)Ls[i].r; rec->Ls[3 * k + 1] = (float)Ls[i].g; rec->Ls[3 * k + 2] = (float)Ls[i].b; }
      }
    }
    for (int i = 0; i < spp; ++i) film.addSample(ix[i], iy[i], Ls[i]);  // sampler_renderer.dart:199-203
  }
  if (out_film) {
    for (size_t p = 0; p < film.weightSum.size(); ++p) {
      out_film[4 * p] = film.Lxyz[3 * p]; out_film[4 * p + 1] = film.Lxyz[3 * p + 1];
      out_film[4 * p + 2] = film.Lxyz[3 * p + 2]; out_film[4 * p + 3] = film.weightSum[p];
    }
  }
  if (out_rgb) {
    for (size_t p = 0; p < film.weightSum.size(); ++p)
      write_pixel(film.Lxyz[3 * p], film.Lxyz[3 * p + 1], film.Lxyz[3 * p + 2], film.weightSum[p], out_rgb + 3 * p);
  }
  return 0;
}

// Replay: radiance of explicit camera samples (sample vectors + recorded in-Li draws).  This is the
// host-buffer protocol of the GPU path.
int orc_li_samples(void* h, const OrcRenderDesc* rd, int64_t n, const int32_t* pixel_xy, const float* sample_vec,
                   int32_t nfloats_stride, const double* tail, const int32_t* tail_count, int32_t max_tail,
                   float* out_Ls) {
  Scene* sc = (Scene*)h;
  CounterScope counterScope(*sc);
  IntegratorCfg cfg;
  Camera cam;
  Film film;
  std::vector<int> n1D, n2D;
  int nFloats, win[4];
  setup_render(*sc, rd, &cfg, &cam, &film, &n1D, &n2D, &nFloats, win);
  if (nfloats_stride < nFloats) return -3;
  int rc = 0;
  for (int64_t k = 0; k < n; ++k) {
    LiRng lr;
    lr.replay = tail ? tail + k * max_tail : nullptr;
    lr.replayN = tail_count ? std::min(tail_count[k], max_tail) : (tail ? max_tail : 0);
    static const D kNoTail = 0.0;
    if (!lr.replay) { lr.replay = &kNoTail; lr.replayN = 0; }
    D ix, iy;
    S Ls = li_one(*sc, cfg, cam, n1D, n2D, pixel_xy[2 * k], pixel_xy[2 * k + 1], sample_vec + k * nfloats_stride, lr,
                  rd->shutter_open, rd->shutter_close, &ix, &iy);
    if (lr.underflow && cfg.kind == 1) rc = 1;
    out_Ls[3 * k] = (float)Ls.r; out_Ls[3 * k + 1] = (float)Ls.g; out_Ls[3 * k + 2] = (float)Ls.b;
  }
  return rc;
}

// Film-only helpers: addSample in order, then writeImage (image_film.dart:99-185,268-299).
int orc_film_accumulate(const OrcRenderDesc* rd, int64_t n, const double* imageXY, const float* Ls, float* out_film,
                        float* out_rgb) {
  Film film;
  film.init(rd->xres, rd->yres, rd->crop, rd->filter_xw, rd->filter_yw, rd->filter_table);
  for (int64_t k = 0; k < n; ++k) film.addSample(imageXY[2 * k], imageXY[2 * k + 1], S{Ls[3 * k], Ls[3 * k + 1], Ls[3 * k + 2]});
  for (size_t p = 0; p < film.weightSum.size(); ++p) {
    if (out_film) {
      out_film[4 * p] = film.Lxyz[3 * p]; out_film[4 * p + 1] = film.Lxyz[3 * p + 1];
      out_film[4 * p + 2] = film.Lxyz[3 * p + 2]; out_film[4 * p + 3] = film.weightSum[p];
    }
    if (out_rgb) write_pixel(film.Lxyz[3 * p], film.Lxyz[3 * p + 1], film.Lxyz[3 * p + 2], film.weightSum[p], out_rgb + 3 * p);
  }
  return 0;
}
// writeImage on an accumulated (X,Y,Z,w) film.
void orc_film_resolve(const float* film, int64_t npix, float* out_rgb) {
  for (int64_t p = 0; p < npix; ++p) write_pixel(film[4 * p], film[4 * p + 1], film[4 * p + 2], film[4 * p + 3], out_rgb + 3 * p);
}

// ---------------------------------------------------------------------------
// Camera set-up helper (host logic; projective_camera.dart:34-53, perspective_camera.dart:46-49,
// transform.dart:301-349, matrix4x4.dart:211-343).  Matrices are f32 stores of f64 expressions.
// ---------------------------------------------------------------------------
namespace {
struct M4 { float m[16]; };
static M4 m4identity() { M4 r; memset(r.m, 0, sizeof(r.m)); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
static M4 m4mul(const M4& a, const M4& b) {  // matrix4x4.dart:193-206
  M4 r;
  for (int i = 0, k = 0; i < 4; ++i, k += 4)
    for (int j = 0; j < 4; ++j)
      r.m[k + j] = (float)((D)a.m[k] * b.m[j] + (D)a.m[k + 1] * b.m[4 + j] + (D)a.m[k + 2] * b.m[8 + j] + (D)a.m[k + 3] * b.m[12 + j]);
  return r;
}
static M4 m4inverse(const M4& in) {  // matrix4x4.dart:232-343 (cofactor expansion)
  const float* d = in.m;
  D n11 = d[0], n12 = d[4], n13 = d[8], n14 = d[12];
  D n21 = d[1], n22 = d[5], n23 = d[9], n24 = d[13];
  D n31 = d[2], n32 = d[6], n33 = d[10], n34 = d[14];
  D n41 = d[3], n42 = d[7], n43 = d[11], n44 = d[15];
  D det = (n14 * n23 * n32 * n41) - (n13 * n24 * n32 * n41) - (n14 * n22 * n33 * n41) + (n12 * n24 * n33 * n41) +
          (n13 * n22 * n34 * n41) - (n12 * n23 * n34 * n41) - (n14 * n23 * n31 * n42) + (n13 * n24 * n31 * n42) +
          (n14 * n21 * n33 * n42) - (n11 * n24 * n33 * n42) - (n13 * n21 * n34 * n42) + (n11 * n23 * n34 * n42) +
          (n14 * n22 * n31 * n43) - (n12 * n24 * n31 * n43) - (n14 * n21 * n32 * n43) + (n11 * n24 * n32 * n43) +
          (n12 * n21 * n34 * n43) - (n11 * n22 * n34 * n43) - (n13 * n22 * n31 * n44) + (n12 * n23 * n31 * n44) +
          (n13 * n21 * n32 * n44) - (n11 * n23 * n32 * n44) - (n12 * n21 * n33 * n44) + (n11 * n22 * n33 * n44);
  M4 r = in;
  if (det == 0.0) return r;
  D invDet = 1.0 / det;
  r.m[0] = (float)((n23 * n34 * n42 - n24 * n33 * n42 + n24 * n32 * n43 - n22 * n34 * n43 - n23 * n32 * n44 + n22 * n33 * n44) * invDet);
  r.m[4] = (float)((n14 * n33 * n42 - n13 * n34 * n42 - n14 * n32 * n43 + n12 * n34 * n43 + n13 * n32 * n44 - n12 * n33 * n44) * invDet);
  r.m[8] = (float)((n13 * n24 * n42 - n14 * n23 * n42 + n14 * n22 * n43 - n12 * n24 * n43 - n13 * n22 * n44 + n12 * n23 * n44) * invDet);
  r.m[12] = (float)((n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34) * invDet);
  r.m[1] = (float)((n24 * n33 * n41 - n23 * n34 * n41 - n24 * n31 * n43 + n21 * n34 * n43 + n23 * n31 * n44 - n21 * n33 * n44) * invDet);
  r.m[5] = (float)((n13 * n34 * n41 - n14 * n33 * n41 + n14 * n31 * n43 - n11 * n34 * n43 - n13 * n31 * n44 + n11 * n33 * n44) * invDet);
  r.m[9] = (float)((n14 * n23 * n41 - n13 * n24 * n41 - n14 * n21 * n43 + n11 * n24 * n43 + n13 * n21 * n44 - n11 * n23 * n44) * invDet);
  r.m[13] = (float)((n13 * n24 * n31 - n14 * n23 * n31 + n14 * n21 * n33 - n11 * n24 * n33 - n13 * n21 * n34 + n11 * n23 * n34) * invDet);
  r.m[2] = (float)((n22 * n34 * n41 - n24 * n32 * n41 + n24 * n31 * n42 - n21 * n34 * n42 - n22 * n31 * n44 + n21 * n32 * n44) * invDet);
  r.m[6] = (float)((n14 * n32 * n41 - n12 * n34 * n41 - n14 * n31 * n42 + n11 * n34 * n42 + n12 * n31 * n44 - n11 * n32 * n44) * invDet);
  r.m[10] = (float)((n12 * n24 * n41 - n14 * n22 * n41 + n14 * n21 * n42 - n11 * n24 * n42 - n12 * n21 * n44 + n11 * n22 * n44) * invDet);
  r.m[14] = (float)((n14 * n22 * n31 - n12 * n24 * n31 - n14 * n21 * n32 + n11 * n24 * n32 + n12 * n21 * n34 - n11 * n22 * n34) * invDet);
  r.m[3] = (float)((n23 * n32 * n41 - n22 * n33 * n41 - n23 * n31 * n42 + n21 * n33 * n42 + n22 * n31 * n43 - n21 * n32 * n43) * invDet);
  r.m[7] = (float)((n12 * n33 * n41 - n13 * n32 * n41 + n13 * n31 * n42 - n11 * n33 * n42 - n12 * n31 * n43 + n11 * n32 * n43) * invDet);
  r.m[11] = (float)((n13 * n22 * n41 - n12 * n23 * n41 - n13 * n21 * n42 + n11 * n23 * n42 + n12 * n21 * n43 - n11 * n22 * n43) * invDet);
  r.m[15] = (float)((n12 * n23 * n31 - n13 * n22 * n31 + n13 * n21 * n32 - n11 * n23 * n32 - n12 * n21 * n33 + n11 * n22 * n33) * invDet);
  return r;
}
struct Xf { M4 m, mInv; };
static Xf xfmul(const Xf& a, const Xf& b) { return Xf{m4mul(a.m, b.m), m4mul(b.mInv, a.mInv)}; }  // transform.dart:83-86
static Xf xfinv(const Xf& a) { return Xf{a.mInv, a.m}; }
static Xf xfscale(D x, D y, D z) {  // transform.dart:226-239
  Xf r{m4identity(), m4identity()};
  r.m.m[0] = (float)x; r.m.m[5] = (float)y; r.m.m[10] = (float)z;
  r.mInv.m[0] = (float)(1.0 / x); r.mInv.m[5] = (float)(1.0 / y); r.mInv.m[10] = (float)(1.0 / z);
  return r;
}
static Xf xftranslate(D x, D y, D z) {  // transform.dart:210-224 (delta is a Vector: f32 components)
  Xf r{m4identity(), m4identity()};
  D dx = r32(x), dy = r32(y), dz = r32(z);
  r.m.m[3] = (float)dx; r.m.m[7] = (float)dy; r.m.m[11] = (float)dz;
  r.mInv.m[3] = (float)-dx; r.mInv.m[7] = (float)-dy; r.mInv.m[11] = (float)-dz;
  return r;
}
}  // namespace

// LookAt + perspective camera -> rasterToCamera, cameraToWorld.
void orc_camera_setup(const float pos[3], const float look[3], const float up[3], float fov, int xres, int yres,
                      float r2c[16], float c2w[16]) {
  // Transform.LookAt (transform.dart:301-329): returns worldToCamera = Transform(Inverse(m), m); the camera uses
  // its inverse (cameraToWorld = m) (dartray.dart camera directive).
  V p = vec(pos[0], pos[1], pos[2]), l = vec(look[0], look[1], look[2]), u = vec(up[0], up[1], up[2]);
  M4 m = m4identity();
  m.m[3] = (float)p.x; m.m[7] = (float)p.y; m.m[11] = (float)p.z; m.m[15] = 1.0f;
  V dir = vnormalize(vsub(l, p));
  V left = vnormalize(vcross(vnormalize(u), dir));
  V newUp = vcross(dir, left);
  m.m[0] = (float)left.x; m.m[4] = (float)left.y; m.m[8] = (float)left.z; m.m[12] = 0.0f;
  m.m[1] = (float)newUp.x; m.m[5] = (float)newUp.y; m.m[9] = (float)newUp.z; m.m[13] = 0.0f;
  m.m[2] = (float)dir.x; m.m[6] = (float)dir.y; m.m[10] = (float)dir.z; m.m[14] = 0.0f;
  memcpy(c2w, m.m, sizeof(m.m));
  // Transform.Perspective(fov, 1e-2, 1000) (transform.dart:338-349)
  D znear = 1.0e-2, zfar = 1000.0;
  M4 persp = m4identity();
  persp.m[10] = (float)(zfar / (zfar - znear));
  persp.m[11] = (float)(-zfar * znear / (zfar - znear));
  persp.m[14] = 1.0f;
  persp.m[15] = 0.0f;
  D invTanAng = 1.0 / std::tan(((kPi / 180.0) * (D)fov) / 2.0);
  Xf cameraToScreen = xfmul(xfscale(invTanAng, invTanAng, 1.0), Xf{persp, m4inverse(persp)});
  // screen window (perspective_camera.dart:152-168)
  D frame = (D)xres / (D)yres;
  D screen[4];
  if (frame > 1.0) { screen[0] = -frame; screen[1] = frame; screen[2] = -1.0; screen[3] = 1.0; }
  else { screen[0] = -1.0; screen[1] = 1.0; screen[2] = -1.0 / frame; screen[3] = 1.0 / frame; }
  // projective_camera.dart:39-52
  Xf screenToRaster = xfmul(xfmul(xfscale((D)xres, (D)yres, 1.0),
                                  xfscale(1.0 / (screen[1] - screen[0]), 1.0 / (screen[2] - screen[3]), 1.0)),
                            xftranslate(-screen[0], -screen[3], 0.0));
  Xf rasterToScreen = xfinv(screenToRaster);
  Xf rasterToCamera = xfmul(xfinv(cameraToScreen), rasterToScreen);
  memcpy(r2c, rasterToCamera.m.m, sizeof(float) * 16);
}

// OrthographicCamera's rasterToCamera (orthographic_camera.dart:44-50: Transform.Orthographic(0, 1),
// transform.dart:333-336; projective_camera.dart:39-52) for the default screen window.
void orc_camera_setup_ortho(int xres, int yres, float r2c[16]) {
  Xf cameraToScreen = xfmul(xfscale(1.0, 1.0, 1.0 / (1.0 - 0.0)), xftranslate(0.0, 0.0, -0.0));
  D frame = (D)xres / (D)yres;
  D screen[4];
  if (frame > 1.0) { screen[0] = -frame; screen[1] = frame; screen[2] = -1.0; screen[3] = 1.0; }
  else { screen[0] = -1.0; screen[1] = 1.0; screen[2] = -1.0 / frame; screen[3] = 1.0 / frame; }
  Xf screenToRaster = xfmul(xfmul(xfscale((D)xres, (D)yres, 1.0),
                                  xfscale(1.0 / (screen[1] - screen[0]), 1.0 / (screen[2] - screen[3]), 1.0)),
                            xftranslate(-screen[0], -screen[3], 0.0));
  Xf rasterToCamera = xfmul(xfinv(cameraToScreen), xfinv(screenToRaster));
  memcpy(r2c, rasterToCamera.m.m, sizeof(float) * 16);
}
// one camera ray (KATs): out = o[3], d[3]
void orc_generate_ray(const OrcRenderDesc* rd, double imageX, double imageY, double lensU, double lensV, double out[6]) {
  Camera cam;
  memcpy(cam.r2c, rd->raster_to_camera, sizeof(cam.r2c));
  memcpy(cam.c2w, rd->camera_to_world, sizeof(cam.c2w));
  cam.lensRadius = rd->lens_radius;
  cam.focalDistance = rd->focal_distance;
  cam.type = rd->camera_type;
  cam.xres = rd->xres;
  cam.yres = rd->yres;
  Ray r = generateRay(cam, imageX, imageY, lensU, lensV, 0.0);
  out[0] = r.o.x; out[1] = r.o.y; out[2] = r.o.z; out[3] = r.d.x; out[4] = r.d.y; out[5] = r.d.z;
}

// ---------------------------------------------------------------------------
// KAT entry points
// ---------------------------------------------------------------------------
double orc_van_der_corput(uint32_t n, uint32_t scramble) { return VanDerCorput(n, scramble); }
double orc_sobol2(uint32_t n, uint32_t scramble) { return Sobol2(n, scramble); }
void orc_concentric_sample_disk(double u1, double u2, double* dx, double* dy) { ConcentricSampleDisk(u1, u2, dx, dy); }
void orc_cosine_sample_hemisphere(double u1, double u2, double out[3]) {
  V v = CosineSampleHemisphere(u1, u2);
  out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
double orc_power_heuristic(int nf, double f, int ng, double g) { return PowerHeuristic(nf, f, ng, g); }
// Pixel filters (lib/filters/*.dart; Filter base: core/filter.dart:26-39).  kind: 0 box (box_filter.dart:33-46),
// 1 gaussian (gaussian_filter.dart:24-37; p0 = alpha), 2 mitchell (mitchell_filter.dart:24-43; p0 = B, p1 = C),
// 3 triangle (triangle_filter.dart:24-30), 4 Lanczos sinc (lanczos_sinc_filter.dart:24-45; p0 = tau).
// ImageFilm tabulates evaluate() at 16 x 16 points (image_film.dart:74-82); the table is what crosses the ABI.
double orc_filter_evaluate(int kind, double xw, double yw, double p0, double p1, double x, double y) {
  const double invX = 1.0 / xw, invY = 1.0 / yw;
  switch (kind) {
    case 0: return 1.0;
    case 1: {
      const double alpha = p0, expX = std::exp(-alpha * xw * xw), expY = std::exp(-alpha * yw * yw);
      auto g = [&](double d, double expv) { return std::max(0.0, std::exp(-alpha * d * d) - expv); };
      return g(x, expX) * g(y, expY);
    }
    case 2: {
      const double b = p0, c = p1;
      auto m = [&](double v) {
        v = std::fabs(2.0 * v);
        if (v > 1.0)
          return ((-b - 6 * c) * v * v * v + (6 * b + 30 * c) * v * v + (-12 * b - 48 * c) * v + (8 * b + 24 * c)) * (1.0 / 6.0);
        return ((12 - 9 * b - 6 * c) * v * v * v + (-18 + 12 * b + 6 * c) * v * v + (6 - 2 * b)) * (1.0 / 6.0);
      };
      return m(x * invX) * m(y * invY);
    }
    case 3: return std::max(0.0, xw - std::fabs(x)) * std::max(0.0, yw - std::fabs(y));
    case 4: {
      const double tau = p0;
      auto sc = [&](double v) {
        v = std::fabs(v);
        if (v < 1e-5) return 1.0;
        if (v > 1.0) return 0.0;
        v *= M_PI;
        const double sinc = std::sin(v) / v;
        const double lanczos = std::sin(v * tau) / (v * tau);
        return sinc * lanczos;
      };
      return sc(x * invX) * sc(y * invY);
    }
  }
  return 0.0;
}
void orc_filter_table(int kind, double xw, double yw, double p0, double p1, float table[256]) {  // image_film.dart:74-82
  int fi = 0;
  for (int y = 0; y < 16; ++y) {
    const double fy = (y + 0.5) * yw / 16;
    for (int x = 0; x < 16; ++x) {
      const double fx = (x + 0.5) * xw / 16;
      table[fi++] = (float)orc_filter_evaluate(kind, xw, yw, p0, p1, fx, fy);
    }
  }
}
void orc_get_sub_window(int w, int h, int num, int count, int32_t ext[4]) {
  int e[4];
  GetSubWindow(w, h, num, count, e);
  for (int i = 0; i < 4; ++i) ext[i] = e[i];
}
void orc_distribution1d(const double* f, int n, float* cdf_out, double* funcInt, const double* u, int nu, int32_t* idx_out) {
  Distribution1D d;
  d.init(std::vector<D>(f, f + n));
  for (int i = 0; i <= n; ++i) cdf_out[i] = d.cdf[i];
  *funcInt = d.funcInt;
  for (int i = 0; i < nu; ++i) idx_out[i] = d.sampleDiscrete(u[i]);
}
void orc_dart_random(int64_t seed, int n, uint32_t* uints, double* floats) {
  DartRandom a(seed), b(seed);
  for (int i = 0; i < n; ++i) { if (uints) uints[i] = a.randomUint(); if (floats) floats[i] = b.randomFloat(); }
}
int64_t orc_counter_key(uint64_t seed, uint64_t a, uint64_t b, uint64_t kind) { return counter_key(seed, a, b, kind); }
// One pixel of the LD sampler: mode 0 = serial stream from DartRandom(seed); mode 1 = counter streams.
void orc_ld_pixel_sample(int mode, int64_t seed, uint64_t pixel_index, int spp, const int32_t* n1D, int c1, const int32_t* n2D,
                         int c2, float* out) {
  std::vector<int> a(n1D, n1D + c1), b(n2D, n2D + c2);
  int nFloats = 5;
  for (int c : a) nFloats += c;
  for (int c : b) nFloats += 2 * c;
  std::vector<float> buf, res;
  if (mode == 0) { DartRandom rng(seed); LDPixelSample(spp, a, b, buf, rng, res, nFloats); }
  else LDPixelSampleCounter(spp, a, b, (uint64_t)seed, pixel_index, res, nFloats);
  memcpy(out, res.data(), res.size() * sizeof(float));
}
// InfiniteAreaLight probes: Le(dir), pdf(dir), sampleLAtPoint(u0,u1) -> wi, pdf, Ls.
int orc_env_probe(void* h, int what, const double in[3], double out[8]) {
  Scene* sc = (Scene*)h;
  if (!sc->hasEnv) return -1;
  V d = vec(in[0], in[1], in[2]);
  if (what == 0) { S l = sc->env.Le(d); out[0] = l.r; out[1] = l.g; out[2] = l.b; }
  else if (what == 1) { out[0] = sc->env.pdfW(d); }
  else { V wi{0, 0, 0}; D pdf = 0; S l = sc->env.sampleL(in[0], in[1], &wi, &pdf);
         out[0] = wi.x; out[1] = wi.y; out[2] = wi.z; out[3] = pdf; out[4] = l.r; out[5] = l.g; out[6] = l.b; }
  return 0;
}
// Single-triangle tests (KATs): returns hit flag; out = {t, b1, b2, p.xyz, nn.xyz}.
int orc_triangle_intersect(const float tri[9], const OrcRay* ray, int reverse, double out[9]) {
  Ray r = to_ray(*ray);
  V a{tri[0], tri[1], tri[2]}, b{tri[3], tri[4], tri[5]}, c{tri[6], tri[7], tri[8]};
  D t, e, b1, b2;
  DG dg;
  if (!tri_intersect(a, b, c, reverse != 0, r, &t, &e, &dg, &b1, &b2)) return 0;
  out[0] = t; out[1] = b1; out[2] = b2;
  out[3] = dg.p.x; out[4] = dg.p.y; out[5] = dg.p.z;
  out[6] = dg.nn.x; out[7] = dg.nn.y; out[8] = dg.nn.z;
  return 1;
}
int orc_triangle_intersectP(const float tri[9], const OrcRay* ray) {
  Ray r = to_ray(*ray);
  V a{tri[0], tri[1], tri[2]}, b{tri[3], tri[4], tri[5]}, c{tri[6], tri[7], tri[8]};
  return tri_intersectP(a, b, c, r) ? 1 : 0;
}
// Slab test of one box (bvh_accel.dart:439-472).
int orc_slab(const float bmin[3], const float bmax[3], const OrcRay* ray) {
  Ray r = to_ray(*ray);
  LinearNode n;
  n.bmin = V{bmin[0], bmin[1], bmin[2]};
  n.bmax = V{bmax[0], bmax[1], bmax[2]};
  V invDir = vec(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
  int neg[3] = {invDir.x < 0 ? 1 : 0, invDir.y < 0 ? 1 : 0, invDir.z < 0 ? 1 : 0};
  return slab(n, r, invDir, neg) ? 1 : 0;
}

}  // extern "C"
