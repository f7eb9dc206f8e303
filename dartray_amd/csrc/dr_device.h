// dr_device.h -- device-side numerics and geometry/shading functions.
//
// Numerics contract of the reference: every arithmetic
// expression is an f64 expression; constructing a Vector/Point/Normal/Spectrum
// or storing into a Float32List rounds to f32.  The device code keeps that
// contract literally: values that live in a Dart Float32List are held as
// `float`, every expression is evaluated in `double`, and this translation unit
// is compiled with -ffp-contract=off.  (A single + - * / of two f32 operands
// done in f32 equals the f64 result rounded to f32 because 53 >= 2*24+2, so
// those are written directly in f32.)
//
// Citations are relative to /root/reference/lib/.
#ifndef DR_DEVICE_H
#define DR_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define DR_DEV __device__ __forceinline__

// Streaming accesses to the per-sample path state (read or written once per stage, ~5 GB per batch).  Plain accesses: non-temporal
// ones were measured on C2 (no gain for k_trace / k_shade_path; scattered nt stores defeat L2 write combining).
#define LDS_STREAM(p) (*(p))
#define STS_STREAM(p, v) (*(p) = (v))

#define DR_INV_PI 0.31830988618379067154  // core/common.dart:23
#define DR_PI 3.141592653589793
#define DR_INV_TWOPI 0.15915494309189533577
#define DR_INF __longlong_as_double(0x7ff0000000000000LL)

// ---- Vector / Point / Normal (core/vector.dart) ---------------------------
struct F3 {
  float x, y, z;
};
DR_DEV F3 f3(double x, double y, double z) { return F3{(float)x, (float)y, (float)z}; }  // vector.dart:29-34
DR_DEV F3 vadd(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }              // :57-60
DR_DEV F3 vsub(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }              // :62-65
DR_DEV F3 vmul(F3 a, double f) { return f3((double)a.x * f, (double)a.y * f, (double)a.z * f); }  // :67-68
DR_DEV F3 vdiv(F3 a, double f) { return f3((double)a.x / f, (double)a.y / f, (double)a.z / f); }  // :70-71
DR_DEV F3 vneg(F3 a) { return F3{-a.x, -a.y, -a.z}; }
DR_DEV double vdot(F3 a, F3 b) {  // :153-155
  return (double)a.x * (double)b.x + (double)a.y * (double)b.y + (double)a.z * (double)b.z;
}
DR_DEV double vlen2(F3 a) { return vdot(a, a); }
DR_DEV double vlen(F3 a) { return sqrt(vlen2(a)); }
DR_DEV F3 vcross(F3 a, F3 b) {  // :161-171
  double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
  return f3((ay * bz) - (az * by), (az * bx) - (ax * bz), (ax * by) - (ay * bx));
}
DR_DEV F3 vnormalize(F3 v) { return vdiv(v, vlen(v)); }  // :173

// ---- Spectrum == RGBColor (core/rgb_color.dart:136-176) --------------------
struct C3 {
  float r, g, b;
};
DR_DEV C3 c3(double r, double g, double b) { return C3{(float)r, (float)g, (float)b}; }
DR_DEV C3 cadd(C3 a, C3 b) { return C3{a.r + b.r, a.g + b.g, a.b + b.b}; }
DR_DEV C3 cmul(C3 a, C3 b) { return C3{a.r * b.r, a.g * b.g, a.b * b.b}; }
DR_DEV C3 cmulD(C3 a, double s) { return c3((double)a.r * s, (double)a.g * s, (double)a.b * s); }
DR_DEV C3 cdivD(C3 a, double s) { return c3((double)a.r / s, (double)a.g / s, (double)a.b / s); }
DR_DEV bool cblack(C3 a) { return !(a.r != 0.0f || a.g != 0.0f || a.b != 0.0f); }
DR_DEV double clum(C3 a) { return 0.212671 * (double)a.r + 0.715160 * (double)a.g + 0.072169 * (double)a.b; }

// ---- device scene ----------------------------------------------------------
// Triangle record, 48 B: the primitive's three world-space vertices gathered
// at upload (12 B of indices + 36 B of vertices per test in the reference,
// triangle_mesh.dart:46-60) plus its GeometricPrimitive attributes.
//   q0 = (p1.x, p1.y, p1.z, p2.x)  q1 = (p2.y, p2.z, p3.x, p3.y)
//   q2 = (p3.z, material, light(int), reverse)
struct DLight {
  float L[3];
  int32_t nsamples;
  uint32_t first_tri, ntris;
  uint32_t cdf_off;  // into lcdf: ntris+1 floats
  uint32_t kind;     // DR_LIGHT_DIFFUSE_AREA / DR_LIGHT_INFINITE / DR_LIGHT_POINT
  double area;
  float pos[3];      // DR_LIGHT_POINT / _SPOT: lightPos (L holds the intensity); DR_LIGHT_DISTANT: lightDir
  float padp;
  float w2l[12];     // DR_LIGHT_SPOT: rows of worldToLight
  double cosTotalWidth, cosFalloffStart;
};
// InfiniteAreaLight: level-0 radiance texels + Distribution2D (montecarlo.dart:222-268) tables.
struct DEnv {
  const float* texels;    // [h][w][3]
  const float* condFunc;  // [h][w]      pConditionalV[v].func
  const float* condCdf;   // [h][w+1]    pConditionalV[v].cdf
  const float* condInt;   // [h]         pConditionalV[v].funcInt (an f32 value)
  const float* margFunc;  // [h]
  const float* margCdf;   // [h+1]
  // Guide table in front of the conditional CDFs' upper_bound (common.dart:304-333): condGuide[v][k] = upper_bound of
  // row v for u = k / guideN, k = 0 .. guideN (guideN a power of two, so k / guideN and floor(u * guideN) are exact):
  // the bound for any u of bucket k lies in [guide[k], guide[k + 1]], a handful of entries instead of w + 1 -- the
  // search returns the same index after ~2 dependent fetches instead of log2(w + 1) = 11.
  const uint16_t* condGuide;  // [h][guideN + 1]
  int32_t guideN;
  float margInt;
  int32_t w, h;
  float L[3];
  float l2w[9], w2l[9];   // rows of the 3x3 part of lightToWorld / worldToLight
};
// One triangle of a light's ShapeSet.  nn / ns / area are functions of the vertices only, so they
// are evaluated once at upload with the reference's arithmetic (host code in dr_api.hip) instead of
// at every path vertex: nn = DifferentialGeometry.nn of a hit on this triangle
// (differential_geometry.dart:84-99), ns = the normal Triangle.sample returns (triangle.dart:376-381),
// area = Triangle.area() (triangle.dart:265-269).
struct DLightTri {  // 72 B
  float p[9];
  uint32_t reverse;
  double area;
  float nn[3];
  float ns[3];
};
// Sphere / Disk (shapes/sphere.dart:23-38, shapes/disk.dart:23-29): affine objectToWorld (rows of the 3x4
// part; the host rejects projective transforms, so transformPoint's w is exactly 1) and the
// constructor-derived doubles.
struct DQuadric {  // 176 B
  float o2w[12], w2o[12];
  double radius, zmin, zmax, thetaMin, thetaMax, phiMax, height, innerRadius;
  int32_t kind, reverse;
};
// Sibling-pair layout used by the v3 traversal: pairs[k] holds the two child records (each a DrBvhNode
// whose `offset` is the child's own pair index if it is interior, the first primitive if it is a leaf)
// of the k-th interior node, 64 bytes, so ONE aligned fetch brings both children's boxes.
#define PREF_DEAD 0xffffffffu
#define PREF_LEAF 0x80000000u
struct DScene {
  const uint4* pairs;   // 4 x uint4 per interior node, or null (then the v2 kernel is used)
  float rootBox[6];     // bounds of node 0
  uint32_t rootRef;     // packed reference of node 0 (see pack_ref)
  uint32_t npairs;
  uint32_t topPairs;    // pair order "top:T" (dr_scene_create): pairs [0, topPairs) are the top T levels of the tree, breadth-first;
                        // every sub-tree below them is one contiguous, depth-first run of records.  0: another order
  // Any-hit rays only (set per launch by launch_trace / launch_intersect from the kernel id: 6 / 7): take the FAR child first.
  // BVHAccel.intersectP never modifies the ray (bvh_accel.dart:167-226): a leaf is reached iff its ancestors' slab tests pass, in
  // whatever order the children are taken, so the boolean is the reference's; only the work of a ray that finds an occluder depends
  // on the order (MEASUREMENTS.md round 6: C5 any-hit -21 %, C2 / C4 -3 ... -4 %; the pilot decides per scene).
  uint32_t anyFarFirst;
  const uint4* nodes;   // 2 x uint4 per node (DrBvhNode)
  const float4* tris;   // 3 x float4 per primitive
  // Shading record of a triangle, 32 B (2 x float4), for scenes of plain triangles with matte materials (the !QUAD
  // shade kernels; null otherwise): everything PathIntegrator.Li needs of a hit primitive that does not depend on the
  // ray -- dg.nn (differential_geometry.dart:84-99, reverseOrientation applied), sn = normalize(dg.dpdu) of the BSDF
  // frame (bsdf.dart:45-51), material and light ids -- evaluated once at upload by the same device functions the
  // shade kernels used per vertex (tri_dg, vnormalize), so the bits are the same.  A 32-byte aligned record never
  // straddles a cache line (the 48-byte vertex record does for a quarter of the primitives) and a path vertex no longer
  // pays two normalisations (6 f64 divisions, 2 square roots) for values that are constants of the triangle.
  //   s0 = (nn.x, nn.y, nn.z, sn.x)  s1 = (sn.y, sn.z, material, light(int))
  const float4* shtris;
  const float4* mats;   // kd.rgb, sigma
  const DLight* lights;
  const DLightTri* ltris;
  const float* lcdf;
  uint32_t nnodes, ntris, nlights, nmats;
  uint32_t nltris, ncdf;  // entries of ltris / lcdf (the shade kernels stage small tables in LDS)
  DEnv env;
  int32_t hasEnv;
  const DQuadric* quads;  // spheres / disks; a primitive record with kind != 0 holds its index in q0.x
  // per-primitive shading record (7 x float4) of meshes with N / S / uv, or null: n0 n1 n2 | s0 s1 s2 | uv0 uv1 uv2 |
  // (flags, xform index); and the mesh transforms (2 x 12 floats: rows of objectToWorld, rows of its inverse)
  const float4* srec;
  const float* xforms;
  uint32_t nquads;
  uint32_t hasSpec;  // some material is not a plain Lambertian matte (general shading kernels)
  // traversal kernel per ray kind (0 closest, 1 any hit): 0 = the default (v2), else 2 / 3 as measured on this
  // scene by the pilot of its first big render (dr_render_device)
  uint32_t traceKernel[2];
};

// Primitive record flags (q2.w): bit 0 = Shape.reverseOrientation, bits 8.. = 0 triangle / DR_QUADRIC_*.
#define PRIM_KIND(w) ((w) >> 8)
struct Tri {
  F3 p1, p2, p3;
  uint32_t mat;
  int32_t light;
  uint32_t reverse;
  uint32_t kind, quad;  // kind != 0: quadric `quad` (p1..p3 are meaningless)
};
DR_DEV Tri load_tri(const DScene& sc, uint32_t prim) {
  const float4* tp = sc.tris + 3 * (size_t)prim;
  float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
  Tri t;
  t.p1 = F3{q0.x, q0.y, q0.z};
  t.p2 = F3{q0.w, q1.x, q1.y};
  t.p3 = F3{q1.z, q1.w, q2.x};
  t.mat = __float_as_uint(q2.y);
  t.light = (int32_t)__float_as_uint(q2.z);
  const uint32_t w = __float_as_uint(q2.w);
  t.reverse = w & 1u;
  t.kind = PRIM_KIND(w);
  t.quad = __float_as_uint(q0.x);
  return t;
}

struct ShTri {
  F3 nn, sn;
  uint32_t mat;
  int32_t light;
};
DR_DEV ShTri load_shtri(const DScene& sc, uint32_t prim) {
  const float4* tp = sc.shtris + 2 * (size_t)prim;
  const float4 s0 = tp[0], s1 = tp[1];
  ShTri t;
  t.nn = F3{s0.x, s0.y, s0.z};
  t.sn = F3{s0.w, s1.x, s1.y};
  t.mat = __float_as_uint(s1.z);
  t.light = (int32_t)__float_as_uint(s1.w);
  return t;
}

// ---- Triangle (shapes/triangle.dart) ----------------------------------------
// Triangle.intersect's hit test (triangle.dart:52-98): f64 scalars on f32 inputs.
DR_DEV bool tri_hit(F3 p1, F3 p2, F3 p3, F3 o, F3 d, double tmin, double tmax, double* tOut, double* b1Out,
                    double* b2Out) {
  double e1x = (double)p2.x - (double)p1.x, e1y = (double)p2.y - (double)p1.y, e1z = (double)p2.z - (double)p1.z;
  double e2x = (double)p3.x - (double)p1.x, e2y = (double)p3.y - (double)p1.y, e2z = (double)p3.z - (double)p1.z;
  double dx = d.x, dy = d.y, dz = d.z;
  double s1x = (dy * e2z) - (dz * e2y);
  double s1y = (dz * e2x) - (dx * e2z);
  double s1z = (dx * e2y) - (dy * e2x);
  double divisor = (s1x * e1x) + (s1y * e1y) + (s1z * e1z);
  if (divisor == 0.0) return false;
  double invDivisor = 1.0 / divisor;
  double sx = (double)o.x - (double)p1.x, sy = (double)o.y - (double)p1.y, sz = (double)o.z - (double)p1.z;
  double b1 = (sx * s1x + sy * s1y + sz * s1z) * invDivisor;
  if (b1 < 0.0 || b1 > 1.0) return false;
  double s2x = (sy * e1z) - (sz * e1y);
  double s2y = (sz * e1x) - (sx * e1z);
  double s2z = (sx * e1y) - (sy * e1x);
  double b2 = ((dx * s2x) + (dy * s2y) + (dz * s2z)) * invDivisor;
  if (b2 < 0.0 || b1 + b2 > 1.0) return false;
  double t = (e2x * s2x + e2y * s2y + e2z * s2z) * invDivisor;
  if (t < tmin || t > tmax) return false;
  *tOut = t;
  *b1Out = b1;
  *b2Out = b2;
  return true;
}
// The same test with the triangle's edges e1 = p2 - p1, e2 = p3 - p1 handed in as the f64 differences the test forms
// first (triangle.dart:52-57): for the few triangles of an emitter they are constants of the light table.
DR_DEV bool tri_hit_e(F3 p1, const double* e, F3 o, F3 d, double tmin, double tmax, double* tOut) {
  const double e1x = e[0], e1y = e[1], e1z = e[2], e2x = e[3], e2y = e[4], e2z = e[5];
  double dx = d.x, dy = d.y, dz = d.z;
  double s1x = (dy * e2z) - (dz * e2y);
  double s1y = (dz * e2x) - (dx * e2z);
  double s1z = (dx * e2y) - (dy * e2x);
  double divisor = (s1x * e1x) + (s1y * e1y) + (s1z * e1z);
  if (divisor == 0.0) return false;
  double invDivisor = 1.0 / divisor;
  double sx = (double)o.x - (double)p1.x, sy = (double)o.y - (double)p1.y, sz = (double)o.z - (double)p1.z;
  double b1 = (sx * s1x + sy * s1y + sz * s1z) * invDivisor;
  if (b1 < 0.0 || b1 > 1.0) return false;
  double s2x = (sy * e1z) - (sz * e1y);
  double s2y = (sz * e1x) - (sx * e1z);
  double s2z = (sx * e1y) - (sy * e1x);
  double b2 = ((dx * s2x) + (dy * s2y) + (dz * s2z)) * invDivisor;
  if (b2 < 0.0 || b1 + b2 > 1.0) return false;
  double t = (e2x * s2x + e2y * s2y + e2z * s2z) * invDivisor;
  if (t < tmin || t > tmax) return false;
  *tOut = t;
  return true;
}
// Triangle.intersectP (triangle.dart:162-194): Vector temporaries rounded to f32.
DR_DEV bool tri_hitP(F3 p1, F3 p2, F3 p3, F3 o, F3 d, double tmin, double tmax) {
  F3 e1 = vsub(p2, p1);
  F3 e2 = vsub(p3, p1);
  F3 s1 = vcross(d, e2);
  double divisor = vdot(s1, e1);
  if (divisor == 0.0) return false;
  double invDivisor = 1.0 / divisor;
  F3 s = vsub(o, p1);
  double b1 = vdot(s, s1) * invDivisor;
  if (b1 < 0.0 || b1 > 1.0) return false;
  F3 s2 = vcross(s, e1);
  double b2 = vdot(d, s2) * invDivisor;
  if (b2 < 0.0 || b1 + b2 > 1.0) return false;
  double t = vdot(e2, s2) * invDivisor;
  if (t < tmin || t > tmax) return false;
  return true;
}

// DifferentialGeometry of a triangle hit (triangle.dart:100-132,154 with the
// default UVs (0,0),(1,0),(1,1) of :255-262; differential_geometry.dart:77-102).
struct DGeo {
  F3 p, dpdu, dpdv, nn;
  double u, v;  // parametric coordinates (only written / read by the uv-aware path)
};
DR_DEV void tri_dg(F3 p1, F3 p2, F3 p3, uint32_t reverse, F3 o, F3 d, double t, DGeo* dg) {
  // du1 = -1, du2 = 0, dv1 = -1, dv2 = -1  =>  determinant = 1, invdet = 1
  const double du1 = 0.0 - 1.0, du2 = 1.0 - 1.0, dv1 = 0.0 - 1.0, dv2 = 0.0 - 1.0;
  F3 dp1 = vsub(p1, p3);
  F3 dp2 = vsub(p2, p3);
  const double invdet = 1.0 / (du1 * dv2 - dv1 * du2);
  dg->dpdu = vmul(vsub(vmul(dp1, dv2), vmul(dp2, dv1)), invdet);
  dg->dpdv = vmul(vadd(vmul(dp1, -du2), vmul(dp2, du1)), invdet);
  dg->p = vadd(o, vmul(d, t));  // Ray.pointAt ray.dart:66-67
  F3 nn = vnormalize(vcross(dg->dpdu, dg->dpdv));
  if (reverse) nn = vmul(nn, -1.0);
  dg->nn = nn;
}
// The same with per-vertex uvs (Triangle.getUVs, triangle.dart:247-263) and the hit's barycentrics: dpdu / dpdv from
// the uv deltas, Vector.CoordinateSystem for a degenerate mapping (:108-127), interpolated (u, v) (:134-137).
DR_DEV void tri_dg_uv(F3 p1, F3 p2, F3 p3, const float* uv, uint32_t reverse, F3 o, F3 d, double t, double b1, double b2, DGeo* dg) {
  const double du1 = (double)uv[0] - (double)uv[4], du2 = (double)uv[2] - (double)uv[4];
  const double dv1 = (double)uv[1] - (double)uv[5], dv2 = (double)uv[3] - (double)uv[5];
  const F3 dp1 = vsub(p1, p3), dp2 = vsub(p2, p3);
  const double determinant = du1 * dv2 - dv1 * du2;
  if (determinant == 0.0) {
    const double e1x = (double)p2.x - (double)p1.x, e1y = (double)p2.y - (double)p1.y, e1z = (double)p2.z - (double)p1.z;
    const double e2x = (double)p3.x - (double)p1.x, e2y = (double)p3.y - (double)p1.y, e2z = (double)p3.z - (double)p1.z;
    const double e3x = (e2y * e1z) - (e2z * e1y), e3y = (e2z * e1x) - (e2x * e1z), e3z = (e2x * e1y) - (e2y * e1x);
    const double len = sqrt(e3x * e3x + e3y * e3y + e3z * e3z);
    const F3 v1 = f3(e3x / len, e3y / len, e3z / len);
    if (fabs((double)v1.x) > fabs((double)v1.y)) {  // Vector.CoordinateSystem (vector.dart:198-214)
      const double invLen = 1.0 / sqrt((double)v1.x * (double)v1.x + (double)v1.z * (double)v1.z);
      dg->dpdu = f3(-(double)v1.z * invLen, 0.0, (double)v1.x * invLen);
    } else {
      const double invLen = 1.0 / sqrt((double)v1.y * (double)v1.y + (double)v1.z * (double)v1.z);
      dg->dpdu = f3(0.0, (double)v1.z * invLen, -(double)v1.y * invLen);
    }
    dg->dpdv = vcross(v1, dg->dpdu);
  } else {
    const double invdet = 1.0 / determinant;
    dg->dpdu = vmul(vsub(vmul(dp1, dv2), vmul(dp2, dv1)), invdet);
    dg->dpdv = vmul(vadd(vmul(dp1, -du2), vmul(dp2, du1)), invdet);
  }
  const double b0 = 1.0 - b1 - b2;
  dg->u = b0 * (double)uv[0] + b1 * (double)uv[2] + b2 * (double)uv[4];
  dg->v = b0 * (double)uv[1] + b1 * (double)uv[3] + b2 * (double)uv[5];
  dg->p = vadd(o, vmul(d, t));
  F3 nn = vnormalize(vcross(dg->dpdu, dg->dpdv));
  if (reverse) nn = vmul(nn, -1.0);
  dg->nn = nn;
}
DR_DEV double tri_area(F3 p1, F3 p2, F3 p3) {  // triangle.dart:265-269
  return 0.5 * vlen(vcross(vsub(p2, p1), vsub(p3, p1)));
}

// ---- Quadrics (shapes/sphere.dart, shapes/disk.dart) ---------------------------
DR_DEV F3 q_point(const float* m, F3 p) {  // Transform.transformPoint, affine (transform.dart:110-129)
  const double x = p.x, y = p.y, z = p.z;
  return f3((double)m[0] * x + (double)m[1] * y + (double)m[2] * z + (double)m[3],
            (double)m[4] * x + (double)m[5] * y + (double)m[6] * z + (double)m[7],
            (double)m[8] * x + (double)m[9] * y + (double)m[10] * z + (double)m[11]);
}
DR_DEV F3 q_vector(const float* m, F3 p) {  // Transform.transformVector (transform.dart:131-145)
  const double x = p.x, y = p.y, z = p.z;
  return f3((double)m[0] * x + (double)m[1] * y + (double)m[2] * z, (double)m[4] * x + (double)m[5] * y + (double)m[6] * z,
            (double)m[8] * x + (double)m[9] * y + (double)m[10] * z);
}
DR_DEV double q_phi(F3 phit) {
  double phi = atan2((double)phit.y, (double)phit.x);
  if (phi < 0.0) phi += 2.0 * DR_PI;
  return phi;
}
DR_DEV bool Quadratic(double A, double B, double C, double* t0, double* t1) {  // common.dart:140-167
  const double discrim = B * B - 4.0 * A * C;
  if (discrim < 0.0) return false;
  const double rootDiscrim = sqrt(discrim);
  double q;
  if (B < 0.0) q = -0.5 * (B - rootDiscrim);
  else q = -0.5 * (B + rootDiscrim);
  *t0 = q / A;
  *t1 = C / q;
  if (*t0 > *t1) {
    const double t = *t0;
    *t0 = *t1;
    *t1 = t;
  }
  return true;
}
DR_DEV bool sphere_clipped(const DQuadric& q, F3 ph, double phi) {
  return (q.zmin > -q.radius && (double)ph.z < q.zmin) || (q.zmax < q.radius && (double)ph.z > q.zmax) || phi > q.phiMax;
}
// Sphere.intersect / intersectP hit test (sphere.dart:40-116,174-247): object-space ray (oo, od)
DR_DEV bool sphere_hit(const DQuadric& q, F3 oo, F3 od, double tmin, double tmax, double* tOut, F3* phitOut) {
  const double dx = od.x, dy = od.y, dz = od.z, ox = oo.x, oy = oo.y, oz = oo.z;
  const double A = dx * dx + dy * dy + dz * dz;
  const double B = 2 * (dx * ox + dy * oy + dz * oz);
  const double C = ox * ox + oy * oy + oz * oz - q.radius * q.radius;
  double t0, t1;
  if (!Quadratic(A, B, C, &t0, &t1)) return false;
  if (t0 > tmax || t1 < tmin) return false;
  double thit = t0;
  if (thit < tmin) {
    thit = t1;
    if (thit > tmax) return false;
  }
  F3 phit = vadd(oo, vmul(od, thit));
  if (phit.x == 0.0f && phit.y == 0.0f) phit.x = (float)(1.0e-5 * q.radius);
  double phi = q_phi(phit);
  if (sphere_clipped(q, phit, phi)) {
    if (thit == t1) return false;
    if (t1 > tmax) return false;
    thit = t1;
    phit = vadd(oo, vmul(od, thit));
    if (phit.x == 0.0f && phit.y == 0.0f) phit.x = (float)(1.0e-5 * q.radius);
    phi = q_phi(phit);
    if (sphere_clipped(q, phit, phi)) return false;
  }
  *tOut = thit;
  *phitOut = phit;
  return true;
}
// Disk.intersect / intersectP hit test (disk.dart:37-67,103-137)
DR_DEV bool disk_hit(const DQuadric& q, F3 oo, F3 od, double tmin, double tmax, double* tOut, F3* phitOut) {
  if (fabs((double)od.z) < 1.0e-7) return false;
  const double thit = (q.height - (double)oo.z) / (double)od.z;
  if (thit < tmin || thit > tmax) return false;
  const F3 phit = vadd(oo, vmul(od, thit));
  const double dist2 = (double)phit.x * (double)phit.x + (double)phit.y * (double)phit.y;
  if (dist2 > q.radius * q.radius || dist2 < q.innerRadius * q.innerRadius) return false;
  if (q_phi(phit) > q.phiMax) return false;
  *tOut = thit;
  *phitOut = phit;
  return true;
}
// world-space ray against a quadric: t and the object-space hit point
DR_DEV bool quadric_hit(const DQuadric& q, F3 o, F3 d, double tmin, double tmax, double* tOut, F3* phitOut) {
  const F3 oo = q_point(q.w2o, o), od = q_vector(q.w2o, d);  // transformRay (transform.dart:180-196)
  return q.kind == DR_QUADRIC_SPHERE ? sphere_hit(q, oo, od, tmin, tmax, tOut, phitOut)
                                     : disk_hit(q, oo, od, tmin, tmax, tOut, phitOut);
}
// DifferentialGeometry of a quadric hit at object-space point phit (sphere.dart:118-165, disk.dart:69-96,
// differential_geometry.dart:77-102).  dndu / dndv are not evaluated: nothing on the path reads them.
DR_DEV void quadric_dg(const DQuadric& q, F3 phit, DGeo* dg) {
  F3 dpdu, dpdv;
  if (q.kind == DR_QUADRIC_SPHERE) {
    double r = (double)phit.z / q.radius;
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    const double theta = acos(r);
    const double zradius = sqrt((double)phit.x * (double)phit.x + (double)phit.y * (double)phit.y);
    const double invzradius = 1.0 / zradius;
    const double cosphi = (double)phit.x * invzradius;
    const double sinphi = (double)phit.y * invzradius;
    dpdu = f3(-q.phiMax * (double)phit.y, q.phiMax * (double)phit.x, 0.0);
    dpdv = vmul(f3((double)phit.z * cosphi, (double)phit.z * sinphi, -q.radius * sin(theta)), q.thetaMax - q.thetaMin);
  } else {
    const double dist2 = (double)phit.x * (double)phit.x + (double)phit.y * (double)phit.y;
    const double oneMinusV = (sqrt(dist2) - q.innerRadius) / (q.radius - q.innerRadius);
    const double invOneMinusV = (oneMinusV > 0.0) ? (1.0 / oneMinusV) : 0.0;
    dpdu = f3(-q.phiMax * (double)phit.y, q.phiMax * (double)phit.x, 0.0);
    dpdv = f3(-(double)phit.x * invOneMinusV, -(double)phit.y * invOneMinusV, 0.0);
    dpdu = vmul(dpdu, q.phiMax * DR_INV_TWOPI);
    dpdv = vmul(dpdv, (q.radius - q.innerRadius) / q.radius);
  }
  dg->p = q_point(q.o2w, phit);
  dg->dpdu = q_vector(q.o2w, dpdu);
  dg->dpdv = q_vector(q.o2w, dpdv);
  F3 nn = vnormalize(vcross(dg->dpdu, dg->dpdv));
  if (q.reverse) nn = vmul(nn, -1.0);
  dg->nn = nn;
}
// The hit a closest-hit query reported (prim, t) turned back into its DifferentialGeometry: the same
// object-space arithmetic as the test itself (transformRay, pointAt, the pole fix-up of sphere.dart:71-73).
DR_DEV void quadric_dg_at(const DQuadric& q, F3 o, F3 d, double t, DGeo* dg) {
  const F3 oo = q_point(q.w2o, o), od = q_vector(q.w2o, d);
  F3 phit = vadd(oo, vmul(od, t));
  if (q.kind == DR_QUADRIC_SPHERE && phit.x == 0.0f && phit.y == 0.0f) phit.x = (float)(1.0e-5 * q.radius);
  quadric_dg(q, phit, dg);
}

// ---- montecarlo.dart ---------------------------------------------------------
DR_DEV void ConcentricSampleDisk(double u1, double u2, double* dx, double* dy) {  // montecarlo.dart:155-201
  double r, theta;
  double sx = 2 * u1 - 1;
  double sy = 2 * u2 - 1;
  if (sx == 0.0 && sy == 0.0) {
    *dx = 0.0;
    *dy = 0.0;
    return;
  }
  if (sx >= -sy) {
    if (sx > sy) {
      r = sx;
      if (sy > 0.0) theta = sy / r;
      else theta = 8.0 + sy / r;
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
  theta *= DR_PI / 4.0;
  *dx = r * cos(theta);
  *dy = r * sin(theta);
}
DR_DEV F3 CosineSampleHemisphere(double u1, double u2) {  // montecarlo.dart:203-209
  double dx, dy;
  ConcentricSampleDisk(u1, u2, &dx, &dy);
  double z = sqrt(fmax(0.0, 1.0 - dx * dx - dy * dy));
  return f3(dx, dy, z);
}
DR_DEV double PowerHeuristic(double fPdf, double gPdf) {  // montecarlo.dart:480-484 with nf = ng = 1
  double f = 1 * fPdf;
  double g = 1 * gPdf;
  return (f * f) / (f * f + g * g);
}

// ---- Sphere as an area light (sphere.dart:255-326) -----------------------------
DR_DEV F3 q_normal(const float* mInv, F3 n) {  // Transform.transformNormal: transpose of the inverse (transform.dart:147-161)
  const double x = n.x, y = n.y, z = n.z;
  return f3((double)mInv[0] * x + (double)mInv[4] * y + (double)mInv[8] * z, (double)mInv[1] * x + (double)mInv[5] * y + (double)mInv[9] * z,
            (double)mInv[2] * x + (double)mInv[6] * y + (double)mInv[10] * z);
}
DR_DEV F3 sphere_sample(const DQuadric& q, double u1, double u2, F3* ns) {  // sphere.dart:255-267
  const double z = 1.0 - 2.0 * u1;  // UniformSampleSphere (montecarlo.dart:113-120)
  const double r = sqrt(fmax(0.0, 1.0 - z * z));
  const double phi = 2.0 * DR_PI * u2;
  const F3 us = f3(r * cos(phi), r * sin(phi), z);
  const F3 p = vadd(F3{0.f, 0.f, 0.f}, vmul(us, q.radius));
  F3 n = vnormalize(q_normal(q.w2o, p));
  if (q.reverse) n = F3{-n.x, -n.y, -n.z};
  *ns = n;
  return q_point(q.o2w, p);
}
DR_DEV double sphere_cos_theta_max(const DQuadric& q, F3 p, F3 Pcenter) {
  const double sinThetaMax2 = q.radius * q.radius / vlen2(vsub(Pcenter, p));
  return sqrt(fmax(0.0, 1.0 - sinThetaMax2));
}
DR_DEV F3 sphere_sample2(const DQuadric& q, F3 p, double u1, double u2, F3* ns) {  // sphere.dart:269-311
  const F3 Pcenter = q_point(q.o2w, F3{0.f, 0.f, 0.f});
  const F3 wc = vnormalize(vsub(Pcenter, p));
  F3 wcX;  // Vector.CoordinateSystem (vector.dart:198-214)
  if (fabs((double)wc.x) > fabs((double)wc.y)) {
    const double invLen = 1.0 / sqrt((double)wc.x * (double)wc.x + (double)wc.z * (double)wc.z);
    wcX = f3(-(double)wc.z * invLen, 0.0, (double)wc.x * invLen);
  } else {
    const double invLen = 1.0 / sqrt((double)wc.y * (double)wc.y + (double)wc.z * (double)wc.z);
    wcX = f3(0.0, (double)wc.z * invLen, -(double)wc.y * invLen);
  }
  const F3 wcY = vcross(wc, wcX);
  if (vlen2(vsub(Pcenter, p)) - q.radius * q.radius < 1.0e-4) return sphere_sample(q, u1, u2, ns);
  const double cosThetaMax = sphere_cos_theta_max(q, p, Pcenter);
  const double costheta = cosThetaMax * (1.0 - u1) + 1.0 * u1;  // Lerp (common.dart:80-81)
  const double sintheta = sqrt(1.0 - costheta * costheta);
  const double phi = u2 * 2.0 * DR_PI;
  const F3 d = vadd(vadd(vmul(wcX, cos(phi) * sintheta), vmul(wcY, sin(phi) * sintheta)), vmul(wc, costheta));
  double thit;
  F3 phit;
  if (!quadric_hit(q, p, d, 1.0e-3, DR_INF, &thit, &phit)) thit = vdot(vsub(Pcenter, p), vnormalize(d));
  const F3 ps = vadd(p, vmul(d, thit));
  F3 n = vnormalize(vsub(ps, Pcenter));
  if (q.reverse) n = F3{-n.x, -n.y, -n.z};
  *ns = n;
  return ps;
}

// ---- per-vertex shading data (triangle.dart:247-263, 271-364) -------------------
struct ShadeRec {  // one primitive's record in DScene::srec
  float n[9], s[9], uv[6];
  uint32_t flags, xform;
};
DR_DEV ShadeRec load_srec(const DScene& sc, uint32_t prim) {
  const float4* r = sc.srec + 7 * (size_t)prim;
  const float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5], g = r[6];
  ShadeRec o;
  o.n[0] = a.x; o.n[1] = a.y; o.n[2] = a.z; o.n[3] = a.w; o.n[4] = b.x; o.n[5] = b.y; o.n[6] = b.z; o.n[7] = b.w; o.n[8] = c.x;
  o.s[0] = c.y; o.s[1] = c.z; o.s[2] = c.w; o.s[3] = d.x; o.s[4] = d.y; o.s[5] = d.z; o.s[6] = d.w; o.s[7] = e.x; o.s[8] = e.y;
  o.uv[0] = e.z; o.uv[1] = e.w; o.uv[2] = f.x; o.uv[3] = f.y; o.uv[4] = f.z; o.uv[5] = f.w;
  o.flags = __float_as_uint(g.x);
  o.xform = __float_as_uint(g.y);
  if (!(o.flags & DR_SHADING_UV)) {  // Triangle.getUVs without a uv array (triangle.dart:255-262)
    o.uv[0] = 0.f; o.uv[1] = 0.f; o.uv[2] = 1.f; o.uv[3] = 0.f; o.uv[4] = 1.f; o.uv[5] = 1.f;
  }
  return o;
}
// Triangle.getShadingGeometry (triangle.dart:271-364) for a mesh with N and / or S: barycentrics back from (u, v)
// (SolveLinearSystem2x2, common.dart:170-185), interpolated normal / tangent through objectToWorld, an orthonormal
// (ss, ts) pair, and DifferentialGeometry.set on them.  dndu / dndv are not evaluated.
DR_DEV void shading_geometry(const DScene& sc, const ShadeRec& sr, uint32_t reverse, const DGeo& dg, DGeo* out) {
  const float* uv = sr.uv;
  const double A0 = (double)uv[2] - (double)uv[0], A1 = (double)uv[4] - (double)uv[0];
  const double A2 = (double)uv[3] - (double)uv[1], A3 = (double)uv[5] - (double)uv[1];
  const double C0 = dg.u - (double)uv[0], C1 = dg.v - (double)uv[1];
  double bx, by = 0.0, bz = 0.0;
  const double det = A0 * A3 - A1 * A2;
  bool ok = !(fabs(det) < 1.0e-10);
  if (ok) {
    by = (A3 * C0 - A1 * C1) / det;
    bz = (A0 * C1 - A2 * C0) / det;
    if (by != by || bz != bz) ok = false;
  }
  if (!ok) bx = by = bz = 1.0 / 3.0;
  else bx = 1.0 - by - bz;
  const float* xf = sc.xforms + 24 * (size_t)sr.xform;
  F3 ns, ss, ts;
  if (sr.flags & DR_SHADING_N) {
    const F3 n0 = F3{sr.n[0], sr.n[1], sr.n[2]}, n1 = F3{sr.n[3], sr.n[4], sr.n[5]}, n2 = F3{sr.n[6], sr.n[7], sr.n[8]};
    ns = vnormalize(q_normal(xf + 12, vadd(vadd(vmul(n0, bx), vmul(n1, by)), vmul(n2, bz))));
  } else {
    ns = dg.nn;
  }
  if (sr.flags & DR_SHADING_S) {
    const F3 s0 = F3{sr.s[0], sr.s[1], sr.s[2]}, s1 = F3{sr.s[3], sr.s[4], sr.s[5]}, s2 = F3{sr.s[6], sr.s[7], sr.s[8]};
    ss = vnormalize(q_vector(xf, vadd(vadd(vmul(s0, bx), vmul(s1, by)), vmul(s2, bz))));
  } else {
    ss = vnormalize(dg.dpdu);
  }
  ts = vcross(ss, ns);
  if (vlen2(ts) > 0.0) {
    ts = vnormalize(ts);
    ss = vcross(ts, ns);
  } else {  // Vector.CoordinateSystem(ns, ss, ts)
    if (fabs((double)ns.x) > fabs((double)ns.y)) {
      const double invLen = 1.0 / sqrt((double)ns.x * (double)ns.x + (double)ns.z * (double)ns.z);
      ss = f3(-(double)ns.z * invLen, 0.0, (double)ns.x * invLen);
    } else {
      const double invLen = 1.0 / sqrt((double)ns.y * (double)ns.y + (double)ns.z * (double)ns.z);
      ss = f3(0.0, (double)ns.z * invLen, -(double)ns.y * invLen);
    }
    ts = vcross(ns, ss);
  }
  out->p = dg.p;
  out->dpdu = ss;
  out->dpdv = ts;
  F3 nn = vnormalize(vcross(ss, ts));
  if (reverse) nn = vmul(nn, -1.0);
  out->nn = nn;
  out->u = dg.u;
  out->v = dg.v;
}
// DifferentialGeometry of the hit (prim, t) of ray (o, d) on a triangle whose mesh has uvs: the barycentrics are
// recomputed with the accepting test's own arithmetic.
DR_DEV void tri_dg_srec(const Tri& tr, const ShadeRec& sr, F3 o, F3 d, double t, DGeo* dg) {
  double tt, b1 = 0.0, b2 = 0.0;
  (void)tri_hit(tr.p1, tr.p2, tr.p3, o, d, -DR_INF, DR_INF, &tt, &b1, &b2);
  tri_dg_uv(tr.p1, tr.p2, tr.p3, sr.uv, tr.reverse, o, d, t, b1, b2, dg);
}

// ---- light tables: global memory, or staged in LDS by the shade kernels ------------
// Every path vertex walks the sampled light's ShapeSet three times (ShapeSet.sample and two ShapeSet.pdf calls each
// intersect EVERY triangle of the set, shape_set.dart:65-89): a chain of dependent fetches of data that is the same
// for all lanes.  From global memory each of them is a ~1000-cycle round trip with nothing else in flight (measured:
// SQ_WAIT_ANY 62 % of the shade kernel's wave-cycles at < 1 vector-memory instruction in flight per wave).  When the
// tables are small -- a handful of emitters, the usual case -- the shade kernels copy them into LDS once per
// workgroup and read them with ds_read (~64 cycles).  Large emissive meshes keep the global tables.
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
struct GlobalLights {
  const DLight* lights;
  const DLightTri* ltris;
  const float* lcdf;
  const float4* mats;
  DR_DEV float4 mat(uint32_t m, int k) const { return mats[4 * (size_t)m + k]; }  // k-th float4 of material m's record
  DR_DEV DLight light(int i) const { return lights[i]; }
  DR_DEV DLightTri ltri(uint32_t i) const { return ltris[i]; }
  DR_DEV float cdf(uint32_t i) const { return lcdf[i]; }
  DR_DEV void edges(uint32_t, const DLightTri& t, double* e) const {  // e1 = p2 - p1, e2 = p3 - p1 (triangle.dart:52-57)
    for (int k = 0; k < 3; ++k) {
      e[k] = (double)t.p[3 + k] - (double)t.p[k];
      e[3 + k] = (double)t.p[6 + k] - (double)t.p[k];
    }
  }
};
template <class T>
DR_DEV T lds_read_struct(lds_cu32* base, uint32_t index) {
  constexpr int W = (int)(sizeof(T) / 4);
  static_assert(sizeof(T) % 4 == 0, "word-sized records only");
  uint32_t w[W];
  lds_cu32* p = base + (size_t)index * W;
#pragma unroll
  for (int k = 0; k < W; ++k) w[k] = p[k];
  T t;
  __builtin_memcpy(&t, w, sizeof(T));
  return t;
}
struct LdsLights {
  lds_cu32* lights;
  lds_cu32* ltris;
  lds_cu32* lcdf;
  lds_cu32* mats;          // null: the material table is too large for LDS, read it from global memory
  const float4* gmats;
  DR_DEV float4 mat(uint32_t m, int k) const {
    if (!mats) return gmats[4 * (size_t)m + k];
    lds_cu32* p = mats + 16 * (size_t)m + 4 * k;
    return make_float4(__uint_as_float(p[0]), __uint_as_float(p[1]), __uint_as_float(p[2]), __uint_as_float(p[3]));
  }
  DR_DEV DLight light(int i) const { return lds_read_struct<DLight>(lights, (uint32_t)i); }
  lds_cu32* ledges;        // 12 words per light triangle: the six f64 edge components, evaluated once per workgroup
  DR_DEV DLightTri ltri(uint32_t i) const { return lds_read_struct<DLightTri>(ltris, i); }
  DR_DEV float cdf(uint32_t i) const { return __uint_as_float(lcdf[i]); }
  DR_DEV void edges(uint32_t i, const DLightTri&, double* e) const {
    lds_cu32* p = ledges + 12 * (size_t)i;
    for (int k = 0; k < 6; ++k) e[k] = __hiloint2double((int)p[2 * k + 1], (int)p[2 * k]);
  }
};

// ---- ShapeSet / DiffuseAreaLight ---------------------------------------------
DR_DEV void ltri_verts(const DLightTri& t, F3* a, F3* b, F3* c) {
  *a = F3{t.p[0], t.p[1], t.p[2]};
  *b = F3{t.p[3], t.p[4], t.p[5]};
  *c = F3{t.p[6], t.p[7], t.p[8]};
}
// Distribution1D.sampleDiscrete (montecarlo.dart:82-92) via upper_bound (common.dart:304-333).
template <class LV>
DR_DEV int sampleDiscrete(const LV& lv, uint32_t cdfOff, int count, double u) {
  int first = 0;
  int cnt = count + 1;
  while (cnt > 0) {
    int step = cnt >> 1;
    int index = first + step;
    if (!(u < (double)lv.cdf(cdfOff + (uint32_t)index))) {
      first = index + 1;
      cnt -= step + 1;
    } else {
      cnt = step;
    }
  }
  int off = first - 1;
  return off < 0 ? 0 : off;
}
// ShapeSet.sample(ls, Ns, p) (shape_set.dart:53-80): pick by area, sample the
// triangle (triangle.dart:366-383), then intersect p->pt with EVERY shape; the
// last hitting shape in list order wins (r.maxDistance is never shrunk).
template <bool QUAD, class LV>
DR_DEV F3 shapeset_sample(const DScene& sc, const LV& lv, const DLight& L, double uPos0, double uPos1, double uComponent, F3* Ns, F3 p) {
  int sn = sampleDiscrete(lv, L.cdf_off, (int)L.ntris, uComponent) % (int)L.ntris;
  F3 a, b, c;
  const DLightTri lt = lv.ltri(L.first_tri + (uint32_t)sn);
  F3 pt;
  if (QUAD && PRIM_KIND(lt.reverse)) {
    const DQuadric& q = sc.quads[__float_as_uint(lt.p[0])];
    if (q.kind == DR_QUADRIC_SPHERE) {
      pt = sphere_sample2(q, p, uPos0, uPos1, Ns);
    } else {
      // Disk.sample (disk.dart:144-155); Shape.sample2 defaults to it (shape.dart:96-98)
      double t0, t1;
      ConcentricSampleDisk(uPos0, uPos1, &t0, &t1);
      pt = q_point(q.o2w, f3(t0 * q.radius, t1 * q.radius, q.height));
      *Ns = F3{lt.ns[0], lt.ns[1], lt.ns[2]};
    }
  } else {
    ltri_verts(lt, &a, &b, &c);
    double su1 = sqrt(uPos0);  // UniformSampleTriangle montecarlo.dart:215-220
    double b1 = 1.0 - su1;
    double b2 = uPos1 * su1;
    pt = vadd(vadd(vmul(a, b1), vmul(b, b2)), vmul(c, (1.0 - b1 - b2)));
    *Ns = F3{lt.ns[0], lt.ns[1], lt.ns[2]};
  }
  F3 rd = vsub(pt, p);
  double thit = 1.0;
  for (uint32_t i = 0; i < L.ntris; ++i) {
    const DLightTri t = lv.ltri(L.first_tri + i);
    double th;
    if (QUAD && PRIM_KIND(t.reverse)) {
      const DQuadric& q = sc.quads[__float_as_uint(t.p[0])];
      F3 phit;
      if (quadric_hit(q, p, rd, 1.0e-3, DR_INF, &th, &phit)) {
        DGeo dg;
        quadric_dg(q, phit, &dg);
        thit = th;
        *Ns = dg.nn;
      }
      continue;
    }
    double e[6];
    lv.edges(L.first_tri + i, t, e);
    if (tri_hit_e(F3{t.p[0], t.p[1], t.p[2]}, e, p, rd, 1.0e-3, DR_INF, &th)) {
      thit = th;
      *Ns = F3{t.nn[0], t.nn[1], t.nn[2]};  // dg.nn of the last hitting shape (shape_set.dart:71-77)
    }
  }
  return vadd(p, vmul(rd, thit));
}
// ShapeSet.pdf(p, wi) (shape_set.dart:82-89) with Shape.pdf2 (shape.dart:100-121).
template <bool QUAD, class LV>
DR_DEV double shapeset_pdf(const DScene& sc, const LV& lv, const DLight& L, F3 p, F3 wi) {
  double pdf = 0.0;
  for (uint32_t i = 0; i < L.ntris; ++i) {
    const DLightTri t = lv.ltri(L.first_tri + i);
    double pdf2;
    double th;
    bool h;
    F3 nn;
    if (QUAD && PRIM_KIND(t.reverse)) {
      const DQuadric& q = sc.quads[__float_as_uint(t.p[0])];
      if (q.kind == DR_QUADRIC_SPHERE) {
        // Sphere.pdf2 (sphere.dart:313-326): the subtended cone's solid angle unless p is inside the sphere
        const F3 Pcenter = q_point(q.o2w, F3{0.f, 0.f, 0.f});
        if (!(vlen2(vsub(Pcenter, p)) - q.radius * q.radius < 1.0e-4)) {
          pdf += t.area * (1.0 / (2.0 * DR_PI * (1.0 - sphere_cos_theta_max(q, p, Pcenter))));  // UniformConePdf
          continue;
        }
      }
      F3 phit;
      h = quadric_hit(q, p, wi, 1.0e-3, DR_INF, &th, &phit);
      if (h) {
        DGeo dg;
        quadric_dg(q, phit, &dg);
        nn = dg.nn;
      }
    } else {
      double e[6];
      lv.edges(L.first_tri + i, t, e);
      h = tri_hit_e(F3{t.p[0], t.p[1], t.p[2]}, e, p, wi, 1.0e-3, DR_INF, &th);
      nn = F3{t.nn[0], t.nn[1], t.nn[2]};
    }
    if (!h) {
      pdf2 = 0.0;
    } else {
      const F3 q = vadd(p, vmul(wi, th));  // ray.pointAt(thit)
      pdf2 = vlen2(vsub(q, p)) / (fabs(vdot(nn, vneg(wi))) * t.area);
      if (isinf(pdf2)) pdf2 = 0.0;
    }
    pdf += t.area * pdf2;
  }
  return pdf / L.area;
}
DR_DEV C3 light_L(const DLight& L, F3 n, F3 w) {  // diffuse_area_light.dart:44-46
  return vdot(n, w) > 0.0 ? C3{L.L[0], L.L[1], L.L[2]} : C3{0.f, 0.f, 0.f};
}

// ---- InfiniteAreaLight (lights/infinite_area_light.dart; core/mipmap.dart; montecarlo.dart:222-268) -------
DR_DEV int emod(int a, int m) {  // Dart's % is Euclidean
  int r = a % m;
  return r < 0 ? r + m : r;
}
DR_DEV C3 env_texel(const DEnv& e, int s, int t) {  // MIPMap.texel, TEXTURE_REPEAT (mipmap.dart:184-207)
  const float* p = e.texels + 3 * ((size_t)emod(t, e.h) * e.w + emod(s, e.w));
  return C3{p[0], p[1], p[2]};
}
// _radiance(u, v) with width 0: MIPMap.lookup takes the `level < 0` branch => triangle(0, s, t)
// (mipmap.dart:209-224,342-355), then * L (infinite_area_light.dart:180-182).
DR_DEV C3 env_radiance(const DEnv& e, double s, double t) {
  s = s * e.w - 0.5;
  t = t * e.h - 0.5;
  const int s0 = (int)floor(s), t0 = (int)floor(t);
  const double ds = s - s0, dt = t - t0;
  C3 v = cadd(cadd(cadd(cmulD(env_texel(e, s0, t0), ((1.0 - ds) * (1.0 - dt))),
                        cmulD(env_texel(e, s0, t0 + 1), ((1.0 - ds) * dt))),
                   cmulD(env_texel(e, s0 + 1, t0), (ds * (1.0 - dt)))),
              cmulD(env_texel(e, s0 + 1, t0 + 1), (ds * dt)));
  return cmul(v, C3{e.L[0], e.L[1], e.L[2]});
}
DR_DEV F3 xf3(const float* m, F3 p) {  // Transform.transformVector (transform.dart:130-144)
  const double x = p.x, y = p.y, z = p.z;
  return f3((double)m[0] * x + (double)m[1] * y + (double)m[2] * z, (double)m[3] * x + (double)m[4] * y + (double)m[5] * z,
            (double)m[6] * x + (double)m[7] * y + (double)m[8] * z);
}
DR_DEV double SphericalTheta(F3 v) {  // vector.dart:195-197
  double z = v.z;
  z = z < -1.0 ? -1.0 : (z > 1.0 ? 1.0 : z);
  return acos(z);
}
DR_DEV double SphericalPhi(F3 v) {  // vector.dart:199-202
  double p = atan2((double)v.y, (double)v.x);
  return (p < 0.0) ? p + 2.0 * DR_PI : p;
}
DR_DEV C3 env_Le(const DEnv& e, F3 dir) {  // infinite_area_light.dart:84-90
  F3 wh = vnormalize(xf3(e.w2l, dir));
  const double s = SphericalPhi(wh) * DR_INV_TWOPI;
  const double t = SphericalTheta(wh) * DR_INV_PI;
  return env_radiance(e, s, t);
}
// upper_bound over cdf[first .. first + cnt) (common.dart:304-333): the first index whose entry is > u
template <class CDF>
DR_DEV int cdf_upper_bound(const CDF& cdf, int first, int cnt, double u) {
  while (cnt > 0) {
    int step = cnt >> 1;
    int index = first + step;
    if (!(u < (double)cdf[index])) {
      first = index + 1;
      cnt -= step + 1;
    } else {
      cnt = step;
    }
  }
  return first;
}
// Distribution1D.sampleContinuous (montecarlo.dart:50-80) once upper_bound's result `first` is known
template <class FUNC, class CDF>
DR_DEV double dist1d_finish(const FUNC& func, const CDF& cdf, double funcInt, int count, double u, int first, double* pdf, int* off) {
  int offset = first - 1 < 0 ? 0 : first - 1;
  if (offset == count) offset = count - 1;
  if (off) *off = offset;
  const double c0 = (double)cdf[offset];
  const double dc = ((double)cdf[offset + 1] - c0);
  double du = 0.0;
  if (dc != 0.0) du = (u - c0) / dc;
  *pdf = (double)func[offset] / funcInt;
  return (offset + du) / count;
}
template <class FUNC, class CDF>
DR_DEV double dist1d_sample(const FUNC& func, const CDF& cdf, double funcInt, int count, double u, double* pdf, int* off) {
  return dist1d_finish(func, cdf, funcInt, count, u, cdf_upper_bound(cdf, 0, count + 1, u), pdf, off);
}
// The same through a guide row (DEnv::condGuide): identical index, found in the bucket's few entries.  Up to four
// candidates are fetched together (one round trip) before any bisection.
DR_DEV double dist1d_sample_guided(const float* func, const float* cdf, double funcInt, int count, double u, const uint16_t* guide, int guideN,
                                   double* pdf) {
  int k = (int)(u * (double)guideN);
  k = k < 0 ? 0 : (k > guideN - 1 ? guideN - 1 : k);
  int lo = guide[k], hi = guide[k + 1];  // upper_bound(u) is in [lo, hi] for every u of bucket k ...
  if (!(u < (double)(k + 1) / (double)guideN)) hi = count + 1;  // ... and a u outside [0, 1) searches to the end
  if (!(u >= (double)k / (double)guideN)) lo = 0;
  int first;
  if (hi - lo <= 4) {
    const int last = count;  // cdf has count + 1 entries
    const float c0 = cdf[lo <= last ? lo : last], c1 = cdf[lo + 1 <= last ? lo + 1 : last], c2 = cdf[lo + 2 <= last ? lo + 2 : last],
                c3 = cdf[lo + 3 <= last ? lo + 3 : last];
    first = hi;
    if (hi - lo > 3 && u < (double)c3) first = lo + 3;
    if (hi - lo > 2 && u < (double)c2) first = lo + 2;
    if (hi - lo > 1 && u < (double)c1) first = lo + 1;
    if (hi - lo > 0 && u < (double)c0) first = lo;
  } else {
    first = cdf_upper_bound(cdf, lo, hi - lo, u);
  }
  return dist1d_finish(func, cdf, funcInt, count, u, first, pdf, nullptr);
}
// sampleLAtPoint (infinite_area_light.dart:92-131): wi, pdf and the radiance; the shadow ray is p + t wi, t < inf.
// MF / MC: the marginal distribution's func and cdf (global pointers, or k_env's LDS copies)
template <class MF, class MC>
DR_DEV C3 env_sample_m(const DEnv& e, const MF& margFunc, const MC& margCdf, double u0, double u1, F3* wi, double* pdf) {
  double pdfs1, pdfs0;
  int voff;
  const double v = dist1d_sample(margFunc, margCdf, (double)e.margInt, e.h, u1, &pdfs1, &voff);
  const float* cf = e.condFunc + (size_t)voff * e.w;
  const float* cc = e.condCdf + (size_t)voff * (e.w + 1);
  const double ci = (double)e.condInt[voff];
  const double u = e.condGuide ? dist1d_sample_guided(cf, cc, ci, e.w, u0, e.condGuide + (size_t)voff * (e.guideN + 1), e.guideN, &pdfs0)
                               : dist1d_sample(cf, cc, ci, e.w, u0, &pdfs0, nullptr);
  const double mapPdf = pdfs0 * pdfs1;
  if (mapPdf == 0.0) {
    *pdf = 0.0;
    return C3{0.f, 0.f, 0.f};
  }
  const double theta = v * DR_PI, phi = u * 2.0 * DR_PI;
  const double costheta = cos(theta), sintheta = sin(theta);
  const double sinphi = sin(phi), cosphi = cos(phi);
  *wi = xf3(e.l2w, f3(sintheta * cosphi, sintheta * sinphi, costheta));
  if (sintheta == 0.0) *pdf = 0.0;
  else *pdf = mapPdf / (2.0 * DR_PI * DR_PI * sintheta);
  return env_radiance(e, u, v);
}
DR_DEV C3 env_sample(const DEnv& e, double u0, double u1, F3* wi, double* pdf) {
  return env_sample_m(e, e.margFunc, e.margCdf, u0, u1, wi, pdf);
}
DR_DEV double env_pdf(const DEnv& e, F3 w) {  // infinite_area_light.dart:190-205 + Distribution2D.pdf (montecarlo.dart:250-263)
  F3 wi = xf3(e.w2l, w);
  const double theta = SphericalTheta(wi), phi = SphericalPhi(wi);
  const double sintheta = sin(theta);
  if (sintheta == 0.0) return 0.0;
  const double uu = phi * DR_INV_TWOPI, vv = theta * DR_INV_PI;
  int iu = (int)(uu * e.w), iv = (int)(vv * e.h);
  iu = iu < 0 ? 0 : (iu > e.w - 1 ? e.w - 1 : iu);
  iv = iv < 0 ? 0 : (iv > e.h - 1 ? e.h - 1 : iv);
  const double ci = (double)e.condInt[iv], mi = (double)e.margInt;
  double p2;
  if (ci * mi == 0.0) p2 = 0.0;
  else p2 = ((double)e.condFunc[(size_t)iv * e.w + iu] * (double)e.margFunc[iv]) / (ci * mi);
  return p2 / (2.0 * DR_PI * DR_PI * sintheta);
}
template <bool NI>
DR_DEV C3 env_Le_x(const DEnv& e, F3 dir) { return env_Le(e, dir); }
template <bool NI>
DR_DEV C3 env_sample_x(const DEnv& e, double u0, double u1, F3* wi, double* pdf) { return env_sample(e, u0, u1, wi, pdf); }
template <bool NI>
DR_DEV double env_pdf_x(const DEnv& e, F3 w) { return env_pdf(e, w); }

// ---- BSDF with one Lambertian lobe -------------------------------------------
// (matte_material.dart:41-65; reflection/bsdf.dart:45-211; bxdf.dart:31-48,84-88;
//  lambertian.dart:31-37)
#define BSDF_REFLECTION 1
#define BSDF_TRANSMISSION 2
#define BSDF_DIFFUSE 4
#define BSDF_GLOSSY 8
#define BSDF_SPECULAR 16
#define BSDF_ALL 31
#define LAMBERT_TYPE (BSDF_REFLECTION | BSDF_DIFFUSE)

struct Bsdf {
  F3 p, nn, sn, tn;  // shading frame
  F3 ng;             // geometric normal (BSDF(dgs, dgGeom.nn)); == nn without per-vertex N / S (triangle.dart:273-276)
  C3 R;
  int nBxDFs;        // number of NON-specular lobes (0 or 1: the Lambertian)
  // specular materials (GEN kernels only): mirror = SpecularReflection(Kr, FresnelNoOp) (mirror_material.dart:38-55),
  // glass = SpecularReflection(Kr, FresnelDielectric(1, ior)) + SpecularTransmission(Kt, 1, ior) (glass_material.dart:44-69)
  int mtype;
  C3 Kr, Kt;
  double ior;
  bool on;       // the non-specular lobe is OrenNayar(R, sigma) with the coefficients below (oren_nayar.dart:24-32)
  double onA, onB;
  // plastic (plastic_material.dart:43-70): a second non-specular lobe Microfacet(Ks, FresnelDielectric(1.5, 1),
  // Blinn(bexp)) after the Lambertian one (either may be absent when its colour is black)
  bool glossy;
  C3 Ks;
  double bexp;
};
DR_DEV bool lambert_matches(int flags) { return (LAMBERT_TYPE & flags) == LAMBERT_TYPE; }
DR_DEV C3 clamp0(float4 m) { return C3{m.x < 0.f ? 0.f : m.x, m.y < 0.f ? 0.f : m.y, m.z < 0.f ? 0.f : m.z}; }
// Material record: 4 x float4 = (Kd, sigma) (Kr, type) (Kt, -) (index as the two halves of a double, -, -)
template <bool GEN, class LV>
DR_DEV Bsdf make_bsdf(const LV& lv, const DGeo& dg, uint32_t mat) {
  Bsdf b;
  b.p = dg.p;
  b.nn = dg.nn;
  b.ng = dg.nn;
  b.sn = vnormalize(dg.dpdu);  // bsdf.dart:45-51
  b.tn = vcross(b.nn, b.sn);
  // Kd.evaluate(dgs).clamp() (matte_material.dart:54)
  C3 r = clamp0(lv.mat(mat, 0));
  b.R = r;
  b.nBxDFs = cblack(r) ? 0 : 1;
  b.mtype = DR_MATERIAL_MATTE;
  b.on = false;
  b.glossy = false;
  if (GEN) {
    const float4 m1 = lv.mat(mat, 1), m3 = lv.mat(mat, 3);
    b.mtype = (int)__float_as_uint(m1.w);
    if (b.mtype == DR_MATERIAL_MATTE) {
      double sig = __hiloint2double((int)__float_as_uint(m3.w), (int)__float_as_uint(m3.z));
      sig = sig < 0.0 ? 0.0 : (sig > 90.0 ? 90.0 : sig);  // sigma.evaluate(dgs).clamp(0, 90) (matte_material.dart:55)
      if (sig != 0.0) {
        const double sigma = (DR_PI / 180.0) * sig;
        const double sigma2 = sigma * sigma;
        b.on = true;
        b.onA = 1.0 - (sigma2 / (2.0 * (sigma2 + 0.33)));
        b.onB = 0.45 * sigma2 / (sigma2 + 0.09);
      }
    } else if (b.mtype == DR_MATERIAL_PLASTIC) {
      b.Ks = clamp0(m1);
      b.glossy = !cblack(b.Ks);
      double e = 1.0 / __hiloint2double((int)__float_as_uint(m3.y), (int)__float_as_uint(m3.x));  // Blinn(1 / roughness)
      if (e > 10000.0 || e != e) e = 10000.0;
      b.bexp = e;
      b.mtype = DR_MATERIAL_MATTE;  // non-specular: shaded through bsdf_f / bsdf_pdf / bsdf_sample_f
    } else {
      const float4 m2 = lv.mat(mat, 2);
      b.nBxDFs = 0;
      b.R = C3{0.f, 0.f, 0.f};
      b.Kr = clamp0(m1);
      b.Kt = clamp0(m2);
      b.ior = __hiloint2double((int)__float_as_uint(m3.y), (int)__float_as_uint(m3.x));
    }
  }
  return b;
}
// The matte BSDF of a plain-triangle hit from its shading record (ShTri): the same values make_bsdf<false> derives
// from the DifferentialGeometry, with nn and sn = normalize(dpdu) read instead of recomputed.
template <class LV>
DR_DEV Bsdf make_bsdf_pre(const LV& lv, F3 p, F3 nn, F3 sn, uint32_t mat) {
  Bsdf b;
  b.p = p;
  b.nn = nn;
  b.ng = nn;
  b.sn = sn;
  b.tn = vcross(b.nn, b.sn);
  C3 r = clamp0(lv.mat(mat, 0));
  b.R = r;
  b.nBxDFs = cblack(r) ? 0 : 1;
  b.mtype = DR_MATERIAL_MATTE;
  b.on = false;
  b.glossy = false;
  return b;
}
DR_DEV F3 bsdf_w2l(const Bsdf& b, F3 v) { return f3(vdot(v, b.sn), vdot(v, b.tn), vdot(v, b.nn)); }  // bsdf.dart:177-179
DR_DEV F3 bsdf_l2w(const Bsdf& b, F3 v) {                                                         // bsdf.dart:181-185
  double vx = v.x, vy = v.y, vz = v.z;
  return f3((double)b.sn.x * vx + (double)b.tn.x * vy + (double)b.nn.x * vz,
            (double)b.sn.y * vx + (double)b.tn.y * vy + (double)b.nn.y * vz,
            (double)b.sn.z * vx + (double)b.tn.z * vy + (double)b.nn.z * vz);
}
DR_DEV double lambert_pdf(F3 wo, F3 wi) {  // bxdf.dart:84-88
  return ((double)wo.z * (double)wi.z > 0.0) ? fabs((double)wi.z) * DR_INV_PI : 0.0;
}
// The diffuse lobe's f(wo, wi) in the local frame: Lambertian (lambertian.dart:35-37) or OrenNayar (oren_nayar.dart:34-60)
DR_DEV double v_sin_theta(F3 v) { return sqrt(fmax(0.0, 1.0 - (double)v.z * (double)v.z)); }  // vector.dart:121-124
DR_DEV C3 diffuse_f(const Bsdf& b, F3 wo, F3 wi) {
  if (!b.on) return cmulD(b.R, DR_INV_PI);
  const double sinthetai = v_sin_theta(wi), sinthetao = v_sin_theta(wo);
  double maxcos = 0.0;
  if (sinthetai > 1e-4 && sinthetao > 1e-4) {
    // Vector.CosPhi / SinPhi (vector.dart:126-140); sintheta != 0 here
    auto cl = [](double x) { return x < -1.0 ? -1.0 : (x > 1.0 ? 1.0 : x); };
    const double cosphii = cl((double)wi.x / sinthetai), sinphii = cl((double)wi.y / sinthetai);
    const double cosphio = cl((double)wo.x / sinthetao), sinphio = cl((double)wo.y / sinthetao);
    const double dcos = cosphii * cosphio + sinphii * sinphio;
    maxcos = fmax(0.0, dcos);
  }
  double sinalpha, tanbeta;
  if (fabs((double)wi.z) > fabs((double)wo.z)) {
    sinalpha = sinthetao;
    tanbeta = sinthetai / fabs((double)wi.z);
  } else {
    sinalpha = sinthetai;
    tanbeta = sinthetao / fabs((double)wo.z);
  }
  return cmulD(b.R, DR_INV_PI * (b.onA + b.onB * maxcos * sinalpha * tanbeta));
}
// FresnelDielectric.evaluate (fresnel_dielectric.dart:30-64); the Spectrum it returns has three equal f32 channels
DR_DEV float fresnel_dielectric(double cosi, double eta_i, double eta_t) {
  cosi = cosi < -1.0 ? -1.0 : (cosi > 1.0 ? 1.0 : cosi);
  const bool entering = cosi > 0.0;
  const double ei = entering ? eta_i : eta_t, et = entering ? eta_t : eta_i;
  const double sint = ei / et * sqrt(fmax(0.0, 1.0 - cosi * cosi));
  if (sint >= 1.0) return 1.0f;
  const double cost = sqrt(fmax(0.0, 1.0 - sint * sint));
  cosi = fabs(cosi);
  const double Rparl = ((et * cosi) - (ei * cost)) / ((et * cosi) + (ei * cost));
  const double Rperp = ((ei * cosi) - (et * cost)) / ((ei * cosi) + (et * cost));
  return (float)((Rparl * Rparl + Rperp * Rperp) / 2.0);
}
#define GLOSSY_TYPE (BSDF_REFLECTION | BSDF_GLOSSY)
DR_DEV bool glossy_matches(int flags) { return (GLOSSY_TYPE & flags) == GLOSSY_TYPE; }
// Blinn.pdf (blinn.dart:62-73) from the half vector's cos(theta) and dot(wo, wh)
DR_DEV double blinn_pdf(double bexp, double costheta, double woDotWh) {
  double p = ((bexp + 1.0) * pow(costheta, bexp)) / (2.0 * DR_PI * 4.0 * woDotWh);
  if (woDotWh <= 0.0) p = 0.0;
  return p;
}
// Microfacet.f (microfacet.dart:27-56) with Blinn.d (blinn.dart:30-33) and FresnelDielectric(1.5, 1.0)
DR_DEV C3 microfacet_f(const Bsdf& b, F3 wo, F3 wi) {
  const double cosThetaO = fabs((double)wo.z), cosThetaI = fabs((double)wi.z);
  if (cosThetaI == 0.0 || cosThetaO == 0.0) return C3{0.f, 0.f, 0.f};
  F3 wh = vadd(wi, wo);
  if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return C3{0.f, 0.f, 0.f};
  wh = vnormalize(wh);
  const double cosThetaH = vdot(wi, wh);
  const float F = fresnel_dielectric(cosThetaH, 1.5, 1.0);
  const double d = (b.bexp + 2.0) * DR_INV_TWOPI * pow(fabs((double)wh.z), b.bexp);
  const double NdotWh = fabs((double)wh.z), WOdotWh = fabs(vdot(wo, wh));
  const double g = fmin(1.0, fmin((2.0 * NdotWh * cosThetaO / WOdotWh), (2.0 * NdotWh * cosThetaI / WOdotWh)));
  return cdivD(cmul(cmulD(b.Ks, d * g), C3{F, F, F}), 4.0 * cosThetaI * cosThetaO);
}
DR_DEV double microfacet_pdf(const Bsdf& b, F3 wo, F3 wi) {  // microfacet.dart:75-80
  if (!((double)wo.z * (double)wi.z > 0.0)) return 0.0;
  const F3 wh = vnormalize(vadd(wo, wi));
  return blinn_pdf(b.bexp, fabs((double)wh.z), vdot(wo, wh));
}
DR_DEV C3 bsdf_f(const Bsdf& b, F3 woW, F3 wiW, int flags) {  // bsdf.dart:187-211
  if (vdot(wiW, b.ng) * vdot(woW, b.ng) > 0) flags = flags & ~BSDF_TRANSMISSION;
  else flags = flags & ~BSDF_REFLECTION;
  C3 f = C3{0.f, 0.f, 0.f};
  if (b.nBxDFs > 0 && lambert_matches(flags)) f = cadd(f, b.on ? diffuse_f(b, bsdf_w2l(b, woW), bsdf_w2l(b, wiW)) : cmulD(b.R, DR_INV_PI));
  if (b.glossy && glossy_matches(flags)) f = cadd(f, microfacet_f(b, bsdf_w2l(b, woW), bsdf_w2l(b, wiW)));
  return f;
}
DR_DEV double bsdf_pdf(const Bsdf& b, F3 woW, F3 wiW, int flags) {  // bsdf.dart:135-156
  if (b.nBxDFs == 0 && !b.glossy) return 0.0;
  F3 wo = bsdf_w2l(b, woW);
  F3 wi = bsdf_w2l(b, wiW);
  double pdf = 0.0;
  int matchingComps = 0;
  if (b.nBxDFs > 0 && lambert_matches(flags)) {
    ++matchingComps;
    pdf += lambert_pdf(wo, wi);
  }
  if (b.glossy && glossy_matches(flags)) {
    ++matchingComps;
    pdf += microfacet_pdf(b, wo, wi);
  }
  return matchingComps > 0 ? pdf / matchingComps : 0.0;
}
DR_DEV C3 bsdf_sample_f(const Bsdf& b, F3 woW, F3* wiW, double uDir0, double uDir1, double uComponent, double* pdf, int flags) {
  // bsdf.dart:53-133 with up to two non-specular lobes: [Lambertian | OrenNayar], Microfacet
  const bool m0 = b.nBxDFs > 0 && lambert_matches(flags), m1 = b.glossy && glossy_matches(flags);
  const int matchingComps = (m0 ? 1 : 0) + (m1 ? 1 : 0);
  *pdf = 0.0;
  if (matchingComps == 0) return C3{0.f, 0.f, 0.f};
  bool pickGlossy = !m0;
  if (matchingComps > 1) {
    int which = (int)floor(uComponent * matchingComps);
    which = which < matchingComps - 1 ? which : matchingComps - 1;
    pickGlossy = which == 1;
  }
  F3 wo = bsdf_w2l(b, woW);
  F3 wi;
  if (pickGlossy) {  // Microfacet.sample_f -> Blinn.sample_f (microfacet.dart:66-73, blinn.dart:35-60)
    const double costheta = pow(uDir0, 1.0 / (b.bexp + 1.0));
    const double sintheta = sqrt(fmax(0.0, 1.0 - costheta * costheta));
    const double phi = uDir1 * 2.0 * DR_PI;
    F3 wh = f3(sintheta * cos(phi), sintheta * sin(phi), costheta);  // Vector.SphericalDirection
    if (!((double)wo.z * (double)wh.z > 0.0)) wh = vneg(wh);
    wi = vadd(vneg(wo), vmul(vmul(wh, 2.0), vdot(wo, wh)));
    *pdf = blinn_pdf(b.bexp, costheta, vdot(wo, wh));
    // (a wi in the other hemisphere makes f black but pdf is kept, microfacet.dart:68-71)
  } else {
    wi = CosineSampleHemisphere(uDir0, uDir1);  // bxdf.dart:37-48
    if (wo.z < 0.0f) wi.z = (float)((double)wi.z * -1.0);
    *pdf = lambert_pdf(wo, wi);
  }
  if (*pdf == 0.0) return C3{0.f, 0.f, 0.f};
  *wiW = bsdf_l2w(b, wi);
  if (matchingComps > 1) {  // add the other lobe's pdf, average (bsdf.dart:102-113)
    *pdf += pickGlossy ? lambert_pdf(wo, wi) : microfacet_pdf(b, wo, wi);
    *pdf /= matchingComps;
  }
  C3 f = C3{0.f, 0.f, 0.f};
  if (vdot(*wiW, b.ng) * vdot(woW, b.ng) > 0) flags = flags & ~BSDF_TRANSMISSION;
  else flags = flags & ~BSDF_REFLECTION;
  if (b.nBxDFs > 0 && lambert_matches(flags)) f = cadd(f, diffuse_f(b, wo, wi));
  if (b.glossy && glossy_matches(flags)) f = cadd(f, microfacet_f(b, wo, wi));
  return f;
}
// BSDF.sample_f(flags = BSDF_ALL) of a mirror / glass BSDF (bsdf.dart:53-133): every lobe is specular, so the
// chosen lobe's f and pdf are returned as they are and pdf is divided by the number of lobes.
DR_DEV C3 spec_sample_f(const Bsdf& b, F3 woW, F3* wiW, double uComponent, double* pdf) {
  const bool hasR = !cblack(b.Kr), hasT = b.mtype == DR_MATERIAL_GLASS && !cblack(b.Kt);
  const int matchingComps = (hasR ? 1 : 0) + (hasT ? 1 : 0);
  *pdf = 0.0;
  if (matchingComps == 0) return C3{0.f, 0.f, 0.f};
  int which = (int)floor(uComponent * matchingComps);
  which = which < matchingComps - 1 ? which : matchingComps - 1;
  const bool reflect = hasR && which == 0;
  const F3 wo = bsdf_w2l(b, woW);
  F3 wi;
  C3 f;
  if (reflect) {  // specular_reflection.dart:33-41
    wi = F3{-wo.x, -wo.y, wo.z};
    *pdf = 1.0;
    const float F = b.mtype == DR_MATERIAL_GLASS ? fresnel_dielectric((double)wo.z, 1.0, b.ior) : 1.0f;
    f = cdivD(cmul(C3{F, F, F}, b.Kr), fabs((double)wi.z));
  } else {  // specular_transmission.dart:37-71
    const bool entering = wo.z > 0.0f;
    const double ei = entering ? 1.0 : b.ior, et = entering ? b.ior : 1.0;
    const double sini2 = fmax(0.0, 1.0 - (double)wo.z * (double)wo.z);
    const double eta = ei / et;
    const double sint2 = eta * eta * sini2;
    if (sint2 >= 1.0) return C3{0.f, 0.f, 0.f};  // total internal reflection (pdf stays 0)
    double cost = sqrt(fmax(0.0, 1.0 - sint2));
    if (entering) cost = -cost;
    wi = f3(eta * -(double)wo.x, eta * -(double)wo.y, cost);
    *pdf = 1.0;
    const float F = fresnel_dielectric((double)wo.z, 1.0, b.ior);
    const float omf = (float)(1.0 - (double)F);
    f = cdivD(cmul(C3{omf, omf, omf}, b.Kt), fabs((double)wi.z));
  }
  *wiW = bsdf_l2w(b, wi);
  if (matchingComps > 1) *pdf /= matchingComps;
  return f;
}

// BSDF.sample_f(flags = BSDF_REFLECTION | BSDF_SPECULAR) or (BSDF_TRANSMISSION | BSDF_SPECULAR) as
// Integrator.SpecularReflect / SpecularTransmit call it (integrator.dart:195,241): only the one specular lobe of that
// hemisphere can match (mirror: reflection; glass: both), so there is no lobe choice and the pdf is the lobe's own.
// A BSDF without such a lobe (matte, plastic) returns black with pdf 0.
DR_DEV C3 spec_lobe_sample_f(const Bsdf& b, F3 woW, F3* wiW, double* pdf, bool reflection) {
  *pdf = 0.0;
  const bool isSpec = b.mtype == DR_MATERIAL_MIRROR || b.mtype == DR_MATERIAL_GLASS;
  const bool hasR = isSpec && !cblack(b.Kr), hasT = b.mtype == DR_MATERIAL_GLASS && !cblack(b.Kt);
  if (reflection ? !hasR : !hasT) return C3{0.f, 0.f, 0.f};
  const F3 wo = bsdf_w2l(b, woW);
  F3 wi;
  C3 f;
  if (reflection) {  // specular_reflection.dart:33-41
    wi = F3{-wo.x, -wo.y, wo.z};
    *pdf = 1.0;
    const float F = b.mtype == DR_MATERIAL_GLASS ? fresnel_dielectric((double)wo.z, 1.0, b.ior) : 1.0f;
    f = cdivD(cmul(C3{F, F, F}, b.Kr), fabs((double)wi.z));
  } else {  // specular_transmission.dart:37-71
    const bool entering = wo.z > 0.0f;
    const double ei = entering ? 1.0 : b.ior, et = entering ? b.ior : 1.0;
    const double sini2 = fmax(0.0, 1.0 - (double)wo.z * (double)wo.z);
    const double eta = ei / et;
    const double sint2 = eta * eta * sini2;
    if (sint2 >= 1.0) return C3{0.f, 0.f, 0.f};  // total internal reflection (pdf stays 0)
    double cost = sqrt(fmax(0.0, 1.0 - sint2));
    if (entering) cost = -cost;
    wi = f3(eta * -(double)wo.x, eta * -(double)wo.y, cost);
    *pdf = 1.0;
    const float F = fresnel_dielectric((double)wo.z, 1.0, b.ior);
    const float omf = (float)(1.0 - (double)F);
    f = cdivD(cmul(C3{omf, omf, omf}, b.Kt), fabs((double)wi.z));
  }
  *wiW = bsdf_l2w(b, wi);
  return f;
}

#endif
