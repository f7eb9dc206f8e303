"""A SECOND, independent restatement of the reference's per-sample path -- written from the Dart text of
/root/reference/lib (file:line cited at every function), NOT from oracle/dartray_oracle.cpp -- in plain Python.

Why it exists: the reference is Dart, no Dart SDK is available in the build image, and the reference ships no golden
vectors, so the C++ oracle cannot be pinned against the reference itself.  HIP kernels and oracle have one author; a
shared misreading of the Dart would go unnoticed.  This module is a separate reading: different language, different
structure (objects mirroring the Dart classes, not the oracle's flat C++), arithmetic in Python floats (IEEE f64, never
fused) with an explicit f32 round at every place the Dart code stores into a Float32List (Vector / Point / Normal /
RGBColor constructors, `data[i] = ...`).  tests/golden/make_restatement_fixtures.py runs it on the recorded serial
sample streams of C1 and C2-small and commits per-sample Li + films; tests compare the oracle and the GPU to them bit
for bit.

Scope (task 5): Triangle.intersect / intersectP, BVHAccel.intersect / intersectP + _intersectP,
GeometricPrimitive.intersect, ShapeSet.sample / pdf, Shape.pdf2, DiffuseAreaLight, EstimateDirect,
UniformSampleOneLight / AllLights, PathIntegrator.Li, DirectLightingIntegrator.Li, BSDF with Lambertian /
SpecularReflection / SpecularTransmission lobes and FresnelDielectric (matte, mirror, glass materials),
InfiniteAreaLight with its MIPMap (pyramid, trilinear lookup) and Distribution2D (built from the texels here),
PerspectiveCamera.generateRayDifferential, SamplerRenderer's guards, ImageFilm.addSample / writeImage.
The low-discrepancy pixel sample (LDPixelSample, the scrambled (0,2) sequences, Shuffle) and the RNG are restated too:
run with a live RNG(taskNum) the module regenerates the serial sample streams themselves.
The SAH build and the flattening of BVHAccel are restated as well (build_bvh): tests compare its node array and
primitive order with the product's dr_bvh_build byte for byte.
The sampler window (getSampleExtent, GetSubWindow, _makeSampler) and the linear / tile pixel orders are restated too.

TEST INFRASTRUCTURE ONLY.
"""
import math
import struct

INFINITY = float("inf")            # common.dart:26 (1.0e500 parses to infinity)
INV_PI = 0.31830988618379067154    # common.dart:23
BSDF_REFLECTION, BSDF_TRANSMISSION, BSDF_DIFFUSE, BSDF_GLOSSY, BSDF_SPECULAR = 1, 2, 4, 8, 16  # bsdf.dart:23-27
BSDF_ALL = 31


_pack, _unpack = struct.Struct("<f").pack, struct.Struct("<f").unpack


def f32(x):
    """A store into a Float32List: round-to-nearest-even f64 -> f32 (overflow to +-inf like the VM)."""
    try:
        return _unpack(_pack(x))[0]
    except OverflowError:
        return math.copysign(INFINITY, x)


# ---------------------------------------------------------------------------------------------------------------
# core/vector.dart, point.dart, normal.dart: every constructor stores three f32 values
# ---------------------------------------------------------------------------------------------------------------
class Vec:
    __slots__ = ("x", "y", "z")

    def __init__(self, x=0.0, y=0.0, z=0.0):                     # vector.dart:29-34
        self.x, self.y, self.z = f32(x), f32(y), f32(z)

    def __add__(self, v):                                          # vector.dart:57-60
        return Vec(self.x + v.x, self.y + v.y, self.z + v.z)

    def __sub__(self, v):                                          # vector.dart:62-65
        return Vec(self.x - v.x, self.y - v.y, self.z - v.z)

    def __mul__(self, f):                                          # vector.dart:67-68
        return Vec(self.x * f, self.y * f, self.z * f)

    def __truediv__(self, f):                                      # vector.dart:70-71
        return Vec(self.x / f, self.y / f, self.z / f)

    def __neg__(self):                                             # vector.dart:73-74
        return Vec(-self.x, -self.y, -self.z)

    def lengthSquared(self):                                       # vector.dart:80-81
        return self.x * self.x + self.y * self.y + self.z * self.z

    def length(self):                                              # vector.dart:83
        return math.sqrt(self.lengthSquared())


def Dot(a, b):                                                     # vector.dart:153-155
    return a.x * b.x + a.y * b.y + a.z * b.z


def AbsDot(a, b):                                                  # vector.dart:157-159
    return abs(a.x * b.x + a.y * b.y + a.z * b.z)


def Cross(a, b):                                                   # vector.dart:161-171
    return Vec((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x))


def Normalize(v):                                                  # vector.dart:173 (v / v.length())
    return v / v.length()


def NormalNormalize(v):                                            # normal.dart:53-55: copy, then invScale(length()): data[i] /= s
    l = v.length()
    return Vec(v.x / l, v.y / l, v.z / l)


# ---------------------------------------------------------------------------------------------------------------
# core/rgb_color.dart:136-176 (Spectrum == RGBColor): three f32 stores per operation
# ---------------------------------------------------------------------------------------------------------------
class RGB:
    __slots__ = ("r", "g", "b")

    def __init__(self, r=0.0, g=None, b=None):
        if g is None:
            g = b = r
        self.r, self.g, self.b = f32(r), f32(g), f32(b)

    def __add__(self, s):
        return RGB(self.r + s.r, self.g + s.g, self.b + s.b)

    def __sub__(self, s):
        return RGB(self.r - s.r, self.g - s.g, self.b - s.b)

    def __mul__(self, s):
        if isinstance(s, RGB):
            return RGB(self.r * s.r, self.g * s.g, self.b * s.b)
        return RGB(self.r * s, self.g * s, self.b * s)

    def clamp(self, low=0.0, high=INFINITY):                        # rgb_color.dart:189-192
        c = lambda v: low if v < low else (high if v > high else v)
        return RGB(c(self.r), c(self.g), c(self.b))

    def __truediv__(self, s):
        return RGB(self.r / s, self.g / s, self.b / s)

    def luminance(self):                                           # rgb_color.dart:165-167
        return 0.212671 * self.r + 0.715160 * self.g + 0.072169 * self.b

    def isBlack(self):                                             # rgb_color.dart:169-174
        return not (self.r != 0.0 or self.g != 0.0 or self.b != 0.0)

    def hasNaNs(self):
        return math.isnan(self.r) or math.isnan(self.g) or math.isnan(self.b)

    def tuple(self):
        return (self.r, self.g, self.b)


class Ray:                                                         # core/ray.dart, ray_differential.dart (no differentials)
    __slots__ = ("o", "d", "mint", "maxt", "depth")

    def __init__(self, o, d, mint=0.0, maxt=INFINITY, depth=0):
        self.o, self.d, self.mint, self.maxt, self.depth = o, d, mint, maxt, depth

    def pointAt(self, t):                                          # ray.dart:66-67: origin + (direction * t)
        return self.o + (self.d * t)


class DG:                                                          # core/differential_geometry.dart:77-102 (`set`)
    __slots__ = ("p", "dpdu", "dpdv", "nn")

    def set(self, p, dpdu, dpdv, reverse):
        self.p, self.dpdu, self.dpdv = p, dpdu, dpdv
        self.nn = NormalNormalize(Cross(dpdu, dpdv))
        if reverse:                                                # reverseOrientation ^ transformSwapsHandedness (identity transforms here)
            self.nn = self.nn * -1.0
        return self


# ---------------------------------------------------------------------------------------------------------------
# shapes/triangle.dart
# ---------------------------------------------------------------------------------------------------------------
class Triangle:
    def __init__(self, p1, p2, p3, reverse=False):
        self.p1, self.p2, self.p3, self.reverse = p1, p2, p3, reverse

    def intersect(self, ray):
        """triangle.dart:44-160: returns (t, rayEpsilon, dg) or None.  All-f64 scalars on the f32 vertex data."""
        p1, p2, p3 = self.p1, self.p2, self.p3
        e1x = p2.x - p1.x; e1y = p2.y - p1.y; e1z = p2.z - p1.z
        e2x = p3.x - p1.x; e2y = p3.y - p1.y; e2z = p3.z - p1.z
        d = ray.d
        s1x = (d.y * e2z) - (d.z * e2y)
        s1y = (d.z * e2x) - (d.x * e2z)
        s1z = (d.x * e2y) - (d.y * e2x)
        divisor = (s1x * e1x) + (s1y * e1y) + (s1z * e1z)
        if divisor == 0.0:
            return None
        invDivisor = 1.0 / divisor
        sx = ray.o.x - p1.x; sy = ray.o.y - p1.y; sz = ray.o.z - p1.z
        b1 = (sx * s1x + sy * s1y + sz * s1z) * invDivisor
        if b1 < 0.0 or b1 > 1.0:
            return None
        s2x = (sy * e1z) - (sz * e1y)
        s2y = (sz * e1x) - (sx * e1z)
        s2z = (sx * e1y) - (sy * e1x)
        b2 = ((d.x * s2x) + (d.y * s2y) + (d.z * s2z)) * invDivisor
        if b2 < 0.0 or b1 + b2 > 1.0:
            return None
        t = (e2x * s2x + e2y * s2y + e2z * s2z) * invDivisor
        if t < ray.mint or t > ray.maxt:
            return None
        # partial derivatives with the default uvs (0,0) (1,0) (1,1) of getUVs (:255-262)
        uvs = (0.0, 0.0, 1.0, 0.0, 1.0, 1.0)
        du1 = uvs[0] - uvs[4]; du2 = uvs[2] - uvs[4]; dv1 = uvs[1] - uvs[5]; dv2 = uvs[3] - uvs[5]
        dp1 = p1 - p3
        dp2 = p2 - p3
        determinant = du1 * dv2 - dv1 * du2
        assert determinant != 0.0
        invdet = 1.0 / determinant
        dpdu = (dp1 * dv2 - dp2 * dv1) * invdet
        dpdv = (dp1 * -du2 + dp2 * du1) * invdet
        dg = DG().set(ray.pointAt(t), dpdu, dpdv, self.reverse)
        return t, 1.0e-3 * t, dg

    def intersectP(self, ray):
        """triangle.dart:162-194: Vector temporaries (f32) where intersect keeps f64 scalars."""
        e1 = self.p2 - self.p1
        e2 = self.p3 - self.p1
        s1 = Cross(ray.d, e2)
        divisor = Dot(s1, e1)
        if divisor == 0.0:
            return False
        invDivisor = 1.0 / divisor
        s = ray.o - self.p1
        b1 = Dot(s, s1) * invDivisor
        if b1 < 0.0 or b1 > 1.0:
            return False
        s2 = Cross(s, e1)
        b2 = Dot(ray.d, s2) * invDivisor
        if b2 < 0.0 or b1 + b2 > 1.0:
            return False
        t = Dot(e2, s2) * invDivisor
        if t < ray.mint or t > ray.maxt:
            return False
        return True

    def area(self):                                                # triangle.dart:265-269
        return 0.5 * Cross(self.p2 - self.p1, self.p3 - self.p1).length()

    def sample(self, u1, u2):
        """triangle.dart:366-383 -> (point, Ns)."""
        su1 = math.sqrt(u1)                                        # UniformSampleTriangle montecarlo.dart:215-220
        b1 = 1.0 - su1
        b2 = u2 * su1
        p = self.p1 * b1 + self.p2 * b2 + self.p3 * (1.0 - b1 - b2)
        n = Cross(self.p2 - self.p1, self.p3 - self.p1)
        Ns = Normalize(n)
        if self.reverse:
            Ns = Vec(Ns.x * -1.0, Ns.y * -1.0, Ns.z * -1.0)
        return p, Ns

    def sample2(self, p, u1, u2):                                  # shape.dart:96-98
        return self.sample(u1, u2)

    def pdf2(self, p, wi):
        """core/shape.dart:100-121."""
        ray = Ray(p, wi, 1.0e-3)
        hit = self.intersect(ray)
        if hit is None:
            return 0.0
        thit, _, dgLight = hit
        q = ray.pointAt(thit)
        dist2 = (q - p).lengthSquared()                            # Vector.DistanceSquared(p, q) = (q - p).lengthSquared()
        denom = AbsDot(dgLight.nn, -wi) * self.area()
        try:
            pdf = dist2 / denom
        except ZeroDivisionError:                                  # a double division by zero is +-infinity / NaN in Dart
            pdf = INFINITY if dist2 > 0 else float("nan")
        if math.isinf(pdf):
            pdf = 0.0
        return pdf


# ---------------------------------------------------------------------------------------------------------------
# core/shape.dart, shapes/sphere.dart, shapes/disk.dart: quadrics keep objectToWorld / worldToObject and transform the ray
# ---------------------------------------------------------------------------------------------------------------
def transformPoint(m, p):                                          # transform.dart:110-129 (m: 16 f32 values, row major)
    x, y, z = p.x, p.y, p.z
    out = Vec(m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7], m[8] * x + m[9] * y + m[10] * z + m[11])
    w = m[12] * x + m[13] * y + m[14] * z + m[15]
    if w != 1.0:
        out = Vec(out.x / w, out.y / w, out.z / w)                  # Point.invScale
    return out


def transformNormal(mInv, n):                                      # transform.dart:147-161: the transpose of the inverse
    x, y, z = n.x, n.y, n.z
    return Vec(mInv[0] * x + mInv[4] * y + mInv[8] * z, mInv[1] * x + mInv[5] * y + mInv[9] * z, mInv[2] * x + mInv[6] * y + mInv[10] * z)


def clamp(x, lo, hi):                                              # dart:core num.clamp
    return lo if x < lo else (hi if x > hi else x)


def Radians(deg):                                                  # common.dart:87-88
    return (math.pi / 180.0) * deg


def Quadratic(A, B, C):                                            # common.dart:140-167 -> (t0, t1) or None
    discrim = B * B - 4.0 * A * C
    if discrim < 0.0:
        return None
    rootDiscrim = math.sqrt(discrim)
    if B < 0.0:
        q = -0.5 * (B - rootDiscrim)
    else:
        q = -0.5 * (B + rootDiscrim)
    t0 = _div(q, A)
    t1 = _div(C, q)
    if t0 > t1:
        t0, t1 = t1, t0
    return t0, t1


def _div(a, b):
    """A Dart double division: x / 0 is +-infinity or NaN, never an exception."""
    try:
        return a / b
    except ZeroDivisionError:
        if a == 0.0 or a != a:
            return float("nan")
        return math.copysign(INFINITY, a) * math.copysign(1.0, b)


def CoordinateSystem(v1):                                          # vector.dart:198-214 -> (v2, v3)
    if abs(v1.x) > abs(v1.y):
        invLen = 1.0 / math.sqrt(v1.x * v1.x + v1.z * v1.z)
        v2 = Vec(-v1.z * invLen, 0.0, v1.x * invLen)
    else:
        invLen = 1.0 / math.sqrt(v1.y * v1.y + v1.z * v1.z)
        v2 = Vec(0.0, v1.z * invLen, -v1.y * invLen)
    return v2, Cross(v1, v2)


def UniformSampleSphere(u1, u2):                                   # montecarlo.dart:113-120
    z = 1.0 - 2.0 * u1
    r = math.sqrt(max(0.0, 1.0 - z * z))
    phi = 2.0 * math.pi * u2
    return Vec(r * math.cos(phi), r * math.sin(phi), z)


def UniformSampleCone2(u1, u2, costhetamax, x, y, z):              # montecarlo.dart:135-142; Lerp(t, v1, v2) = v1 * (1 - t) + v2 * t
    costheta = costhetamax * (1.0 - u1) + 1.0 * u1
    sintheta = math.sqrt(1.0 - costheta * costheta)
    phi = u2 * 2.0 * math.pi
    return x * (math.cos(phi) * sintheta) + y * (math.sin(phi) * sintheta) + z * costheta


def UniformConePdf(cosThetaMax):                                   # montecarlo.dart:144-146
    return _div(1.0, 2.0 * math.pi * (1.0 - cosThetaMax))


class Shape:
    """core/shape.dart:26-124: the transforms (as the 16 f32 values of their matrices), the orientation flag and the
    defaults a shape inherits.  transformSwapsHandedness stays false (shape.dart:30; nothing ever assigns it)."""

    def __init__(self, o2w, w2o, ro):
        self.o2w, self.w2o, self.reverse = [float(v) for v in o2w], [float(v) for v in w2o], bool(ro)

    def _objectRay(self, r):                                       # Transform.transformRay (transform.dart:180-196) by worldToObject
        return Ray(transformPoint(self.w2o, r.o), transformVector(self.w2o, r.d), r.mint, r.maxt, r.depth)

    def worldBound(self):                                          # shape.dart:37-39, Transform.transformBBox (transform.dart:163-178)
        b = self.objectBound()
        lo, hi = b.mn, b.mx
        out = BBox()
        for q in (lo, Vec(hi.x, lo.y, lo.z), Vec(lo.x, hi.y, lo.z), Vec(lo.x, lo.y, hi.z), Vec(lo.x, hi.y, hi.z), Vec(hi.x, hi.y, lo.z),
                  Vec(hi.x, lo.y, hi.z), hi):
            out = BBox.UnionPoint(out, transformPoint(self.o2w, q))  # setPoint, then unionPoint
        return out

    def sample2(self, p, u1, u2):                                  # shape.dart:96-98
        return self.sample(u1, u2)

    def pdf2(self, p, wi):
        """core/shape.dart:100-121."""
        ray = Ray(p, wi, 1.0e-3)
        hit = self.intersect(ray)
        if hit is None:
            return 0.0
        thit, _, dgLight = hit
        dist2 = (ray.pointAt(thit) - p).lengthSquared()            # Vector.DistanceSquared(a, b) = (b - a).lengthSquared()
        pdf = _div(dist2, AbsDot(dgLight.nn, -wi) * self.area())
        if math.isinf(pdf):
            pdf = 0.0
        return pdf


class Sphere(Shape):
    def __init__(self, o2w, w2o, ro, radius, z0, z1, pm):          # sphere.dart:24-32 (all members are Dart doubles)
        super().__init__(o2w, w2o, ro)
        self.radius = radius
        self.zmin = clamp(min(z0, z1), -radius, radius)
        self.zmax = clamp(max(z0, z1), -radius, radius)
        self.thetaMin = math.acos(clamp(self.zmin / radius, -1.0, 1.0))
        self.thetaMax = math.acos(clamp(self.zmax / radius, -1.0, 1.0))
        self.phiMax = Radians(clamp(pm, 0.0, 360.0))

    def objectBound(self):                                         # sphere.dart:33-36
        return BBox(Vec(-self.radius, -self.radius, self.zmin), Vec(self.radius, self.radius, self.zmax))

    def _hit(self, r):
        """The part intersect (sphere.dart:38-116) and intersectP (:168-241) share: the object-space ray, the accepted
        root and its point and azimuth, or None."""
        radius, zmin, zmax, phiMax = self.radius, self.zmin, self.zmax, self.phiMax
        ray = self._objectRay(r)
        d, o = ray.d, ray.o
        A = d.x * d.x + d.y * d.y + d.z * d.z
        B = 2 * (d.x * o.x + d.y * o.y + d.z * o.z)
        C = o.x * o.x + o.y * o.y + o.z * o.z - radius * radius
        roots = Quadratic(A, B, C)
        if roots is None:
            return None
        t0, t1 = roots
        if t0 > ray.maxt or t1 < ray.mint:
            return None
        thit = t0
        if thit < ray.mint:
            thit = t1
            if thit > ray.maxt:
                return None

        def at(t):
            phit = ray.pointAt(t)
            if phit.x == 0.0 and phit.y == 0.0:
                phit.x = f32(1.0e-5 * radius)
            phi = math.atan2(phit.y, phit.x)
            if phi < 0.0:
                phi += 2.0 * math.pi
            return phit, phi

        def clipped(phit, phi):
            return (zmin > -radius and phit.z < zmin) or (zmax < radius and phit.z > zmax) or phi > phiMax

        phit, phi = at(thit)
        if clipped(phit, phi):
            if thit == t1:
                return None
            if t1 > ray.maxt:
                return None
            thit = t1
            phit, phi = at(thit)
            if clipped(phit, phi):
                return None
        return thit, phit, phi

    def intersect(self, r):
        """sphere.dart:38-166 -> (t, rayEpsilon, dg) or None.  dndu / dndv (:137-155) are left out: they only feed the
        ray differentials and bump mapping, which are not on the path."""
        h = self._hit(r)
        if h is None:
            return None
        thit, phit, phi = h
        radius, phiMax, thetaMin, thetaMax = self.radius, self.phiMax, self.thetaMin, self.thetaMax
        theta = math.acos(clamp(phit.z / radius, -1.0, 1.0))
        zradius = math.sqrt(phit.x * phit.x + phit.y * phit.y)
        invzradius = _div(1.0, zradius)
        cosphi = phit.x * invzradius
        sinphi = phit.y * invzradius
        dpdu = Vec(-phiMax * phit.y, phiMax * phit.x, 0.0)
        dpdv = Vec(phit.z * cosphi, phit.z * sinphi, -radius * math.sin(theta)) * (thetaMax - thetaMin)
        dg = DG().set(transformPoint(self.o2w, phit), transformVector(self.o2w, dpdu), transformVector(self.o2w, dpdv), self.reverse)
        return thit, 5.0e-4 * thit, dg

    def intersectP(self, r):                                       # sphere.dart:168-241
        return self._hit(r) is not None

    def area(self):                                                # sphere.dart:243-245
        return self.phiMax * self.radius * (self.zmax - self.zmin)

    def sample(self, u1, u2):                                      # sphere.dart:247-260 -> (point, Ns)
        p = Vec() + UniformSampleSphere(u1, u2) * self.radius
        ns = NormalNormalize(transformNormal(self.w2o, Vec(p.x, p.y, p.z)))   # objectToWorld.mInv == worldToObject.m
        if self.reverse:
            ns = Vec(-ns.x, -ns.y, -ns.z)
        return transformPoint(self.o2w, p), ns

    def sample2(self, p, u1, u2):                                  # sphere.dart:262-299
        Pcenter = transformPoint(self.o2w, Vec())
        wc = Normalize(Pcenter - p)
        wcX, wcY = CoordinateSystem(wc)
        if (Pcenter - p).lengthSquared() - self.radius * self.radius < 1.0e-4:
            return self.sample(u1, u2)
        sinThetaMax2 = self.radius * self.radius / (Pcenter - p).lengthSquared()
        cosThetaMax = math.sqrt(max(0.0, 1.0 - sinThetaMax2))
        r = Ray(p, UniformSampleCone2(u1, u2, cosThetaMax, wcX, wcY, wc), 1.0e-3)
        hit = self.intersect(r)
        if hit is None:
            thit = Dot(Pcenter - p, Normalize(r.d))
        else:
            thit = hit[0]
        ps = r.pointAt(thit)
        ns = Normalize(ps - Pcenter)
        if self.reverse:
            ns = Vec(-ns.x, -ns.y, -ns.z)
        return ps, ns

    def pdf2(self, p, wi):                                         # sphere.dart:301-312
        Pcenter = transformPoint(self.o2w, Vec())
        if (Pcenter - p).lengthSquared() - self.radius * self.radius < 1.0e-4:
            return Shape.pdf2(self, p, wi)
        sinThetaMax2 = self.radius * self.radius / (Pcenter - p).lengthSquared()
        cosThetaMax = math.sqrt(max(0.0, 1.0 - sinThetaMax2))
        return UniformConePdf(cosThetaMax)


class Disk(Shape):
    def __init__(self, o2w, w2o, ro, height, radius, innerRadius, phiMax):   # disk.dart:24-29
        super().__init__(o2w, w2o, ro)
        self.height, self.radius, self.innerRadius = height, radius, innerRadius
        self.phiMax = Radians(clamp(phiMax, 0.0, 360.0))

    def objectBound(self):                                         # disk.dart:31-34
        return BBox(Vec(-self.radius, -self.radius, self.height), Vec(self.radius, self.radius, self.height))

    def _hit(self, r):
        """Shared by intersect (disk.dart:38-68) and intersectP (:103-137)."""
        ray = self._objectRay(r)
        if abs(ray.d.z) < 1.0e-7:
            return None
        thit = (self.height - ray.o.z) / ray.d.z
        if thit < ray.mint or thit > ray.maxt:
            return None
        phit = ray.pointAt(thit)
        dist2 = phit.x * phit.x + phit.y * phit.y
        if dist2 > self.radius * self.radius or dist2 < self.innerRadius * self.innerRadius:
            return None
        phi = math.atan2(phit.y, phit.x)
        if phi < 0.0:
            phi += 2.0 * math.pi
        if phi > self.phiMax:
            return None
        return thit, phit, dist2, phi

    def intersect(self, r):                                        # disk.dart:38-101
        h = self._hit(r)
        if h is None:
            return None
        thit, phit, dist2, phi = h
        radius, innerRadius, phiMax = self.radius, self.innerRadius, self.phiMax
        oneMinusV = (math.sqrt(dist2) - innerRadius) / (radius - innerRadius)
        invOneMinusV = (1.0 / oneMinusV) if oneMinusV > 0.0 else 0.0
        dpdu = Vec(-phiMax * phit.y, phiMax * phit.x, 0.0)
        dpdv = Vec(-phit.x * invOneMinusV, -phit.y * invOneMinusV, 0.0)
        dpdu = dpdu * (phiMax * INV_TWOPI)
        dpdv = dpdv * ((radius - innerRadius) / radius)
        dg = DG().set(transformPoint(self.o2w, phit), transformVector(self.o2w, dpdu), transformVector(self.o2w, dpdv), self.reverse)
        return thit, 5.0e-4 * thit, dg

    def intersectP(self, r):                                       # disk.dart:103-137
        return self._hit(r) is not None

    def area(self):                                                # disk.dart:139-142
        return self.phiMax * 0.5 * (self.radius * self.radius - self.innerRadius * self.innerRadius)

    def sample(self, u1, u2):                                      # disk.dart:144-155 -> (point, Ns)
        dx, dy = ConcentricSampleDisk(u1, u2)
        p = Vec(dx * self.radius, dy * self.radius, self.height)
        ns = NormalNormalize(transformNormal(self.w2o, Vec(0.0, 0.0, 1.0)))
        if self.reverse:
            ns = ns * -1.0                                         # Normal.scale(-1.0)
        return transformPoint(self.o2w, p), ns


# ---------------------------------------------------------------------------------------------------------------
# accelerators/bvh_accel.dart
# ---------------------------------------------------------------------------------------------------------------
class Prim:                                                        # GeometricPrimitive (geometric_primitive.dart)
    def __init__(self, shape, material, light=None):
        # material: a Kd triple (matte), or ("matte", Kd) / ("mirror", Kr) / ("glass", Kr, Kt, index)
        if not isinstance(material[0], str):
            material = ("matte", tuple(material))
        self.shape, self.material, self.light = shape, material, light


class Isect:
    __slots__ = ("dg", "prim", "rayEpsilon")


class BVH:
    def __init__(self, nodes, prims):
        """nodes: list of (bmin xyz, bmax xyz, offset, nprims, axis) of the flattened tree; prims in BVH order."""
        self.nodes, self.prims = nodes, prims

    @staticmethod
    def _intersectP(node, ray, invDir, dirIsNeg):
        """bvh_accel.dart:439-472 (bounds[0] = pMin, bounds[1] = pMax)."""
        b = (node[0], node[1])
        tmin = (b[dirIsNeg[0]][0] - ray.o.x) * invDir.x
        tmax = (b[1 - dirIsNeg[0]][0] - ray.o.x) * invDir.x
        tymin = (b[dirIsNeg[1]][1] - ray.o.y) * invDir.y
        tymax = (b[1 - dirIsNeg[1]][1] - ray.o.y) * invDir.y
        if (tmin > tymax) or (tymin > tmax):
            return False
        if tymin > tmin:
            tmin = tymin
        if tymax < tmax:
            tmax = tymax
        tzmin = (b[dirIsNeg[2]][2] - ray.o.z) * invDir.z
        tzmax = (b[1 - dirIsNeg[2]][2] - ray.o.z) * invDir.z
        if (tmin > tzmax) or (tzmin > tmax):
            return False
        if tzmin > tmin:
            tmin = tzmin
        if tzmax < tmax:
            tmax = tzmax
        return (tmin < ray.maxt) and (tmax > ray.mint)

    @staticmethod
    def _invdir(ray):
        def inv(v):
            return math.copysign(INFINITY, v) if v == 0.0 else 1.0 / v   # IEEE 1.0 / +-0.0
        invDir = Vec(inv(ray.d.x), inv(ray.d.y), inv(ray.d.z))            # a Vector: f32 (:109-111)
        return invDir, (1 if invDir.x < 0 else 0, 1 if invDir.y < 0 else 0, 1 if invDir.z < 0 else 0)

    def intersect(self, ray):
        """bvh_accel.dart:101-165 -> Isect or None; shrinks ray.maxt (geometric_primitive.dart:47-61)."""
        if not self.nodes:
            return None
        found = None
        invDir, dirIsNeg = self._invdir(ray)
        todo, nodeNum = [], 0
        while True:
            node = self.nodes[nodeNum]
            if self._intersectP(node, ray, invDir, dirIsNeg):
                if node[3] > 0:
                    for i in range(node[3]):
                        prim = self.prims[node[2] + i]
                        hit = prim.shape.intersect(ray)
                        if hit is not None:
                            thit, eps, dg = hit
                            found = Isect()
                            found.dg, found.prim, found.rayEpsilon = dg, prim, eps
                            ray.maxt = thit
                    if not todo:
                        break
                    nodeNum = todo.pop()
                else:
                    if dirIsNeg[node[4]] != 0:
                        todo.append(nodeNum + 1)
                        nodeNum = node[2]
                    else:
                        todo.append(node[2])
                        nodeNum = nodeNum + 1
            else:
                if not todo:
                    break
                nodeNum = todo.pop()
        return found

    def intersectP(self, ray):
        """bvh_accel.dart:167-226."""
        if not self.nodes:
            return False
        invDir, dirIsNeg = self._invdir(ray)
        todo, nodeNum = [], 0
        while True:
            node = self.nodes[nodeNum]
            if self._intersectP(node, ray, invDir, dirIsNeg):
                if node[3] > 0:
                    for i in range(node[3]):
                        if self.prims[node[2] + i].shape.intersectP(ray):
                            return True
                    if not todo:
                        break
                    nodeNum = todo.pop()
                else:
                    if dirIsNeg[node[4]] != 0:
                        todo.append(nodeNum + 1)
                        nodeNum = node[2]
                    else:
                        todo.append(node[2])
                        nodeNum = nodeNum + 1
            else:
                if not todo:
                    break
                nodeNum = todo.pop()
        return False


# ---------------------------------------------------------------------------------------------------------------
# core/montecarlo.dart
# ---------------------------------------------------------------------------------------------------------------
class Distribution1D:
    def __init__(self, f):                                         # montecarlo.dart:25-52
        count = len(f)
        self.count = count
        self.func = [f32(v) for v in f]
        cdf = [0.0] * (count + 1)
        for i in range(1, count + 1):
            cdf[i] = f32(cdf[i - 1] + self.func[i - 1] / count)
        self.funcInt = cdf[count]
        if self.funcInt == 0.0:
            for i in range(1, count + 1):
                cdf[i] = f32(i / count)
        else:
            for i in range(1, count + 1):
                cdf[i] = f32(cdf[i] / self.funcInt)
        self.cdf = cdf

    def sampleDiscrete(self, u):                                   # montecarlo.dart:82-92 via upper_bound (common.dart:304-333)
        lst, first, count = self.cdf, 0, self.count + 1
        if len(lst) == 1:
            ptr = 0
        else:
            while count > 0:
                index = first
                step = count >> 1
                index += step
                if not (u < lst[index]):
                    index += 1
                    first = index
                    count -= step + 1
                else:
                    count = step
            ptr = first
        return max(0, ptr - 1)

    def sampleContinuous(self, u):
        """montecarlo.dart:54-80 -> (x, pdf, offset)."""
        ptr = upper_bound(self.cdf, u, self.count + 1)
        offset = max(0, ptr - 1)
        if offset == self.count:
            offset = self.count - 1
        dc = self.cdf[offset + 1] - self.cdf[offset]
        du = 0.0
        if dc != 0.0:
            du = (u - self.cdf[offset]) / dc
        pdf = self.func[offset] / self.funcInt
        return (offset + du) / self.count, pdf, offset


def upper_bound(lst, value, last):                                 # common.dart:304-333 (first = 0, compare = less_than)
    if len(lst) == 0:
        return -1
    if len(lst) == 1:
        return 0
    first, count = 0, last
    while count > 0:
        index = first
        step = count >> 1
        index += step
        if not (value < lst[index]):
            index += 1
            first = index
            count -= step + 1
        else:
            count = step
    return first


class Distribution2D:                                              # montecarlo.dart:222-268
    def __init__(self, data, nu, nv):
        self.pConditionalV = [Distribution1D(data[v * nu:v * nu + nu]) for v in range(nv)]
        marginalFunc = [f32(c.funcInt) for c in self.pConditionalV]
        self.pMarginal = Distribution1D(marginalFunc)

    def sampleContinuous(self, u0, u1):
        """-> (u, v, pdf)."""
        v, pdfs1, iv = self.pMarginal.sampleContinuous(u1)
        u, pdfs0, _ = self.pConditionalV[iv].sampleContinuous(u0)
        return u, v, pdfs0 * pdfs1

    def pdf(self, u, v):
        nu, nv = self.pConditionalV[0].count, self.pMarginal.count
        iu = min(max(int(u * nu), 0), nu - 1)                      # .toInt() truncates toward zero
        iv = min(max(int(v * nv), 0), nv - 1)
        if self.pConditionalV[iv].funcInt * self.pMarginal.funcInt == 0.0:
            return 0.0
        return (self.pConditionalV[iv].func[iu] * self.pMarginal.func[iv]) / (self.pConditionalV[iv].funcInt * self.pMarginal.funcInt)


def ConcentricSampleDisk(u1, u2):                                  # montecarlo.dart:155-201
    sx = 2 * u1 - 1
    sy = 2 * u2 - 1
    if sx == 0.0 and sy == 0.0:
        return 0.0, 0.0
    if sx >= -sy:
        if sx > sy:
            r = sx
            theta = sy / r if sy > 0.0 else 8.0 + sy / r
        else:
            r = sy
            theta = 2.0 - sx / r
    else:
        if sx <= sy:
            r = -sx
            theta = 4.0 - sy / r
        else:
            r = -sy
            theta = 6.0 + sx / r
    theta *= math.pi / 4.0
    return r * math.cos(theta), r * math.sin(theta)


def CosineSampleHemisphere(u1, u2):                                # montecarlo.dart:203-209
    dx, dy = ConcentricSampleDisk(u1, u2)
    z = math.sqrt(max(0.0, 1.0 - dx * dx - dy * dy))
    return Vec(dx, dy, z)


def PowerHeuristic(nf, fPdf, ng, gPdf):                            # montecarlo.dart:480-484
    f = nf * fPdf
    g = ng * gPdf
    return (f * f) / (f * f + g * g)


# ---------------------------------------------------------------------------------------------------------------
# core/light/shape_set.dart + lights/diffuse_area_light.dart
# ---------------------------------------------------------------------------------------------------------------
class ShapeSet:
    def __init__(self, shapes):                                    # shape_set.dart:24-51 (shapes already in refine order)
        self.shapes = shapes
        self.areas = [s.area() for s in shapes]
        self.area = 0.0
        for a in self.areas:
            self.area += a
        self.areaDistribution = Distribution1D(self.areas)

    def sample(self, uPos, uComponent, p):
        """shape_set.dart:53-80 (the overload with a reference point) -> (point, Ns)."""
        sn = self.areaDistribution.sampleDiscrete(uComponent) % len(self.shapes)
        pt, Ns = self.shapes[sn].sample2(p, uPos[0], uPos[1])      # Shape.sample2 defaults to sample (shape.dart:96-98)
        r = Ray(p, pt - p, 1.0e-3, INFINITY)
        thit = 1.0
        anyHit, dgn = False, None
        for s in self.shapes:
            hit = s.intersect(r)                                   # Shape.intersect never shrinks r.maxDistance: the LAST hit wins
            if hit is not None:
                thit, dgn = hit[0], hit[2].nn
                anyHit = True
        if anyHit:
            Ns = dgn
        return r.pointAt(thit), Ns

    def pdf(self, p, wi):                                          # shape_set.dart:82-89
        pdf = 0.0
        for a, s in zip(self.areas, self.shapes):
            pdf += a * s.pdf2(p, wi)
        return pdf / self.area


class DiffuseAreaLight:
    def __init__(self, Lemit, shapes):
        self.Lemit = RGB(*Lemit)
        self.shapeSet = ShapeSet(shapes)

    def L(self, n, w):                                             # diffuse_area_light.dart:44-46
        return self.Lemit if Dot(n, w) > 0.0 else RGB(0.0)

    def pdf(self, p, w):
        return self.shapeSet.pdf(p, w)

    def Le(self, ray):                                             # light.dart:70-72: an area light adds nothing to an escaped ray
        return RGB(0.0)

    def sampleLAtPoint(self, p, pEpsilon, uPos, uComponent):
        """diffuse_area_light.dart:60-70 -> (Ls, wi, pdf, shadow ray)."""
        ps, ns = self.shapeSet.sample(uPos, uComponent, p)
        wo = Normalize(ps - p)
        pdf = self.shapeSet.pdf(p, wo)
        dist = (ps - p).length()                                   # VisibilityTester.setSegment (visibility_tester.dart:26-29)
        shadow = Ray(p, (ps - p) / dist, pEpsilon, dist * (1.0 - 1.0e-3))
        return self.L(ns, -wo), wo, pdf, shadow


# ---------------------------------------------------------------------------------------------------------------
# core/mipmap.dart (RGB image, TEXTURE_REPEAT; other sizes than powers of two are resampled first) and lights/infinite_area_light.dart
# ---------------------------------------------------------------------------------------------------------------
_invLog2 = 1.0 / math.log(2.0)                                     # common.dart:98


def Log2(x):                                                       # common.dart:101-103
    return math.log(x) * _invLog2


def Lanczos(x, tau=2.0):                                           # texture.dart:27-39
    x = abs(x)
    if x < 1.0e-5:
        return 1.0
    if x > 1.0:
        return 0.0
    x *= math.pi
    s_ = math.sin(x * tau) / (x * tau)
    lanczos = math.sin(x) / x
    return s_ * lanczos


def RoundUpPow2(v):                                                # common.dart:105-113
    v -= 1
    for k in (1, 2, 4, 8, 16):
        v |= v >> k
    return v + 1


def _resampleWeights(oldres, newres):                              # mipmap.dart:360-384 -> [(firstTexel, [w0, w1, w2, w3])]
    wt = []
    filterwidth = 2.0
    for i in range(newres):
        center = (i + 0.5) * oldres / newres
        firstTexel = math.floor((center - filterwidth) + 0.5)
        weight = [Lanczos(((firstTexel + j + 0.5) - center) / filterwidth) for j in range(4)]
        invSumWts = 1.0 / (weight[0] + weight[1] + weight[2] + weight[3])
        wt.append((firstTexel, [w * invSumWts for w in weight]))
    return wt


def dart_clamp(v, low, high):
    """num.clamp (from memory of the SDK's double.dart): compareTo orders -0.0 below 0.0 and NaN above everything, NaN is returned as is."""
    if v != v:
        return v
    if v < low or (v == 0.0 and low == 0.0 and math.copysign(1.0, v) < 0.0):
        return low
    if v > high:
        return high
    return v


def resampleToPow2(img, xres, yres):
    """The resampling branch of MIPMap.texture (mipmap.dart:71-138), wrapMode TEXTURE_REPEAT.  img: yres rows of xres RGB.
    -> (resampledImage as a list of RGB, sPow2, tPow2)."""
    sPow2, tPow2 = RoundUpPow2(xres), RoundUpPow2(yres)
    sWeights = _resampleWeights(xres, sPow2)
    res = [RGB(0.0)] * (sPow2 * tPow2)
    for t in range(yres):                                          # apply sWeights to zoom in s direction
        for s_ in range(sPow2):
            acc = RGB(0.0)                                         # resampledImage[p] = zero
            first, weight = sWeights[s_]
            for j in range(4):
                origS = (first + j) % xres                         # TEXTURE_REPEAT (Dart % is non-negative: so is Python's)
                if 0 <= origS < xres:
                    px = img[t * xres + origS] * weight[j]
                    acc = acc + px                                 # resampledImage[t * sPow2 + s] += px
            res[t * sPow2 + s_] = acc
    tWeights = _resampleWeights(yres, tPow2)
    for s_ in range(sPow2):                                        # resample image in t direction
        workData = []
        for t in range(tPow2):
            w = RGB(0.0)
            first, weight = tWeights[t]
            for j in range(4):
                offset = (first + j) % yres
                if 0 <= offset < yres:
                    w = w + res[offset * sPow2 + s_] * weight[j]
            workData.append(w)
        for t in range(tPow2):
            w = workData[t]
            res[t * sPow2 + s_] = RGB(dart_clamp(w.r, 0.0, INFINITY), dart_clamp(w.g, 0.0, INFINITY), dart_clamp(w.b, 0.0, INFINITY))   # _clamp(workData[t])
    return res, sPow2, tPow2


class MIPMap:
    def __init__(self, texels, width, height):
        """MIPMap.texture (mipmap.dart:64-160): `texels` = height rows of width (r, g, b)."""
        level0 = [RGB(*t) for t in texels]
        if width & (width - 1) or height & (height - 1):           # !IsPowerOf2(xres) || !IsPowerOf2(yres)
            level0, width, height = resampleToPow2(level0, width, height)
        self.width, self.height = width, height
        self.levels = 1 + int(Log2(max(width, height)))            # .toInt()
        self.pyramid = [(width, height, level0)]
        for i in range(1, self.levels):
            pw, ph, _ = self.pyramid[i - 1]
            sRes, tRes = max(1, pw // 2), max(1, ph // 2)
            lvl = []
            for t in range(tRes):
                for s_ in range(sRes):
                    lvl.append((self.texel(i - 1, 2 * s_, 2 * t) + self.texel(i - 1, 2 * s_ + 1, 2 * t) +
                                self.texel(i - 1, 2 * s_, 2 * t + 1) + self.texel(i - 1, 2 * s_ + 1, 2 * t + 1)) * 0.25)
            self.pyramid.append((sRes, tRes, lvl))

    def texel(self, level, s_, t):                                 # mipmap.dart:184-207 (TEXTURE_REPEAT; Dart % is non-negative)
        w, h, l = self.pyramid[level]
        return l[(t % h) * w + (s_ % w)]

    def triangle(self, level, s_, t):                              # mipmap.dart:342-355
        level = min(max(level, 0), self.levels - 1)
        w, h, _ = self.pyramid[level]
        s_ = s_ * w - 0.5
        t = t * h - 0.5
        s0, t0 = math.floor(s_), math.floor(t)
        ds, dt = s_ - s0, t - t0
        return (self.texel(level, s0, t0) * ((1.0 - ds) * (1.0 - dt)) + self.texel(level, s0, t0 + 1) * ((1.0 - ds) * dt) +
                self.texel(level, s0 + 1, t0) * (ds * (1.0 - dt)) + self.texel(level, s0 + 1, t0 + 1) * (ds * dt))

    def lookup(self, s_, t, width=0.0):                            # mipmap.dart:209-224
        level = self.levels - 1 + Log2(max(width, 1.0e-8))
        if level < 0:
            return self.triangle(0, s_, t)
        elif level >= self.levels - 1:
            return self.texel(self.levels - 1, 0, 0)
        iLevel = math.floor(level)
        delta = level - iLevel
        return self.triangle(iLevel, s_, t) * (1.0 - delta) + self.triangle(iLevel + 1, s_, t) * delta


def transformVector(m, v):                                         # transform.dart:131-145 (m: 16 f32 values, row major)
    return Vec(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z)


def SphericalTheta(v):                                             # vector.dart:185-187
    return math.acos(min(max(v.z, -1.0), 1.0))


def SphericalPhi(v):                                               # vector.dart:189-192
    p = math.atan2(v.y, v.x)
    return p + 2.0 * math.pi if p < 0.0 else p


INV_TWOPI = 0.15915494309189533577                                 # common.dart:24


class InfiniteAreaLight:
    """infinite_area_light.dart:37-68, 262-285: the radiance map, and the Distribution2D over luminance * sin(theta)
    of the map filtered at 1 / max(width, height)."""

    def __init__(self, lightToWorld, worldToLight, L, texels, width, height):
        self.lightToWorld, self.worldToLight = [float(x) for x in lightToWorld], [float(x) for x in worldToLight]
        self.L = RGB(*L)
        self.radianceMap = MIPMap(texels, width, height)
        width, height = self.radianceMap.width, self.radianceMap.height   # _setRadianceMap reads the map's own (resampled) size (:283-285)
        filt = 1.0 / max(width, height)
        img = [0.0] * (width * height)
        for v in range(height):
            vp = v / height
            sinTheta = math.sin(math.pi * (v + 0.5) / height)
            for u in range(width):
                up = u / width
                x = f32(self._radiance(up, vp, filt).luminance())  # img is a Float32List: two stores
                img[u + v * width] = f32(x * sinTheta)
        self.img = img
        self.distribution = Distribution2D(img, width, height)

    def _radiance(self, u, v, width=0.0):                          # :180-182
        return self.radianceMap.lookup(u, v, width) * self.L

    def Le(self, ray):                                             # :84-90
        wh = Normalize(transformVector(self.worldToLight, ray.d))
        s_ = SphericalPhi(wh) * INV_TWOPI
        t = SphericalTheta(wh) * INV_PI
        return self._radiance(s_, t)

    def sampleLAtPoint(self, p, pEpsilon, uPos, uComponent):
        """:92-131 -> (Ls, wi, pdf, shadow ray)."""
        u, v, mapPdf = self.distribution.sampleContinuous(uPos[0], uPos[1])
        if mapPdf == 0.0:
            return RGB(0.0), Vec(), 0.0, None
        theta, phi = v * math.pi, u * 2.0 * math.pi
        costheta, sintheta = math.cos(theta), math.sin(theta)
        sinphi, cosphi = math.sin(phi), math.cos(phi)
        wi = transformVector(self.lightToWorld, Vec(sintheta * cosphi, sintheta * sinphi, costheta))
        pdf = 0.0 if sintheta == 0.0 else mapPdf / (2.0 * math.pi * math.pi * sintheta)
        shadow = Ray(p, wi, pEpsilon, INFINITY)                    # visibility.setRay (visibility_tester.dart:31-33)
        return self._radiance(u, v), wi, pdf, shadow

    def pdf(self, p, w):                                           # :184-200
        wi = transformVector(self.worldToLight, w)
        theta, phi = SphericalTheta(wi), SphericalPhi(wi)
        sintheta = math.sin(theta)
        if sintheta == 0.0:
            return 0.0
        return self.distribution.pdf(phi * INV_TWOPI, theta * INV_PI) / (2.0 * math.pi * math.pi * sintheta)


# ---------------------------------------------------------------------------------------------------------------
# core/reflection/*.dart: BxDF, Lambertian, SpecularReflection, SpecularTransmission, Fresnel*, BSDF; and the three
# materials that build them (materials/matte_material.dart, mirror_material.dart, glass_material.dart)
# ---------------------------------------------------------------------------------------------------------------
def CosTheta(w):                                                   # vector.dart:117
    return w.z


def AbsCosTheta(w):                                                # vector.dart:119
    return abs(w.z)


def SinTheta2(w):                                                  # vector.dart:121-122
    return max(0.0, 1.0 - CosTheta(w) * CosTheta(w))


class BxDF:                                                        # bxdf.dart:23-91
    type = 0

    def matchesFlags(self, flags):
        return (self.type & flags) == self.type

    def sample_f(self, wo, u1, u2):
        """-> (f, wi, pdf): cosine-sample the hemisphere, flipping the direction if necessary (bxdf.dart:37-48)."""
        wi = CosineSampleHemisphere(u1, u2)
        if wo.z < 0.0:
            wi = Vec(wi.x, wi.y, wi.z * -1.0)
        return self.f(wo, wi), wi, self.pdf(wo, wi)

    def pdf(self, wo, wi):                                         # bxdf.dart:84-88, Vector.SameHemisphere (vector.dart:194-196)
        return AbsCosTheta(wi) * INV_PI if wo.z * wi.z > 0.0 else 0.0


class Lambertian(BxDF):                                            # lambertian.dart:23-40
    type = BSDF_REFLECTION | BSDF_DIFFUSE

    def __init__(self, R):
        self.R = R

    def f(self, wo, wi):
        return self.R * INV_PI


class FresnelNoOp:                                                 # fresnel_no_op.dart:23-27
    def evaluate(self, cosi):
        return RGB(1.0)


class FresnelDielectric:                                           # fresnel_dielectric.dart:30-68
    def __init__(self, eta_i, eta_t):
        self.eta_i, self.eta_t = eta_i, eta_t

    def evaluate(self, cosi):
        cosi = -1.0 if cosi < -1.0 else (1.0 if cosi > 1.0 else cosi)
        entering = cosi > 0.0
        ei, et = self.eta_i, self.eta_t
        if not entering:
            ei, et = et, ei
        sint = ei / et * math.sqrt(max(0.0, 1.0 - cosi * cosi))    # Snell's law
        if sint >= 1.0:
            return RGB(1.0)                                        # total internal reflection
        cost = math.sqrt(max(0.0, 1.0 - sint * sint))
        cosi = abs(cosi)
        Rparl = ((et * cosi) - (ei * cost)) / ((et * cosi) + (ei * cost))
        Rperp = ((ei * cosi) - (et * cost)) / ((ei * cosi) + (et * cost))
        return RGB((Rparl * Rparl + Rperp * Rperp) / 2.0)


class SpecularReflection(BxDF):                                    # specular_reflection.dart:23-49
    type = BSDF_REFLECTION | BSDF_SPECULAR

    def __init__(self, R, fresnel):
        self.R, self.fresnel = R, fresnel

    def f(self, wo, wi):
        return RGB(0.0)

    def pdf(self, wo, wi):
        return 0.0

    def sample_f(self, wo, u1, u2):
        wi = Vec(-wo.x, -wo.y, wo.z)                               # perfect specular reflection direction
        return (self.fresnel.evaluate(CosTheta(wo)) * self.R) / AbsCosTheta(wi), wi, 1.0


class SpecularTransmission(BxDF):                                  # specular_transmission.dart:23-79
    type = BSDF_TRANSMISSION | BSDF_SPECULAR

    def __init__(self, T, ei, et):
        self.T, self.etai, self.etat = T, ei, et
        self.fresnel = FresnelDielectric(ei, et)

    def f(self, wo, wi):
        return RGB(0.0)

    def pdf(self, wo, wi):
        return 0.0

    def sample_f(self, wo, u1, u2):
        entering = CosTheta(wo) > 0.0
        ei, et = self.etai, self.etat
        if not entering:
            ei, et = et, ei
        sini2 = SinTheta2(wo)
        eta = ei / et
        sint2 = eta * eta * sini2
        if sint2 >= 1.0:                                           # total internal reflection: pdf stays 0 (bsdf.dart:84-95)
            return RGB(0.0), Vec(), 0.0
        cost = math.sqrt(max(0.0, 1.0 - sint2))
        if entering:
            cost = -cost
        sintOverSini = eta
        wi = Vec(sintOverSini * -wo.x, sintOverSini * -wo.y, cost)
        F = self.fresnel.evaluate(CosTheta(wo))
        return ((RGB(1.0) - F) * self.T) / AbsCosTheta(wi), wi, 1.0


class BSDF:
    def __init__(self, dgs, ngeom, material):
        self.p, self.nn, self.ng = dgs.p, dgs.nn, ngeom             # bsdf.dart:45-51
        self.sn = Normalize(dgs.dpdu)
        self.tn = Cross(self.nn, self.sn)
        self.bxdfs = []
        kind = material[0]
        if kind == "matte":                                         # matte_material.dart:41-65 (sigma == 0)
            r = RGB(*material[1]).clamp()
            if not r.isBlack():
                self.bxdfs.append(Lambertian(r))
        elif kind == "mirror":                                      # mirror_material.dart:38-54
            R = RGB(*material[1]).clamp()
            if not R.isBlack():
                self.bxdfs.append(SpecularReflection(R, FresnelNoOp()))
        elif kind == "glass":                                       # glass_material.dart:44-68
            ior = float(material[3])
            R, T = RGB(*material[1]).clamp(), RGB(*material[2]).clamp()
            if not R.isBlack():
                self.bxdfs.append(SpecularReflection(R, FresnelDielectric(1.0, ior)))
            if not T.isBlack():
                self.bxdfs.append(SpecularTransmission(T, 1.0, ior))
        else:
            raise ValueError(kind)

    def worldToLocal(self, v):                                      # bsdf.dart:177-179
        return Vec(Dot(v, self.sn), Dot(v, self.tn), Dot(v, self.nn))

    def localToWorld(self, v):                                      # bsdf.dart:181-185
        sn, tn, nn = self.sn, self.tn, self.nn
        return Vec(sn.x * v.x + tn.x * v.y + nn.x * v.z, sn.y * v.x + tn.y * v.y + nn.y * v.z, sn.z * v.x + tn.z * v.y + nn.z * v.z)

    def numComponents(self, flags):                                 # bsdf.dart:162-175
        return sum(1 for b in self.bxdfs if b.matchesFlags(flags))

    def f(self, woW, wiW, flags):                                   # bsdf.dart:187-211
        wi, wo = self.worldToLocal(wiW), self.worldToLocal(woW)
        if Dot(wiW, self.ng) * Dot(woW, self.ng) > 0:
            flags = flags & ~BSDF_TRANSMISSION                      # ignore BTDFs
        else:
            flags = flags & ~BSDF_REFLECTION                        # ignore BRDFs
        f = RGB(0.0)
        for b in self.bxdfs:
            if b.matchesFlags(flags):
                f = f + b.f(wo, wi)
        return f

    def pdf(self, woW, wiW, flags):                                 # bsdf.dart:135-156
        if not self.bxdfs:
            return 0.0
        wo, wi = self.worldToLocal(woW), self.worldToLocal(wiW)
        pdf, matching = 0.0, 0
        for b in self.bxdfs:
            if b.matchesFlags(flags):
                matching += 1
                pdf += b.pdf(wo, wi)
        return pdf / matching if matching > 0 else 0.0

    def sample_f(self, woW, uDir, uComponent, flags):
        """bsdf.dart:53-133 -> (f, wiW, pdf, sampledType)."""
        matching = self.numComponents(flags)
        if matching == 0:
            return RGB(0.0), Vec(), 0.0, 0
        which = min(math.floor(uComponent * matching), matching - 1)
        bxdf, count = None, which
        for b in self.bxdfs:
            if b.matchesFlags(flags):
                if count == 0:
                    bxdf = b
                    break
                count -= 1
        wo = self.worldToLocal(woW)
        f, wi, pdf = bxdf.sample_f(wo, uDir[0], uDir[1])
        if pdf == 0.0:
            return RGB(0.0), Vec(), 0.0, 0
        wiW = self.localToWorld(wi)
        if not (bxdf.type & BSDF_SPECULAR) and matching > 1:        # overall pdf with all matching lobes
            for b in self.bxdfs:
                if b is not bxdf and b.matchesFlags(flags):
                    pdf += b.pdf(wo, wi)
        if matching > 1:
            pdf /= matching
        if (bxdf.type & BSDF_SPECULAR) == 0:                        # value of the BSDF for the sampled direction
            f = RGB(0.0)
            if Dot(wiW, self.ng) * Dot(woW, self.ng) > 0:
                flags = flags & ~BSDF_TRANSMISSION
            else:
                flags = flags & ~BSDF_REFLECTION
            for b in self.bxdfs:
                if b.matchesFlags(flags):
                    f = f + b.f(wo, wi)
        return f, wiW, pdf, bxdf.type


# ---------------------------------------------------------------------------------------------------------------
# core/integrator.dart, surface_integrators/*.dart, renderers/sampler_renderer.dart
# ---------------------------------------------------------------------------------------------------------------
class Scene:
    def __init__(self, bvh, lights):
        self.bvh, self.lights = bvh, lights


def isect_Le(isect, wo):                                            # intersection.dart:60-63
    area = isect.prim.light
    return area.L(isect.dg.nn, wo) if area is not None else RGB(0.0)


def getBSDF(isect):
    """Intersection.getBSDF -> GeometricPrimitive.getBSDF -> Triangle.getShadingGeometry (a copy without per-vertex
    N / S, triangle.dart:273-276) -> Material.getBSDF (matte / mirror / glass)."""
    return BSDF(isect.dg, isect.dg.nn, isect.prim.material)


def EstimateDirect(scene, light, p, n, wo, rayEpsilon, bsdf, lightSample, bsdfSample, flags):
    """integrator.dart:119-185.  lightSample = (uPos0, uPos1, uComponent), bsdfSample = (uDir0, uDir1, uComponent)."""
    Ld = RGB(0.0)
    Li, wi, lightPdf, shadow = light.sampleLAtPoint(p, rayEpsilon, (lightSample[0], lightSample[1]), lightSample[2])
    if lightPdf > 0.0 and not Li.isBlack():
        f = bsdf.f(wo, wi, flags)
        if not f.isBlack() and not scene.bvh.intersectP(shadow):
            Li = Li * RGB(1.0)                                      # visibility.transmittance: no VolumeRegion
            bsdfPdf = bsdf.pdf(wo, wi, flags)
            weight = PowerHeuristic(1, lightPdf, 1, bsdfPdf)
            Ld = Ld + f * Li * ((AbsDot(wi, n) * weight / lightPdf))
    f, wi, bsdfPdf, sampledType = bsdf.sample_f(wo, (bsdfSample[0], bsdfSample[1]), bsdfSample[2], flags)
    if not f.isBlack() and bsdfPdf > 0.0:
        weight = 1.0
        if (sampledType & BSDF_SPECULAR) == 0:
            lightPdf = light.pdf(p, wi)
            if lightPdf == 0.0:
                return Ld
            weight = PowerHeuristic(1, bsdfPdf, 1, lightPdf)
        Li = RGB(0.0)
        ray = Ray(p, wi, rayEpsilon, INFINITY)
        lightIsect = scene.bvh.intersect(ray)
        if lightIsect is not None:
            if lightIsect.prim.light is light:
                Li = isect_Le(lightIsect, -wi)
        else:
            Li = light.Le(ray)                                      # 0 for an area light (light.dart:70-72)
        if not Li.isBlack():
            Li = Li * RGB(1.0)                                      # renderer.transmittance
            Ld = Ld + f * Li * (AbsDot(wi, n) * weight / bsdfPdf)
    return Ld


class Draws:
    """The RNG.randomFloat() values drawn inside Li, in order (recorded by the serial run)."""

    def __init__(self, values):
        self.v, self.pos = values, 0

    def randomFloat(self):
        x = self.v[self.pos]
        self.pos += 1
        return float(x)


def UniformSampleOneLight(scene, p, n, wo, rayEpsilon, bsdf, rng, lightNumU=None, lightSample=None, bsdfSample=None):
    """integrator.dart:79-117."""
    nLights = len(scene.lights)
    if nLights == 0:
        return RGB(0.0)
    if lightNumU is not None:
        lightNum = math.floor(lightNumU * nLights)
    else:
        lightNum = math.floor(rng.randomFloat() * nLights)
    lightNum = min(lightNum, nLights - 1)
    light = scene.lights[lightNum]
    if lightSample is None:
        lightSample = (f32(rng.randomFloat()), f32(rng.randomFloat()), rng.randomFloat())   # LightSample.random (light_sample.dart:46-51)
        bsdfSample = (f32(rng.randomFloat()), f32(rng.randomFloat()), rng.randomFloat())    # BSDFSample.random (bsdf_sample.dart:37-42)
    return EstimateDirect(scene, light, p, n, wo, rayEpsilon, bsdf, lightSample, bsdfSample, BSDF_ALL & ~BSDF_SPECULAR) * float(nLights)


SAMPLE_DEPTH = 3                                                    # path_integrator.dart:139


def PathLi(scene, r, isect, sv, rng, maxDepth):
    """path_integrator.dart:29-122.  sv: the flat sample vector (5 camera floats, 14 1-D, 9 2-D entries): per bounce b
    the 1-D slots (light component, light number, bsdf component, path component) = oneD[4b .. 4b+3] and the 2-D slots
    (light position, bsdf direction, path direction) = twoD[3b .. 3b+2] -- the order of requestSamples (:124-131)."""
    oneD = lambda k: float(sv[5 + k])
    twoD = lambda k: (float(sv[5 + 14 + 2 * k]), float(sv[5 + 14 + 2 * k + 1]))
    pathThroughput = RGB(1.0)
    L = RGB(0.0)
    ray = Ray(r.o, r.d, r.mint, r.maxt, r.depth)
    specularBounce = False
    isectP = isect
    bounces = 0
    while True:
        if bounces == 0 or specularBounce:
            L = L + pathThroughput * isect_Le(isectP, -ray.d)
        bsdf = getBSDF(isectP)
        p, n = bsdf.p, bsdf.nn
        wo = -ray.d
        if bounces < SAMPLE_DEPTH:
            lp = twoD(3 * bounces)
            bd = twoD(3 * bounces + 1)
            L = L + pathThroughput * UniformSampleOneLight(
                scene, p, n, wo, isectP.rayEpsilon, bsdf, rng, oneD(4 * bounces + 1),
                (lp[0], lp[1], oneD(4 * bounces)), (bd[0], bd[1], oneD(4 * bounces + 2)))
        else:
            L = L + pathThroughput * UniformSampleOneLight(scene, p, n, wo, isectP.rayEpsilon, bsdf, rng)
        if bounces < SAMPLE_DEPTH:
            pd = twoD(3 * bounces + 2)
            outgoing = (pd[0], pd[1], oneD(4 * bounces + 3))
        else:
            outgoing = (f32(rng.randomFloat()), f32(rng.randomFloat()), rng.randomFloat())
        f, wi, pdf, flags = bsdf.sample_f(wo, (outgoing[0], outgoing[1]), outgoing[2], BSDF_ALL)
        if f.isBlack() or pdf == 0.0:
            break
        specularBounce = (flags & BSDF_SPECULAR) != 0
        pathThroughput = pathThroughput * (f * AbsDot(wi, n) / pdf)
        ray = Ray(p, wi, isectP.rayEpsilon, INFINITY, ray.depth + 1)  # RayDifferential.child
        if bounces > 3:
            continueProbability = min(0.5, pathThroughput.luminance())
            if rng.randomFloat() > continueProbability:
                break
            pathThroughput = pathThroughput / continueProbability
        if bounces == maxDepth:
            break
        localIsect = scene.bvh.intersect(ray)
        if localIsect is None:
            if specularBounce:                                      # path_integrator.dart:106-110
                for light in scene.lights:
                    L = L + pathThroughput * light.Le(ray)
            break
        pathThroughput = pathThroughput * RGB(1.0)                  # renderer.transmittance
        isectP = localIsect
        bounces += 1
    return L


def DirectLi(scene, ray, isect, sv, rng, maxDepth, nSamplesPerLight, strategy="all"):
    """direct_lighting_integrator.dart:30-68 with UniformSampleAllLights (strategy "all", integrator.dart:39-77) or
    UniformSampleOneLight (strategy "one", :79-117).
    Sample layout (requestSamples :70-87).  "all": per light a LightSampleOffsets then a BSDFSampleOffsets, each add1D(n) +
    add2D(n).  "one": LightSampleOffsets(1), lightNumOffset = add1D(1), BSDFSampleOffsets(1) -- oneD = [light component,
    light number, bsdf component], twoD = [light position, bsdf direction].  The volume integrator's two 1-D slots follow."""
    bsdf = getBSDF(isect)
    wo = -ray.d
    p, n = bsdf.p, bsdf.nn
    L = RGB(0.0)
    L = L + isect_Le(isect, wo)
    if scene.lights and strategy == "one":                          # SAMPLE_ONE_UNIFORM (:51-55)
        n1D = 5
        ls = (float(sv[5 + n1D]), float(sv[5 + n1D + 1]), float(sv[5]))
        bs = (float(sv[5 + n1D + 2]), float(sv[5 + n1D + 3]), float(sv[5 + 2]))
        L = L + UniformSampleOneLight(scene, p, n, wo, isect.rayEpsilon, bsdf, rng, float(sv[5 + 1]), ls, bs)
    elif scene.lights:
        n1D = 2 + sum(2 * k for k in nSamplesPerLight)
        Lall = RGB(0.0)
        o1, o2 = 0, 0
        for i, light in enumerate(scene.lights):
            ns = nSamplesPerLight[i]
            lc, lp = o1, o2            # light sample: component slot, position slot
            bc, bd = o1 + ns, o2 + ns  # bsdf sample
            o1 += 2 * ns
            o2 += 2 * ns
            Ld = RGB(0.0)
            for j in range(ns):
                ls = (float(sv[5 + n1D + 2 * (lp + j)]), float(sv[5 + n1D + 2 * (lp + j) + 1]), float(sv[5 + lc + j]))
                bs = (float(sv[5 + n1D + 2 * (bd + j)]), float(sv[5 + n1D + 2 * (bd + j) + 1]), float(sv[5 + bc + j]))
                Ld = Ld + EstimateDirect(scene, light, p, n, wo, isect.rayEpsilon, bsdf, ls, bs, BSDF_ALL & ~BSDF_SPECULAR)
            Lall = Lall + Ld / float(ns)
        L = L + Lall
    if ray.depth + 1 < maxDepth:
        # trace rays for specular reflection and refraction (direct_lighting_integrator.dart:59-65)
        li = lambda rd: RendererLi(scene, "directone" if strategy == "one" else "direct", maxDepth, rd, sv, rng, nSamplesPerLight)
        L = L + SpecularBounce(ray, bsdf, rng, isect, li, BSDF_REFLECTION | BSDF_SPECULAR)
        L = L + SpecularBounce(ray, bsdf, rng, isect, li, BSDF_TRANSMISSION | BSDF_SPECULAR)
    return L


def SpecularBounce(ray, bsdf, rng, isect, rendererLi, flags):
    """Integrator.SpecularReflect / SpecularTransmit (integrator.dart:187-232 / :234-290); the two differ in `flags` and in
    the ray differentials, which nothing on this path reads (no textures).  Each draws a BSDFSample.random(rng) whether
    or not the BSDF has such a lobe."""
    wo = -ray.d
    p, n = bsdf.p, bsdf.nn
    bs = (f32(rng.randomFloat()), f32(rng.randomFloat()), rng.randomFloat())   # BSDFSample.random (bsdf_sample.dart:37-42)
    f, wi, pdf, _ = bsdf.sample_f(wo, (bs[0], bs[1]), bs[2], flags)
    L = RGB(0.0)
    if pdf > 0.0 and not f.isBlack() and AbsDot(wi, n) != 0.0:
        rd = Ray(p, wi, isect.rayEpsilon, INFINITY, ray.depth + 1)            # RayDifferential.child
        Li = rendererLi(rd)
        L = f * Li * (AbsDot(wi, n) / pdf)
    return L


def RendererLi(scene, integrator, maxDepth, ray, sv, rng, nSamplesPerLight=None):
    """SamplerRenderer.Li (sampler_renderer.dart:67-98) without the guards of the task loop: the surface integrator at the
    hit, or the lights' Le along an escaped ray; the volume integrator adds 0 and transmits 1."""
    isect = scene.bvh.intersect(ray)
    if isect is not None:
        if integrator == "path":
            Li = PathLi(scene, ray, isect, sv, rng, maxDepth)
        else:
            Li = DirectLi(scene, ray, isect, sv, rng, maxDepth, nSamplesPerLight, "one" if integrator == "directone" else "all")
    else:
        Li = RGB(0.0)
        for light in scene.lights:
            Li = Li + light.Le(ray)
    return RGB(1.0) * Li + RGB(0.0)                                 # T * Li + Lvi


# ---------------------------------------------------------------------------------------------------------------
# core/matrix4x4.dart, core/transform.dart, core/projective_camera.dart:34-53, cameras/perspective_camera.dart:46-57,
# 139-182: the camera matrices
# ---------------------------------------------------------------------------------------------------------------
def mat_values(*v):                                                # Matrix4x4.values: sixteen stores into a Float32List
    return [f32(x) for x in v]


def mat_mul(m1, m2):                                               # matrix4x4.dart:197-210 (row major)
    r = [0.0] * 16
    for i in range(4):
        k = 4 * i
        for j in range(4):
            r[k + j] = f32(m1[k] * m2[j] + m1[k + 1] * m2[4 + j] + m1[k + 2] * m2[8 + j] + m1[k + 3] * m2[12 + j])
    return r


def mat_inverse(m):
    """Matrix4x4.Inverse = copy + invert (matrix4x4.dart:212-214, 242-354): cofactors over the determinant, every
    element one f64 expression stored as f32; the names nRC read the array COLUMN-wise (n12 = data[4])."""
    n11, n12, n13, n14 = m[0], m[4], m[8], m[12]
    n21, n22, n23, n24 = m[1], m[5], m[9], m[13]
    n31, n32, n33, n34 = m[2], m[6], m[10], m[14]
    n41, n42, n43, n44 = m[3], m[7], m[11], m[15]
    det = ((n14 * n23 * n32 * n41) - (n13 * n24 * n32 * n41) - (n14 * n22 * n33 * n41) + (n12 * n24 * n33 * n41) +
           (n13 * n22 * n34 * n41) - (n12 * n23 * n34 * n41) - (n14 * n23 * n31 * n42) + (n13 * n24 * n31 * n42) +
           (n14 * n21 * n33 * n42) - (n11 * n24 * n33 * n42) - (n13 * n21 * n34 * n42) + (n11 * n23 * n34 * n42) +
           (n14 * n22 * n31 * n43) - (n12 * n24 * n31 * n43) - (n14 * n21 * n32 * n43) + (n11 * n24 * n32 * n43) +
           (n12 * n21 * n34 * n43) - (n11 * n22 * n34 * n43) - (n13 * n22 * n31 * n44) + (n12 * n23 * n31 * n44) +
           (n13 * n21 * n32 * n44) - (n11 * n23 * n32 * n44) - (n12 * n21 * n33 * n44) + (n11 * n22 * n33 * n44))
    if det == 0.0:
        return list(m)
    invDet = 1.0 / det
    d = [0.0] * 16
    d[0] = (n23 * n34 * n42 - n24 * n33 * n42 + n24 * n32 * n43 - n22 * n34 * n43 - n23 * n32 * n44 + n22 * n33 * n44) * invDet
    d[4] = (n14 * n33 * n42 - n13 * n34 * n42 - n14 * n32 * n43 + n12 * n34 * n43 + n13 * n32 * n44 - n12 * n33 * n44) * invDet
    d[8] = (n13 * n24 * n42 - n14 * n23 * n42 + n14 * n22 * n43 - n12 * n24 * n43 - n13 * n22 * n44 + n12 * n23 * n44) * invDet
    d[12] = (n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34) * invDet
    d[1] = (n24 * n33 * n41 - n23 * n34 * n41 - n24 * n31 * n43 + n21 * n34 * n43 + n23 * n31 * n44 - n21 * n33 * n44) * invDet
    d[5] = (n13 * n34 * n41 - n14 * n33 * n41 + n14 * n31 * n43 - n11 * n34 * n43 - n13 * n31 * n44 + n11 * n33 * n44) * invDet
    d[9] = (n14 * n23 * n41 - n13 * n24 * n41 - n14 * n21 * n43 + n11 * n24 * n43 + n13 * n21 * n44 - n11 * n23 * n44) * invDet
    d[13] = (n13 * n24 * n31 - n14 * n23 * n31 + n14 * n21 * n33 - n11 * n24 * n33 - n13 * n21 * n34 + n11 * n23 * n34) * invDet
    d[2] = (n22 * n34 * n41 - n24 * n32 * n41 + n24 * n31 * n42 - n21 * n34 * n42 - n22 * n31 * n44 + n21 * n32 * n44) * invDet
    d[6] = (n14 * n32 * n41 - n12 * n34 * n41 - n14 * n31 * n42 + n11 * n34 * n42 + n12 * n31 * n44 - n11 * n32 * n44) * invDet
    d[10] = (n12 * n24 * n41 - n14 * n22 * n41 + n14 * n21 * n42 - n11 * n24 * n42 - n12 * n21 * n44 + n11 * n22 * n44) * invDet
    d[14] = (n14 * n22 * n31 - n12 * n24 * n31 - n14 * n21 * n32 + n11 * n24 * n32 + n12 * n21 * n34 - n11 * n22 * n34) * invDet
    d[3] = (n23 * n32 * n41 - n22 * n33 * n41 - n23 * n31 * n42 + n21 * n33 * n42 + n22 * n31 * n43 - n21 * n32 * n43) * invDet
    d[7] = (n12 * n33 * n41 - n13 * n32 * n41 + n13 * n31 * n42 - n11 * n33 * n42 - n12 * n31 * n43 + n11 * n32 * n43) * invDet
    d[11] = (n13 * n22 * n41 - n12 * n23 * n41 - n13 * n21 * n42 + n11 * n23 * n42 + n12 * n21 * n43 - n11 * n22 * n43) * invDet
    d[15] = (n12 * n23 * n31 - n13 * n22 * n31 + n13 * n21 * n32 - n11 * n23 * n32 - n12 * n21 * n33 + n11 * n22 * n33) * invDet
    return [f32(x) for x in d]


class Transform:                                                   # transform.dart:31-35: m and mInv, both copied
    def __init__(self, m, inv=None):
        self.m = list(m)
        self.mInv = mat_inverse(m) if inv is None else list(inv)

    def __mul__(self, t2):                                         # :83-86
        return Transform(mat_mul(self.m, t2.m), mat_mul(t2.mInv, self.mInv))

    @staticmethod
    def Inverse(t):                                                # :58-60
        return Transform(t.mInv, t.m)

    @staticmethod
    def Translate(delta):                                          # :214-227
        return Transform(mat_values(1.0, 0.0, 0.0, delta.x, 0.0, 1.0, 0.0, delta.y, 0.0, 0.0, 1.0, delta.z, 0.0, 0.0, 0.0, 1.0),
                         mat_values(1.0, 0.0, 0.0, -delta.x, 0.0, 1.0, 0.0, -delta.y, 0.0, 0.0, 1.0, -delta.z, 0.0, 0.0, 0.0, 1.0))

    @staticmethod
    def Scale(x, y, z):                                            # :229-241
        return Transform(mat_values(x, 0.0, 0.0, 0.0, 0.0, y, 0.0, 0.0, 0.0, 0.0, z, 0.0, 0.0, 0.0, 0.0, 1.0),
                         mat_values(1.0 / x, 0.0, 0.0, 0.0, 0.0, 1.0 / y, 0.0, 0.0, 0.0, 0.0, 1.0 / z, 0.0, 0.0, 0.0, 0.0, 1.0))

    @staticmethod
    def Rotate(angle, axis):                                       # :276-303 (the inverse is Matrix4x4.Transpose(m))
        a = Normalize(axis)
        s_ = math.sin(Radians(angle))
        c = math.cos(Radians(angle))
        m = mat_values(a.x * a.x + (1.0 - a.x * a.x) * c, a.x * a.y * (1.0 - c) - a.z * s_, a.x * a.z * (1.0 - c) + a.y * s_, 0.0,
                       a.x * a.y * (1.0 - c) + a.z * s_, a.y * a.y + (1.0 - a.y * a.y) * c, a.y * a.z * (1.0 - c) - a.x * s_, 0.0,
                       a.x * a.z * (1.0 - c) - a.y * s_, a.y * a.z * (1.0 - c) + a.x * s_, a.z * a.z + (1.0 - a.z * a.z) * c, 0.0,
                       0.0, 0.0, 0.0, 1.0)
        return Transform(m, [m[4 * (i % 4) + i // 4] for i in range(16)])

    @staticmethod
    def LookAt(pos, look, up):                                     # :305-331 -> world-to-camera; its mInv is the matrix built here
        m = [0.0] * 16
        m[3], m[7], m[11], m[15] = pos.x, pos.y, pos.z, 1.0
        d = Normalize(look - pos)
        left = Normalize(Cross(Normalize(up), d))
        newUp = Cross(d, left)
        m[0], m[4], m[8], m[12] = left.x, left.y, left.z, 0.0
        m[1], m[5], m[9], m[13] = newUp.x, newUp.y, newUp.z, 0.0
        m[2], m[6], m[10], m[14] = d.x, d.y, d.z, 0.0
        return Transform(mat_inverse(m), m)

    @staticmethod
    def Perspective(fov, znear, zfar):                             # :338-349
        persp = mat_values(1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, zfar / (zfar - znear), -zfar * znear / (zfar - znear), 0.0, 0.0, 1.0, 0.0)
        invTanAng = 1.0 / math.tan(((math.pi / 180.0) * fov) / 2.0)  # Radians (common.dart:87-88)
        return Transform.Scale(invTanAng, invTanAng, 1.0) * Transform(persp)


def perspective_camera_matrices(pos, look, up, fov, xres, yres):
    """LookAt + Camera "perspective" with the default screen window (perspective_camera.dart:139-182) ->
    (rasterToCamera.m, cameraToWorld.m): projective_camera.dart:34-53."""
    cam2world = Transform.Inverse(Transform.LookAt(Vec(*pos), Vec(*look), Vec(*up)))
    frame = xres / yres
    screen = [-frame, frame, -1.0, 1.0] if frame > 1.0 else [-1.0, 1.0, -1.0 / frame, 1.0 / frame]
    cameraToScreen = Transform.Perspective(fov, 1.0e-2, 1000.0)
    screenToRaster = (Transform.Scale(float(xres), float(yres), 1.0) *
                      Transform.Scale(1.0 / (screen[1] - screen[0]), 1.0 / (screen[2] - screen[3]), 1.0) *
                      Transform.Translate(Vec(-screen[0], -screen[3], 0.0)))
    rasterToScreen = Transform.Inverse(screenToRaster)
    rasterToCamera = Transform.Inverse(cameraToScreen) * rasterToScreen
    return rasterToCamera.m, cam2world.m


class PerspectiveCamera:
    def __init__(self, rasterToCamera, cameraToWorld, lensRadius=0.0, focalDistance=1.0e30):
        self.r2c = [float(v) for v in rasterToCamera]              # Matrix4x4.data: Float32List (row major)
        self.c2w = [float(v) for v in cameraToWorld]
        self.lensRadius, self.focalDistance = float(lensRadius), float(focalDistance)

    @staticmethod
    def _point(m, p):                                               # transform.dart:110-129
        x, y, z = p.x, p.y, p.z
        out = Vec(m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7], m[8] * x + m[9] * y + m[10] * z + m[11])
        w = m[12] * x + m[13] * y + m[14] * z + m[15]
        if w != 1.0:
            out = Vec(out.x / w, out.y / w, out.z / w)              # invScale
        return out

    @staticmethod
    def _vector(m, p):                                              # transform.dart:131-145
        x, y, z = p.x, p.y, p.z
        return Vec(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z)

    def generateRay(self, imageX, imageY, lensU=0.0, lensV=0.0):
        """perspective_camera.dart:93-132 (the ray differentials are not computed: nothing on the path reads them)."""
        Pras = Vec(imageX, imageY, 0.0)
        Pcamera = self._point(self.r2c, Pras)
        d = Normalize(Pcamera)
        o = Vec(0.0, 0.0, 0.0)
        if self.lensRadius > 0.0:                                  # depth of field (:104-119)
            lu, lv = ConcentricSampleDisk(lensU, lensV)
            lu *= self.lensRadius
            lv *= self.lensRadius
            ft = self.focalDistance / d.z
            Pfocus = Ray(o, d).pointAt(ft)
            o = Vec(lu, lv, 0.0)
            d = Normalize(Pfocus - o)
        return Ray(self._point(self.c2w, o), self._vector(self.c2w, d), 0.0, INFINITY, 0)


def renderer_Li(scene, integrator, maxDepth, camera, px, py, sv, draws, nSamplesPerLight=None):
    """SamplerRenderer: generateRayDifferential, Li (sampler_renderer.dart:67-98), the guards (:181-193).
    -> (Ls, imageX, imageY)."""
    imageX = px + float(sv[0])                                      # montecarlo.dart:451-452
    imageY = py + float(sv[1])
    ray = camera.generateRay(imageX, imageY, float(sv[2]), float(sv[3]))
    rng = draws if hasattr(draws, "randomFloat") else Draws(draws)   # a live RNG (serial mode) or the recorded draws
    pos0 = rng.pos
    Ls = RendererLi(scene, integrator, maxDepth, ray, sv, rng, nSamplesPerLight) * 1.0   # Li, then * rayWeight (:170-172)
    if Ls.hasNaNs():
        Ls = RGB(0.0)
    elif Ls.luminance() < -1e-5:
        Ls = RGB(0.0)
    elif math.isinf(Ls.luminance()):
        Ls = RGB(0.0)
    return Ls, imageX, imageY, rng.pos - pos0


# ---------------------------------------------------------------------------------------------------------------
# accelerators/bvh_accel.dart:41-91, 228-437 (the SAH build and the depth-first flattening), core/bbox.dart,
# core/common.dart:255-297 (partition, nth_element), core/primitive.dart:71-84 (fullyRefine)
# ---------------------------------------------------------------------------------------------------------------
def dart_min(a, b):
    """dart:math min on doubles (from memory of sdk/lib/math/math.dart): for zeros of different sign the result is -0.0."""
    if a > b:
        return b
    if a < b:
        return a
    if a == 0.0:
        return (a + b) * a * b
    return b if b != b else a


def dart_max(a, b):
    if a > b:
        return a
    if a < b:
        return b
    if a == 0.0:
        return a + b
    return b if b != b else a


class BBox:                                                        # bbox.dart:27-60: pMin / pMax are Points (f32)
    __slots__ = ("mn", "mx")

    def __init__(self, p1=None, p2=None):
        if p1 is None:
            self.mn, self.mx = Vec(INFINITY, INFINITY, INFINITY), Vec(-INFINITY, -INFINITY, -INFINITY)
        else:
            self.mn = Vec(dart_min(p1.x, p2.x), dart_min(p1.y, p2.y), dart_min(p1.z, p2.z))
            self.mx = Vec(dart_max(p1.x, p2.x), dart_max(p1.y, p2.y), dart_max(p1.z, p2.z))

    @staticmethod
    def Union(b, b2):                                              # bbox.dart:146-155, 203-205
        r = BBox()
        r.mn = Vec(dart_min(b.mn.x, b2.mn.x), dart_min(b.mn.y, b2.mn.y), dart_min(b.mn.z, b2.mn.z))
        r.mx = Vec(dart_max(b.mx.x, b2.mx.x), dart_max(b.mx.y, b2.mx.y), dart_max(b.mx.z, b2.mx.z))
        return r

    @staticmethod
    def UnionPoint(b, q):                                          # bbox.dart:135-144, 199-201
        r = BBox()
        r.mn = Vec(dart_min(b.mn.x, q.x), dart_min(b.mn.y, q.y), dart_min(b.mn.z, q.z))
        r.mx = Vec(dart_max(b.mx.x, q.x), dart_max(b.mx.y, q.y), dart_max(b.mx.z, q.z))
        return r

    def center(self):                                              # bbox.dart:71
        return (self.mn * 0.5) + (self.mx * 0.5)

    def surfaceArea(self):                                         # bbox.dart:166-169
        d = self.mx - self.mn
        return 2.0 * (d.x * d.y + d.x * d.z + d.y * d.z)

    def maximumExtent(self):                                       # bbox.dart:176-185
        d = self.mx - self.mn
        if d.x > d.y and d.x > d.z:
            return 0
        return 1 if d.y > d.z else 2


def _axis(v, dim):
    return (v.x, v.y, v.z)[dim]


def partition(lst, pred, first, last):                             # common.dart:255-283
    while first < last:
        while pred(lst[first]):
            first += 1
            if first == last:
                return first
        while True:
            last -= 1
            if first == last:
                return first
            if pred(lst[last]):
                break
        lst[first], lst[last] = lst[last], lst[first]
        first += 1
    return first


def nth_element(lst, first, nth, last, pred):
    """common.dart:286-294: the whole range is SORTED with the comparator (a, b) => pred(a, b) ? -1 : 1, which never
    answers 0.  dart:core List.sort on at most 32 elements (all this path ever passes: nPrimitives <= 4) is the SDK's
    insertion sort -- `while (j > left && compare(a[j - 1], el) > 0)` -- so an element moves in front of its equals."""
    l = lst[first:last]
    assert len(l) <= 32
    compare = lambda a, b: -1 if pred(a, b) else 1
    for i in range(1, len(l)):
        el = l[i]
        j = i
        while j > 0 and compare(l[j - 1], el) > 0:
            l[j] = l[j - 1]
            j -= 1
        l[j] = el
    lst[first:last] = l


def build_bvh(tris, maxPrims=4):
    """BVHAccel(p, maxPrims, SPLIT_SAH) for already refined primitives `tris` = [(p1, p2, p3) | BBox, ...] (Vec, world space)
    -> (flattened nodes [(bmin, bmax, offset, nPrimitives, axis)], the primitive order as indices into `tris`)."""
    maxPrimsInNode = min(255, maxPrims)
    info = []                                                      # _BVHPrimitiveInfo: (primitiveNumber, centroid, bounds)
    for i, t in enumerate(tris):
        if isinstance(t, BBox):                                    # an intersectable shape kept whole: its Shape.worldBound
            bounds = t
        else:
            a, b, c = t
            bounds = BBox.UnionPoint(BBox(a, b), c)                # Triangle.worldBound (triangle.dart:39-42)
        info.append((i, bounds.center(), bounds))
    ordered, total = [], [0]

    def leaf(start, end, bbox):
        first = len(ordered)
        for i in range(start, end):
            ordered.append(info[i][0])
        return {"bounds": bbox, "first": first, "n": end - start, "children": None, "axis": 0}

    def recursiveBuild(start, end):                                # :228-418
        total[0] += 1
        bbox = BBox()
        for i in range(start, end):
            bbox = BBox.Union(bbox, info[i][2])
        nPrimitives = end - start
        if nPrimitives == 1:
            return leaf(start, end, bbox)
        centroidBounds = BBox()
        for i in range(start, end):
            centroidBounds = BBox.UnionPoint(centroidBounds, info[i][1])
        dim = centroidBounds.maximumExtent()
        mid = (start + end) // 2
        cmin, cmax = _axis(centroidBounds.mn, dim), _axis(centroidBounds.mx, dim)
        if cmax == cmin:
            return leaf(start, end, bbox)
        comparePoints = lambda a, b: _axis(a[1], dim) < _axis(b[1], dim)
        if nPrimitives <= 4:
            mid = (start + end) // 2
            nth_element(info, start, mid, end, comparePoints)
        else:
            nBuckets = 12
            count = [0] * nBuckets
            bb = [BBox() for _ in range(nBuckets)]
            for i in range(start, end):
                b = int(nBuckets * ((_axis(info[i][1], dim) - cmin) / (cmax - cmin)))
                if b == nBuckets:
                    b = nBuckets - 1
                count[b] += 1
                bb[b] = BBox.Union(bb[b], info[i][2])
            cost = [0.0] * (nBuckets - 1)                          # a Float32List
            for i in range(nBuckets - 1):
                b0, b1, count0, count1 = BBox(), BBox(), 0, 0
                for j in range(i + 1):
                    b0 = BBox.Union(b0, bb[j])
                    count0 += count[j]
                for j in range(i + 1, nBuckets):
                    b1 = BBox.Union(b1, bb[j])
                    count1 += count[j]
                cost[i] = f32(0.125 + (count0 * b0.surfaceArea() + count1 * b1.surfaceArea()) / bbox.surfaceArea())
            minCost, minCostSplit = cost[0], 0
            for i in range(1, nBuckets - 1):
                if cost[i] < minCost:
                    minCost, minCostSplit = cost[i], i
            if nPrimitives > maxPrimsInNode or minCost < nPrimitives:
                def compareToBucket(q):
                    b = math.floor(nBuckets * ((_axis(q[1], dim) - cmin) / (cmax - cmin)))
                    if b == nBuckets:
                        b = nBuckets - 1
                    return b <= minCostSplit
                mid = partition(info, compareToBucket, start, end)
            else:
                return leaf(start, end, bbox)
        c2 = recursiveBuild(mid, end)                              # the SECOND child is built first (:411-415)
        c1 = recursiveBuild(start, mid)
        return {"bounds": BBox.Union(c1["bounds"], c2["bounds"]), "first": 0, "n": 0, "children": (c1, c2), "axis": dim}

    if not tris:
        return [], []
    root = recursiveBuild(0, len(tris))
    nodes = [None] * total[0]
    offset = [0]

    def flatten(node):                                             # :420-437
        my = offset[0]
        offset[0] += 1
        bmin, bmax = node["bounds"].mn, node["bounds"].mx
        if node["n"] > 0:
            nodes[my] = ((bmin.x, bmin.y, bmin.z), (bmax.x, bmax.y, bmax.z), node["first"], node["n"], 0)
        else:
            flatten(node["children"][0])
            second = flatten(node["children"][1])
            nodes[my] = ((bmin.x, bmin.y, bmin.z), (bmax.x, bmax.y, bmax.z), second, 0, node["axis"])
        return my

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 10000))
    flatten(root)
    return nodes, ordered


# ---------------------------------------------------------------------------------------------------------------
# core/rng.dart over dart:math Random (the Dart VM's generator: SURVEY.md Appendix E -- the SDK is not part of the
# reference tree), core/montecarlo.dart:294-303, 407-551: the low-discrepancy pixel sample
# ---------------------------------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


class DartRandom:
    """dart:math Random(seed) on the VM: a 64-bit multiply-with-carry state, A = 0xffffda61, seeded through a 64-bit mix
    (0 -> 0x5A17) and four warm-up steps."""

    def __init__(self, seed):
        n = seed & _M64
        n = ((~n) + (n << 21)) & _M64
        n ^= n >> 24
        n = (n * 265) & _M64
        n ^= n >> 14
        n = (n * 21) & _M64
        n ^= n >> 28
        n = (n + (n << 31)) & _M64
        if n == 0:
            n = 0x5A17
        self.lo, self.hi = n & 0xffffffff, n >> 32
        for _ in range(4):
            self._nextState()

    def _nextState(self):
        st = (0xffffda61 * self.lo + self.hi) & _M64
        self.lo, self.hi = st & 0xffffffff, st >> 32

    def nextInt(self, mx):
        if mx & (mx - 1) == 0:                                     # power of two: mask
            self._nextState()
            return self.lo & (mx - 1)
        while True:
            self._nextState()
            rnd32 = self.lo
            result = rnd32 % mx
            if not (rnd32 - result + mx > (1 << 32)):
                return result

    def nextDouble(self):
        return (self.nextInt(1 << 26) * float(1 << 27) + self.nextInt(1 << 27)) / float(1 << 53)


class RNG:                                                         # rng.dart:27-43
    def __init__(self, seed=5489):
        self.random = DartRandom(seed)
        self.pos = 0                                               # randomFloat calls so far (bookkeeping for the tests)

    def randomFloat(self):
        self.pos += 1
        return self.random.nextDouble()

    def randomUint(self):
        return self.random.nextInt(0xffffffff)


ONE_MINUS_EPSILON = 0.9999999403953552                             # montecarlo.dart:23


def Sobol2(n, scramble):                                           # montecarlo.dart:486-493 (64-bit ints: no wrap at bit 31)
    v = 1 << 31
    while n != 0:
        if n & 0x1 != 0:
            scramble ^= v
        n >>= 1
        v ^= v >> 1
    return min(((scramble >> 8) & 0xffffff) / (1 << 24), ONE_MINUS_EPSILON)


def VanDerCorput(n, scramble):                                     # montecarlo.dart:495-504
    n = (n << 16) | (n >> 16)
    n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8)
    n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4)
    n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2)
    n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1)
    n ^= scramble
    return min(((n >> 8) & 0xffffff) / (1 << 24), ONE_MINUS_EPSILON)


def Shuffle(samples, offset, count, dims, rng):                    # montecarlo.dart:294-303
    for i in range(count):
        other = i + (rng.randomUint() % (count - i))
        for j in range(dims):
            a, b = offset + dims * i + j, offset + dims * other + j
            samples[a], samples[b] = samples[b], samples[a]


def LDShuffleScrambled1D(nSamples, nPixel, samples, rng):         # montecarlo.dart:523-534 (`samples`: a Float32List view)
    scramble = rng.randomUint()
    for i in range(nSamples * nPixel):
        samples[i] = f32(VanDerCorput(i, scramble))
    for i in range(nPixel):
        Shuffle(samples, i * nSamples, nSamples, 1, rng)
    Shuffle(samples, 0, nPixel, nSamples, rng)


def LDShuffleScrambled2D(nSamples, nPixel, samples, rng):         # montecarlo.dart:537-548
    scramble = [rng.randomUint(), rng.randomUint()]
    for i in range(nSamples * nPixel):
        samples[2 * i] = f32(VanDerCorput(i, scramble[0]))         # Sample02 (:507-511)
        samples[2 * i + 1] = f32(Sobol2(i, scramble[1]))
    for i in range(nPixel):
        Shuffle(samples, 2 * i * nSamples, nSamples, 2, rng)
    Shuffle(samples, 0, nPixel, 2 * nSamples, rng)


def LDPixelSample(shutterOpen, shutterClose, nPixelSamples, n1D, n2D, rng):
    """montecarlo.dart:407-472 -> nPixelSamples flat sample vectors [image x, y offsets, lens u, v, time, the 1-D slots,
    the 2-D slots] (the pixel position is added by the caller, like samples[i].imageX = xPos + ...)."""
    imageSamples = [0.0] * (2 * nPixelSamples)
    lensSamples = [0.0] * (2 * nPixelSamples)
    timeSamples = [0.0] * nPixelSamples
    oneD = [[0.0] * (n * nPixelSamples) for n in n1D]
    twoD = [[0.0] * (2 * n * nPixelSamples) for n in n2D]
    LDShuffleScrambled2D(1, nPixelSamples, imageSamples, rng)
    LDShuffleScrambled2D(1, nPixelSamples, lensSamples, rng)
    LDShuffleScrambled1D(1, nPixelSamples, timeSamples, rng)
    for i in range(len(n1D)):
        LDShuffleScrambled1D(n1D[i], nPixelSamples, oneD[i], rng)
    for i in range(len(n2D)):
        LDShuffleScrambled2D(n2D[i], nPixelSamples, twoD[i], rng)
    out = []
    for i in range(nPixelSamples):
        t = timeSamples[i]
        v = [imageSamples[2 * i], imageSamples[2 * i + 1], lensSamples[2 * i], lensSamples[2 * i + 1],
             shutterOpen * (1.0 - t) + shutterClose * t]            # Lerp (common.dart:80-81): v1 * (1 - t) + v2 * t
        for j in range(len(n1D)):
            v.extend(oneD[j][n1D[j] * i:n1D[j] * (i + 1)])
        for j in range(len(n2D)):
            v.extend(twoD[j][2 * n2D[j] * i:2 * n2D[j] * (i + 1)])
        out.append(v)
    return out


# ---------------------------------------------------------------------------------------------------------------
# The sampler window and the pixel order: film/image_film.dart:247-252, core/common.dart:52-73 (GetSubWindow),
# dartray/dartray.dart:1009-1023 (_makeSampler), pixel_samplers/linear_pixel_sampler.dart, tile_pixel_sampler.dart
# ---------------------------------------------------------------------------------------------------------------
def getSampleExtent(left, top, width, height, xWidth, yWidth):     # image_film.dart:247-252
    return [math.floor(left + 0.5 - xWidth), math.ceil(left + 0.5 + width + xWidth),
            math.floor(top + 0.5 - yWidth), math.ceil(top + 0.5 + height + yWidth)]


def GetSubWindow(w, h, num, count):                                # common.dart:52-73
    nx, ny = count, 1
    while (nx & 0x1) == 0 and 2 * w * ny < h * nx:
        nx >>= 1
        ny <<= 1
    xo, yo = num % nx, num // nx
    tx0, tx1 = xo / nx, (xo + 1) / nx
    ty0, ty1 = yo / ny, (yo + 1) / ny
    lerp = lambda t, v1, v2: v1 * (1.0 - t) + v2 * t
    return [math.floor(lerp(tx0, 0, w)), min(math.floor(lerp(tx1, 0, w)), w), math.floor(lerp(ty0, 0, h)), min(math.floor(lerp(ty1, 0, h)), h)]


def sampler_window(extent, taskNum, taskCount):
    """_makeSampler (dartray.dart:1009-1023): GetSubWindow's extents -- computed from 0, not from the sample extent's own
    origin -- go to the sampler unchanged -> (x, y, width, height)."""
    e = GetSubWindow(extent[1] - extent[0], extent[3] - extent[2], taskNum, taskCount)
    return e[0], e[2], e[1] - e[0], e[3] - e[2]


def linear_pixels(x, y, width, height):                            # linear_pixel_sampler.dart:29-40
    return [(px, py) for py in range(y, y + height) for px in range(x, x + width)]


def tile_pixels(x, y, width, height, tileSize=32, randomize=True):
    """tile_pixel_sampler.dart:36-100: tiles in row-major order, shuffled from tile 1 on by the sampler's own RNG()
    (seed 5489), pixels row by row inside a tile."""
    left, top, right, bottom = x, y, x + width - 1, y + height - 1
    numXTiles = width // tileSize + (0 if width % tileSize == 0 else 1)
    numYTiles = height // tileSize + (0 if height % tileSize == 0 else 1)
    tiles = [[xi, yi] for yi in range(numYTiles) for xi in range(numXTiles)]
    numTiles = len(tiles)
    if randomize:
        rng = RNG()
        for ti in range(1, numTiles):
            r = rng.randomUint() % numTiles
            tiles[ti], tiles[r] = tiles[r], tiles[ti]
    out = []
    for tx, ty in tiles:
        sx, sy = left + tx * tileSize, top + ty * tileSize
        for yi in range(tileSize):
            py = sy + yi
            if py > bottom:
                break
            for xi in range(tileSize):
                px = sx + xi
                if px > right:
                    break
                out.append((px, py))
    return out


# ---------------------------------------------------------------------------------------------------------------
# film/image_film.dart
# ---------------------------------------------------------------------------------------------------------------
class ImageFilm:
    FILTER_TABLE_SIZE = 16

    def __init__(self, xres, yres, xWidth, yWidth, filterTable):
        self.left, self.top, self.width, self.height = 0, 0, xres, yres      # cropWindow [0,1,0,1] (image_film.dart:61-65)
        self.xWidth, self.yWidth = xWidth, yWidth
        self.invX, self.invY = 1.0 / xWidth, 1.0 / yWidth                      # filter.dart:33-37
        self.table = [float(v) for v in filterTable]                          # Float32List
        self.Lxyz = [0.0] * (3 * xres * yres)                                  # Float32List
        self.weightSum = [0.0] * (xres * yres)

    def addSample(self, imageX, imageY, L):
        """image_film.dart:99-150 (the preview repaint is UI only)."""
        dimageX = imageX - 0.5
        dimageY = imageY - 0.5
        x0 = math.ceil(dimageX - self.xWidth)
        x1 = math.floor(dimageX + self.xWidth)
        y0 = math.ceil(dimageY - self.yWidth)
        y1 = math.floor(dimageY + self.yWidth)
        x0 = max(x0, self.left)
        x1 = min(x1, self.left + self.width - 1)
        y0 = max(y0, self.top)
        y1 = min(y1, self.top + self.height - 1)
        if (x1 - x0) < 0 or (y1 - y0) < 0:
            return
        xyz = (f32(0.412453 * L.r + 0.357580 * L.g + 0.180423 * L.b),       # L.toXYZ(): XYZColor.from (xyz_color.dart:39-42;
               f32(0.212671 * L.r + 0.715160 * L.g + 0.072169 * L.b),       # spectrum.dart:294-298), stored in a Float32List
               f32(0.019334 * L.r + 0.119193 * L.g + 0.950227 * L.b))
        T = self.FILTER_TABLE_SIZE
        ifx = [min(math.floor(abs((x - dimageX) * self.invX * T)), T - 1) for x in range(x0, x1 + 1)]
        ify = [min(math.floor(abs((y - dimageY) * self.invY * T)), T - 1) for y in range(y0, y1 + 1)]
        for y in range(y0, y1 + 1):
            for x in range(x0, x1 + 1):
                filterWt = self.table[ify[y - y0] * T + ifx[x - x0]]
                pi = (y - self.top) * self.width + (x - self.left)
                self.Lxyz[3 * pi] = f32(self.Lxyz[3 * pi] + filterWt * xyz[0])
                self.Lxyz[3 * pi + 1] = f32(self.Lxyz[3 * pi + 1] + filterWt * xyz[1])
                self.Lxyz[3 * pi + 2] = f32(self.Lxyz[3 * pi + 2] + filterWt * xyz[2])
                self.weightSum[pi] = f32(self.weightSum[pi] + filterWt)

    def writeImage(self):
        """image_film.dart:268-299 with splatScale * splatRGB == +0."""
        rgb = [0.0] * (3 * self.width * self.height)
        for pi in range(self.width * self.height):
            X, Y, Z = self.Lxyz[3 * pi], self.Lxyz[3 * pi + 1], self.Lxyz[3 * pi + 2]
            c0 = 3.240479 * X - 1.537150 * Y - 0.498535 * Z             # spectrum.dart:287-291
            c1 = -0.969256 * X + 1.875991 * Y + 0.041556 * Z
            c2 = 0.055648 * X - 0.204043 * Y + 1.057311 * Z
            w = self.weightSum[pi]
            if w != 0.0:
                invWt = 1.0 / w
                rgb[3 * pi] = f32(max(0.0, c0 * invWt))
                rgb[3 * pi + 1] = f32(max(0.0, c1 * invWt))
                rgb[3 * pi + 2] = f32(max(0.0, c2 * invWt))
            for k in range(3):
                rgb[3 * pi + k] = f32(rgb[3 * pi + k] + 1.0 * 0.0)
        return rgb
