"""Host-side mirror of the reference's plugin interface for the hot path.

The reference is Dart; no Dart SDK exists in this image, so the host code above
the C ABI is written in Python with the reference's class names, constructor
arguments and error behaviour (a failing native call raises, like LogSevere,
lib/core/log.dart:42-47).  Each class cites the Dart class it mirrors; the
compute itself happens in libdartray_hip.so.

    prims   = [GeometricPrimitive(TriangleMesh(...), MatteMaterial(Kd), DiffuseAreaLight(L) or None), ...]
    accel   = BVHAccel(prims)                               # lib/accelerators/bvh_accel.dart
    scene   = Scene(accel, lights)                          # lib/core/scene.dart
    film    = ImageFilm(64, 64, BoxFilter(0.5, 0.5))        # lib/film/image_film.dart
    camera  = PerspectiveCamera.lookAt(pos, look, up, fov, film)
    sampler = LowDiscrepancySampler(camera, 4)
    out     = SamplerRenderer(sampler, camera, PathIntegrator(5), EmissionIntegrator()).render(scene)
"""
import ctypes as C
import collections
import math
import os

import numpy as np

from . import _abi
from ._abi import DartRayHipError  # noqa: F401  (re-export)


# ---------------------------------------------------------------------------
# shapes, materials, lights (lib/shapes/triangle_mesh.dart, lib/materials/matte_material.dart,
# lib/lights/diffuse_area_light.dart, lib/core/primitive/geometric_primitive.dart)
# ---------------------------------------------------------------------------
class TriangleMesh:
    """shapes/triangle_mesh.dart:23-36.  P is already in world space (the
    reference pre-transforms vertices to world space, f32).  Optional per-vertex shading data: `n` (normals)
    and `s` (tangents) stay in OBJECT space and are transformed by objectToWorld at shading time
    (triangle.dart:303-317), so a mesh that has them also carries its transform; `uvs` [nverts,2] replace the
    default (0,0),(1,0),(1,1) parametrisation (triangle.dart:247-263)."""

    def __init__(self, vertexIndex, P, reverseOrientation=False, n=None, s=None, uvs=None, objectToWorld=None,
                 worldToObject=None):
        self.vertexIndex = np.ascontiguousarray(vertexIndex, dtype=np.uint32).reshape(-1, 3)
        self.P = np.ascontiguousarray(P, dtype=np.float32).reshape(-1, 3)
        self.reverseOrientation = bool(reverseOrientation)
        nv = len(self.P)
        self.n = None if n is None else np.ascontiguousarray(n, dtype=np.float32).reshape(nv, 3)
        self.s = None if s is None else np.ascontiguousarray(s, dtype=np.float32).reshape(nv, 3)
        self.uvs = None if uvs is None else np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1)[:2 * nv].reshape(nv, 2)
        eye = np.eye(4, dtype=np.float32)
        self.objectToWorld = eye if objectToWorld is None else np.ascontiguousarray(np.asarray(objectToWorld, np.float32).reshape(4, 4))
        self.worldToObject = (eye if objectToWorld is None else _inv(self.objectToWorld)) if worldToObject is None else \
            np.ascontiguousarray(np.asarray(worldToObject, np.float32).reshape(4, 4))
        if self.vertexIndex.size and int(self.vertexIndex.max()) >= len(self.P):
            raise ValueError("TriangleMesh has out of-bounds vertex index")  # triangle_mesh.dart:160-166

    @property
    def ntris(self):
        return len(self.vertexIndex)

    def canIntersect(self):
        return False  # triangle_mesh.dart:79-81

    def refine(self):
        """Triangle order after Primitive.fullyRefine / ShapeSet: the todo list
        is a LIFO stack, so triangles come out reversed (primitive.dart:71-84,
        shape_set.dart:25-35)."""
        return np.arange(self.ntris - 1, -1, -1, dtype=np.int64)


def _bbox_transform(m, lo, hi):
    """Transform.transformBBox (transform.dart:163-178): union of the 8 transformed corners, f32."""
    m = np.asarray(m, np.float32).astype(np.float64)
    lo = np.asarray(lo, np.float64).astype(np.float32).astype(np.float64)  # Point(...) stores f32
    hi = np.asarray(hi, np.float64).astype(np.float32).astype(np.float64)
    lo, hi = np.minimum(lo, hi), np.maximum(lo, hi)                         # BBox(p1, p2) bbox.dart:36-40
    corners = np.array([[(hi if (k >> a) & 1 else lo)[a] for a in range(3)] for k in range(8)])
    pts = transform_points(m, corners)
    return pts.min(0), pts.max(0)


def transform_points(m, P):
    """Transform.transformPoint (transform.dart:110-129) over an [n,3] array: the reference's left-to-right
    f64 sums (no fused multiply-add, which a BLAS matmul may use), stored f32; w != 1 divides (Point.invScale)."""
    m = np.asarray(m, np.float32).astype(np.float64).reshape(4, 4)
    P = np.asarray(P, np.float64).astype(np.float32).astype(np.float64)
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    out = np.stack([m[r, 0] * x + m[r, 1] * y + m[r, 2] * z + m[r, 3] for r in range(3)], axis=1).astype(np.float32)
    w = m[3, 0] * x + m[3, 1] * y + m[3, 2] * z + m[3, 3]
    sel = w != 1.0
    if np.any(sel):
        out[sel] = (out[sel].astype(np.float64) / w[sel, None]).astype(np.float32)
    return out


class _Quadric:
    """Common part of the quadric shapes: they keep objectToWorld and transform the ray per test
    (shape.dart:24-39) instead of pre-transforming geometry, and are intersectable as they are."""

    kind = 0

    def __init__(self, o2w, w2o, reverseOrientation):
        self.objectToWorld = np.ascontiguousarray(np.asarray(o2w, np.float32).reshape(4, 4))
        self.worldToObject = np.ascontiguousarray(np.asarray(w2o, np.float32).reshape(4, 4))
        self.reverseOrientation = bool(reverseOrientation)

    def canIntersect(self):
        return True

    def worldBound(self):  # shape.dart:37-39
        lo, hi = self.objectBound()
        return _bbox_transform(self.objectToWorld, lo, hi)


class Sphere(_Quadric):
    """shapes/sphere.dart:23-38 (constructor arguments as in Sphere.Create :313-321; phiMax in degrees)."""

    kind = _abi.DR_QUADRIC_SPHERE

    def __init__(self, o2w, w2o, ro, radius=1.0, z0=None, z1=None, phiMax=360.0):
        super().__init__(o2w, w2o, ro)
        self.radius = float(radius)
        z0 = -self.radius if z0 is None else float(z0)
        z1 = self.radius if z1 is None else float(z1)
        self.params = (self.radius, z0, z1, float(phiMax))
        self.zmin = min(max(min(z0, z1), -self.radius), self.radius)
        self.zmax = min(max(max(z0, z1), -self.radius), self.radius)

    def objectBound(self):  # sphere.dart:35-38
        return (-self.radius, -self.radius, self.zmin), (self.radius, self.radius, self.zmax)


class Disk(_Quadric):
    """shapes/disk.dart:23-29 (arguments as in Disk.Create :157-165; phiMax in degrees)."""

    kind = _abi.DR_QUADRIC_DISK

    def __init__(self, o2w, w2o, ro, height=0.0, radius=1.0, innerRadius=0.0, phiMax=360.0):
        super().__init__(o2w, w2o, ro)
        self.height, self.radius, self.innerRadius = float(height), float(radius), float(innerRadius)
        self.params = (self.height, self.radius, self.innerRadius, float(phiMax))

    def objectBound(self):  # disk.dart:31-34
        return (-self.radius, -self.radius, self.height), (self.radius, self.radius, self.height)


class MatteMaterial:
    """materials/matte_material.dart:37-77 with constant textures."""

    kind = _abi.DR_MATERIAL_MATTE

    def __init__(self, Kd=(0.5, 0.5, 0.5), sigma=0.0):
        self.Kd = np.asarray(Kd, dtype=np.float32).reshape(3)
        self.sigma = float(sigma)


class MirrorMaterial:
    """materials/mirror_material.dart:35-62 with a constant Kr: one SpecularReflection(Kr, FresnelNoOp) lobe."""

    kind = _abi.DR_MATERIAL_MIRROR

    def __init__(self, Kr=(0.9, 0.9, 0.9)):
        self.Kr = np.asarray(Kr, dtype=np.float32).reshape(3)


class GlassMaterial:
    """materials/glass_material.dart:41-85 with constant textures: SpecularReflection(Kr, FresnelDielectric(1, index))
    + SpecularTransmission(Kt, 1, index)."""

    kind = _abi.DR_MATERIAL_GLASS

    def __init__(self, Kr=(1.0, 1.0, 1.0), Kt=(1.0, 1.0, 1.0), index=1.5):
        self.Kr = np.asarray(Kr, dtype=np.float32).reshape(3)
        self.Kt = np.asarray(Kt, dtype=np.float32).reshape(3)
        self.index = float(index)


class PlasticMaterial:
    """materials/plastic_material.dart:40-85 with constant textures: Lambertian(Kd) + Microfacet(Ks,
    FresnelDielectric(1.5, 1.0), Blinn(1 / roughness))."""

    kind = _abi.DR_MATERIAL_PLASTIC

    def __init__(self, Kd=(0.25, 0.25, 0.25), Ks=(0.25, 0.25, 0.25), roughness=0.1):
        self.Kd = np.asarray(Kd, dtype=np.float32).reshape(3)
        self.Ks = np.asarray(Ks, dtype=np.float32).reshape(3)
        self.roughness = float(roughness)
        self.Kr, self.index = self.Ks, self.roughness  # the fields the C ABI / oracle carry them in


class PointLight:
    """lights/point_light.dart:35-47: an isotropic delta light at lightToWorld(0,0,0) with intensity I."""

    def __init__(self, light2world=None, I=(1.0, 1.0, 1.0)):
        m = np.eye(4, dtype=np.float32) if light2world is None else np.asarray(light2world, np.float32).reshape(4, 4)
        self.lightToWorld = m
        self.lightPos = transform_points(m, np.zeros((1, 3), np.float32))[0]
        self.intensity = np.asarray(I, dtype=np.float32).reshape(3)
        self.nSamples = 1
        self.shape = None

    def isDeltaLight(self):
        return True


class SpotLight(PointLight):
    """lights/spot_light.dart:40-85: a point light with a smooth-step cone about light-space +z; `width` and `fall`
    are the total cone angle and the falloff start, degrees."""

    def __init__(self, light2world=None, I=(1.0, 1.0, 1.0), width=30.0, fall=25.0, world2light=None):
        super().__init__(light2world, I)
        self.worldToLight = _inv(self.lightToWorld) if world2light is None else np.asarray(world2light, np.float32).reshape(4, 4)
        self.width, self.fall = float(width), float(fall)


class DistantLight:
    """lights/distant_light.dart:37-61: radiance L arriving from direction lightDir = normalize(lightToWorld(dir))."""

    def __init__(self, light2world=None, L=(1.0, 1.0, 1.0), dir=(0.0, 0.0, -1.0)):
        m = np.eye(4, dtype=np.float32) if light2world is None else np.asarray(light2world, np.float32).reshape(4, 4)
        self.lightToWorld = m
        d = np.asarray(dir, np.float64).astype(np.float32).astype(np.float64)
        v = np.array([m[r, 0] * d[0] + m[r, 1] * d[1] + m[r, 2] * d[2] for r in range(3)], np.float64).astype(np.float32)
        self.lightDir = _normalize(v)
        self.lightPos = self.lightDir  # the field the C ABI / oracle carry it in
        self.intensity = np.asarray(L, dtype=np.float32).reshape(3)
        self.nSamples = 1
        self.shape = None

    def isDeltaLight(self):
        return True


class DiffuseAreaLight:
    """lights/diffuse_area_light.dart:36-43."""

    def __init__(self, L=(1.0, 1.0, 1.0), nSamples=1, shape=None):
        self.Lemit = np.asarray(L, dtype=np.float32).reshape(3)
        self.nSamples = max(1, int(nSamples))
        self.shape = shape


class InfiniteAreaLight:
    """lights/infinite_area_light.dart:36-68.  `texels` is the radiance map's image [H, W, 3] f32 as MIPMap.texture receives it (a
    size that is no power of two is resampled up to the next one, mipmap.dart:71-138: dr_scene_create does that) -- None gives the
    1x1 white map of the no-'mapname' case;
    `L` the factor _radiance() multiplies in (:180-182).  NB when the reference loads a map from a file it
    ALSO pre-multiplies the texels by L (:44-49), i.e. L is applied twice; callers that want that pass
    pre-multiplied texels."""

    def __init__(self, light2world=None, L=(1.0, 1.0, 1.0), nSamples=1, texels=None):
        m = np.eye(4, dtype=np.float32) if light2world is None else np.asarray(light2world, np.float32).reshape(4, 4)
        self.lightToWorld = m
        self.worldToLight = _inv(m)  # Transform.Inverse (light.dart:30) of a Transform made from the matrix alone (transform.dart:31-35)
        self.L = np.asarray(L, dtype=np.float32).reshape(3)
        self.Lemit = self.L
        self.nSamples = max(1, int(nSamples))
        if texels is None:
            texels = np.ones((1, 1, 3), dtype=np.float32)
        self.texels = np.ascontiguousarray(texels, dtype=np.float32)
        self.shape = None

    def isDeltaLight(self):
        return False


class GeometricPrimitive:
    """core/primitive/geometric_primitive.dart:27-29."""

    def __init__(self, shape, material, areaLight=None):
        self.shape = shape
        self.material = material
        self.areaLight = areaLight
        if areaLight is not None and areaLight.shape is None:
            areaLight.shape = shape

    def getAreaLight(self):
        return self.areaLight


class Ray:
    """Batch of core/ray.dart rays (o, d f32; minDistance/maxDistance f64)."""

    def __init__(self, origin, direction, minDistance=0.0, maxDistance=math.inf):
        self.origin = np.ascontiguousarray(origin, dtype=np.float32).reshape(-1, 3)
        self.direction = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
        n = len(self.origin)
        self.minDistance = np.broadcast_to(np.asarray(minDistance, dtype=np.float64), (n,)).copy()
        self.maxDistance = np.broadcast_to(np.asarray(maxDistance, dtype=np.float64), (n,)).copy()

    def __len__(self):
        return len(self.origin)

    def to_abi(self):
        arr = (_abi.DrRay * len(self))()
        buf = np.frombuffer(arr, dtype=np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("tmin", "<f8"), ("tmax", "<f8")]))
        buf["o"] = self.origin
        buf["d"] = self.direction
        buf["tmin"] = self.minDistance
        buf["tmax"] = self.maxDistance
        return arr


HIT_DTYPE = np.dtype([("prim", "<i4"), ("pad", "<i4"), ("t", "<f8"), ("b1", "<f8"), ("b2", "<f8")])
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<u4"), ("nprims", "<u2"),
                       ("axis", "u1"), ("pad", "u1")])


# ---------------------------------------------------------------------------
# BVHAccel (lib/accelerators/bvh_accel.dart) -- the Aggregate of the scene
# ---------------------------------------------------------------------------
def build_bvh_arrays(verts, refined, quadric_bounds, nquadrics, max_prims, builder=None):
    """BVHAccel's constructor (bvh_accel.dart:41-91,228-437) through the C ABI: (nodes, order, nnodes, depth, builder that ran).
    builder: "device" = dr_bvh_build_device (HIP; needs an initialised GPU), "host" = dr_bvh_build_mixed (C++ threads);
    None = the environment's DARTRAY_BVH_BUILDER, else the device builder whenever a GPU has been selected.  Both write
    the same bytes (tests/test_gpu_bvh_device.py)."""
    n = len(refined)
    lib = _abi.lib()
    if builder is None:
        builder = os.environ.get("DARTRAY_BVH_BUILDER") or ("device" if _abi._initialised is not None else "host")
    if builder not in ("device", "host"):
        raise ValueError("builder must be 'device' or 'host'")
    nodes = np.zeros(max(2 * n - 1, 1), dtype=NODE_DTYPE)
    order = np.zeros(max(n, 1), dtype=np.uint32)
    nn = C.c_uint64(0)
    depth = C.c_uint32(0)
    fn = lib.dr_bvh_build_device if builder == "device" else lib.dr_bvh_build_mixed
    _abi.check(fn(verts.ctypes.data, len(verts), refined.ctypes.data, n, quadric_bounds.ctypes.data, nquadrics, max_prims,
                  nodes.ctypes.data, C.byref(nn), order.ctypes.data, C.byref(depth)))
    return nodes, order, int(nn.value), int(depth.value), builder


class BVHAccel:
    """Aggregate 'bvh' (accelerators/bvh_accel.dart:36-91).

    The constructor refines the primitives, runs the SAH build (host C++ behind
    dr_bvh_build) and keeps the flattened arrays a Dart-side shim would marshal:
    `nodes` (32-byte _LinearBVHNode records) and the per-primitive tables in
    `primitives` order."""

    def __init__(self, p, maxPrims=4, splitMethod="sah", builder=None):
        if splitMethod != "sah":
            raise NotImplementedError("only the default 'sah' split method is on the path")
        self.maxPrimsInNode = min(255, int(maxPrims))
        self.prims_in = list(p)
        verts, tri, mat, lightOf, rev = [], [], [], [], []
        vn, vs, vuv, shade, xform = [], [], [], [], []  # optional per-vertex shading data (triangle_mesh.dart:195-203)
        self.mesh_xforms = []
        base = 0
        self.materials = []
        self._lights = []  # DiffuseAreaLight objects in first-seen order
        self.quadrics = []  # Sphere / Disk shapes: intersectable, so fullyRefine keeps them whole (primitive.dart:71-84)
        for gp in self.prims_in:
            mesh = gp.shape
            mid = len(self.materials)
            self.materials.append(gp.material)
            li = -1
            if gp.areaLight is not None:
                if gp.areaLight not in self._lights:
                    self._lights.append(gp.areaLight)
                li = self._lights.index(gp.areaLight)
            sflags, xf = 0, 0
            if isinstance(mesh, _Quadric):
                tri.append(np.array([[_abi.DR_PRIM_QUADRIC, len(self.quadrics), 0]], dtype=np.uint32))
                self.quadrics.append(mesh)
                nprim = 1
            else:
                order = mesh.refine()
                verts.append(mesh.P)
                tri.append(mesh.vertexIndex[order].astype(np.uint32) + np.uint32(base))
                base += len(mesh.P)
                nprim = len(order)
                n, s, uv = (getattr(mesh, a, None) for a in ("n", "s", "uvs"))
                sflags = (_abi.DR_SHADING_N if n is not None else 0) | (_abi.DR_SHADING_S if s is not None else 0) | \
                    (_abi.DR_SHADING_UV if uv is not None else 0)
                vn.append(n if n is not None else np.zeros((len(mesh.P), 3), np.float32))
                vs.append(s if s is not None else np.zeros((len(mesh.P), 3), np.float32))
                vuv.append(uv if uv is not None else np.zeros((len(mesh.P), 2), np.float32))
                if sflags & (_abi.DR_SHADING_N | _abi.DR_SHADING_S):
                    xf = len(self.mesh_xforms)
                    self.mesh_xforms.append((mesh.objectToWorld, mesh.worldToObject))
            shade.append(np.full(nprim, sflags, dtype=np.uint8))
            xform.append(np.full(nprim, xf, dtype=np.uint32))
            mat.append(np.full(nprim, mid, dtype=np.uint32))
            lightOf.append(np.full(nprim, li, dtype=np.int32))
            rev.append(np.full(nprim, 1 if mesh.reverseOrientation else 0, dtype=np.uint8))
        self.verts = np.ascontiguousarray(np.concatenate(verts) if verts else np.zeros((0, 3), np.float32))
        refined = np.ascontiguousarray(np.concatenate(tri) if tri else np.zeros((0, 3), np.uint32))
        mat = np.concatenate(mat) if mat else np.zeros(0, np.uint32)
        lightOf = np.concatenate(lightOf) if lightOf else np.zeros(0, np.int32)
        rev = np.concatenate(rev) if rev else np.zeros(0, np.uint8)
        shade = np.concatenate(shade) if shade else np.zeros(0, np.uint8)
        xform = np.concatenate(xform) if xform else np.zeros(0, np.uint32)
        self.has_shading = bool(shade.any())
        if self.has_shading:
            self.vert_normals = np.ascontiguousarray(np.concatenate(vn), np.float32)
            self.vert_tangents = np.ascontiguousarray(np.concatenate(vs), np.float32)
            self.vert_uvs = np.ascontiguousarray(np.concatenate(vuv), np.float32)
        n = len(refined)
        qb = np.zeros((max(len(self.quadrics), 1), 6), dtype=np.float32)
        for i, q in enumerate(self.quadrics):
            lo, hi = q.worldBound()
            qb[i, :3], qb[i, 3:] = lo, hi
        import time as _time
        _t0 = _time.perf_counter()
        nodes, order, nn, depth, self.builder = build_bvh_arrays(self.verts, refined, qb, len(self.quadrics), self.maxPrimsInNode, builder)
        self.build_ms = (_time.perf_counter() - _t0) * 1e3  # the constructor proper (host pointers in and out)
        order = order[:n]
        self.nodes = nodes[:nn] if n else None  # bvh_accel.dart:50-53
        self.depth = int(depth)
        self.order = order
        # BVHAccel.primitives (orderedPrims, bvh_accel.dart:69-76)
        self.tri_idx = np.ascontiguousarray(refined[order])
        self.tri_material = np.ascontiguousarray(mat[order])
        self.tri_light = np.ascontiguousarray(lightOf[order])
        self.tri_reverse = np.ascontiguousarray(rev[order])
        self.tri_shading = np.ascontiguousarray(shade[order])
        self.tri_xform = np.ascontiguousarray(xform[order])
        self._scene = None
        self._scene_key = None

    @staticmethod
    def Create(prims, ps=None):  # bvh_accel.dart:474-482
        ps = ps or {}
        return BVHAccel(prims, ps.get("maxnodeprims", 4), ps.get("splitmethod", "sah"))

    @property
    def primitives(self):
        return self.tri_idx

    def canIntersect(self):
        return True

    def worldBound(self):  # bvh_accel.dart:93-95
        if self.nodes is None:
            return (np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32))
        return (self.nodes[0]["bmin"].copy(), self.nodes[0]["bmax"].copy())

    def lights(self):
        """One DiffuseAreaLight per emissive shape (dartray.dart:398-401)."""
        return list(self._lights)

    # --- device scene (created lazily, shared with Scene) ---
    def _device_scene(self, lights=None):
        """The uploaded DrScene of this aggregate + light list.  One aggregate can serve several Scenes whose light
        lists differ (e.g. with and without an InfiniteAreaLight): the cache is keyed on the list's identity."""
        want = self._lights if lights is None else lights
        key = tuple(id(l) for l in want)
        cache = self.__dict__.setdefault("_scenes", collections.OrderedDict())
        if key in cache:
            cache.move_to_end(key)
        else:
            # a few light lists keep their own upload (alternating callers do not evict each other), but not without
            # bound: every entry holds the whole geometry in HBM (C4: 1.1 GB), so the least recently used one goes first
            while len(cache) >= self._SCENE_CACHE:
                cache.popitem(last=False)[1].destroy()
            cache[key] = _DeviceScene(self, want)
        self._scene, self._scene_key = cache[key], key
        return cache[key]

    _SCENE_CACHE = 3

    def intersect(self, ray):
        """Aggregate.intersect (bvh_accel.dart:101-165) on a batch: returns a
        structured array (prim, t, b1, b2); prim == -1 is a miss."""
        return self._device_scene().intersect(ray, any_hit=False)

    def intersectP(self, ray):
        """Aggregate.intersectP (bvh_accel.dart:167-226): bool per ray."""
        return self._device_scene().intersect(ray, any_hit=True)["prim"] >= 0

    def stats(self):
        return self._device_scene().stats()


def delta_light_kind(L):
    return _abi.DR_LIGHT_SPOT if isinstance(L, SpotLight) else (_abi.DR_LIGHT_DISTANT if isinstance(L, DistantLight) else _abi.DR_LIGHT_POINT)


class _DeviceScene:
    """Owns the DrScene handle (scene arrays resident in HBM)."""

    def __init__(self, accel, lights):
        _abi.init(_abi._initialised if _abi._initialised is not None else 0)
        lib = _abi.lib()
        self.lights = list(lights)  # (no reference back to the aggregate: the cache there would make it a cycle)
        # the general shading kernels (not the plain-triangle matte ones) run this scene: dr_api.hip's `general`
        self.general = bool(accel.quadrics) or accel.has_shading or any(isinstance(L, (PointLight, DistantLight)) for L in self.lights) or \
            any(getattr(m, "kind", _abi.DR_MATERIAL_MATTE) != _abi.DR_MATERIAL_MATTE or float(getattr(m, "sigma", 0.0)) != 0.0 for m in accel.materials)
        mats = (_abi.DrMaterial * max(len(accel.materials), 1))()
        for i, m in enumerate(accel.materials):
            mats[i].type = getattr(m, "kind", _abi.DR_MATERIAL_MATTE)
            if mats[i].type == _abi.DR_MATERIAL_MATTE:
                mats[i].kd[:] = [float(x) for x in m.Kd]
                mats[i].sigma = float(m.sigma)
            elif mats[i].type == _abi.DR_MATERIAL_PLASTIC:
                mats[i].kd[:] = [float(x) for x in m.Kd]
                mats[i].kr[:] = [float(x) for x in m.Ks]
                mats[i].index = m.roughness
            else:
                mats[i].kr[:] = [float(x) for x in m.Kr]
                if mats[i].type == _abi.DR_MATERIAL_GLASS:
                    mats[i].kt[:] = [float(x) for x in m.Kt]
                    mats[i].index = m.index
        # lights: ShapeSet triangle lists in refine (reversed) order (shape_set.dart:25-35)
        dl = (_abi.DrAreaLight * max(len(self.lights), 1))()
        ltris = []
        base_of = {}
        base = 0
        quad_of = {id(q): i for i, q in enumerate(accel.quadrics)}
        for gp in accel.prims_in:
            if isinstance(gp.shape, _Quadric):
                continue
            base_of[id(gp.shape)] = base
            base += len(gp.shape.P)
        envs = []
        for i, L in enumerate(self.lights):
            if isinstance(L, (PointLight, DistantLight)):
                dl[i].L[:] = [float(x) for x in L.intensity]
                dl[i].nsamples = 1
                dl[i].kind = delta_light_kind(L)
                dl[i].position[:] = [float(x) for x in L.lightPos]
                if isinstance(L, SpotLight):
                    dl[i].world_to_light[:] = [float(x) for x in L.worldToLight.reshape(-1)]
                    if getattr(L, "marshal_cosines", False):
                        # what a host that only holds the constructed SpotLight has (spot_light.dart:46-47): the two cosines
                        dl[i].kind = _abi.DR_LIGHT_SPOT_COS
                        dl[i].cone_width = math.cos(math.radians(L.width))
                        dl[i].cone_falloff_start = math.cos(math.radians(L.fall))
                    else:
                        dl[i].cone_width, dl[i].cone_falloff_start = L.width, L.fall
                continue
            if isinstance(L, InfiniteAreaLight):
                e = _abi.DrEnvMap()
                e.texels = L.texels.ctypes.data
                e.height, e.width = L.texels.shape[0], L.texels.shape[1]
                e.light_to_world[:] = [float(v) for v in L.lightToWorld.reshape(-1)]
                e.world_to_light[:] = [float(v) for v in L.worldToLight.reshape(-1)]
                dl[i].L[:] = [float(x) for x in L.L]
                dl[i].nsamples = L.nSamples
                dl[i].kind = _abi.DR_LIGHT_INFINITE
                dl[i].env_index = len(envs)
                envs.append(e)
                continue
            mesh = L.shape
            first = len(ltris)
            if isinstance(mesh, _Quadric):  # ShapeSet keeps an intersectable shape whole (shape_set.dart:25-35)
                ltris.append((_abi.DR_PRIM_QUADRIC, quad_of[id(mesh)], 0, 1 if mesh.reverseOrientation else 0))
            else:
                lflags = (1 if mesh.reverseOrientation else 0) | (2 if getattr(mesh, "uvs", None) is not None else 0)
                for t in mesh.refine():
                    v = mesh.vertexIndex[t] + base_of[id(mesh)]
                    ltris.append((int(v[0]), int(v[1]), int(v[2]), lflags))
            dl[i].L[:] = [float(x) for x in L.Lemit]
            dl[i].nsamples = L.nSamples
            dl[i].first_tri = first
            dl[i].ntris = len(ltris) - first
        lt = (_abi.DrLightTri * max(len(ltris), 1))()
        for i, t in enumerate(ltris):
            lt[i].v[:] = t[:3]
            lt[i].reverse_orientation = t[3]
        d = _abi.DrSceneDesc()
        n = len(accel.tri_idx)
        env_arr = (_abi.DrEnvMap * max(len(envs), 1))(*envs)
        d.env_maps = C.cast(env_arr, C.c_void_p)
        d.nenv_maps = len(envs)
        qa = (_abi.DrQuadric * max(len(accel.quadrics), 1))()
        for i, q in enumerate(accel.quadrics):
            qa[i].kind = q.kind
            qa[i].object_to_world[:] = [float(v) for v in q.objectToWorld.reshape(-1)]
            qa[i].world_to_object[:] = [float(v) for v in q.worldToObject.reshape(-1)]
            qa[i].params[:] = [float(v) for v in q.params]
        d.quadrics = C.cast(qa, C.c_void_p)
        d.nquadrics = len(accel.quadrics)
        xa = (_abi.DrMeshXform * max(len(accel.mesh_xforms), 1))()
        if accel.has_shading:
            for i, (o2w, w2o) in enumerate(accel.mesh_xforms):
                xa[i].object_to_world[:] = [float(v) for v in np.asarray(o2w, np.float32).reshape(-1)]
                xa[i].world_to_object[:] = [float(v) for v in np.asarray(w2o, np.float32).reshape(-1)]
            d.vert_normals = accel.vert_normals.ctypes.data
            d.vert_tangents = accel.vert_tangents.ctypes.data
            d.vert_uvs = accel.vert_uvs.ctypes.data
            d.tri_shading = accel.tri_shading.ctypes.data
            d.tri_xform = accel.tri_xform.ctypes.data
            d.mesh_xforms = C.cast(xa, C.c_void_p)
            d.nmesh_xforms = len(accel.mesh_xforms)
        self._keep = (mats, dl, lt, env_arr, qa, xa)
        d.nodes = accel.nodes.ctypes.data if accel.nodes is not None else None
        d.nnodes = len(accel.nodes) if accel.nodes is not None else 0
        d.verts = accel.verts.ctypes.data
        d.nverts = len(accel.verts)
        d.tri_idx = accel.tri_idx.ctypes.data
        d.ntris = n
        d.tri_material = accel.tri_material.ctypes.data
        # per-primitive area-light index = position of its DiffuseAreaLight in Scene.lights
        pos = np.full(len(accel._lights) + 1, -1, dtype=np.int32)
        for i, al in enumerate(accel._lights):
            if al not in self.lights:
                raise ValueError("an emissive primitive's area light is missing from Scene.lights")
            pos[i] = self.lights.index(al)
        tri_light = np.ascontiguousarray(np.where(accel.tri_light >= 0, pos[accel.tri_light], -1).astype(np.int32))
        self._keep = self._keep + (tri_light,)
        d.tri_light = tri_light.ctypes.data
        d.tri_reverse = accel.tri_reverse.ctypes.data
        d.materials = C.cast(mats, C.c_void_p)
        d.nmaterials = len(accel.materials)
        d.lights = C.cast(dl, C.c_void_p)
        d.nlights = len(self.lights)
        d.light_tris = C.cast(lt, C.c_void_p)
        d.nlight_tris = len(ltris)
        d.bvh_depth = accel.depth
        h = C.c_void_p()
        _abi.check(lib.dr_scene_create(C.byref(d), C.byref(h)))
        self.handle = h

    def destroy(self):
        """dr_scene_destroy now (an evicted cache entry must not wait for a garbage collection)."""
        if getattr(self, "handle", None):
            _abi.lib().dr_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def intersect(self, ray, any_hit):
        n = len(ray)
        out = np.zeros(n, dtype=HIT_DTYPE)
        if n:
            arr = ray.to_abi()
            _abi.check(_abi.lib().dr_intersect(self.handle, C.cast(arr, C.c_void_p), n, out.ctypes.data, 1 if any_hit else 0))
        return out

    def stats(self):
        s = _abi.DrRenderStats()
        _abi.check(_abi.lib().dr_get_stats(self.handle, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _abi.DrRenderStats._fields_}

    def reset_stats(self):
        _abi.check(_abi.lib().dr_reset_stats(self.handle))

    def state_layout(self, layout=None):
        """Get (layout, alive share at the second bounce) of this scene's path renders (0 = not measured yet), or store a
        layout (64 / 4; 0 makes the next big render measure again)."""
        if layout is not None:
            _abi.check(_abi.lib().dr_scene_set_state_layout(self.handle, int(layout)))
        lay, dens = C.c_int32(0), C.c_float(0.0)
        _abi.check(_abi.lib().dr_scene_get_state_layout(self.handle, C.byref(lay), C.byref(dens)))
        return int(lay.value), float(dens.value)

    def trace_kernels(self, kernels=None):
        """Get (closest, any-hit) traversal kernels of this scene (0 = not measured yet), or set them (2 / 3, closest-hit also 5,
        any-hit also 6 / 7 = 2 / 3 with the far child first; (0, 0) makes the next big render measure again)."""
        arr = (C.c_uint32 * 2)(*(kernels or (0, 0)))
        if kernels is not None:
            _abi.check(_abi.lib().dr_scene_set_trace_kernels(self.handle, C.byref(arr)))
        _abi.check(_abi.lib().dr_scene_get_trace_kernels(self.handle, C.byref(arr)))
        return int(arr[0]), int(arr[1])

    def last_render_info(self):
        """What the last render_device call ran with (dr_scene_last_render_info)."""
        arr = (C.c_int32 * 8)()
        _abi.check(_abi.lib().dr_scene_last_render_info(self.handle, C.byref(arr)))
        return {"state_layout": int(arr[0]), "closest_kernel": int(arr[1]), "any_hit_kernel": int(arr[2]),
                "pilot_batches": int(arr[4]), "batches": int(arr[5]), "trace_wg_per_cu": int(arr[6]), "overlap_any": int(arr[7]) & 1,
                "coherent_camera": (int(arr[7]) >> 1) & 1, "lazy_gen": (int(arr[7]) >> 3) & 1}

    def workspace_bytes(self):
        """Device memory the scene's path-state workspace holds right now (dr_scene_workspace_bytes)."""
        n = C.c_uint64(0)
        _abi.check(_abi.lib().dr_scene_workspace_bytes(self.handle, C.byref(n)))
        return int(n.value)

    def coherent_stats(self):
        """The part of stats()' closest-hit totals that k_trace_pk (coherent waves: the camera rays) traced."""
        arr = (C.c_double * 5)()
        _abi.check(_abi.lib().dr_scene_get_coherent_stats(self.handle, C.byref(arr)))
        return {"rays": int(arr[0]), "nodes": int(arr[1]), "tris": int(arr[2]), "launches": int(arr[3]), "ms": float(arr[4])}

    def sampler_stats(self):
        """(pixel, LD block) pairs the device sampler generated / that the integrator's reads name (lazy generation: fewer where paths end early)."""
        arr = (C.c_double * 2)()
        _abi.check(_abi.lib().dr_scene_get_sampler_stats(self.handle, C.byref(arr)))
        return {"generated": int(arr[0]), "named": int(arr[1])}

    def pilot(self):
        """What the traversal pilot measured, ms per algorithmic GB: {"closest": {2: .., 3: .., 5: ..}, "any_hit": {2: .., 3: ..}}
        (0.0 = that candidate was not timed)."""
        arr = (C.c_float * 6)()
        _abi.check(_abi.lib().dr_scene_get_pilot(self.handle, C.byref(arr)))
        return {"closest": {2: float(arr[0]), 3: float(arr[1]), 5: float(arr[2])}, "any_hit": {2: float(arr[3]), 3: float(arr[4])},
                "far_first": float(arr[5])}  # any-hit rays: far child first over the reference order, time per ray (0 = not measured)


class Scene:
    """core/scene.dart:26-45."""

    def __init__(self, aggregate, lights, volumeRegion=None):
        if volumeRegion is not None:
            raise NotImplementedError("participating media are not on the path")
        self.aggregate = aggregate
        self.lights = list(lights)
        self.volumeRegion = None
        self.worldBound = aggregate.worldBound()

    def _device(self):
        return self.aggregate._device_scene(self.lights)

    def intersect(self, ray):  # scene.dart:51-56 (through this Scene's own device scene: one upload per Scene)
        return self._device().intersect(ray, any_hit=False)

    def intersectP(self, ray):  # scene.dart:63-68
        return self._device().intersect(ray, any_hit=True)["prim"] >= 0


# ---------------------------------------------------------------------------
# filter, film, camera (lib/filters/box_filter.dart, lib/film/image_film.dart,
# lib/cameras/perspective_camera.dart, lib/core/projective_camera.dart)
# ---------------------------------------------------------------------------
class Filter:
    """core/filter.dart:26-39.  ImageFilm tabulates evaluate() at 16 x 16 points (image_film.dart:74-82); the table and
    the two widths are what cross the C ABI (DrFilm.filter_table), so every filter below runs on the device path."""
    def __init__(self, xw, yw):
        self.xWidth = float(xw)
        self.yWidth = float(yw)
        self.invXWidth = 1.0 / self.xWidth
        self.invYWidth = 1.0 / self.yWidth


class BoxFilter(Filter):
    def __init__(self, xw=0.5, yw=0.5):  # box_filter.dart:33-46
        super().__init__(xw, yw)

    def evaluate(self, x, y):
        return 1.0


class GaussianFilter(Filter):
    def __init__(self, xw=2.0, yw=2.0, alpha=2.0):  # gaussian_filter.dart:24-47
        super().__init__(xw, yw)
        self.alpha = float(alpha)
        self.expX = math.exp(-self.alpha * self.xWidth * self.xWidth)
        self.expY = math.exp(-self.alpha * self.yWidth * self.yWidth)

    def _gaussian(self, d, expv):
        return max(0.0, math.exp(-self.alpha * d * d) - expv)

    def evaluate(self, x, y):
        return self._gaussian(x, self.expX) * self._gaussian(y, self.expY)


class MitchellFilter(Filter):
    def __init__(self, b=1.0 / 3.0, c=1.0 / 3.0, xw=2.0, yw=2.0):  # mitchell_filter.dart:24-53
        super().__init__(xw, yw)
        self.b = float(b)
        self.c = float(c)

    def _mitchell1D(self, x):
        b, c = self.b, self.c
        x = abs(2.0 * x)
        if x > 1.0:
            return ((-b - 6 * c) * x * x * x + (6 * b + 30 * c) * x * x + (-12 * b - 48 * c) * x + (8 * b + 24 * c)) * (1.0 / 6.0)
        return ((12 - 9 * b - 6 * c) * x * x * x + (-18 + 12 * b + 6 * c) * x * x + (6 - 2 * b)) * (1.0 / 6.0)

    def evaluate(self, x, y):
        return self._mitchell1D(x * self.invXWidth) * self._mitchell1D(y * self.invYWidth)


class TriangleFilter(Filter):
    def __init__(self, xw=2.0, yw=2.0):  # triangle_filter.dart:24-38
        super().__init__(xw, yw)

    def evaluate(self, x, y):
        return max(0.0, self.xWidth - abs(x)) * max(0.0, self.yWidth - abs(y))


class LanczosSincFilter(Filter):
    def __init__(self, xw=4.0, yw=4.0, tau=3.0):  # lanczos_sinc_filter.dart:24-56
        super().__init__(xw, yw)
        self.tau = float(tau)

    def _sinc1D(self, x):
        x = abs(x)
        if x < 1e-5:
            return 1.0
        if x > 1.0:
            return 0.0
        x *= math.pi
        sinc = math.sin(x) / x
        lanczos = math.sin(x * self.tau) / (x * self.tau)
        return sinc * lanczos

    def evaluate(self, x, y):
        return self._sinc1D(x * self.invXWidth) * self._sinc1D(y * self.invYWidth)


FILTER_TABLE_SIZE = 16  # image_film.dart:307


class ImageFilm:
    def __init__(self, xres, yres, filter=None, cropWindow=(0.0, 1.0, 0.0, 1.0)):
        self.xResolution = int(xres)
        self.yResolution = int(yres)
        self.filter = filter or BoxFilter()
        self.cropWindow = tuple(float(c) for c in cropWindow)
        # image_film.dart:61-65
        self.left = math.ceil(self.xResolution * self.cropWindow[0])
        self.width = max(1, math.ceil(self.xResolution * self.cropWindow[1]) - self.left)
        self.top = math.ceil(self.yResolution * self.cropWindow[2])
        self.height = max(1, math.ceil(self.yResolution * self.cropWindow[3]) - self.top)
        # image_film.dart:74-82
        t = np.zeros(FILTER_TABLE_SIZE * FILTER_TABLE_SIZE, dtype=np.float32)
        fi = 0
        for y in range(FILTER_TABLE_SIZE):
            fy = (y + 0.5) * self.filter.yWidth / FILTER_TABLE_SIZE
            for x in range(FILTER_TABLE_SIZE):
                fx = (x + 0.5) * self.filter.xWidth / FILTER_TABLE_SIZE
                t[fi] = self.filter.evaluate(fx, fy)
                fi += 1
        self.filterTable = t

    def getSampleExtent(self):  # image_film.dart:247-252
        return (math.floor(self.left + 0.5 - self.filter.xWidth),
                math.ceil(self.left + 0.5 + self.width + self.filter.xWidth),
                math.floor(self.top + 0.5 - self.filter.yWidth),
                math.ceil(self.top + 0.5 + self.height + self.filter.yWidth))

    def to_abi(self, f):
        f.xres, f.yres = self.xResolution, self.yResolution
        f.crop[:] = self.cropWindow
        f.filter_xw, f.filter_yw = self.filter.xWidth, self.filter.yWidth
        f.filter_table[:] = [float(v) for v in self.filterTable]


def _m4(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32).reshape(4, 4)


def _mul(a, b):  # Matrix4x4.Mul: left-to-right f64 sums, f32 store (matrix4x4.dart:193-206)
    a, b = a.astype(np.float64), b.astype(np.float64)
    r = np.empty((4, 4), dtype=np.float64)
    for i in range(4):
        r[i] = a[i, 0] * b[0] + a[i, 1] * b[1] + a[i, 2] * b[2] + a[i, 3] * b[3]
    return r.astype(np.float32)


def _inv(a):
    """Matrix4x4.Inverse (matrix4x4.dart:212-214, 242-354): the reference's own formula -- cofactors over the determinant,
    every element ONE f64 expression in the reference's term order, stored f32; a singular matrix comes back unchanged.
    (A general-purpose inverse such as numpy.linalg.inv differs in the last bits and leaves 1e-17 where this leaves 0.)"""
    d = [float(v) for v in np.asarray(a, np.float32).reshape(-1)]
    # the reference names the elements column-wise: nRC = data[4 * (C - 1) + (R - 1)]
    n11, n12, n13, n14 = d[0], d[4], d[8], d[12]
    n21, n22, n23, n24 = d[1], d[5], d[9], d[13]
    n31, n32, n33, n34 = d[2], d[6], d[10], d[14]
    n41, n42, n43, n44 = d[3], d[7], d[11], d[15]
    det = ((n14 * n23 * n32 * n41) - (n13 * n24 * n32 * n41) - (n14 * n22 * n33 * n41) + (n12 * n24 * n33 * n41) +
           (n13 * n22 * n34 * n41) - (n12 * n23 * n34 * n41) - (n14 * n23 * n31 * n42) + (n13 * n24 * n31 * n42) +
           (n14 * n21 * n33 * n42) - (n11 * n24 * n33 * n42) - (n13 * n21 * n34 * n42) + (n11 * n23 * n34 * n42) +
           (n14 * n22 * n31 * n43) - (n12 * n24 * n31 * n43) - (n14 * n21 * n32 * n43) + (n11 * n24 * n32 * n43) +
           (n12 * n21 * n34 * n43) - (n11 * n22 * n34 * n43) - (n13 * n22 * n31 * n44) + (n12 * n23 * n31 * n44) +
           (n13 * n21 * n32 * n44) - (n11 * n23 * n32 * n44) - (n12 * n21 * n33 * n44) + (n11 * n22 * n33 * n44))
    if det == 0.0:
        return np.asarray(a, np.float32).reshape(4, 4).copy()
    i = 1.0 / det
    r = [0.0] * 16
    r[0] = (n23 * n34 * n42 - n24 * n33 * n42 + n24 * n32 * n43 - n22 * n34 * n43 - n23 * n32 * n44 + n22 * n33 * n44) * i
    r[4] = (n14 * n33 * n42 - n13 * n34 * n42 - n14 * n32 * n43 + n12 * n34 * n43 + n13 * n32 * n44 - n12 * n33 * n44) * i
    r[8] = (n13 * n24 * n42 - n14 * n23 * n42 + n14 * n22 * n43 - n12 * n24 * n43 - n13 * n22 * n44 + n12 * n23 * n44) * i
    r[12] = (n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34) * i
    r[1] = (n24 * n33 * n41 - n23 * n34 * n41 - n24 * n31 * n43 + n21 * n34 * n43 + n23 * n31 * n44 - n21 * n33 * n44) * i
    r[5] = (n13 * n34 * n41 - n14 * n33 * n41 + n14 * n31 * n43 - n11 * n34 * n43 - n13 * n31 * n44 + n11 * n33 * n44) * i
    r[9] = (n14 * n23 * n41 - n13 * n24 * n41 - n14 * n21 * n43 + n11 * n24 * n43 + n13 * n21 * n44 - n11 * n23 * n44) * i
    r[13] = (n13 * n24 * n31 - n14 * n23 * n31 + n14 * n21 * n33 - n11 * n24 * n33 - n13 * n21 * n34 + n11 * n23 * n34) * i
    r[2] = (n22 * n34 * n41 - n24 * n32 * n41 + n24 * n31 * n42 - n21 * n34 * n42 - n22 * n31 * n44 + n21 * n32 * n44) * i
    r[6] = (n14 * n32 * n41 - n12 * n34 * n41 - n14 * n31 * n42 + n11 * n34 * n42 + n12 * n31 * n44 - n11 * n32 * n44) * i
    r[10] = (n12 * n24 * n41 - n14 * n22 * n41 + n14 * n21 * n42 - n11 * n24 * n42 - n12 * n21 * n44 + n11 * n22 * n44) * i
    r[14] = (n14 * n22 * n31 - n12 * n24 * n31 - n14 * n21 * n32 + n11 * n24 * n32 + n12 * n21 * n34 - n11 * n22 * n34) * i
    r[3] = (n23 * n32 * n41 - n22 * n33 * n41 - n23 * n31 * n42 + n21 * n33 * n42 + n22 * n31 * n43 - n21 * n32 * n43) * i
    r[7] = (n12 * n33 * n41 - n13 * n32 * n41 + n13 * n31 * n42 - n11 * n33 * n42 - n12 * n31 * n43 + n11 * n32 * n43) * i
    r[11] = (n13 * n22 * n41 - n12 * n23 * n41 - n13 * n21 * n42 + n11 * n23 * n42 + n12 * n21 * n43 - n11 * n22 * n43) * i
    r[15] = (n12 * n23 * n31 - n13 * n22 * n31 + n13 * n21 * n32 - n11 * n23 * n32 - n12 * n21 * n33 + n11 * n22 * n33) * i
    return np.asarray(r, dtype=np.float64).astype(np.float32).reshape(4, 4)


def _normalize(v):
    v = np.asarray(v, dtype=np.float32).astype(np.float64)
    return (v / math.sqrt(float(v @ v))).astype(np.float32)


def look_at(pos, look, up):
    """Transform.LookAt (transform.dart:301-329): returns camera-to-world."""
    pos = np.asarray(pos, np.float32)
    look = np.asarray(look, np.float32)
    d = _normalize((look.astype(np.float64) - pos.astype(np.float64)).astype(np.float32))
    upn = _normalize(up)
    left = _normalize(np.cross(upn.astype(np.float64), d.astype(np.float64)).astype(np.float32))
    new_up = np.cross(d.astype(np.float64), left.astype(np.float64)).astype(np.float32)
    m = np.eye(4, dtype=np.float32)
    m[:3, 0] = left
    m[:3, 1] = new_up
    m[:3, 2] = d
    m[:3, 3] = pos
    return m


class PerspectiveCamera:
    """cameras/perspective_camera.dart:46-57 + core/projective_camera.dart:34-53."""

    def __init__(self, cam2world, screenWindow, sopen, sclose, lensr, focald, fov, film):
        self.cameraToWorld = _m4(cam2world)
        self.shutterOpen, self.shutterClose = float(sopen), float(sclose)
        self.lensRadius, self.focalDistance = float(lensr), float(focald)
        self.film = film
        znear, zfar = 1.0e-2, 1000.0
        persp = _m4([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, zfar / (zfar - znear), -zfar * znear / (zfar - znear)], [0, 0, 1, 0]])
        inv_tan = 1.0 / math.tan(((math.pi / 180.0) * fov) / 2.0)  # transform.dart:338-349
        scale = _m4(np.diag([inv_tan, inv_tan, 1.0, 1.0]))
        scale_inv = _m4(np.diag([1.0 / inv_tan, 1.0 / inv_tan, 1.0, 1.0]))
        c2s = _mul(scale, persp)
        c2s_inv = _mul(_inv(persp), scale_inv)
        sw = [float(s) for s in screenWindow]
        s1 = _m4(np.diag([float(film.xResolution), float(film.yResolution), 1.0, 1.0]))
        s1i = _m4(np.diag([1.0 / film.xResolution, 1.0 / film.yResolution, 1.0, 1.0]))
        s2 = _m4(np.diag([1.0 / (sw[1] - sw[0]), 1.0 / (sw[2] - sw[3]), 1.0, 1.0]))
        s2i = _m4(np.diag([1.0 / (1.0 / (sw[1] - sw[0])), 1.0 / (1.0 / (sw[2] - sw[3])), 1.0, 1.0]))
        tr = np.eye(4, dtype=np.float32)
        tr[0, 3], tr[1, 3] = np.float32(-sw[0]), np.float32(-sw[3])
        tri = np.eye(4, dtype=np.float32)
        tri[0, 3], tri[1, 3] = -tr[0, 3], -tr[1, 3]
        # screenToRaster = Scale * Scale * Translate; its inverse multiplies the inverses in reverse order
        r2s = _mul(tri, _mul(s2i, s1i))
        self.rasterToCamera = _mul(c2s_inv, r2s)
        self.cameraToScreen = c2s
        del s1, s2  # forward matrices are not needed on the path

    @staticmethod
    def lookAt(pos, look, up, fov, film, lensradius=0.0, focaldistance=1.0e30, shutteropen=0.0, shutterclose=1.0):
        """'LookAt' + Camera "perspective" defaults (perspective_camera.dart:134-183)."""
        frame = film.xResolution / film.yResolution
        if frame > 1.0:
            screen = [-frame, frame, -1.0, 1.0]
        else:
            screen = [-1.0, 1.0, -1.0 / frame, 1.0 / frame]
        return PerspectiveCamera(look_at(pos, look, up), screen, shutteropen, shutterclose, lensradius, focaldistance,
                                 fov, film)

    cameraType = 0  # DR_CAMERA_PERSPECTIVE

    def to_abi(self, c):
        c.raster_to_camera[:] = [float(v) for v in self.rasterToCamera.reshape(-1)]
        c.camera_to_world[:] = [float(v) for v in self.cameraToWorld.reshape(-1)]
        c.lens_radius, c.focal_distance = self.lensRadius, self.focalDistance
        c.shutter_open, c.shutter_close = self.shutterOpen, self.shutterClose
        c.type = self.cameraType


def _default_screen_window(film):
    frame = film.xResolution / film.yResolution   # perspective_camera.dart:152-168 (the same in all three cameras)
    return [-frame, frame, -1.0, 1.0] if frame > 1.0 else [-1.0, 1.0, -1.0 / frame, 1.0 / frame]


def _raster_to_screen(film, screenWindow):
    """Inverse of ProjectiveCamera's screenToRaster = Scale(xres, yres, 1) * Scale(1/(sw1-sw0), 1/(sw2-sw3), 1) *
    Translate(-sw0, -sw3, 0) (projective_camera.dart:39-52): the inverses multiplied in reverse order, every factor
    and product rounded to f32 like Matrix4x4."""
    sw = [float(s) for s in screenWindow]
    s1i = _m4(np.diag([1.0 / film.xResolution, 1.0 / film.yResolution, 1.0, 1.0]))
    s2i = _m4(np.diag([1.0 / (1.0 / (sw[1] - sw[0])), 1.0 / (1.0 / (sw[2] - sw[3])), 1.0, 1.0]))
    tri = np.eye(4, dtype=np.float32)
    tri[0, 3], tri[1, 3] = -np.float32(-sw[0]), -np.float32(-sw[3])
    return _mul(tri, _mul(s2i, s1i))


class OrthographicCamera(PerspectiveCamera):
    """cameras/orthographic_camera.dart:44-80: a ProjectiveCamera over Transform.Orthographic(0, 1)
    (transform.dart:333-336); rays leave the raster point along +z of camera space."""
    cameraType = 1  # DR_CAMERA_ORTHOGRAPHIC

    def __init__(self, cam2world, screenWindow, sopen, sclose, lensr, focald, film):
        self.cameraToWorld = _m4(cam2world)
        self.shutterOpen, self.shutterClose = float(sopen), float(sclose)
        self.lensRadius, self.focalDistance = float(lensr), float(focald)
        self.film = film
        znear, zfar = 0.0, 1.0
        tr = np.eye(4, dtype=np.float32)
        tr[2, 3] = np.float32(-znear)
        tri = np.eye(4, dtype=np.float32)
        tri[2, 3] = np.float32(znear)
        sc = _m4(np.diag([1.0, 1.0, 1.0 / (zfar - znear), 1.0]))
        sci = _m4(np.diag([1.0, 1.0, 1.0 / (1.0 / (zfar - znear)), 1.0]))
        self.cameraToScreen = _mul(sc, tr)
        self.rasterToCamera = _mul(_mul(tri, sci), _raster_to_screen(film, screenWindow))

    @staticmethod
    def lookAt(pos, look, up, film, lensradius=0.0, focaldistance=1.0e30, shutteropen=0.0, shutterclose=1.0, screenWindow=None):
        return OrthographicCamera(look_at(pos, look, up), screenWindow or _default_screen_window(film), shutteropen,
                                  shutterclose, lensradius, focaldistance, film)


class EnvironmentCamera(PerspectiveCamera):
    """cameras/environment_camera.dart:38-52: every raster point maps to a lat-long direction from the camera origin;
    no lens, no projection matrix (rasterToCamera is unused)."""
    cameraType = 2  # DR_CAMERA_ENVIRONMENT

    def __init__(self, cam2world, sopen, sclose, film):
        self.cameraToWorld = _m4(cam2world)
        self.shutterOpen, self.shutterClose = float(sopen), float(sclose)
        self.lensRadius, self.focalDistance = 0.0, 1.0e30
        self.film = film
        self.rasterToCamera = np.eye(4, dtype=np.float32)
        self.cameraToScreen = np.eye(4, dtype=np.float32)

    @staticmethod
    def lookAt(pos, look, up, film, shutteropen=0.0, shutterclose=1.0):
        return EnvironmentCamera(look_at(pos, look, up), shutteropen, shutterclose, film)


# ---------------------------------------------------------------------------
# samplers and integrators
# ---------------------------------------------------------------------------
def RoundUpPow2(v):  # common.dart:113-123
    v -= 1
    v |= v >> 1
    v |= v >> 2
    v |= v >> 4
    v |= v >> 8
    v |= v >> 16
    return v + 1


class DartRandom:
    """dart:math Random(seed) of the Dart VM as used by core/rng.dart:27-43 (multiply-with-carry, A = 0xffffda61,
    Thomas-Wang seeding, four warm-up steps) -- host side only: the pixel samplers below shuffle
    with their own RNG(5489)."""
    _M64 = (1 << 64) - 1

    def __init__(self, seed=5489):
        n = seed & self._M64
        n = ((~n) + (n << 21)) & self._M64
        n ^= n >> 24
        n = (n * 265) & self._M64
        n ^= n >> 14
        n = (n * 21) & self._M64
        n ^= n >> 28
        n = (n + (n << 31)) & self._M64
        n = n or 0x5A17
        self.lo, self.hi = n & 0xffffffff, n >> 32
        for _ in range(4):
            self._step()

    def _step(self):
        s = (0xffffda61 * self.lo + self.hi) & self._M64
        self.lo, self.hi = s & 0xffffffff, s >> 32

    def randomUint(self):  # Random.nextInt(0xffffffff): only lo == 0xffffffff is rejected
        while True:
            self._step()
            if self.lo != 0xffffffff:
                return self.lo


class LinearPixelSampler:
    """Pixels "linear" (pixel_samplers/linear_pixel_sampler.dart:29-40): rows top to bottom."""
    kind, tileSize, randomize = 0, 32, False

    def setup(self, x, y, width, height):
        ys, xs = np.meshgrid(np.arange(y, y + height, dtype=np.int32), np.arange(x, x + width, dtype=np.int32), indexing="ij")
        return np.stack([xs, ys], axis=-1).reshape(-1, 2)


class TilePixelSampler(LinearPixelSampler):
    """Pixels "tile" (tile_pixel_sampler.dart:33-100), the reference's default: tileSize^2 tiles in row-major order,
    shuffled (from tile 1 on, each with a uniformly drawn partner) by the sampler's own RNG(5489)."""
    kind = 1

    def __init__(self, tileSize=32, randomize=True):
        self.tileSize, self.randomize = int(tileSize), bool(randomize)

    def setup(self, x, y, width, height):
        ts = self.tileSize
        nx = width // ts + (0 if width % ts == 0 else 1)
        ny = height // ts + (0 if height % ts == 0 else 1)
        tiles = [(xi, yi) for yi in range(ny) for xi in range(nx)]
        if self.randomize:
            rng = DartRandom()
            for ti in range(1, len(tiles)):
                r = rng.randomUint() % len(tiles)
                tiles[ti], tiles[r] = tiles[r], tiles[ti]
        right, bottom = x + width - 1, y + height - 1
        out = []
        for tx, ty in tiles:
            sx, sy = x + tx * ts, y + ty * ts
            xs = np.arange(sx, min(sx + ts - 1, right) + 1, dtype=np.int32)
            ys = np.arange(sy, min(sy + ts - 1, bottom) + 1, dtype=np.int32)
            gy, gx = np.meshgrid(ys, xs, indexing="ij")
            out.append(np.stack([gx, gy], axis=-1).reshape(-1, 2))
        return np.concatenate(out) if out else np.zeros((0, 2), np.int32)


class RandomPixelSampler(LinearPixelSampler):
    """Pixels "random" (random_pixel_sampler.dart:27-58): the linear list, entry i swapped with a uniformly drawn one."""
    kind = 2

    def setup(self, x, y, width, height):
        p = LinearPixelSampler.setup(self, x, y, width, height).copy()
        rng = DartRandom()
        n = len(p)
        for i in range(n):
            l = rng.randomUint() % n
            p[[i, l]] = p[[l, i]]
        return p


class LowDiscrepancySampler:
    """samplers/low_discrepancy_sampler.dart:32-88.  The reference threads ONE
    serial RNG through sampler and integrator (sampler_renderer.dart:137); on
    the device every (pixel, LD block) owns a keyed stream instead
    (DR_SAMPLER_COUNTER), or the caller supplies recorded sample vectors
    (HostBufferSampler)."""

    def __init__(self, camera, nsamp=4, seed=5489, pixels=None):
        self.camera = camera
        self.samplesPerPixel = RoundUpPow2(int(nsamp))
        self.seed = int(seed)
        # PixelSampler: the ORDER in which pixels are sampled only matters to the serial reference stream (which RNG
        # numbers a pixel gets); the device's keyed streams give every pixel the same samples in any order
        self.pixelSampler = pixels or LinearPixelSampler()

    def roundSize(self, size):
        return RoundUpPow2(size)


class HostBufferSampler:
    """Explicit camera samples: pixel_xy [npix,2] int32, sample_vec [npix*spp, nfloats] f32
    (imageU, imageV, lensU, lensV, time, oneD..., twoD...), tail [npix*spp, max_tail] f64 =
    the RNG.randomFloat() values PathIntegrator.Li draws for bounces >= 3.  tail_count [npix*spp] (how many of its
    max_tail slots each sample actually drew: the oracle's recording has it) selects the PACKED form of the C ABI
    (DrRenderDesc.tail_offsets): only the drawn values cross the host link."""

    def __init__(self, camera, spp, pixel_xy, sample_vec, tail=None, tail_count=None):
        self.camera = camera
        self.samplesPerPixel = int(spp)
        self.pixel_xy = np.ascontiguousarray(pixel_xy, dtype=np.int32).reshape(-1, 2)
        self.sample_vec = np.ascontiguousarray(sample_vec, dtype=np.float32)
        self.tail = None if tail is None else np.ascontiguousarray(tail, dtype=np.float64)
        self.tail_offsets = None
        if len(self.sample_vec) != len(self.pixel_xy) * self.samplesPerPixel:
            raise ValueError("sample_vec must hold spp vectors per pixel")
        if tail_count is not None and self.tail is not None:
            cnt = np.minimum(np.asarray(tail_count, dtype=np.int64), self.tail.shape[1])
            if len(cnt) != len(self.sample_vec):
                raise ValueError("tail_count must hold one entry per sample")
            self.max_tail = int(self.tail.shape[1])
            self.tail_offsets = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
            self.tail = np.ascontiguousarray(self.tail[np.arange(self.tail.shape[1])[None, :] < cnt[:, None]])  # row-major: draw order per sample
            if len(self.tail) == 0:
                self.tail = np.zeros(1, np.float64)


class PathIntegrator:
    """surface_integrators/path_integrator.dart:26-27,133-136."""

    def __init__(self, maxDepth=5):
        self.maxDepth = int(maxDepth)

    kind = _abi.DR_INTEGRATOR_PATH


class DirectLightingIntegrator:
    """surface_integrators/direct_lighting_integrator.dart:23-28: strategy 'all' (UniformSampleAllLights, the default) or 'one'
    (UniformSampleOneLight with the integrator's own lightNum slot, :51-55,82-87)."""
    SAMPLE_ALL_UNIFORM = 0
    SAMPLE_ONE_UNIFORM = 1

    def __init__(self, strategy=0, maxDepth=5):
        if strategy not in (self.SAMPLE_ALL_UNIFORM, self.SAMPLE_ONE_UNIFORM):
            raise ValueError("DirectLightingIntegrator strategy must be SAMPLE_ALL_UNIFORM (0) or SAMPLE_ONE_UNIFORM (1)")
        self.strategy = strategy
        self.maxDepth = int(maxDepth)

    @property
    def kind(self):
        return _abi.DR_INTEGRATOR_DIRECT_ONE if self.strategy == self.SAMPLE_ONE_UNIFORM else _abi.DR_INTEGRATOR_DIRECT_ALL


class EmissionIntegrator:
    """volume_integrators/emission_integrator.dart with no VolumeRegion: T = 1,
    Lv = 0; its only effect on the path is the two 1-D sample slots it requests."""

    def __init__(self, stepSize=1.0):
        self.stepSize = stepSize


class OutputImage:
    """core/output_image.dart:35-55."""

    def __init__(self, xOffset, yOffset, width, height, rgb, film=None):
        self.xOffset, self.yOffset, self.width, self.height = xOffset, yOffset, width, height
        self.imageWidth, self.imageHeight = width, height
        self.rgb = rgb
        self.film = film  # (X, Y, Z, weightSum) per pixel: ImageFilm._Lxyz/_weightSum


class SamplerRenderer:
    """renderers/sampler_renderer.dart:28-31: Renderer.render(Scene) -> OutputImage."""

    def __init__(self, sampler, camera, surfaceIntegrator, volumeIntegrator=None, taskNum=0, taskCount=1,
                 tileRank=0, tileCount=1, tileSize=32):
        self.sampler = sampler
        self.camera = camera
        self.surfaceIntegrator = surfaceIntegrator
        self.volumeIntegrator = volumeIntegrator
        self.taskNum, self.taskCount = int(taskNum), int(taskCount)
        self.tileRank, self.tileCount, self.tileSize = int(tileRank), int(tileCount), int(tileSize)
        self.last_stats = None

    def describe(self):
        """DrRenderDesc for this renderer (plus the arrays it points into)."""
        d = _abi.DrRenderDesc()
        self.camera.to_abi(d.camera)
        self.camera.film.to_abi(d.film)
        d.integrator = self.surfaceIntegrator.kind
        d.max_depth = self.surfaceIntegrator.maxDepth
        d.spp = self.sampler.samplesPerPixel
        d.task_num, d.task_count = self.taskNum, self.taskCount
        d.tile_rank, d.tile_count, d.tile_size = self.tileRank, self.tileCount, self.tileSize
        keep = []
        if isinstance(self.sampler, HostBufferSampler):
            s = self.sampler
            d.sampler_mode = _abi.DR_SAMPLER_HOST_BUFFER
            d.nsamples = len(s.sample_vec)
            d.pixel_xy = s.pixel_xy.ctypes.data
            d.sample_vec = s.sample_vec.ctypes.data
            d.sample_stride = s.sample_vec.shape[1]
            if s.tail is not None:
                d.tail = s.tail.ctypes.data
                if s.tail_offsets is not None:
                    d.max_tail = s.max_tail
                    d.tail_offsets = s.tail_offsets.ctypes.data
                else:
                    d.max_tail = s.tail.shape[1]
            keep = [s.pixel_xy, s.sample_vec, s.tail, s.tail_offsets]
        else:
            d.sampler_mode = _abi.DR_SAMPLER_COUNTER
            d.seed = self.sampler.seed
        return d, keep

    def render(self, scene):
        film = self.camera.film
        d, keep = self.describe()
        out_film = np.zeros((film.height, film.width, 4), dtype=np.float32)
        out_rgb = np.zeros((film.height, film.width, 3), dtype=np.float32)
        dev = scene._device()
        dev.reset_stats()
        _abi.check(_abi.lib().dr_render(dev.handle, C.byref(d), out_film.ctypes.data, out_rgb.ctypes.data))
        del keep
        self.last_stats = dev.stats()
        return OutputImage(film.left, film.top, film.width, film.height, out_rgb, out_film)

    def render_sharded(self, scene, root=0):
        """One rank's part of a sharded render (dr_render_sharded): this renderer's tile / task share, ONE film reduce over
        the communicator of dr_comm_init, and on the root rank the OutputImage; other ranks return None.  What
        RenderManager's fan-out and rectangle merge do in the host (render_manager.dart:100-141), as one C call."""
        film = self.camera.film
        d, keep = self.describe()
        lib = _abi.lib()
        is_root = lib.dr_comm_world() <= 1 or lib.dr_comm_rank() == root
        out_film = np.zeros((film.height, film.width, 4), dtype=np.float32) if is_root else None
        out_rgb = np.zeros((film.height, film.width, 3), dtype=np.float32) if is_root else None
        dev = scene._device()
        _abi.check(lib.dr_render_sharded(dev.handle, C.byref(d), root, out_film.ctypes.data if is_root else None,
                                         out_rgb.ctypes.data if is_root else None))
        del keep
        return OutputImage(film.left, film.top, film.width, film.height, out_rgb, out_film) if is_root else None

    def pixels(self):
        """Raster pixels this renderer's task / tile share traces, in trace order (host-only)."""
        d, keep = self.describe()
        n = C.c_uint64(0)
        _abi.check(_abi.lib().dr_enumerate_pixels(C.byref(d), None, 0, C.byref(n)))
        out = np.zeros((n.value, 2), dtype=np.int32)
        _abi.check(_abi.lib().dr_enumerate_pixels(C.byref(d), out.ctypes.data, n.value, C.byref(n)))
        return out

    def render_device(self, scene, film_ptr, stream=0):
        """Renderer.render with the film left in HBM: accumulates this renderer's share into the
        [height, width, 4] f32 device buffer at `film_ptr` on HIP stream `stream` (asynchronous)."""
        d, keep = self.describe()
        dev = scene._device()
        _abi.check(_abi.lib().dr_render_device(dev.handle, C.byref(d), film_ptr, stream))
        self._keep = keep
        return dev

    def Li(self, *a, **k):  # a per-ray FFI seam is far too fine grained
        raise NotImplementedError("SamplerRenderer.Li is evaluated on the device inside render()")


# ---------------------------------------------------------------------------
# Plugin registry (lib/core/plugin.dart:23-180): names the reference registers
# in RegisterStandardPlugins (render_manager_interface.dart:37-157) for the path.
# ---------------------------------------------------------------------------
class Plugin:
    _reg = {"accelerator": {}, "surfaceIntegrator": {}, "renderer": {}, "sampler": {}, "film": {}, "filter": {},
            "camera": {}, "material": {}, "shape": {}, "areaLight": {}, "volumeIntegrator": {}, "pixelSampler": {}}

    @classmethod
    def register(cls, kind, name, creator):
        cls._reg[kind][name] = creator

    @classmethod
    def get(cls, kind, name):
        return cls._reg[kind].get(name)


def RegisterStandardPlugins():
    Plugin.register("accelerator", "bvh", BVHAccel.Create)
    Plugin.register("surfaceIntegrator", "path", lambda ps=None: PathIntegrator((ps or {}).get("maxdepth", 5)))
    Plugin.register("surfaceIntegrator", "directlighting",
                    lambda ps=None: DirectLightingIntegrator(1 if (ps or {}).get("strategy", "all") == "one" else 0, (ps or {}).get("maxdepth", 5)))
    Plugin.register("volumeIntegrator", "emission", lambda ps=None: EmissionIntegrator((ps or {}).get("stepsize", 1.0)))
    Plugin.register("renderer", "sampler", SamplerRenderer)
    Plugin.register("sampler", "lowdiscrepancy", LowDiscrepancySampler)
    Plugin.register("film", "image", ImageFilm)
    Plugin.register("pixelSampler", "linear", lambda ps=None: LinearPixelSampler())
    Plugin.register("pixelSampler", "tile", lambda ps=None: TilePixelSampler((ps or {}).get("tilesize", 32), (ps or {}).get("random", True)))
    Plugin.register("pixelSampler", "random", lambda ps=None: RandomPixelSampler())
    Plugin.register("filter", "box", lambda ps=None: BoxFilter((ps or {}).get("xwidth", 0.5), (ps or {}).get("ywidth", 0.5)))
    Plugin.register("filter", "gaussian", lambda ps=None: GaussianFilter((ps or {}).get("xwidth", 2.0), (ps or {}).get("ywidth", 2.0),
                                                                           (ps or {}).get("alpha", 2.0)))
    Plugin.register("filter", "sinc", lambda ps=None: LanczosSincFilter((ps or {}).get("xwidth", 4.0), (ps or {}).get("ywidth", 4.0),
                                                                          (ps or {}).get("tau", 3.0)))
    Plugin.register("filter", "mitchell", lambda ps=None: MitchellFilter((ps or {}).get("B", 1.0 / 3.0), (ps or {}).get("C", 1.0 / 3.0),
                                                                           (ps or {}).get("xwidth", 2.0), (ps or {}).get("ywidth", 2.0)))
    Plugin.register("filter", "triangle", lambda ps=None: TriangleFilter((ps or {}).get("xwidth", 2.0), (ps or {}).get("ywidth", 2.0)))
    Plugin.register("camera", "perspective", PerspectiveCamera)
    Plugin.register("camera", "orthographic", OrthographicCamera)
    Plugin.register("camera", "environment", EnvironmentCamera)
    Plugin.register("material", "matte", MatteMaterial)
    Plugin.register("shape", "trianglemesh", TriangleMesh)
    Plugin.register("areaLight", "diffuse", DiffuseAreaLight)


RegisterStandardPlugins()
