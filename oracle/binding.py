"""ctypes binding of the CPU oracle (oracle/dartray_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  Nothing under dartray_amd/ may import this module.
PARITY UNPINNED: see the header of dartray_oracle.cpp.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libdartray_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "dartray_oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


class OrcMesh(C.Structure):
    _fields_ = [("P", C.c_void_p), ("idx", C.c_void_p), ("nverts", C.c_int32), ("ntris", C.c_int32),
                ("Kd", C.c_float * 3), ("sigma", C.c_float), ("reverse_orientation", C.c_int32),
                ("has_light", C.c_int32), ("L", C.c_float * 3), ("light_nsamples", C.c_int32),
                ("kind", C.c_int32), ("o2w", C.c_float * 16), ("w2o", C.c_float * 16), ("params", C.c_double * 4),
                ("mat_type", C.c_int32), ("Kr", C.c_float * 3), ("Kt", C.c_float * 3), ("ior", C.c_double),
                ("sigma_d", C.c_double), ("N", C.c_void_p), ("S", C.c_void_p), ("uv", C.c_void_p)]


class OrcSceneDesc(C.Structure):
    _fields_ = [("nmeshes", C.c_int32), ("meshes", C.POINTER(OrcMesh)), ("max_prims_in_node", C.c_int32),
                ("has_env", C.c_int32), ("env_texels", C.c_void_p), ("env_w", C.c_int32), ("env_h", C.c_int32),
                ("env_L", C.c_float * 3), ("env_l2w", C.c_float * 16), ("env_w2l", C.c_float * 16),
                ("env_nsamples", C.c_int32), ("env_before_mesh", C.c_int32),
                ("npoint_lights", C.c_int32), ("point_pos", C.c_void_p), ("point_intensity", C.c_void_p),
                ("point_before_mesh", C.c_void_p), ("point_kind", C.c_void_p), ("spot_w2l", C.c_void_p),
                ("spot_angles", C.c_void_p)]


class OrcRenderDesc(C.Structure):
    _fields_ = [("xres", C.c_int32), ("yres", C.c_int32), ("crop", C.c_double * 4),
                ("filter_xw", C.c_double), ("filter_yw", C.c_double), ("filter_table", C.c_float * 256),
                ("raster_to_camera", C.c_float * 16), ("camera_to_world", C.c_float * 16),
                ("lens_radius", C.c_double), ("focal_distance", C.c_double),
                ("shutter_open", C.c_double), ("shutter_close", C.c_double), ("camera_type", C.c_int32),
                ("integrator", C.c_int32), ("max_depth", C.c_int32), ("spp", C.c_int32), ("sampler_mode", C.c_int32),
                ("seed", C.c_int64), ("task_num", C.c_int32), ("task_count", C.c_int32),
                ("npixels", C.c_int32), ("pixels", C.c_void_p),
                ("pixel_order", C.c_int32), ("tile_size", C.c_int32), ("tile_random", C.c_int32), ("pad_po", C.c_int32)]


class OrcRecord(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("count", C.c_int64), ("nfloats", C.c_int32), ("max_tail", C.c_int32),
                ("pixel_xy", C.c_void_p), ("sample_vec", C.c_void_p), ("nfloats_cap", C.c_int32),
                ("tail", C.c_void_p), ("tail_count", C.c_void_p), ("Ls", C.c_void_p)]


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris",
                                          "any_tris", "light_tris", "camera_samples", "max_stack")]


RAY_DTYPE = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("tmin", "<f8"), ("tmax", "<f8")])
HIT_DTYPE = np.dtype([("prim", "<i4"), ("pad", "<i4"), ("t", "<f8"), ("b1", "<f8"), ("b2", "<f8")])
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<u4"), ("nprims", "<u2"),
                       ("axis", "u1"), ("pad", "u1")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(LIB_PATH)
        l.orc_scene_create.restype = C.c_void_p
        l.orc_scene_create.argtypes = [C.POINTER(OrcSceneDesc)]
        l.orc_scene_destroy.argtypes = [C.c_void_p]
        l.orc_scene_info.argtypes = [C.c_void_p, C.c_void_p]
        l.orc_scene_get_bvh.argtypes = [C.c_void_p] * 5
        l.orc_scene_get_verts.argtypes = [C.c_void_p, C.c_void_p]
        l.orc_counters.argtypes = [C.c_void_p, C.POINTER(OrcCounters), C.c_int]
        l.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        l.orc_intersect_brute.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        l.orc_sample_floats.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.orc_order_study.argtypes = [C.c_int, C.c_void_p]
        l.orc_resample_pow2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.orc_render.argtypes = [C.c_void_p, C.POINTER(OrcRenderDesc), C.c_void_p, C.c_void_p, C.POINTER(OrcRecord)]
        l.orc_li_samples.argtypes = [C.c_void_p, C.POINTER(OrcRenderDesc), C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        l.orc_film_accumulate.argtypes = [C.POINTER(OrcRenderDesc), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.orc_film_resolve.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        l.orc_camera_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        l.orc_van_der_corput.restype = C.c_double
        l.orc_van_der_corput.argtypes = [C.c_uint32, C.c_uint32]
        l.orc_sobol2.restype = C.c_double
        l.orc_sobol2.argtypes = [C.c_uint32, C.c_uint32]
        l.orc_concentric_sample_disk.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        l.orc_cosine_sample_hemisphere.argtypes = [C.c_double, C.c_double, C.c_void_p]
        l.orc_power_heuristic.restype = C.c_double
        l.orc_power_heuristic.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double]
        l.orc_get_sub_window.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        l.orc_pixel_order.argtypes = [C.c_int] * 7 + [C.c_void_p]
        l.orc_camera_setup_ortho.argtypes = [C.c_int, C.c_int, C.c_void_p]
        l.orc_generate_ray.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
        l.orc_filter_evaluate.restype = C.c_double
        l.orc_filter_evaluate.argtypes = [C.c_int] + [C.c_double] * 6
        l.orc_filter_table.argtypes = [C.c_int] + [C.c_double] * 4 + [C.c_void_p]
        l.orc_distribution1d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_int, C.c_void_p]
        l.orc_dart_random.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        l.orc_counter_key.restype = C.c_int64
        l.orc_counter_key.argtypes = [C.c_uint64] * 4
        l.orc_ld_pixel_sample.argtypes = [C.c_int, C.c_int64, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        l.orc_triangle_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        l.orc_triangle_intersectP.argtypes = [C.c_void_p, C.c_void_p]
        l.orc_slab.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        l.orc_env_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        l.orc_version.restype = C.c_char_p
        _lib = l
    return _lib


def make_rays(o, d, tmin=0.0, tmax=np.inf):
    o = np.asarray(o, np.float32).reshape(-1, 3)
    r = np.zeros(len(o), dtype=RAY_DTYPE)
    r["o"] = o
    r["d"] = np.asarray(d, np.float32).reshape(-1, 3)
    r["tmin"] = tmin
    r["tmax"] = tmax
    return r


def order_study(enable=None):
    """Measurement only (oracle: OrderStudy): enable True / False starts (and resets) / stops counting; returns the counts so far."""
    out = (C.c_ulonglong * 17)()
    lib().orc_order_study(-1 if enable is None else int(bool(enable)), out)
    keys = ("rays", "occluded", "nodes_all", "tris_all", "nodes_occ_ref", "tris_occ_ref", "nodes_occ_far_first", "tris_occ_far_first",
            "nodes_occ_larger_area_first", "tris_occ_larger_area_first", "ideal_nodes_occ", "nodes_occ_longer_interval_first",
            "tris_occ_longer_interval_first", "nodes_occ_leaf_then_far_first", "tris_occ_leaf_then_far_first", "nodes_occ_smaller_area_first",
            "tris_occ_smaller_area_first")
    return dict(zip(keys, (int(v) for v in out)))


def resample_pow2(texels):
    """MIPMap.texture's resampling of an [H, W, 3] f32 image to power-of-two size (mipmap.dart:71-138), as the oracle restates it."""
    t = np.ascontiguousarray(texels, np.float32)
    h, w = t.shape[:2]
    p2 = lambda v: 1 << (int(v) - 1).bit_length()
    out = np.zeros((p2(h), p2(w), 3), np.float32)
    ow, oh = C.c_int(0), C.c_int(0)
    if lib().orc_resample_pow2(t.ctypes.data, w, h, out.ctypes.data, C.byref(ow), C.byref(oh)) != 0:
        raise RuntimeError("orc_resample_pow2 failed")
    assert (oh.value, ow.value) == out.shape[:2]
    return out


class OracleScene:
    """Scene built by the oracle from the same GeometricPrimitive list the product takes
    (objects with .shape.{P,vertexIndex,reverseOrientation}, .material.{Kd,sigma}, .areaLight)."""

    def __init__(self, prims, max_prims=4, env=None, env_before=None, points=()):
        """env: optional InfiniteAreaLight-like object (.texels [H,W,3] f32, .L, .lightToWorld, .worldToLight, .nSamples);
        env_before: index of the primitive whose area light follows it in Scene.lights (None: the env light is last);
        points: [(PointLight-like with .lightPos / .intensity, index of the primitive whose area light follows it or None)]."""
        l = lib()
        meshes = (OrcMesh * max(len(prims), 1))()
        self._keep = []
        for i, gp in enumerate(prims):
            m = meshes[i]
            if getattr(gp.shape, "kind", 0):  # Sphere (1) / Disk (2): .objectToWorld, .worldToObject, .params
                m.kind = gp.shape.kind
                m.o2w[:] = [float(x) for x in np.asarray(gp.shape.objectToWorld, np.float32).reshape(-1)]
                m.w2o[:] = [float(x) for x in np.asarray(gp.shape.worldToObject, np.float32).reshape(-1)]
                m.params[:] = [float(x) for x in gp.shape.params]
            else:
                P = np.ascontiguousarray(gp.shape.P, np.float32)
                idx = np.ascontiguousarray(gp.shape.vertexIndex, np.uint32)
                self._keep += [P, idx]
                m.P, m.idx = P.ctypes.data, idx.ctypes.data
                m.nverts, m.ntris = len(P), len(idx)
                for field, attr in (("N", "n"), ("S", "s"), ("uv", "uvs")):  # optional per-vertex shading data
                    a = getattr(gp.shape, attr, None)
                    if a is not None:
                        a = np.ascontiguousarray(a, np.float32)
                        self._keep.append(a)
                        setattr(m, field, a.ctypes.data)
                if getattr(gp.shape, "n", None) is not None or getattr(gp.shape, "s", None) is not None:
                    m.o2w[:] = [float(x) for x in np.asarray(gp.shape.objectToWorld, np.float32).reshape(-1)]
                    m.w2o[:] = [float(x) for x in np.asarray(gp.shape.worldToObject, np.float32).reshape(-1)]
            mt = getattr(gp.material, "kind", 0)  # 0 matte, 1 mirror, 2 glass
            m.mat_type = mt
            if mt == 0:
                m.Kd[:] = [float(x) for x in gp.material.Kd]
                m.sigma_d = float(gp.material.sigma)
            elif mt == 3:  # plastic: Ks travels in Kr, roughness in ior
                m.Kd[:] = [float(x) for x in gp.material.Kd]
                m.Kr[:] = [float(x) for x in gp.material.Ks]
                m.ior = float(gp.material.roughness)
            else:
                m.Kr[:] = [float(x) for x in gp.material.Kr]
                if mt == 2:
                    m.Kt[:] = [float(x) for x in gp.material.Kt]
                    m.ior = float(gp.material.index)
            m.reverse_orientation = 1 if gp.shape.reverseOrientation else 0
            if gp.areaLight is not None:
                m.has_light = 1
                m.L[:] = [float(x) for x in gp.areaLight.Lemit]
                m.light_nsamples = gp.areaLight.nSamples
        d = OrcSceneDesc(len(prims), meshes, max_prims)
        d.env_before_mesh = -1 if env_before is None else int(env_before)
        if points:
            pos = np.ascontiguousarray([p.lightPos for p, _ in points], np.float32)
            inten = np.ascontiguousarray([p.intensity for p, _ in points], np.float32)
            before = np.ascontiguousarray([-1 if b is None else int(b) for _, b in points], np.int32)
            # 2 point, 3 spot (has .worldToLight, .width, .fall), 4 distant (has .lightDir; carried in lightPos)
            kind = np.ascontiguousarray([3 if hasattr(p, "width") else (4 if hasattr(p, "lightDir") else 2) for p, _ in points], np.int32)
            w2l = np.ascontiguousarray([np.asarray(getattr(p, "worldToLight", np.eye(4)), np.float32).reshape(-1) for p, _ in points], np.float32)
            ang = np.ascontiguousarray([(getattr(p, "width", 0.0), getattr(p, "fall", 0.0)) for p, _ in points], np.float64)
            self._keep += [pos, inten, before, kind, w2l, ang]
            d.npoint_lights = len(points)
            d.point_pos, d.point_intensity, d.point_before_mesh = pos.ctypes.data, inten.ctypes.data, before.ctypes.data
            d.point_kind, d.spot_w2l, d.spot_angles = kind.ctypes.data, w2l.ctypes.data, ang.ctypes.data
        if env is not None:
            tex = np.ascontiguousarray(env.texels, np.float32)
            self._keep.append(tex)
            d.has_env = 1
            d.env_texels = tex.ctypes.data
            d.env_h, d.env_w = tex.shape[0], tex.shape[1]
            d.env_L[:] = [float(x) for x in env.L]
            d.env_l2w[:] = [float(x) for x in np.asarray(env.lightToWorld, np.float32).reshape(-1)]
            d.env_w2l[:] = [float(x) for x in np.asarray(env.worldToLight, np.float32).reshape(-1)]
            d.env_nsamples = env.nSamples
        self.h = l.orc_scene_create(C.byref(d))
        if not self.h:
            raise RuntimeError("orc_scene_create failed (an empty environment map, or a sphere used as an area light)")
        info = (C.c_int64 * 6)()
        l.orc_scene_info(self.h, info)
        self.nnodes, self.nprims, self.depth, self.nlights, self.nverts, self.nlighttris = [int(v) for v in info]

    def __del__(self):
        try:
            if self.h:
                lib().orc_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def bvh(self):
        nodes = np.zeros(self.nnodes, dtype=NODE_DTYPE)
        tri = np.zeros((self.nprims, 3), dtype=np.uint32)
        mesh = np.zeros(self.nprims, dtype=np.int32)
        src = np.zeros(self.nprims, dtype=np.int32)
        lib().orc_scene_get_bvh(self.h, nodes.ctypes.data, tri.ctypes.data, mesh.ctypes.data, src.ctypes.data)
        return nodes, tri, mesh, src

    def verts(self):
        P = np.zeros((self.nverts, 3), np.float32)
        lib().orc_scene_get_verts(self.h, P.ctypes.data)
        return P

    def counters(self, reset=False):
        c = OrcCounters()
        lib().orc_counters(self.h, C.byref(c), 1 if reset else 0)
        return {n: getattr(c, n) for n, _ in OrcCounters._fields_}

    def intersect(self, rays, any_hit=False, brute=False):
        rays = np.ascontiguousarray(rays)
        out = np.zeros(len(rays), dtype=HIT_DTYPE)
        fn = lib().orc_intersect_brute if brute else lib().orc_intersect
        fn(self.h, rays.ctypes.data, len(rays), out.ctypes.data, 1 if any_hit else 0)
        return out

    def env_probe(self, what, a, b=0.0, c=0.0):
        """what: 0 = Le(dir) -> rgb, 1 = pdf(dir), 2 = sampleLAtPoint(u0, u1) -> (wi, pdf, Ls)."""
        i = np.array([a, b, c], np.float64)
        o = np.zeros(8, np.float64)
        if lib().orc_env_probe(self.h, what, i.ctypes.data, o.ctypes.data) != 0:
            raise RuntimeError("scene has no infinite light")
        return o

    def sample_floats(self, integrator, max_depth):
        return lib().orc_sample_floats(self.h, integrator, max_depth)

    def render(self, rd, record=0, max_tail=40, want_film=True):
        """SamplerRenderer.render.  Returns dict(rgb, film, and -- if record > 0 -- the per-sample
        recording: pixel_xy, sample_vec, tail, tail_count, Ls)."""
        l = lib()
        # ImageFilm window
        import math
        left = math.ceil(rd.xres * rd.crop[0]); width = max(1, math.ceil(rd.xres * rd.crop[1]) - left)
        top = math.ceil(rd.yres * rd.crop[2]); height = max(1, math.ceil(rd.yres * rd.crop[3]) - top)
        rgb = np.zeros((height, width, 3), np.float32)
        film = np.zeros((height, width, 4), np.float32)
        rec = None
        out = {}
        if record > 0:
            nf = self.sample_floats(rd.integrator, rd.max_depth)
            rec = OrcRecord()
            rec.capacity = record
            rec.max_tail = max_tail
            rec.nfloats_cap = nf
            out["pixel_xy"] = np.zeros((record, 2), np.int32)
            out["sample_vec"] = np.zeros((record, nf), np.float32)
            out["tail"] = np.zeros((record, max_tail), np.float64)
            out["tail_count"] = np.zeros(record, np.int32)
            out["Ls"] = np.zeros((record, 3), np.float32)
            rec.pixel_xy = out["pixel_xy"].ctypes.data
            rec.sample_vec = out["sample_vec"].ctypes.data
            rec.tail = out["tail"].ctypes.data
            rec.tail_count = out["tail_count"].ctypes.data
            rec.Ls = out["Ls"].ctypes.data
        rc = l.orc_render(self.h, C.byref(rd), rgb.ctypes.data, film.ctypes.data if want_film else None,
                          C.byref(rec) if rec is not None else None)
        if rc != 0:
            raise RuntimeError("orc_render failed: %d" % rc)
        out["rgb"], out["film"] = rgb, film
        if rec is not None:
            n = rec.count
            out["count"] = n
            for k in ("pixel_xy", "sample_vec", "tail", "tail_count", "Ls"):
                out[k] = out[k][:n]
        return out

    def li_samples(self, rd, pixel_xy, sample_vec, tail=None, tail_count=None):
        pixel_xy = np.ascontiguousarray(pixel_xy, np.int32)
        sample_vec = np.ascontiguousarray(sample_vec, np.float32)
        n = len(sample_vec)
        out = np.zeros((n, 3), np.float32)
        if tail is not None:
            tail = np.ascontiguousarray(tail, np.float64)
        if tail_count is not None:
            tail_count = np.ascontiguousarray(tail_count, np.int32)
        rc = lib().orc_li_samples(self.h, C.byref(rd), n, pixel_xy.ctypes.data, sample_vec.ctypes.data,
                                  sample_vec.shape[1], tail.ctypes.data if tail is not None else None,
                                  tail_count.ctypes.data if tail_count is not None else None,
                                  tail.shape[1] if tail is not None else 0, out.ctypes.data)
        if rc < 0:
            raise RuntimeError("orc_li_samples failed: %d" % rc)
        return out


def render_desc(renderer, sampler_mode=None, pixels=None):
    """OrcRenderDesc from a dartray_amd.core.SamplerRenderer-like object (duck typed: no import of
    the product package happens here)."""
    cam, film = renderer.camera, renderer.camera.film
    rd = OrcRenderDesc()
    rd.xres, rd.yres = film.xResolution, film.yResolution
    rd.crop[:] = film.cropWindow
    rd.filter_xw, rd.filter_yw = film.filter.xWidth, film.filter.yWidth
    rd.filter_table[:] = [float(v) for v in film.filterTable]
    rd.raster_to_camera[:] = [float(v) for v in cam.rasterToCamera.reshape(-1)]
    rd.camera_to_world[:] = [float(v) for v in cam.cameraToWorld.reshape(-1)]
    rd.lens_radius, rd.focal_distance = cam.lensRadius, cam.focalDistance
    rd.camera_type = getattr(cam, "cameraType", 0)
    rd.shutter_open, rd.shutter_close = cam.shutterOpen, cam.shutterClose
    rd.integrator = renderer.surfaceIntegrator.kind
    rd.max_depth = renderer.surfaceIntegrator.maxDepth
    rd.spp = renderer.sampler.samplesPerPixel
    rd.sampler_mode = 1 if sampler_mode is None else sampler_mode
    rd.seed = getattr(renderer.sampler, "seed", 0)
    rd.task_num, rd.task_count = renderer.taskNum, renderer.taskCount
    ps = getattr(renderer.sampler, "pixelSampler", None)
    if ps is not None:
        rd.pixel_order, rd.tile_size, rd.tile_random = ps.kind, ps.tileSize, int(ps.randomize)
    if pixels is not None:
        pixels = np.ascontiguousarray(pixels, np.int32).reshape(-1, 2)
        rd._pixels_keep = pixels
        rd.npixels = len(pixels)
        rd.pixels = pixels.ctypes.data
    return rd
