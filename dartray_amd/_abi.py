"""ctypes view of include/dartray_hip.h and the loader of libdartray_hip.so.

The product path never falls back to a CPU implementation: if the HIP library
is missing, loading raises and every operator fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DARTRAY_LIB") or os.path.join(_HERE, "libdartray_hip.so")  # DARTRAY_LIB: A/B builds

DR_OK = 0
DR_CAMERA_PERSPECTIVE, DR_CAMERA_ORTHOGRAPHIC, DR_CAMERA_ENVIRONMENT = 0, 1, 2
DR_INTEGRATOR_DIRECT_ALL = 0
DR_INTEGRATOR_PATH = 1
DR_INTEGRATOR_DIRECT_ONE = 2
DR_SAMPLER_HOST_BUFFER = 0
DR_SAMPLER_COUNTER = 1
DR_LIGHT_DIFFUSE_AREA = 0
DR_LIGHT_INFINITE = 1
DR_LIGHT_POINT = 2
DR_LIGHT_SPOT = 3
DR_LIGHT_DISTANT = 4
DR_LIGHT_SPOT_COS = 5


class DrBvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("offset", C.c_uint32),
                ("nprims", C.c_uint16), ("axis", C.c_uint8), ("pad", C.c_uint8)]


class DrMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("kd", C.c_float * 3), ("kr", C.c_float * 3), ("kt", C.c_float * 3),
                ("sigma", C.c_double), ("index", C.c_double)]


DR_MATERIAL_MATTE, DR_MATERIAL_MIRROR, DR_MATERIAL_GLASS, DR_MATERIAL_PLASTIC = 0, 1, 2, 3


class DrAreaLight(C.Structure):
    _fields_ = [("L", C.c_float * 3), ("nsamples", C.c_int32), ("first_tri", C.c_uint32), ("ntris", C.c_uint32),
                ("kind", C.c_uint32), ("env_index", C.c_uint32), ("position", C.c_float * 3), ("pad", C.c_float),
                ("world_to_light", C.c_float * 16), ("cone_width", C.c_double), ("cone_falloff_start", C.c_double)]


class DrEnvMap(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("light_to_world", C.c_float * 16), ("world_to_light", C.c_float * 16)]


class DrLightTri(C.Structure):
    _fields_ = [("v", C.c_uint32 * 3), ("reverse_orientation", C.c_uint32)]


DR_PRIM_QUADRIC = 0xFFFFFFFF
DR_QUADRIC_SPHERE, DR_QUADRIC_DISK = 1, 2


class DrQuadric(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pad", C.c_int32), ("object_to_world", C.c_float * 16),
                ("world_to_object", C.c_float * 16), ("params", C.c_double * 4)]


class DrMeshXform(C.Structure):
    _fields_ = [("object_to_world", C.c_float * 16), ("world_to_object", C.c_float * 16)]


DR_SHADING_N, DR_SHADING_S, DR_SHADING_UV = 1, 2, 4


class DrSceneDesc(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("nnodes", C.c_uint64),
                ("verts", C.c_void_p), ("nverts", C.c_uint64),
                ("tri_idx", C.c_void_p), ("ntris", C.c_uint64),
                ("tri_material", C.c_void_p), ("tri_light", C.c_void_p), ("tri_reverse", C.c_void_p),
                ("materials", C.c_void_p), ("nmaterials", C.c_uint32),
                ("lights", C.c_void_p), ("nlights", C.c_uint32),
                ("light_tris", C.c_void_p), ("nlight_tris", C.c_uint32),
                ("bvh_depth", C.c_uint32),
                ("env_maps", C.c_void_p), ("nenv_maps", C.c_uint32),
                ("quadrics", C.c_void_p), ("nquadrics", C.c_uint32),
                ("vert_normals", C.c_void_p), ("vert_tangents", C.c_void_p), ("vert_uvs", C.c_void_p),
                ("tri_shading", C.c_void_p), ("tri_xform", C.c_void_p), ("mesh_xforms", C.c_void_p),
                ("nmesh_xforms", C.c_uint32)]


class DrRay(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("tmin", C.c_double), ("tmax", C.c_double)]


class DrHit(C.Structure):
    _fields_ = [("prim", C.c_int32), ("pad", C.c_int32), ("t", C.c_double), ("b1", C.c_double), ("b2", C.c_double)]


class DrCamera(C.Structure):
    _fields_ = [("raster_to_camera", C.c_float * 16), ("camera_to_world", C.c_float * 16),
                ("lens_radius", C.c_double), ("focal_distance", C.c_double),
                ("shutter_open", C.c_double), ("shutter_close", C.c_double), ("type", C.c_int32), ("pad", C.c_int32)]


class DrFilm(C.Structure):
    _fields_ = [("xres", C.c_int32), ("yres", C.c_int32), ("crop", C.c_double * 4),
                ("filter_xw", C.c_double), ("filter_yw", C.c_double), ("filter_table", C.c_float * 256)]


class DrRenderDesc(C.Structure):
    _fields_ = [("camera", DrCamera), ("film", DrFilm),
                ("integrator", C.c_int32), ("max_depth", C.c_int32), ("spp", C.c_int32), ("sampler_mode", C.c_int32),
                ("seed", C.c_int64),
                ("task_num", C.c_int32), ("task_count", C.c_int32),
                ("tile_rank", C.c_int32), ("tile_count", C.c_int32), ("tile_size", C.c_int32),
                ("nsamples", C.c_int64),
                ("pixel_xy", C.c_void_p), ("sample_vec", C.c_void_p), ("sample_stride", C.c_int32),
                ("tail", C.c_void_p), ("max_tail", C.c_int32), ("tail_offsets", C.c_void_p)]


class DrRenderStats(C.Structure):
    _fields_ = [("camera_samples", C.c_uint64), ("film_samples", C.c_uint64),
                ("closest_rays", C.c_uint64), ("any_rays", C.c_uint64),
                ("closest_nodes", C.c_uint64), ("any_nodes", C.c_uint64),
                ("closest_tris", C.c_uint64), ("any_tris", C.c_uint64),
                ("trace_launches", C.c_uint64), ("trace_ms", C.c_double), ("total_ms", C.c_double),
                ("batches", C.c_uint64),
                ("closest_launches", C.c_uint64), ("any_launches", C.c_uint64),
                ("closest_ms", C.c_double), ("any_ms", C.c_double),
                ("shade_ms", C.c_double), ("gen_ms", C.c_double), ("film_ms", C.c_double),
                ("shade_items", C.c_uint64), ("shade_vertices", C.c_uint64), ("shade_cont", C.c_uint64),
                ("shade_mis", C.c_uint64), ("shade_shadow", C.c_uint64), ("pilot_ms", C.c_double)]


DR_COMM_ID_BYTES = 128
DR_ABI_VERSION = 7  # include/dartray_hip.h (tests/test_host_logic.py compares the two); lib() refuses a library of another version


# name -> (restype, argtypes): every symbol include/dartray_hip.h declares.
EXPORTS = {
    "dr_init": (C.c_int, [C.c_int]),
    "dr_bvh_build": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p,
                               C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint32)]),
    "dr_bvh_build_mixed": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int32,
                                     C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint32)]),
    "dr_bvh_build_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int32,
                                      C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint32)]),
    "dr_scene_create": (C.c_int, [C.POINTER(DrSceneDesc), C.POINTER(C.c_void_p)]),
    "dr_scene_destroy": (None, [C.c_void_p]),
    "dr_scene_get_trace_kernels": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32 * 2)]),
    "dr_scene_set_trace_kernels": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32 * 2)]),
    "dr_scene_get_pilot": (C.c_int, [C.c_void_p, C.POINTER(C.c_float * 6)]),
    "dr_scene_last_render_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32 * 8)]),
    "dr_scene_get_coherent_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double * 5)]),
    "dr_scene_get_sampler_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double * 2)]),
    "dr_intersect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]),
    "dr_sample_floats": (C.c_int32, [C.c_int32, C.c_uint32]),
    "dr_scene_sample_floats": (C.c_int32, [C.c_void_p, C.c_int32]),
    "dr_render": (C.c_int, [C.c_void_p, C.POINTER(DrRenderDesc), C.c_void_p, C.c_void_p]),
    "dr_render_device": (C.c_int, [C.c_void_p, C.POINTER(DrRenderDesc), C.c_void_p, C.c_void_p]),
    "dr_render_sharded": (C.c_int, [C.c_void_p, C.POINTER(DrRenderDesc), C.c_int32, C.c_void_p, C.c_void_p]),
    "dr_film_resolve_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "dr_enumerate_pixels": (C.c_int, [C.POINTER(DrRenderDesc), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "dr_get_stats": (C.c_int, [C.c_void_p, C.POINTER(DrRenderStats)]),
    "dr_reset_stats": (C.c_int, [C.c_void_p]),
    "dr_copy_bandwidth": (C.c_int, [C.c_uint64, C.c_int32, C.POINTER(C.c_double)]),
    "dr_scene_get_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "dr_scene_get_state_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "dr_scene_set_state_layout": (C.c_int, [C.c_void_p, C.c_int32]),
    "dr_comm_available": (C.c_int, []),
    "dr_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "dr_comm_unique_id": (C.c_int, [C.c_void_p, C.c_uint64]),
    "dr_comm_init": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_uint64]),
    "dr_film_reduce": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dr_comm_allreduce_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dr_comm_rank": (C.c_int, []),
    "dr_comm_world": (C.c_int, []),
    "dr_comm_destroy": (C.c_int, []),
    "dr_last_error": (C.c_char_p, []),
    "dr_version": (C.c_char_p, []),
    "dr_abi_version": (C.c_int32, []),
    "dr_scene_workspace_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
}

_lib = None


class DartRayHipError(RuntimeError):
    """Raised for every non-zero return code of the C ABI (the Dart shim maps
    it to LogSevere -> Exception, lib/core/log.dart:42-47)."""


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so /
    libhsa-runtime64.so; if this library pulled in /opt/rocm's copy first, torch's later
    initialisation would find "No HIP GPUs" (two HSA runtimes cannot share the device).  torch is
    this package's plumbing for device memory, streams and RCCL, so when it is installed its HIP
    runtime is loaded first (by path, without importing torch) and libdartray_hip.so's
    DT_NEEDED libamdhip64.so.7 then resolves to that same object."""
    import importlib.util
    override = os.environ.get("DARTRAY_HIP_RUNTIME")
    if override:
        C.CDLL(override, mode=C.RTLD_GLOBAL)
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _check_build_info():
    """The shipped .so must be the one built from the sources next to it: __graft_entry__.build() records the sha256
    of every source and header in libdartray_hip.buildinfo.json; a mismatch (a stale library) fails loudly."""
    if os.environ.get("DARTRAY_LIB"):
        return  # an explicitly chosen A/B build
    import hashlib
    import json
    info_path = os.path.join(_HERE, "libdartray_hip.buildinfo.json")
    csrc = os.path.join(_HERE, "csrc")
    if not os.path.isdir(csrc):
        return  # a binary-only deployment
    if not os.path.exists(info_path):
        raise DartRayHipError("libdartray_hip.buildinfo.json missing: rebuild with __graft_entry__.build()")
    want = json.load(open(info_path)).get("sources", {})
    for rel, digest in want.items():
        path = os.path.join(csrc, rel)
        if not os.path.exists(path) or hashlib.sha256(open(path, "rb").read()).hexdigest() != digest:
            raise DartRayHipError("libdartray_hip.so is stale: %s changed since it was built (run __graft_entry__.build())" % rel)


def lib():
    """Load libdartray_hip.so (built in-tree by __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DartRayHipError(
                "HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
        _check_build_info()
        _share_hip_runtime_with_torch()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.dr_abi_version() != DR_ABI_VERSION:
            raise DartRayHipError("libdartray_hip.so has ABI version %d, this binding was written against %d" % (l.dr_abi_version(), DR_ABI_VERSION))
        _lib = l
    return _lib


def check(rc):
    if rc != DR_OK:
        raise DartRayHipError("dartray_hip error %d: %s" % (rc, lib().dr_last_error().decode()))


_initialised = None


def init(device=0):
    global _initialised
    if _initialised != device:
        check(lib().dr_init(device))
        _initialised = device
