/****************************************************************************
 * hip_sampler_renderer.dart -- DartRay's Renderer seam on an MI355X.
 *
 * Drop this file into lib/renderers/ of brendan-duncan/dartray and add the
 * `hipsampler` branch of INTEGRATION.md section 2 to DartRay._makeRenderer
 * (lib/dartray/dartray.dart:662-761).  `HipSamplerRenderer.render(Scene)`
 * (Renderer.render, lib/core/renderer.dart:27-35) flattens the Scene the
 * Dart API built -- BVHAccel.nodes / .primitives (bvh_accel.dart:486-487,
 * :533-538), every GeometricPrimitive's shape / material / areaLight
 * (geometric_primitive.dart:83-85), the world-space vertices of its
 * TriangleMesh (triangle_mesh.dart:39-42), each DiffuseAreaLight's ShapeSet
 * in ShapeSet order (shape_set.dart:27-35), scene.lights in list order --
 * into the POD structs of include/dartray_hip.h and renders it with
 * libdartray_hip.so through dart:ffi.  The OutputImage it returns
 * (output_image.dart:35-55) is what SamplerRenderer.render returns.
 *
 * Structs are written field by field into native memory at the byte offsets
 * the header itself asserts (DR_ABI_SIZE / DR_ABI_OFFSET at the end of
 * dartray_hip.h); tests/test_abi_c_host.py checks every offset constant of
 * this file against those asserts.  No Dart SDK exists in the image this was
 * written in: the file has been checked against the header mechanically,
 * not executed.  tests/abi/c1_from_c.c is the same call sequence from C and
 * IS executed on the GPU.
 *
 * SDK window: the reference is pre-null-safety Dart (no `environment` in its pubspec.yaml) and dart:ffi with
 * package:ffi's `calloc` / `Utf8.toDartString` needs Dart >= 2.12: Dart 2.12 ... 2.19 with `// @dart=2.9`-style
 * unsound null safety (`dart --no-sound-null-safety`), package:ffi ^1.0.0.  Dart 3 rejects the reference itself.
 *
 * Scope: what the device path covers (SURVEY.md section 8): BVHAccel over
 * TriangleMesh / Sphere / Disk shapes, matte / mirror / glass / plastic
 * materials with constant textures, DiffuseAreaLight / InfiniteAreaLight /
 * point, spot and distant lights, PathIntegrator and DirectLightingIntegrator
 * ("all"), LowDiscrepancySampler, ImageFilm with any Filter, the three
 * cameras.  Anything else is reported through LogSevere (log.dart:42-47),
 * never approximated.
 ****************************************************************************/
library hip_sampler_renderer;

import 'dart:async';
import 'dart:ffi';
import 'dart:typed_data';

import 'package:ffi/ffi.dart';

import '../accelerators/accelerators.dart';
import '../cameras/cameras.dart';
import '../core/core.dart';
import '../film/film.dart';
import '../lights/lights.dart';
import '../materials/materials.dart';
import '../samplers/samplers.dart';
import '../shapes/shapes.dart';
import '../surface_integrators/surface_integrators.dart';

// ---- sizes and byte offsets of include/dartray_hip.h (DR_ABI_SIZE / DR_ABI_OFFSET) ----
const int SIZEOF_DrBvhNode = 32;
const int OFF_DrBvhNode_bmin = 0;
const int OFF_DrBvhNode_bmax = 12;
const int OFF_DrBvhNode_offset = 24;
const int OFF_DrBvhNode_nprims = 28;
const int OFF_DrBvhNode_axis = 30;
const int SIZEOF_DrMaterial = 56;
const int OFF_DrMaterial_type = 0;
const int OFF_DrMaterial_kd = 4;
const int OFF_DrMaterial_kr = 16;
const int OFF_DrMaterial_kt = 28;
const int OFF_DrMaterial_sigma = 40;
const int OFF_DrMaterial_index = 48;
const int SIZEOF_DrAreaLight = 128;
const int OFF_DrAreaLight_L = 0;
const int OFF_DrAreaLight_nsamples = 12;
const int OFF_DrAreaLight_first_tri = 16;
const int OFF_DrAreaLight_ntris = 20;
const int OFF_DrAreaLight_kind = 24;
const int OFF_DrAreaLight_env_index = 28;
const int OFF_DrAreaLight_position = 32;
const int OFF_DrAreaLight_world_to_light = 48;
const int OFF_DrAreaLight_cone_width = 112;
const int OFF_DrAreaLight_cone_falloff_start = 120;
const int SIZEOF_DrEnvMap = 144;
const int OFF_DrEnvMap_texels = 0;
const int OFF_DrEnvMap_width = 8;
const int OFF_DrEnvMap_height = 12;
const int OFF_DrEnvMap_light_to_world = 16;
const int OFF_DrEnvMap_world_to_light = 80;
const int SIZEOF_DrLightTri = 16;
const int OFF_DrLightTri_v = 0;
const int OFF_DrLightTri_reverse_orientation = 12;
const int SIZEOF_DrMeshXform = 128;
const int OFF_DrMeshXform_object_to_world = 0;
const int OFF_DrMeshXform_world_to_object = 64;
const int SIZEOF_DrQuadric = 168;
const int OFF_DrQuadric_kind = 0;
const int OFF_DrQuadric_object_to_world = 8;
const int OFF_DrQuadric_world_to_object = 72;
const int OFF_DrQuadric_params = 136;
const int SIZEOF_DrSceneDesc = 208;
const int OFF_DrSceneDesc_nodes = 0;
const int OFF_DrSceneDesc_nnodes = 8;
const int OFF_DrSceneDesc_verts = 16;
const int OFF_DrSceneDesc_nverts = 24;
const int OFF_DrSceneDesc_tri_idx = 32;
const int OFF_DrSceneDesc_ntris = 40;
const int OFF_DrSceneDesc_tri_material = 48;
const int OFF_DrSceneDesc_tri_light = 56;
const int OFF_DrSceneDesc_tri_reverse = 64;
const int OFF_DrSceneDesc_materials = 72;
const int OFF_DrSceneDesc_nmaterials = 80;
const int OFF_DrSceneDesc_lights = 88;
const int OFF_DrSceneDesc_nlights = 96;
const int OFF_DrSceneDesc_light_tris = 104;
const int OFF_DrSceneDesc_nlight_tris = 112;
const int OFF_DrSceneDesc_bvh_depth = 116;
const int OFF_DrSceneDesc_env_maps = 120;
const int OFF_DrSceneDesc_nenv_maps = 128;
const int OFF_DrSceneDesc_quadrics = 136;
const int OFF_DrSceneDesc_nquadrics = 144;
const int OFF_DrSceneDesc_vert_normals = 152;
const int OFF_DrSceneDesc_vert_tangents = 160;
const int OFF_DrSceneDesc_vert_uvs = 168;
const int OFF_DrSceneDesc_tri_shading = 176;
const int OFF_DrSceneDesc_tri_xform = 184;
const int OFF_DrSceneDesc_mesh_xforms = 192;
const int OFF_DrSceneDesc_nmesh_xforms = 200;
const int SIZEOF_DrCamera = 168;
const int OFF_DrCamera_raster_to_camera = 0;
const int OFF_DrCamera_camera_to_world = 64;
const int OFF_DrCamera_lens_radius = 128;
const int OFF_DrCamera_focal_distance = 136;
const int OFF_DrCamera_shutter_open = 144;
const int OFF_DrCamera_shutter_close = 152;
const int OFF_DrCamera_type = 160;
const int SIZEOF_DrFilm = 1080;
const int OFF_DrFilm_xres = 0;
const int OFF_DrFilm_yres = 4;
const int OFF_DrFilm_crop = 8;
const int OFF_DrFilm_filter_xw = 40;
const int OFF_DrFilm_filter_yw = 48;
const int OFF_DrFilm_filter_table = 56;
const int SIZEOF_DrRenderDesc = 1352;
const int OFF_DrRenderDesc_camera = 0;
const int OFF_DrRenderDesc_film = 168;
const int OFF_DrRenderDesc_integrator = 1248;
const int OFF_DrRenderDesc_max_depth = 1252;
const int OFF_DrRenderDesc_spp = 1256;
const int OFF_DrRenderDesc_sampler_mode = 1260;
const int OFF_DrRenderDesc_seed = 1264;
const int OFF_DrRenderDesc_task_num = 1272;
const int OFF_DrRenderDesc_task_count = 1276;
const int OFF_DrRenderDesc_tile_rank = 1280;
const int OFF_DrRenderDesc_tile_count = 1284;
const int OFF_DrRenderDesc_tile_size = 1288;
const int OFF_DrRenderDesc_nsamples = 1296;
const int OFF_DrRenderDesc_pixel_xy = 1304;
const int OFF_DrRenderDesc_sample_vec = 1312;
const int OFF_DrRenderDesc_sample_stride = 1320;
const int OFF_DrRenderDesc_tail = 1328;
const int OFF_DrRenderDesc_max_tail = 1336;
const int OFF_DrRenderDesc_tail_offsets = 1344;  // host-buffer sampler only: stays NULL here (calloc'd descriptor)

// enums of the header
const int DR_MATERIAL_MATTE = 0, DR_MATERIAL_MIRROR = 1, DR_MATERIAL_GLASS = 2, DR_MATERIAL_PLASTIC = 3;
const int DR_LIGHT_DIFFUSE_AREA = 0, DR_LIGHT_INFINITE = 1, DR_LIGHT_POINT = 2, DR_LIGHT_SPOT = 3, DR_LIGHT_DISTANT = 4;
const int DR_LIGHT_SPOT_COS = 5;
const int DR_SHADING_N = 1, DR_SHADING_S = 2, DR_SHADING_UV = 4;
const int DR_COMM_ID_BYTES = 128;
const int DR_PRIM_QUADRIC = 0xFFFFFFFF, DR_QUADRIC_SPHERE = 1, DR_QUADRIC_DISK = 2;
const int DR_CAMERA_PERSPECTIVE = 0, DR_CAMERA_ORTHOGRAPHIC = 1, DR_CAMERA_ENVIRONMENT = 2;
const int DR_INTEGRATOR_DIRECT_ALL = 0, DR_INTEGRATOR_PATH = 1, DR_INTEGRATOR_DIRECT_ONE = 2;
const int DR_SAMPLER_COUNTER = 1;

typedef _InitC = Int32 Function(Int32);
typedef _InitD = int Function(int);
typedef _SceneCreateC = Int32 Function(Pointer<Uint8>, Pointer<Pointer<Void>>);
typedef _SceneCreateD = int Function(Pointer<Uint8>, Pointer<Pointer<Void>>);
typedef _RenderC = Int32 Function(Pointer<Void>, Pointer<Uint8>, Pointer<Float>, Pointer<Float>);
typedef _RenderD = int Function(Pointer<Void>, Pointer<Uint8>, Pointer<Float>, Pointer<Float>);
typedef _RenderShardedC = Int32 Function(Pointer<Void>, Pointer<Uint8>, Int32, Pointer<Float>, Pointer<Float>);
typedef _RenderShardedD = int Function(Pointer<Void>, Pointer<Uint8>, int, Pointer<Float>, Pointer<Float>);
typedef _RenderDeviceC = Int32 Function(Pointer<Void>, Pointer<Uint8>, Pointer<Void>, Pointer<Void>);
typedef _RenderDeviceD = int Function(Pointer<Void>, Pointer<Uint8>, Pointer<Void>, Pointer<Void>);
typedef _FilmReduceC = Int32 Function(Pointer<Void>, Int64, Int32, Pointer<Void>);
typedef _FilmReduceD = int Function(Pointer<Void>, int, int, Pointer<Void>);
typedef _FilmResolveC = Int32 Function(Pointer<Void>, Int64, Pointer<Void>, Pointer<Void>);
typedef _FilmResolveD = int Function(Pointer<Void>, int, Pointer<Void>, Pointer<Void>);
typedef _CommIdC = Int32 Function(Pointer<Uint8>, Uint64);
typedef _CommIdD = int Function(Pointer<Uint8>, int);
typedef _CommInitC = Int32 Function(Int32, Int32, Pointer<Uint8>, Uint64);
typedef _CommInitD = int Function(int, int, Pointer<Uint8>, int);
typedef _CommVoidC = Int32 Function();
typedef _SetOptionC = Int32 Function(Pointer<Utf8>, Pointer<Utf8>);
typedef _SetOptionD = int Function(Pointer<Utf8>, Pointer<Utf8>);
typedef _SetLayoutC = Int32 Function(Pointer<Void>, Int32);
typedef _SetLayoutD = int Function(Pointer<Void>, int);
typedef _CommVoidD = int Function();
typedef _KernelsC = Int32 Function(Pointer<Void>, Pointer<Uint32>);
typedef _KernelsD = int Function(Pointer<Void>, Pointer<Uint32>);
typedef _GetLayoutC = Int32 Function(Pointer<Void>, Pointer<Int32>, Pointer<Float>);
typedef _GetLayoutD = int Function(Pointer<Void>, Pointer<Int32>, Pointer<Float>);
typedef _DestroyC = Void Function(Pointer<Void>);
typedef _DestroyD = void Function(Pointer<Void>);
typedef _ErrC = Pointer<Utf8> Function();
typedef _ErrD = Pointer<Utf8> Function();

/// A zero-initialised block of native memory with typed little-endian writers: one per C struct (array).
class _Blob {
  final int length;
  final Pointer<Uint8> ptr;
  ByteData _view;

  _Blob(int bytes)
      : length = bytes < 1 ? 1 : bytes,
        ptr = calloc<Uint8>(bytes < 1 ? 1 : bytes) {
    _view = ByteData.view(ptr.asTypedList(length).buffer, ptr.asTypedList(length).offsetInBytes, length);
  }

  void i32(int off, int v) => _view.setInt32(off, v, Endian.little);
  void u32(int off, int v) => _view.setUint32(off, v, Endian.little);
  void u16(int off, int v) => _view.setUint16(off, v, Endian.little);
  void u8(int off, int v) => _view.setUint8(off, v);
  void i64(int off, int v) => _view.setInt64(off, v, Endian.little);
  void u64(int off, int v) => _view.setUint64(off, v, Endian.little);
  void f32(int off, double v) => _view.setFloat32(off, v, Endian.little);
  void f64(int off, double v) => _view.setFloat64(off, v, Endian.little);
  void addr(int off, _Blob b) => _view.setUint64(off, b == null ? 0 : b.ptr.address, Endian.little);
  void f32s(int off, List<double> v) {
    for (int i = 0; i < v.length; ++i) {
      f32(off + 4 * i, v[i]);
    }
  }

  void free() => calloc.free(ptr);
}

class HipSamplerRenderer extends Renderer {
  static final DynamicLibrary _lib = _open();
  static final _InitD _init = _lib.lookupFunction<_InitC, _InitD>('dr_init');
  /// DR_ABI_VERSION of the include/dartray_hip.h these offsets were written against: the structs carry no size field, so a
  /// library of another layout version is refused before any struct crosses the boundary.
  static const int ABI_VERSION = 7;
  static DynamicLibrary _open() {
    DynamicLibrary l = DynamicLibrary.open('libdartray_hip.so');
    _CommVoidD abi = l.lookupFunction<_CommVoidC, _CommVoidD>('dr_abi_version');
    if (abi() != ABI_VERSION) {
      LogSevere('libdartray_hip.so has ABI version ${abi()}, this binding was written against $ABI_VERSION');
    }
    return l;
  }
  static final _SceneCreateD _sceneCreate = _lib.lookupFunction<_SceneCreateC, _SceneCreateD>('dr_scene_create');
  static final _RenderD _render = _lib.lookupFunction<_RenderC, _RenderD>('dr_render');
  static final _DestroyD _destroy = _lib.lookupFunction<_DestroyC, _DestroyD>('dr_scene_destroy');
  // the multi-GPU half (include/dartray_hip.h): one process per GPU, ONE film reduce per render
  static final _RenderShardedD _renderSharded = _lib.lookupFunction<_RenderShardedC, _RenderShardedD>('dr_render_sharded');
  static final _CommIdD _commUniqueId = _lib.lookupFunction<_CommIdC, _CommIdD>('dr_comm_unique_id');
  static final _CommInitD _commInit = _lib.lookupFunction<_CommInitC, _CommInitD>('dr_comm_init');
  static final _CommVoidD _commDestroy = _lib.lookupFunction<_CommVoidC, _CommVoidD>('dr_comm_destroy');
  static final _CommVoidD _commWorld = _lib.lookupFunction<_CommVoidC, _CommVoidD>('dr_comm_world');
  static final _CommVoidD _commRank = _lib.lookupFunction<_CommVoidC, _CommVoidD>('dr_comm_rank');
  static final _CommVoidD _commAvailable = _lib.lookupFunction<_CommVoidC, _CommVoidD>('dr_comm_available');
  static final _SetOptionD _setOption = _lib.lookupFunction<_SetOptionC, _SetOptionD>('dr_set_option');
  static final _SetLayoutD _setStateLayout = _lib.lookupFunction<_SetLayoutC, _SetLayoutD>('dr_scene_set_state_layout');
  static final _GetLayoutD _getStateLayout = _lib.lookupFunction<_GetLayoutC, _GetLayoutD>('dr_scene_get_state_layout');
  static final _KernelsD _getTraceKernels = _lib.lookupFunction<_KernelsC, _KernelsD>('dr_scene_get_trace_kernels');
  static final _KernelsD _setTraceKernels = _lib.lookupFunction<_KernelsC, _KernelsD>('dr_scene_set_trace_kernels');

  /// What the library's pilot picked for a scene: [closest-hit kernel, any-hit kernel, state layout] (include/dartray_hip.h:
  /// dr_scene_get_trace_kernels / dr_scene_get_state_layout).  A render of a big scene measures them on its first batches; a host
  /// that renders the same scene again (animation frames: this renderer builds a device scene per render, like the reference's
  /// isolates load the scene per task) or a manager that knows rank 0's picks hands them in through [pilotPicks] and the render
  /// runs no calibration batches -- and, on N GPUs, every rank runs the same kernels (a step is the slowest rank's: the Python
  /// host's dartray_amd/dist.py share_pilot does the same over its control plane).  Every kernel and layout is bit-exact: the
  /// image does not depend on them.  [lastPilotPicks] is what the last render of this renderer ended up with.
  List<int> pilotPicks;
  List<int> lastPilotPicks;

  void _applyPilotPicks(Pointer<Void> scene) {
    if (pilotPicks == null || pilotPicks.length != 3) {
      return;
    }
    if (pilotPicks[0] != 0 && pilotPicks[1] != 0) {
      Pointer<Uint32> k = calloc<Uint32>(2);
      try {
        k[0] = pilotPicks[0];
        k[1] = pilotPicks[1];
        _check(_setTraceKernels(scene, k));
      } finally {
        calloc.free(k);
      }
    }
    if (pilotPicks[2] != 0) {
      _check(_setStateLayout(scene, pilotPicks[2]));
    }
  }

  void _readPilotPicks(Pointer<Void> scene) {
    Pointer<Uint32> k = calloc<Uint32>(2);
    Pointer<Int32> lay = calloc<Int32>(1);
    try {
      _check(_getTraceKernels(scene, k));
      _check(_getStateLayout(scene, lay, nullptr));
      lastPilotPicks = [k[0], k[1], lay[0]];
    } finally {
      calloc.free(k);
      calloc.free(lay);
    }
  }

  /// Can this process join a multi-GPU render?  Binds librccl and checks its version WITHOUT talking to another rank, so
  /// that the workers can agree on it before anyone blocks in [commInit] (hip_render_manager.dart).
  static bool commAvailable(int device) {
    _checkStatic(_init(device));
    return _commAvailable() == 0;
  }

  /// A tuning / diagnostic switch of the library (the DARTRAY_* names of include/dartray_hip.h, without setenv): read
  /// at every use, so it applies from the next render on; value null returns the switch to the environment's value.
  static void setOption(String name, String value) {
    Pointer<Utf8> n = name.toNativeUtf8();
    Pointer<Utf8> v = value == null ? nullptr : value.toNativeUtf8();
    try {
      _checkStatic(_setOption(n, v));
    } finally {
      calloc.free(n);
      if (v != nullptr) calloc.free(v);
    }
  }
  // for hosts that keep the film on the device themselves (device pointers come from their own HIP binding)
  static final _RenderDeviceD renderDevice = _lib.lookupFunction<_RenderDeviceC, _RenderDeviceD>('dr_render_device');
  static final _FilmReduceD filmReduce = _lib.lookupFunction<_FilmReduceC, _FilmReduceD>('dr_film_reduce');
  static final _FilmResolveD filmResolveDevice = _lib.lookupFunction<_FilmResolveC, _FilmResolveD>('dr_film_resolve_device');

  /// Rank 0 of a multi-GPU render: the 128 opaque bytes every other rank needs for [commInit] (carried by the host:
  /// a file, a pipe, a socket -- hip_render_manager.dart uses a file).
  static Uint8List commUniqueId(int device) {
    _checkStatic(_init(device));
    Pointer<Uint8> id = calloc<Uint8>(DR_COMM_ID_BYTES);
    try {
      _checkStatic(_commUniqueId(id, DR_COMM_ID_BYTES));
      return new Uint8List.fromList(id.asTypedList(DR_COMM_ID_BYTES));
    } finally {
      calloc.free(id);
    }
  }

  /// Every rank, once, after dr_init(device): joins the RCCL communicator of `world` ranks.
  static void commInit(int device, int rank, int world, Uint8List id) {
    _checkStatic(_init(device));
    Pointer<Uint8> p = calloc<Uint8>(DR_COMM_ID_BYTES);
    try {
      p.asTypedList(DR_COMM_ID_BYTES).setAll(0, id);
      _checkStatic(_commInit(rank, world, p, DR_COMM_ID_BYTES));
    } finally {
      calloc.free(p);
    }
  }

  static void commDestroy() => _checkStatic(_commDestroy());

  static void _checkStatic(int rc) {
    if (rc != 0) {
      LogSevere('dartray_hip error $rc: ${_err().toDartString()}');
    }
  }
  static final _ErrD _err = _lib.lookupFunction<_ErrC, _ErrD>('dr_last_error');

  /// tileRank / tileCount: this process's share of the image's 32 x 32 tiles (round-robin) when a render is spread
  /// over several GPUs (one process each, see hip_render_manager.dart); the default is the whole image.
  HipSamplerRenderer(this.sampler, this.camera, this.surfaceIntegrator, this.volumeIntegrator,
                     [this.taskNum = 0, this.taskCount = 1, this.device = 0, this.seed = 5489,
                      this.tileRank = 0, this.tileCount = 1]);

  /// Every failure of the library becomes a LogSevere, i.e. an Exception (log.dart:42-47), which
  /// DartRay.worldEnd turns into completeError (dartray.dart:573-583).
  void _check(int rc) {
    if (rc != 0) {
      LogSevere('dartray_hip error $rc: ${_err().toDartString()}');
    }
  }

  static void _unsupported(String what) {
    LogSevere('HipSamplerRenderer: $what is outside the device path');
  }

  static List<double> _rgb(Spectrum s) {
    RGBColor c = s.toRGB();
    return [c.c[0], c.c[1], c.c[2]];
  }

  /// Kd / Kr / Kt / sigma / index are Textures; the device path takes constants (evaluate() of a constant texture
  /// ignores the DifferentialGeometry).
  static final DifferentialGeometry _dg0 = new DifferentialGeometry();

  Future<OutputImage> render(Scene scene) {
    Completer<OutputImage> completer = new Completer<OutputImage>();
    List<_Blob> blobs = [];
    Pointer<Pointer<Void>> handle = calloc<Pointer<Void>>();
    Pointer<Float> lxyzw = nullptr;
    Pointer<Float> rgb = nullptr;
    try {
      _check(_init(device));
      if (scene.aggregate is! BVHAccel) {
        _unsupported('aggregate ${scene.aggregate.runtimeType} (only "bvh")');
      }
      if (scene.volumeRegion != null) {
        _unsupported('a VolumeRegion');
      }
      BVHAccel bvh = scene.aggregate;

      // ---- BVHAccel.nodes: _LinearBVHNode { bounds, offset, nPrimitives, axis } (bvh_accel.dart:533-538) ----
      final int nnodes = bvh.nodes == null ? 0 : bvh.nodes.length;
      _Blob nodes = new _Blob(nnodes * SIZEOF_DrBvhNode);
      blobs.add(nodes);
      for (int i = 0; i < nnodes; ++i) {
        var n = bvh.nodes[i];
        int o = i * SIZEOF_DrBvhNode;
        nodes.f32s(o + OFF_DrBvhNode_bmin, [n.bounds.pMin.x, n.bounds.pMin.y, n.bounds.pMin.z]);
        nodes.f32s(o + OFF_DrBvhNode_bmax, [n.bounds.pMax.x, n.bounds.pMax.y, n.bounds.pMax.z]);
        nodes.u32(o + OFF_DrBvhNode_offset, n.offset);
        nodes.u16(o + OFF_DrBvhNode_nprims, n.nPrimitives);
        nodes.u8(o + OFF_DrBvhNode_axis, n.nPrimitives > 0 ? 0 : n.axis);
      }

      // ---- BVHAccel.primitives, in BVH order: shapes, materials, area lights ----
      final int nprims = bvh.primitives.length;
      Map<TriangleMesh, int> meshBase = {};      // first vertex of each mesh in the shared vertex array
      List<TriangleMesh> meshes = [];
      int nverts = 0;
      List<Shape> quadrics = [];
      Map<Shape, int> quadricIndex = {};
      List<Material> materials = [];
      Map<Material, int> materialIndex = {};
      Map<Light, int> lightIndex = {};
      for (int i = 0; i < scene.lights.length; ++i) {
        lightIndex[scene.lights[i]] = i;
      }
      int baseOf(TriangleMesh m) {
        if (!meshBase.containsKey(m)) {
          meshBase[m] = nverts;
          meshes.add(m);
          nverts += m.nverts;
        }
        return meshBase[m];
      }
      int quadricOf(Shape s) {
        if (!quadricIndex.containsKey(s)) {
          quadricIndex[s] = quadrics.length;
          quadrics.add(s);
        }
        return quadricIndex[s];
      }

      _Blob triIdx = new _Blob(nprims * 12);
      _Blob triMaterial = new _Blob(nprims * 4);
      _Blob triLight = new _Blob(nprims * 4);
      _Blob triReverse = new _Blob(nprims);
      _Blob triShading = new _Blob(nprims);
      _Blob triXform = new _Blob(nprims * 4);
      bool anyShading = false;
      List<TriangleMesh> xformMeshes = [];
      Map<TriangleMesh, int> xformIndex = {};
      blobs.addAll([triIdx, triMaterial, triLight, triReverse, triShading, triXform]);
      for (int i = 0; i < nprims; ++i) {
        if (bvh.primitives[i] is! GeometricPrimitive) {
          _unsupported('primitive ${bvh.primitives[i].runtimeType} (instancing)');
        }
        GeometricPrimitive gp = bvh.primitives[i];
        Shape sh = gp.shape;
        if (sh is Triangle) {
          if (sh.mesh.alphaTexture != null) {
            _unsupported('an alpha texture');
          }
          int b = baseOf(sh.mesh);
          // per-vertex N / S / uv (triangle_mesh.dart:195-203): which attributes the primitive's mesh has, and the
          // mesh transform that takes N / S to world space at shading time (triangle.dart:303-317)
          int bits = (sh.mesh.n != null ? DR_SHADING_N : 0) | (sh.mesh.s != null ? DR_SHADING_S : 0) |
                     (sh.mesh.uvs != null ? DR_SHADING_UV : 0);
          triShading.u8(i, bits);
          if (bits != 0) {
            anyShading = true;
          }
          if ((bits & (DR_SHADING_N | DR_SHADING_S)) != 0) {
            if (!xformIndex.containsKey(sh.mesh)) {
              xformIndex[sh.mesh] = xformMeshes.length;
              xformMeshes.add(sh.mesh);
            }
            triXform.u32(4 * i, xformIndex[sh.mesh]);
          }
          for (int k = 0; k < 3; ++k) {
            triIdx.u32(12 * i + 4 * k, b + sh.mesh.vertexIndex[sh.index + k]);   // Triangle.v(k) (triangle.dart:241)
          }
        } else if (sh is Sphere || sh is Disk) {
          triIdx.u32(12 * i, DR_PRIM_QUADRIC);
          triIdx.u32(12 * i + 4, quadricOf(sh));
        } else {
          _unsupported('shape ${sh.runtimeType}');
        }
        if (!materialIndex.containsKey(gp.material)) {
          materialIndex[gp.material] = materials.length;
          materials.add(gp.material);
        }
        triMaterial.u32(4 * i, materialIndex[gp.material]);
        triLight.i32(4 * i, gp.areaLight == null ? -1 : lightIndex[gp.areaLight]);
        triReverse.u8(i, sh.reverseOrientation ? 1 : 0);
      }

      // world-space vertices: TriangleMesh.point(i) (triangle_mesh.dart:39-42)
      _Blob verts = new _Blob(nverts * 12);
      blobs.add(verts);
      for (TriangleMesh m in meshes) {
        int b = meshBase[m];
        for (int i = 0; i < m.nverts; ++i) {
          Point p = m.point(i);
          verts.f32s(12 * (b + i), [p.x, p.y, p.z]);
        }
      }

      // per-vertex normals / tangents (OBJECT space: TriangleMesh keeps them as given) and uvs, indexed like verts
      _Blob vertN, vertS, vertUV, meshXforms;
      if (anyShading) {
        vertN = new _Blob(nverts * 12);
        vertS = new _Blob(nverts * 12);
        vertUV = new _Blob(nverts * 8);
        meshXforms = new _Blob(xformMeshes.length * SIZEOF_DrMeshXform);
        blobs.addAll([vertN, vertS, vertUV, meshXforms]);
        for (TriangleMesh m in meshes) {
          int b = meshBase[m];
          for (int i = 0; i < m.nverts; ++i) {
            if (m.n != null) {
              vertN.f32s(12 * (b + i), [m.n[i].x, m.n[i].y, m.n[i].z]);
            }
            if (m.s != null) {
              vertS.f32s(12 * (b + i), [m.s[i].x, m.s[i].y, m.s[i].z]);
            }
            if (m.uvs != null) {
              vertUV.f32s(8 * (b + i), [m.uvs[2 * i], m.uvs[2 * i + 1]]);
            }
          }
        }
        for (int i = 0; i < xformMeshes.length; ++i) {
          meshXforms.f32s(i * SIZEOF_DrMeshXform + OFF_DrMeshXform_object_to_world, xformMeshes[i].objectToWorld.m.data);
          meshXforms.f32s(i * SIZEOF_DrMeshXform + OFF_DrMeshXform_world_to_object, xformMeshes[i].worldToObject.m.data);
        }
      }

      // quadrics: constructor arguments as the Dart doubles they are (sphere.dart:313-321, disk.dart:157-165)
      _Blob quads = new _Blob(quadrics.length * SIZEOF_DrQuadric);
      blobs.add(quads);
      for (int i = 0; i < quadrics.length; ++i) {
        Shape s = quadrics[i];
        int o = i * SIZEOF_DrQuadric;
        quads.f32s(o + OFF_DrQuadric_object_to_world, s.objectToWorld.m.data);
        quads.f32s(o + OFF_DrQuadric_world_to_object, s.worldToObject.m.data);
        if (s is Sphere) {
          quads.i32(o + OFF_DrQuadric_kind, DR_QUADRIC_SPHERE);
          // (phiMax is stored in radians, sphere.dart:313-321: Degrees(Radians(x)) is x up to an ulp, exact for 360)
          List<double> p = [s.radius, s.zmin, s.zmax, Degrees(s.phiMax)];
          for (int k = 0; k < 4; ++k) {
            quads.f64(o + OFF_DrQuadric_params + 8 * k, p[k]);
          }
        } else if (s is Disk) {
          quads.i32(o + OFF_DrQuadric_kind, DR_QUADRIC_DISK);
          List<double> p = [s.height, s.radius, s.innerRadius, Degrees(s.phiMax)];
          for (int k = 0; k < 4; ++k) {
            quads.f64(o + OFF_DrQuadric_params + 8 * k, p[k]);
          }
        }
      }

      // ---- materials with constant textures ----
      _Blob mats = new _Blob(materials.length * SIZEOF_DrMaterial);
      blobs.add(mats);
      for (int i = 0; i < materials.length; ++i) {
        Material m = materials[i];
        int o = i * SIZEOF_DrMaterial;
        if (m is MatteMaterial) {                         // matte_material.dart:41-65
          if (m.bumpMap != null) {
            _unsupported('a bump map');
          }
          mats.i32(o + OFF_DrMaterial_type, DR_MATERIAL_MATTE);
          mats.f32s(o + OFF_DrMaterial_kd, _rgb(m.Kd.evaluate(_dg0)));
          mats.f64(o + OFF_DrMaterial_sigma, m.sigma.evaluate(_dg0));
        } else if (m is MirrorMaterial) {                 // mirror_material.dart:38-55
          mats.i32(o + OFF_DrMaterial_type, DR_MATERIAL_MIRROR);
          mats.f32s(o + OFF_DrMaterial_kr, _rgb(m.Kr.evaluate(_dg0)));
        } else if (m is GlassMaterial) {                  // glass_material.dart:44-69
          mats.i32(o + OFF_DrMaterial_type, DR_MATERIAL_GLASS);
          mats.f32s(o + OFF_DrMaterial_kr, _rgb(m.Kr.evaluate(_dg0)));
          mats.f32s(o + OFF_DrMaterial_kt, _rgb(m.Kt.evaluate(_dg0)));
          mats.f64(o + OFF_DrMaterial_index, m.index.evaluate(_dg0));
        } else if (m is PlasticMaterial) {                // plastic_material.dart:43-70
          mats.i32(o + OFF_DrMaterial_type, DR_MATERIAL_PLASTIC);
          mats.f32s(o + OFF_DrMaterial_kd, _rgb(m.Kd.evaluate(_dg0)));
          mats.f32s(o + OFF_DrMaterial_kr, _rgb(m.Ks.evaluate(_dg0)));
          mats.f64(o + OFF_DrMaterial_index, m.roughness.evaluate(_dg0));
        } else {
          _unsupported('material ${m.runtimeType}');
        }
      }

      // ---- scene.lights, in list order (the order lightNum indexes) ----
      final int nlights = scene.lights.length;
      _Blob lights = new _Blob(nlights * SIZEOF_DrAreaLight);
      blobs.add(lights);
      List<List<int>> lightTris = [];   // (v0, v1, v2, reverse) or (DR_PRIM_QUADRIC, quadric, 0, reverse)
      _Blob envMaps;
      _Blob envTexels;
      for (int i = 0; i < nlights; ++i) {
        Light l = scene.lights[i];
        int o = i * SIZEOF_DrAreaLight;
        if (l is DiffuseAreaLight) {                      // diffuse_area_light.dart:36-43
          lights.u32(o + OFF_DrAreaLight_kind, DR_LIGHT_DIFFUSE_AREA);
          lights.f32s(o + OFF_DrAreaLight_L, _rgb(l.Lemit));
          lights.i32(o + OFF_DrAreaLight_nsamples, l.nSamples);
          lights.u32(o + OFF_DrAreaLight_first_tri, lightTris.length);
          for (Shape s in l.shapeSet.shapes) {            // ShapeSet order: the LIFO refine order (shape_set.dart:27-35)
            if (s is Triangle) {
              int b = baseOf(s.mesh);                     // (an emitter's mesh is also a primitive's mesh: already placed)
              // bit 1: the mesh has uvs -- they decide dpdu x dpdv, i.e. the side the emitter shines from
              lightTris.add([b + s.mesh.vertexIndex[s.index], b + s.mesh.vertexIndex[s.index + 1],
                             b + s.mesh.vertexIndex[s.index + 2],
                             (s.reverseOrientation ? 1 : 0) | (s.mesh.uvs != null ? 2 : 0)]);
            } else if (s is Sphere || s is Disk) {
              lightTris.add([DR_PRIM_QUADRIC, quadricOf(s), 0, s.reverseOrientation ? 1 : 0]);
            } else {
              _unsupported('emissive shape ${s.runtimeType}');
            }
          }
          lights.u32(o + OFF_DrAreaLight_ntris, lightTris.length - lights._view.getUint32(o + OFF_DrAreaLight_first_tri, Endian.little));
        } else if (l is InfiniteAreaLight) {              // infinite_area_light.dart
          if (envMaps != null) {
            _unsupported('more than one infinite light');
          }
          lights.u32(o + OFF_DrAreaLight_kind, DR_LIGHT_INFINITE);
          lights.f32s(o + OFF_DrAreaLight_L, _rgb(l.L));
          lights.i32(o + OFF_DrAreaLight_nsamples, l.nSamples);
          lights.u32(o + OFF_DrAreaLight_env_index, 0);
          // level 0 of the radiance MIPMap (mipmap.dart:139): width x height RGB texels
          final int w = l.radianceMap.width, h = l.radianceMap.height;
          envTexels = new _Blob(w * h * 12);
          for (int t = 0; t < h; ++t) {
            for (int s = 0; s < w; ++s) {
              envTexels.f32s(12 * (t * w + s), _rgb(l.radianceMap.texel(0, s, t)));
            }
          }
          envMaps = new _Blob(SIZEOF_DrEnvMap);
          envMaps.addr(OFF_DrEnvMap_texels, envTexels);
          envMaps.i32(OFF_DrEnvMap_width, w);
          envMaps.i32(OFF_DrEnvMap_height, h);
          envMaps.f32s(OFF_DrEnvMap_light_to_world, l.lightToWorld.m.data);
          envMaps.f32s(OFF_DrEnvMap_world_to_light, l.worldToLight.m.data);
          blobs.addAll([envTexels, envMaps]);
        } else if (l is SpotLight) {
          // A constructed SpotLight only keeps the two cosines (spot_light.dart:46-47); DR_LIGHT_SPOT_COS takes
          // exactly those doubles (DR_LIGHT_SPOT takes the constructor's degrees).  SpotLight extends Light, not
          // PointLight: this branch precedes the PointLight one only for readability.
          lights.u32(o + OFF_DrAreaLight_kind, DR_LIGHT_SPOT_COS);
          lights.f32s(o + OFF_DrAreaLight_L, _rgb(l.intensity));
          lights.i32(o + OFF_DrAreaLight_nsamples, 1);
          lights.f32s(o + OFF_DrAreaLight_position, [l.lightPos.x, l.lightPos.y, l.lightPos.z]);
          lights.f32s(o + OFF_DrAreaLight_world_to_light, l.worldToLight.m.data);
          lights.f64(o + OFF_DrAreaLight_cone_width, l.cosTotalWidth);
          lights.f64(o + OFF_DrAreaLight_cone_falloff_start, l.cosFalloffStart);
        } else if (l is PointLight) {                     // point_light.dart:36-39
          lights.u32(o + OFF_DrAreaLight_kind, DR_LIGHT_POINT);
          lights.f32s(o + OFF_DrAreaLight_L, _rgb(l.intensity));
          lights.i32(o + OFF_DrAreaLight_nsamples, 1);
          lights.f32s(o + OFF_DrAreaLight_position, [l.lightPos.x, l.lightPos.y, l.lightPos.z]);
        } else if (l is DistantLight) {                   // distant_light.dart:38-42
          lights.u32(o + OFF_DrAreaLight_kind, DR_LIGHT_DISTANT);
          lights.f32s(o + OFF_DrAreaLight_L, _rgb(l.L));
          lights.i32(o + OFF_DrAreaLight_nsamples, 1);
          lights.f32s(o + OFF_DrAreaLight_position, [l.lightDir.x, l.lightDir.y, l.lightDir.z]);
        } else {
          _unsupported('light ${l.runtimeType}');
        }
      }
      _Blob ltris = new _Blob(lightTris.length * SIZEOF_DrLightTri);
      blobs.add(ltris);
      for (int i = 0; i < lightTris.length; ++i) {
        for (int k = 0; k < 4; ++k) {
          ltris.u32(i * SIZEOF_DrLightTri + 4 * k, lightTris[i][k]);
        }
      }

      // ---- DrSceneDesc ----
      _Blob sd = new _Blob(SIZEOF_DrSceneDesc);
      blobs.add(sd);
      sd.addr(OFF_DrSceneDesc_nodes, nodes);
      sd.u64(OFF_DrSceneDesc_nnodes, nnodes);
      sd.addr(OFF_DrSceneDesc_verts, verts);
      sd.u64(OFF_DrSceneDesc_nverts, nverts);
      sd.addr(OFF_DrSceneDesc_tri_idx, triIdx);
      sd.u64(OFF_DrSceneDesc_ntris, nprims);
      sd.addr(OFF_DrSceneDesc_tri_material, triMaterial);
      sd.addr(OFF_DrSceneDesc_tri_light, triLight);
      sd.addr(OFF_DrSceneDesc_tri_reverse, triReverse);
      sd.addr(OFF_DrSceneDesc_materials, mats);
      sd.u32(OFF_DrSceneDesc_nmaterials, materials.length);
      sd.addr(OFF_DrSceneDesc_lights, lights);
      sd.u32(OFF_DrSceneDesc_nlights, nlights);
      sd.addr(OFF_DrSceneDesc_light_tris, ltris);
      sd.u32(OFF_DrSceneDesc_nlight_tris, lightTris.length);
      sd.u32(OFF_DrSceneDesc_bvh_depth, 0);               // a Dart BVHAccel does not record its depth: the library measures it
      sd.addr(OFF_DrSceneDesc_env_maps, envMaps);
      sd.u32(OFF_DrSceneDesc_nenv_maps, envMaps == null ? 0 : 1);
      sd.addr(OFF_DrSceneDesc_quadrics, quads);
      sd.u32(OFF_DrSceneDesc_nquadrics, quadrics.length);
      if (anyShading) {
        sd.addr(OFF_DrSceneDesc_vert_normals, vertN);
        sd.addr(OFF_DrSceneDesc_vert_tangents, vertS);
        sd.addr(OFF_DrSceneDesc_vert_uvs, vertUV);
        sd.addr(OFF_DrSceneDesc_tri_shading, triShading);
        sd.addr(OFF_DrSceneDesc_tri_xform, triXform);
        sd.addr(OFF_DrSceneDesc_mesh_xforms, meshXforms);
        sd.u32(OFF_DrSceneDesc_nmesh_xforms, xformMeshes.length);
      }
      _check(_sceneCreate(sd.ptr, handle));
      _applyPilotPicks(handle.value);

      // ---- DrRenderDesc: camera, film, integrator, sampler, task (sampler_renderer.dart:29-31,36-65) ----
      _Blob rd = new _Blob(SIZEOF_DrRenderDesc);
      blobs.add(rd);
      int cam = OFF_DrRenderDesc_camera;
      rd.f32s(cam + OFF_DrCamera_camera_to_world, camera.cameraToWorld.startTransform.m.data);
      rd.f64(cam + OFF_DrCamera_shutter_open, camera.shutterOpen);
      rd.f64(cam + OFF_DrCamera_shutter_close, camera.shutterClose);
      if (camera is PerspectiveCamera || camera is OrthographicCamera) {
        ProjectiveCamera pc = camera;
        rd.f32s(cam + OFF_DrCamera_raster_to_camera, pc.rasterToCamera.m.data);
        rd.f64(cam + OFF_DrCamera_lens_radius, pc.lensRadius);
        rd.f64(cam + OFF_DrCamera_focal_distance, pc.focalDistance);
        rd.i32(cam + OFF_DrCamera_type, camera is PerspectiveCamera ? DR_CAMERA_PERSPECTIVE : DR_CAMERA_ORTHOGRAPHIC);
      } else if (camera is EnvironmentCamera) {
        rd.i32(cam + OFF_DrCamera_type, DR_CAMERA_ENVIRONMENT);
      } else {
        _unsupported('camera ${camera.runtimeType}');
      }
      if (camera.film is! ImageFilm) {
        _unsupported('film ${camera.film.runtimeType}');
      }
      ImageFilm film = camera.film;
      int fo = OFF_DrRenderDesc_film;
      rd.i32(fo + OFF_DrFilm_xres, film.xResolution);
      rd.i32(fo + OFF_DrFilm_yres, film.yResolution);
      for (int k = 0; k < 4; ++k) {
        rd.f64(fo + OFF_DrFilm_crop + 8 * k, film.cropWindow[k]);
      }
      rd.f64(fo + OFF_DrFilm_filter_xw, film.filter.xWidth);
      rd.f64(fo + OFF_DrFilm_filter_yw, film.filter.yWidth);
      // ImageFilm._filterTable is private: the same 16 x 16 table from the same formula (image_film.dart:74-82)
      const int FILTER_TABLE_SIZE = 16;
      int fi = 0;
      for (int y = 0; y < FILTER_TABLE_SIZE; ++y) {
        double fy = (y + 0.5) * film.filter.yWidth / FILTER_TABLE_SIZE;
        for (int x = 0; x < FILTER_TABLE_SIZE; ++x) {
          double fx = (x + 0.5) * film.filter.xWidth / FILTER_TABLE_SIZE;
          rd.f32(fo + OFF_DrFilm_filter_table + 4 * (fi++), film.filter.evaluate(fx, fy));
        }
      }
      if (surfaceIntegrator is PathIntegrator) {
        PathIntegrator pi = surfaceIntegrator;
        rd.i32(OFF_DrRenderDesc_integrator, DR_INTEGRATOR_PATH);
        rd.i32(OFF_DrRenderDesc_max_depth, pi.maxDepth);
      } else if (surfaceIntegrator is DirectLightingIntegrator) {
        DirectLightingIntegrator di = surfaceIntegrator;
        rd.i32(OFF_DrRenderDesc_integrator,
               di.strategy == DirectLightingIntegrator.SAMPLE_ONE_UNIFORM ? DR_INTEGRATOR_DIRECT_ONE : DR_INTEGRATOR_DIRECT_ALL);
        rd.i32(OFF_DrRenderDesc_max_depth, di.maxDepth);
      } else {
        _unsupported('surface integrator ${surfaceIntegrator.runtimeType}');
      }
      if (sampler is! LowDiscrepancySampler) {
        _unsupported('sampler ${sampler.runtimeType}');
      }
      rd.i32(OFF_DrRenderDesc_spp, sampler.samplesPerPixel);
      // The device runs the same LD sampler with one keyed RNG stream per (pixel, LD block) / (pixel, sample) instead
      // of the task's single serial Random(taskNum) (sampler_renderer.dart:137): same estimator, different numbers.
      rd.i32(OFF_DrRenderDesc_sampler_mode, DR_SAMPLER_COUNTER);
      rd.i64(OFF_DrRenderDesc_seed, seed);
      rd.i32(OFF_DrRenderDesc_task_num, taskNum);         // GetSubWindow rectangle (common.dart:52-73)
      rd.i32(OFF_DrRenderDesc_task_count, taskCount);
      rd.i32(OFF_DrRenderDesc_tile_rank, tileRank);       // this process's share of the 32 x 32 tiles (round-robin)
      rd.i32(OFF_DrRenderDesc_tile_count, tileCount);
      rd.i32(OFF_DrRenderDesc_tile_size, 32);

      // ---- Renderer.render ----
      final int npix = film.width * film.height;
      lxyzw = calloc<Float>(4 * npix);
      rgb = calloc<Float>(3 * npix);
      if (tileCount > 1) {
        // one rank of a multi-GPU render: its tiles, ONE ncclReduce(sum) of the full-frame film onto rank 0, which
        // resolves it (dr_render_sharded; commInit has been called by the process).  Ranks > 0 complete with null,
        // like a RenderTask whose output the manager does not merge (render_manager.dart:108-125).
        _check(_renderSharded(handle.value, rd.ptr, 0, lxyzw, rgb));
        if (_commRank() > 0) {
          completer.complete(null);
          return completer.future;
        }
      } else {
        _check(_render(handle.value, rd.ptr, lxyzw, rgb));
      }
      _readPilotPicks(handle.value);
      OutputImage out = new OutputImage(film.left, film.top, film.width, film.height,
                                        film.xResolution, film.yResolution,
                                        new Float32List.fromList(rgb.asTypedList(3 * npix)));
      completer.complete(out);
    } catch (e) {
      completer.completeError(e);
    } finally {
      if (handle.value != nullptr) {
        _destroy(handle.value);
      }
      calloc.free(handle);
      if (lxyzw != nullptr) {
        calloc.free(lxyzw);
      }
      if (rgb != nullptr) {
        calloc.free(rgb);
      }
      for (_Blob b in blobs) {
        b.free();
      }
    }
    return completer.future;
  }

  // Per-ray seams are far too fine grained for FFI (one call per ray); like AggregateTestRenderer
  // (aggregate_test_renderer.dart:120-128) they are not part of this renderer.
  Spectrum Li(Scene scene, RayDifferential ray, Sample sample, RNG rng,
              [Intersection isect, Spectrum T]) => new Spectrum(0.0);

  Spectrum transmittance(Scene scene, RayDifferential ray, Sample sample, RNG rng) =>
      new Spectrum(1.0);

  int taskNum;
  int taskCount;
  int device;
  int seed;
  int tileRank;
  int tileCount;
  Sampler sampler;
  Camera camera;
  SurfaceIntegrator surfaceIntegrator;
  VolumeIntegrator volumeIntegrator;
}
