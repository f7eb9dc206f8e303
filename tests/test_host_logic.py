"""Host logic of the product above the C ABI (no GPU needed): the library loads and exports every
symbol of include/dartray_hip.h, the host BVH builder reproduces the oracle's tree bit for bit, and
the work-split helpers match the reference's semantics."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from dartray_amd import _abi, core, scenes
from test_oracle_bvh import SCENES


def test_library_exports_every_declared_symbol(hip):
    hdr = open(os.path.join(ROOT, "include", "dartray_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dr_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(_abi.EXPORTS), (declared ^ set(_abi.EXPORTS))
    l = _abi.lib()
    for name in declared:
        assert getattr(l, name) is not None
    assert b"gfx950" in l.dr_version()
    # the ABI version a host checks at load: the header's define, the library's answer and the ctypes binding's constant agree
    header = open(os.path.join(ROOT, "include", "dartray_hip.h")).read()
    want = int(re.search(r"#define DR_ABI_VERSION (\d+)", header).group(1))
    assert l.dr_abi_version() == want == hip.DR_ABI_VERSION


def test_abi_struct_sizes_match_the_header(hip):
    assert C.sizeof(_abi.DrBvhNode) == 32 and C.sizeof(_abi.DrRay) == 40 and C.sizeof(_abi.DrHit) == 32 and C.sizeof(_abi.DrCamera) == 168
    assert C.sizeof(_abi.DrMaterial) == 56 and C.sizeof(_abi.DrQuadric) == 168 and C.sizeof(_abi.DrAreaLight) == 128 and C.sizeof(_abi.DrMeshXform) == 128 and C.sizeof(_abi.DrEnvMap) == 144 and C.sizeof(_abi.DrLightTri) == 16
    assert core.NODE_DTYPE.itemsize == 32 and core.HIT_DTYPE.itemsize == 32


def test_render_without_a_gpu_fails_loudly(hip):
    """No CPU fallback exists: without a device the product raises (the Dart shim would LogSevere)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_abi.DartRayHipError):
        _abi.check(_abi.lib().dr_init(0))
    prims, mk = scenes.config("C1")
    with pytest.raises(_abi.DartRayHipError):
        mk().render(scenes.make_scene(prims))


@pytest.mark.parametrize("name", list(SCENES))
def test_host_bvh_builder_equals_oracle(ob, hip, name):
    """dr_bvh_build (product, C++) vs the oracle's restatement of bvh_accel.dart:228-437: identical node
    array (32-byte records), identical primitive order, identical per-primitive tables."""
    prims = SCENES[name]()
    acc = core.BVHAccel(prims)
    osc = ob.OracleScene(prims)
    nodes, tri, mesh, src = osc.bvh()
    assert acc.nodes.tobytes() == nodes.tobytes()
    assert np.array_equal(acc.tri_idx, tri)
    assert np.array_equal(acc.tri_material, mesh.astype(np.uint32))  # one material per GeometricPrimitive
    assert acc.depth == osc.depth
    assert np.array_equal(acc.verts, osc.verts())
    lights = [i for i, gp in enumerate(prims) if gp.areaLight is not None]
    exp_light = np.array([lights.index(m) if m in lights else -1 for m in mesh], np.int32)
    assert np.array_equal(acc.tri_light, exp_light)


def test_parallel_builder_is_thread_count_invariant(hip, monkeypatch):
    """the parallel SAH builder (concurrent sub-tree tasks) yields the same bytes for any
    thread count -- the tree only depends on the reference's algorithm, not on scheduling."""
    prims = scenes.cornell_prims(scenes.blob_prim(300, 150))  # 90k triangles: above the task cut-off
    out = []
    for th in ("1", "3", "8"):
        monkeypatch.setenv("DARTRAY_BUILD_THREADS", th)
        acc = core.BVHAccel(prims)
        out.append((acc.nodes.tobytes(), acc.tri_idx.tobytes(), acc.depth))
    assert out[0] == out[1] == out[2]


def test_refine_reverses_triangle_order():
    # Primitive.fullyRefine pops a LIFO stack (primitive.dart:71-84)
    m = core.TriangleMesh(np.arange(12).reshape(4, 3), np.zeros((12, 3)))
    assert list(m.refine()) == [3, 2, 1, 0]


def test_empty_scene_builds(hip):
    acc = core.BVHAccel([])
    assert acc.nodes is None and len(acc.tri_idx) == 0  # bvh_accel.dart:50-53


def test_camera_matrices_match_the_oracle(ob):
    """Host camera set-up (projective_camera.dart:34-53, transform.dart:301-349).  The matrices are inputs
    that cross the ABI, so the product's numpy construction only has to agree to f32 precision."""
    for (xres, yres, fov) in [(64, 64, 35.0), (1024, 1024, 35.0), (320, 240, 55.0), (240, 320, 20.0)]:
        cam = scenes.cornell_camera(xres, yres) if fov == 35.0 else core.PerspectiveCamera.lookAt(
            (1, 2, -30), (0.5, 0, 0), (0, 1, 0), fov, core.ImageFilm(xres, yres))
        pos = np.array((0, 0, -35) if fov == 35.0 else (1, 2, -30), np.float32)
        look = np.array((0, 0, 0) if fov == 35.0 else (0.5, 0, 0), np.float32)
        up = np.array((0, 1, 0), np.float32)
        r2c = np.zeros(16, np.float32)
        c2w = np.zeros(16, np.float32)
        ob.lib().orc_camera_setup(pos.ctypes.data, look.ctypes.data, up.ctypes.data, fov, xres, yres, r2c.ctypes.data, c2w.ctypes.data)
        assert np.allclose(cam.cameraToWorld.reshape(-1), c2w, rtol=0, atol=1e-6)
        assert np.allclose(cam.rasterToCamera.reshape(-1), r2c, rtol=2e-6, atol=1e-9)


def test_film_window_and_sample_extent():
    f = core.ImageFilm(64, 48)
    assert (f.left, f.top, f.width, f.height) == (0, 0, 64, 48)
    assert f.getSampleExtent() == (0, 65, 0, 49)  # one pixel larger than the film (Appendix D.16)
    f = core.ImageFilm(100, 100, core.BoxFilter(0.5, 0.5), (0.25, 0.75, 0.5, 1.0))
    assert (f.left, f.top, f.width, f.height) == (25, 50, 50, 50)
    assert np.all(f.filterTable == 1.0) and f.filterTable.shape == (256,)


def test_sample_floats(ob, hip):
    l = _abi.lib()
    assert l.dr_sample_floats(_abi.DR_INTEGRATOR_PATH, 1) == 37  # Appendix B
    assert l.dr_sample_floats(_abi.DR_INTEGRATOR_DIRECT_ALL, 1) == 13
    osc = ob.OracleScene(scenes.cornell_c1_prims())
    assert osc.sample_floats(1, 5) == 37 and osc.sample_floats(0, 5) == 13


def test_pixel_enumeration_tasks_and_tiles(ob, hip):
    prims, mk = scenes.config("C2", xres=100, yres=70, spp=4, blob=(8, 4))
    full = mk().pixels()
    assert len(full) == 101 * 71 and tuple(full[0]) == (0, 0) and tuple(full[-1]) == (100, 70)
    assert np.array_equal(full[:101, 0], np.arange(101)) and np.all(full[:101, 1] == 0)  # linear order
    # tasks: the GetSubWindow rectangles partition the sampler window and match the oracle's restatement
    seen = np.zeros((71, 101), np.int32)
    for t in range(8):
        _, mk_t = scenes.config("C2", xres=100, yres=70, spp=4, blob=(8, 4), taskNum=t, taskCount=8)
        p = mk_t().pixels()
        ext = np.zeros(4, np.int32)
        ob.lib().orc_get_sub_window(101, 71, t, 8, ext.ctypes.data)
        assert len(p) == (ext[1] - ext[0]) * (ext[3] - ext[2])
        assert p[:, 0].min() == ext[0] and p[:, 0].max() == ext[1] - 1 and p[:, 1].min() == ext[2] and p[:, 1].max() == ext[3] - 1
        seen[p[:, 1], p[:, 0]] += 1
    assert np.all(seen == 1)
    # tiles: round-robin 32x32 tiles over 3 ranks partition the window; a pixel's owner is its tile index mod 3
    seen[:] = 0
    for rnk in range(3):
        _, mk_t = scenes.config("C2", xres=100, yres=70, spp=4, blob=(8, 4), tileRank=rnk, tileCount=3)
        p = mk_t().pixels()
        owner = ((p[:, 1] // 32) * 4 + p[:, 0] // 32) % 3
        assert np.all(owner == rnk)
        seen[p[:, 1], p[:, 0]] += 1
    assert np.all(seen == 1)


def test_plugin_registry_names():
    # the names RegisterStandardPlugins registers for the path (render_manager_interface.dart:37-157)
    assert core.Plugin.get("accelerator", "bvh") is not None
    assert isinstance(core.Plugin.get("surfaceIntegrator", "path")({"maxdepth": 7}), core.PathIntegrator)
    assert core.Plugin.get("surfaceIntegrator", "path")().maxDepth == 5
    assert core.Plugin.get("renderer", "sampler") is core.SamplerRenderer
    assert core.Plugin.get("accelerator", "kdtree") is None
    assert core.RoundUpPow2(5) == 8 and core.RoundUpPow2(256) == 256
