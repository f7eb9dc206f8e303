"""The second, independently written restatement of the reference path (tests/golden/dart_restatement.py: plain
Python, from the Dart text) against the C++ oracle and the GPU.

The reference is Dart, cannot run here and ships no golden vectors, so nothing in this image can pin the oracle to
the reference itself ("parity unpinned").  What these tests remove is the single-reader risk: two readings of the
same Dart functions, in different languages and with a different structure, must agree bit for bit -- per-sample
radiance, film, written image, the number of RNG draws per sample, and hit records on 4000 stress rays.  The cases:
C1 (DirectLighting), C2-small (PathIntegrator, matte) and a mirror + glass scene (specular lobes, FresnelDielectric,
two-lobe component selection, the specularBounce rule: SURVEY section 8 row f4) and an open scene under an
InfiniteAreaLight (rows a25 / f2: the MIP pyramid, its trilinear lookup and the Distribution2D are rebuilt from the
texels by the restatement, then Le / sampleL / pdf); and the mirror + glass scene under the reference's default
integrator, DirectLighting, whose SpecularReflect / SpecularTransmit recurse through Renderer.Li; a thin-lens camera;
DirectLighting with several samples per light; and a box of spheres and disks (Shape / Sphere / Disk / Quadratic and the
Transform methods they use, as BVH primitives and as emitters) under both integrators.
  CPU: oracle (live, serial mode) == committed restatement fixtures; the fixtures are reproducible from the script.
  GPU: the recorded serial streams replayed through DR_SAMPLER_HOST_BUFFER == the restatement's films."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from dartray_amd import _abi, core, pbrt, scenes

sys.path.insert(0, GOLDEN)
import dart_restatement as dr  # noqa: E402
import make_restatement_fixtures as mrf  # noqa: E402


def _cases():
    return {c[0]: c for c in mrf.cases()}


@pytest.mark.parametrize("name,spp,record", [("restatement_c1.npz", 4, 65 * 65 * 4), ("restatement_c2small.npz", 8, 17 * 17 * 8),
                                             ("restatement_cspec.npz", 8, 17 * 17 * 8), ("restatement_cenv.npz", 8, 17 * 17 * 8),
                                             ("restatement_cdlspec.npz", 4, 17 * 17 * 4), ("restatement_clens.npz", 4, 17 * 17 * 4),
                                             ("restatement_cdl2.npz", 4, 17 * 17 * 4), ("restatement_cquad.npz", 8, 17 * 17 * 8),
                                             ("restatement_cquaddl.npz", 4, 17 * 17 * 4), ("restatement_cdlone.npz", 4, 17 * 17 * 4),
                                             ("restatement_cenvnp2.npz", 8, 17 * 17 * 8)])
def test_oracle_equals_the_independent_restatement(ob, name, spp, record):
    """The oracle, run live in the reference's serial mode, against what the Python restatement computed from the
    same sample vectors and RNG draws."""
    _, prims, r, golden, integ, _ = _cases()[name]
    fx = np.load(os.path.join(GOLDEN, name))
    g = np.load(os.path.join(GOLDEN, golden))
    env = getattr(r, "env", None)
    osc = ob.OracleScene(prims, env=env) if env is not None else ob.OracleScene(prims)
    rec = osc.render(ob.render_desc(r, sampler_mode=0), record=record, max_tail=40 if integ == "path" else (200 if ("dlspec" in name or "quaddl" in name or "dlone" in name) else 8))
    assert np.array_equal(rec["sample_vec"], g["sample_vec"])          # same inputs as the fixtures were made from
    assert np.array_equal(rec["Ls"], fx["Ls"])                          # per-sample Li
    assert np.array_equal(rec["film"], fx["film"])                      # ImageFilm.addSample, in reference order
    assert np.array_equal(rec["rgb"], fx["rgb"])                        # ImageFilm.writeImage
    assert np.array_equal(rec["tail_count"], fx["draws_used"])          # RNG consumption inside Li
    assert (fx["Ls"].max(axis=1) > 0).mean() > 0.1


def test_fixtures_are_what_the_script_computes():
    """Re-runs the restatement on the head of each stream: the committed fixtures are its output, not hand edits."""
    for name, prims, r, golden, integ, nspl in mrf.cases():
        fx = np.load(os.path.join(GOLDEN, name))
        out = mrf.run(name, prims, r, golden, integ, nspl, limit=600)
        assert np.array_equal(out["Ls"], fx["Ls"][:600]) and np.array_equal(out["draws_used"], fx["draws_used"][:600])


def test_live_restatement_regenerates_the_serial_streams():
    """Nothing recorded: the restatement's own port of the Dart VM generator, seeded RNG(taskNum = 0) like
    sampler_renderer.dart:137, feeds its LDPixelSample and its Li in turn (one serial stream, as the reference consumes
    it).  The sample vectors, the per-sample radiance and the number of in-Li draws it produces must equal the golden
    serial streams (recorded from the C++ oracle) bit for bit -- sampler (a2 / a3), RNG (a22) and integrators read twice."""
    for name, prims, r, golden, integ, nspl in mrf.cases():
        g = np.load(os.path.join(GOLDEN, golden))
        sv, pix, spp = g["sample_vec"], g["pixel_xy"], r.sampler.samplesPerPixel
        if integ == "path":
            n1D, n2D = [1] * 14, [1] * 9                       # requestSamples (path_integrator.dart:124-131) + the volume integrator's two slots
        elif integ == "directone":
            n1D, n2D = [1] * 5, [1] * 2                        # strategy "one" (direct_lighting_integrator.dart:82-87): light, lightNum, bsdf + tau, scatter
        else:
            n2D = [k for ns in nspl for k in (ns, ns)]          # per light: LightSampleOffsets, BSDFSampleOffsets (add1D(n) + add2D(n) each)
            n1D = n2D + [1, 1]
        scene = mrf.build_scene(prims, getattr(r, "env", None))
        cam = mrf.restated_camera(r.camera)
        rng = dr.RNG(0)
        k = 0
        for px, py in pix[:400 if name == "restatement_c1.npz" else len(pix)]:
            for v in dr.LDPixelSample(0.0, 1.0, spp, n1D, n2D, rng):
                assert np.array_equal(np.array(v, np.float32), sv[k]), (name, k)
                L, _, _, nd = dr.renderer_Li(scene, integ, r.surfaceIntegrator.maxDepth, cam, int(px), int(py), v, rng, nspl)
                assert np.array_equal(np.array(L.tuple(), np.float32), g["Ls"][k]), (name, k)
                assert nd == int(g["tail_count"][k]), (name, k)
                k += 1
        assert k >= 1100


@pytest.mark.parametrize("res,fw,tasks", [((64, 64), 0.5, 1), ((100, 70), 0.5, 4), ((33, 95), 2.0, 8), ((16, 16), 0.5, 1)])
def test_restated_sampler_window_and_pixel_orders(hip, res, fw, tasks):
    """getSampleExtent + GetSubWindow + _makeSampler and the linear / tile pixel samplers (the tile sampler shuffles
    its tiles with its own RNG(5489)) restated from the Dart against the product -- the C library's enumeration of a
    task's pixels (dr_enumerate_pixels) and the host pixel-sampler classes -- and, for the fixtures' resolution, against
    the pixel order recorded in the golden serial streams."""
    film = core.ImageFilm(res[0], res[1], core.BoxFilter(fw, fw))
    cam = core.PerspectiveCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), 35.0, film)
    ext = dr.getSampleExtent(0, 0, res[0], res[1], fw, fw)
    assert tuple(ext) == tuple(film.getSampleExtent())
    for num in range(tasks):
        x, y, w, h = dr.sampler_window(ext, num, tasks)
        r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 4, 5489), cam, core.PathIntegrator(5), core.EmissionIntegrator(),
                                 taskNum=num, taskCount=tasks)
        assert np.array_equal(np.array(dr.linear_pixels(x, y, w, h), np.int32).reshape(-1, 2), r.pixels()), num   # the task's window, linear order
        for ts in (32, 7):   # the reference's default order (what the serial replay feeds through host buffers)
            assert np.array_equal(np.array(dr.tile_pixels(x, y, w, h, ts), np.int32).reshape(-1, 2), core.TilePixelSampler(ts).setup(x, y, w, h)), (num, ts)
    if res == (16, 16):
        g = np.load(os.path.join(GOLDEN, "c2small_path_serial.npz"))
        x, y, w, h = dr.sampler_window(ext, 0, 1)
        assert np.array_equal(np.array(dr.linear_pixels(x, y, w, h), np.int32), g["pixel_xy"])


@pytest.mark.parametrize("pos,look,up,fov,res", [((0.0, 0.0, -35.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 35.0, (16, 16)),
                                                  ((0.0, 14.0, -72.0), (0.0, 3.0, 0.0), (0.0, 1.0, 0.0), 45.0, (2048, 2048)),
                                                  ((3.5, 2.25, -9.0), (0.5, -1.0, 4.0), (0.1, 1.0, 0.2), 63.7, (100, 70)),
                                                  ((-1.0, 8.0, 2.0), (4.0, 0.0, 3.0), (0.0, 0.0, 1.0), 20.0, (70, 100))])
def test_restated_camera_matrices_equal_the_product(pos, look, up, fov, res):
    """Transform.LookAt / Perspective / Scale / Translate, Matrix4x4.Mul / Inverse (f32 storage, f64 expressions) and the
    ProjectiveCamera constructor restated from the Dart: rasterToCamera and cameraToWorld equal the product's host
    camera (whose matrices test_camera_matrices_match_the_oracle ties to the oracle) bit for bit."""
    film = core.ImageFilm(res[0], res[1], core.BoxFilter(0.5, 0.5))
    cam = core.PerspectiveCamera.lookAt(pos, look, up, fov, film)
    r2c, c2w = dr.perspective_camera_matrices(pos, look, up, fov, res[0], res[1])
    assert np.array_equal(np.array(r2c, np.float32), cam.rasterToCamera.reshape(-1))
    assert np.array_equal(np.array(c2w, np.float32), cam.cameraToWorld.reshape(-1))


@pytest.mark.parametrize("angle,axis", [(90.0, (1.0, 0.0, 0.0)), (-60.0, (1.0, 0.0, 0.0)), (30.0, (0.0, 0.0, 1.0)), (137.5, (0.3, -2.0, 0.7))])
def test_restated_rotate_and_transform_products(angle, axis):
    """Transform.Rotate (transform.dart:276-303: f32 normalised axis, inverse = transpose) and products with Translate,
    as the quadric cases place their shapes, against the scene-file front end's Transform (bit for bit, m and mInv)."""
    a = dr.Transform.Translate(dr.Vec(4.5, -5.5, -3.0)) * dr.Transform.Rotate(angle, dr.Vec(*axis)) * dr.Transform.Rotate(30.0, dr.Vec(0.0, 0.0, 1.0))
    b = pbrt.Transform.Translate(4.5, -5.5, -3.0) * pbrt.Transform.Rotate(angle, *axis) * pbrt.Transform.Rotate(30.0, 0.0, 0.0, 1.0)
    assert np.array_equal(np.array(a.m, np.float32), np.asarray(b.m, np.float32).reshape(-1))
    assert np.array_equal(np.array(a.mInv, np.float32), np.asarray(b.mInv, np.float32).reshape(-1))


def _refined_triangles(prims):
    """fullyRefine (primitive.dart:71-84): a mesh's triangles come off the todo stack in reverse order."""
    tris, vid, base, nq = [], [], 0, 0
    for gp in prims:
        mesh = gp.shape
        if isinstance(mesh, (core.Sphere, core.Disk)):   # canIntersect: kept whole, bounded by Shape.worldBound
            cls = dr.Sphere if isinstance(mesh, core.Sphere) else dr.Disk
            q = cls(mesh.objectToWorld.reshape(-1), mesh.worldToObject.reshape(-1), mesh.reverseOrientation, *mesh.params)
            tris.append(q.worldBound())
            vid.append((_abi.DR_PRIM_QUADRIC, nq, 0))
            nq += 1
            continue
        for t in range(len(mesh.vertexIndex) - 1, -1, -1):
            a, b, c = (int(v) for v in mesh.vertexIndex[t])
            tris.append(tuple(dr.Vec(*map(float, mesh.P[i])) for i in (a, b, c)))
            vid.append((a + base, b + base, c + base))
        base += len(mesh.P)
    return tris, vid


@pytest.mark.parametrize("case", ["c1", "c2small", "cspec", "cenv", "cquad", "blob6k"])  # (cdlspec is cspec's scene)
def test_restated_sah_build_equals_the_product_builder(hip, case):
    """BVHAccel's SAH build + flattening (bvh_accel.dart:41-91, 228-437; partition / nth_element of common.dart) restated
    in Python against dr_bvh_build (which the oracle's serial builder equals byte for byte,
    test_host_bvh_builder_equals_oracle): node boxes, child offsets, leaf ranges, split axes and the primitive order."""
    if case == "blob6k":
        prims, _ = scenes.config("C2", xres=8, yres=8, spp=1, blob=(80, 40))
    else:
        prims = _cases()["restatement_%s.npz" % case][1]
    acc = core.BVHAccel(prims)
    tris, vid = _refined_triangles(prims)
    nodes, order = dr.build_bvh(tris, 4)
    assert len(nodes) == len(acc.nodes)
    for i, n in enumerate(nodes):
        m = acc.nodes[i]
        assert tuple(np.float32(n[0])) == tuple(m["bmin"]) and tuple(np.float32(n[1])) == tuple(m["bmax"]), i
        assert n[2] == int(m["offset"]) and n[3] == int(m["nprims"]) and (n[3] > 0 or n[4] == int(m["axis"])), i
    assert np.array_equal(np.array([vid[i] for i in order], np.uint32), acc.tri_idx)


def test_restated_traversal_reproduces_the_golden_hit_records():
    """BVHAccel.intersect / intersectP + Triangle.intersect / intersectP restated in Python against the golden hit
    records of tests/golden/c2small_hits.npz (4000 AggregateTestRenderer-style rays: axis-parallel directions,
    origins on surfaces, finite and infinite extents)."""
    g = np.load(os.path.join(GOLDEN, "c2small_hits.npz"))
    prims, _ = scenes.config("C2", xres=16, yres=16, spp=8, blob=(32, 16))
    scene = mrf.build_scene(prims)
    bad = 0
    for k in range(len(g["o"])):
        mk = lambda: dr.Ray(dr.Vec(*map(float, g["o"][k])), dr.Vec(*map(float, g["d"][k])), float(g["tmin"][k]), float(g["tmax"][k]))
        ray = mk()
        hit = scene.bvh.intersect(ray)
        prim = -1 if hit is None else scene.bvh.prims.index(hit.prim)
        if prim != int(g["prim"][k]) or (hit is not None and ray.maxt != float(g["t"][k])):
            bad += 1
        if scene.bvh.intersectP(mk()) != bool(g["occluded"][k]):
            bad += 1
    assert bad == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,spp", [("restatement_c1.npz", 4), ("restatement_c2small.npz", 8), ("restatement_cspec.npz", 8), ("restatement_cenv.npz", 8),
                                      ("restatement_cdlspec.npz", 4), ("restatement_clens.npz", 4), ("restatement_cdl2.npz", 4),
                                      ("restatement_cquad.npz", 8), ("restatement_cquaddl.npz", 4), ("restatement_cdlone.npz", 4),
                                      ("restatement_cenvnp2.npz", 8)])
def test_gpu_replay_equals_the_independent_restatement(gpu, name, spp):
    _, prims, r, golden, integ, _ = _cases()[name]
    fx = np.load(os.path.join(GOLDEN, name))
    g = np.load(os.path.join(GOLDEN, golden))
    r.sampler = core.HostBufferSampler(r.camera, spp, g["pixel_xy"], g["sample_vec"], g["tail"] if "tail" in g.files else None)
    out = r.render(scenes.make_scene(prims, getattr(r, "env", None)))
    assert np.array_equal(out.film, fx["film"]) and np.array_equal(out.rgb, fx["rgb"])
