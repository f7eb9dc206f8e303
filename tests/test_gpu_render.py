"""GPU parity of SamplerRenderer.render (the whole wavefront pipeline) against the oracle, through the
C ABI (dr_render / dr_render_device).  Tolerance: north_star asks for 1e-4 relative per pixel
(SURVEY.md section 8d metric); the f64-faithful kernels are in practice bit-exact, which is what most of
these tests assert -- rel_tol below is the contractual bound."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from dartray_amd import _abi, core, scenes
from util import rel_err_image

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4  # per-pixel: max_c |gpu - ref| / max(max_c |ref|, 1e-6)


def _check(out, ref, exact=True):
    err = rel_err_image(out.rgb, ref["rgb"]).max()
    assert err <= REL_TOL, err
    if exact:
        assert np.array_equal(out.film, ref["film"])
        assert np.array_equal(out.rgb, ref["rgb"])


def test_c1_direct_lighting_counter_mode(ob, gpu):
    """BASELINE config 0: Cornell floor + quad emitter, DirectLighting, 64x64, 4 spp."""
    prims, mk = scenes.config("C1")
    r = mk()
    out = r.render(scenes.make_scene(prims))
    osc = ob.OracleScene(prims)
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    _check(out, ref)
    c, st = osc.counters(), r.last_stats
    for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris", "camera_samples"):
        assert st[k] == c[k], k
    assert st["film_samples"] == 64 * 64 * 4 and st["camera_samples"] == 65 * 65 * 4


def test_c1_reference_serial_stream_via_host_buffers(gpu):
    """The reference-faithful run: sample vectors produced by ONE serial DartRandom(0) stream (golden,
    recorded by the oracle) are handed to the GPU as host buffers; the image must equal the serial golden."""
    g = np.load(os.path.join(GOLDEN, "c1_serial.npz"))
    prims, mk = scenes.config("C1")
    r = mk()
    r.sampler = core.HostBufferSampler(r.camera, 4, g["pixel_xy"], g["sample_vec"])
    out = r.render(scenes.make_scene(prims))
    assert np.array_equal(out.rgb, g["rgb"]) and np.array_equal(out.film, g["film"])


def test_path_serial_stream_with_recorded_rng_tail(gpu):
    """PathIntegrator with the recorded in-Li RNG draws (bounces >= 3, Russian roulette) as host buffers."""
    g = np.load(os.path.join(GOLDEN, "c2small_path_serial.npz"))
    prims, mk = scenes.config("C2", xres=16, yres=16, spp=8, blob=(32, 16))
    r = mk()
    r.sampler = core.HostBufferSampler(r.camera, 8, g["pixel_xy"], g["sample_vec"], g["tail"])
    out = r.render(scenes.make_scene(prims))
    assert rel_err_image(out.rgb, g["rgb"]).max() <= REL_TOL
    assert np.array_equal(out.rgb, g["rgb"]) and np.array_equal(out.film, g["film"])
    # missing tail buffer is an error, not a silent zero
    r.sampler = core.HostBufferSampler(r.camera, 8, g["pixel_xy"], g["sample_vec"], None)
    with pytest.raises(_abi.DartRayHipError):
        r.render(scenes.make_scene(prims))


def test_reference_default_tile_order_replayed_through_host_buffers(ob, gpu):
    """The reference's default pixel order is Pixels "tile" (32 x 32 tiles shuffled by RNG(5489),
    tile_pixel_sampler.dart:38-95): the oracle walks the window in that order with the task's serial RNG, records every
    sample's inputs, and the device -- given the recording in the same order -- reproduces the serial image exactly."""
    prims, mk = scenes.config("C2", xres=70, yres=40, spp=4, blob=(24, 12))
    r = mk()
    r.sampler.pixelSampler = core.TilePixelSampler()
    rec = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=0), record=71 * 41 * 4, max_tail=40)
    order = core.TilePixelSampler().setup(0, 0, 71, 41)
    assert np.array_equal(rec["pixel_xy"][::4], order)            # the recording follows the tile order
    r.sampler = core.HostBufferSampler(r.camera, 4, rec["pixel_xy"][::4], rec["sample_vec"], rec["tail"])
    out = r.render(scenes.make_scene(prims))
    assert np.array_equal(out.film, rec["film"]) and np.array_equal(out.rgb, rec["rgb"])


def test_path_counter_mode_golden(gpu):
    g = np.load(os.path.join(GOLDEN, "c2small_path_counter.npz"))
    prims, mk = scenes.config("C2", xres=16, yres=16, spp=8, blob=(32, 16))
    out = mk().render(scenes.make_scene(prims))
    assert np.array_equal(out.rgb, g["rgb"]) and np.array_equal(out.film, g["film"])


@pytest.mark.parametrize("cfg", [dict(xres=64, yres=64, spp=16, blob=(40, 20)),
                                 dict(xres=48, yres=32, spp=64, blob=(100, 50)),
                                 dict(xres=20, yres=20, spp=256, blob=(16, 8))])
def test_path_counter_mode_vs_oracle(ob, gpu, cfg):
    prims, mk = scenes.config("C2", **cfg)
    r = mk()
    out = r.render(scenes.make_scene(prims))
    osc = ob.OracleScene(prims)
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    _check(out, ref)
    c, st = osc.counters(), r.last_stats
    for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris"):
        assert st[k] == c[k], k


@pytest.mark.parametrize("spp", [1, 2, 64, 128, 256, 512, 1024, 2048, 4096])
def test_sample_counts_at_the_ends_of_the_range(ob, gpu, spp):
    """spp 1 / 2 (degenerate LD blocks, row-major k_gen_samples), 64..256 (k_gen_samples_lm, u8), 512 / 1024 (u16, the
    two-wave kernel) and 2048 / 4096 (u16, 32 / 16 pixels per sampler group)."""
    prims, mk = scenes.config("C2", xres=8, yres=6, spp=spp, blob=(16, 8))
    r = mk()
    out = r.render(scenes.make_scene(prims))
    ref = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=1))
    _check(out, ref)


@pytest.mark.parametrize("crop,fw,res", [((0.25, 0.75, 0.5, 1.0), 0.5, (40, 24)), ((0.0, 1.0, 0.0, 1.0), 1.0, (17, 29)),
                                          ((0.1, 0.33, 0.2, 0.9), 1.5, (31, 20))])
def test_crop_windows_and_wide_box_filters(ob, gpu, crop, fw, res):
    """ImageFilm crop windows (image_film.dart:61-65) and filters wider than a pixel: samples then spread over
    several pixels (float atomics on the device: the sum order differs, so the film is compared to 1e-6).  The
    reference samples the window [0,w) x [0,h) whatever the extent's origin: reproduced, so a
    cropped image is only partly covered -- in the oracle and on the device alike."""
    prims = scenes.cornell_prims(scenes.blob_prim(16, 8))
    film = core.ImageFilm(res[0], res[1], core.BoxFilter(fw, fw), crop)
    cam = core.PerspectiveCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), 35.0, film)
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 8), cam, core.PathIntegrator(3), core.EmissionIntegrator())
    out = r.render(scenes.make_scene(prims))
    ref = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=1))
    assert out.rgb.shape == ref["rgb"].shape == (film.height, film.width, 3)
    assert rel_err_image(out.rgb, ref["rgb"]).max() <= REL_TOL
    assert np.allclose(out.film, ref["film"], rtol=1e-5, atol=1e-6)
    if fw == 0.5:
        assert np.array_equal(out.film, ref["film"])


def test_c4_class_hairball(ob, gpu):
    """BASELINE config 3 at reduced size: thin strands => deep, unbalanced tree, many leaf tests per ray."""
    prims, mk = scenes.config("C4", xres=32, yres=32, spp=8, hair=(200, 40))
    r = mk()
    out = r.render(scenes.make_scene(prims))
    osc = ob.OracleScene(prims)
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    _check(out, ref)
    c, st = osc.counters(), r.last_stats
    for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris"):
        assert st[k] == c[k], k


@pytest.mark.parametrize("depth", [0, 1, 3, 4, 8])
def test_path_max_depth(ob, gpu, depth):
    prims, mk = scenes.config("C2", xres=24, yres=24, spp=8, blob=(16, 8))
    r = mk()
    r.surfaceIntegrator = core.PathIntegrator(depth)
    out = r.render(scenes.make_scene(prims))
    ref = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=1))
    _check(out, ref)


def test_scene_without_lights_and_black_material(ob, gpu):
    # nLights == 0: UniformSampleOneLight returns before any draw (integrator.dart:87-90); Kd == 0: empty BSDF
    walls = scenes.cornell_walls()
    walls[0].material = core.MatteMaterial((0.0, 0.0, 0.0))
    for prims in (walls, walls + [scenes.emitter_quad()]):
        film = core.ImageFilm(16, 16)
        cam = core.PerspectiveCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), 35.0, film)
        r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 8), cam, core.PathIntegrator(5), core.EmissionIntegrator())
        out = r.render(scenes.make_scene(prims))
        ref = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=1))
        _check(out, ref)


def test_thin_lens_camera(ob, gpu):
    prims = scenes.cornell_prims(scenes.blob_prim(16, 8))
    film = core.ImageFilm(24, 24)
    cam = core.PerspectiveCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), 35.0, film, lensradius=0.5, focaldistance=30.0)
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 8), cam, core.PathIntegrator(3), core.EmissionIntegrator())
    out = r.render(scenes.make_scene(prims))
    ref = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=1))
    _check(out, ref)


def test_two_emitters_direct_and_path(ob, gpu):
    e2 = scenes._quad((-9.9, -2, -2), (-9.9, 2, -2), (-9.9, 2, 2), (-9.9, -2, 2), (0.5, 0.5, 0.5),
                      core.DiffuseAreaLight((5.0, 9.0, 3.0), 1))
    prims = scenes.cornell_prims(scenes.blob_prim(16, 8)) + [e2]
    for integ in (core.DirectLightingIntegrator(0, 5), core.PathIntegrator(5)):
        film = core.ImageFilm(24, 24)
        cam = core.PerspectiveCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), 35.0, film)
        r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 16), cam, integ, core.EmissionIntegrator())
        out = r.render(scenes.make_scene(prims))
        ref = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=1))
        _check(out, ref)
        assert out.rgb.mean() > 0


@pytest.mark.parametrize("spp", [1, 8, 64])
def test_direct_lighting_strategy_one(ob, gpu, spp):
    """DirectLightingIntegrator with SAMPLE_ONE_UNIFORM (direct_lighting_integrator.dart:51-55): UniformSampleOneLight with the
    integrator's own slots -- LightSampleOffsets(1), lightNumOffset = add1D(1), BSDFSampleOffsets(1) (:82-87: 5 + 5 + 4 = 14 floats) --
    over three lights (two quads of different radiance and a constant sky, so that the choice shows) with a mirror blob whose
    SpecularReflect recursion reads the same slots again at the deeper vertices.  Counter streams against the oracle (film and visit
    counters), then the reference's one serial stream replayed through host buffers."""
    e2 = scenes._quad((-9.9, -2, -2), (-9.9, 2, -2), (-9.9, 2, 2), (-9.9, -2, 2), (0.5, 0.5, 0.5), core.DiffuseAreaLight((5.0, 9.0, 3.0), 3))
    prims = scenes.cornell_prims(scenes.blob_prim(16, 8)) + [e2, core.GeometricPrimitive(scenes.blob_mesh(12, 6, radius=2.5, centre=(5.5, -6.5, -3.0)),
                                                                                     core.MirrorMaterial((0.9, 0.9, 0.9)))]
    env = core.InfiniteAreaLight(scenes.SKY_TO_WORLD, (0.2, 0.3, 0.5), 2, None)
    film = core.ImageFilm(20, 16)
    cam = core.PerspectiveCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), 35.0, film)
    one = core.DirectLightingIntegrator(core.DirectLightingIntegrator.SAMPLE_ONE_UNIFORM, 5)
    assert one.kind == _abi.DR_INTEGRATOR_DIRECT_ONE
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, spp), cam, one, core.EmissionIntegrator())
    scene = scenes.make_scene(prims, env)
    out = r.render(scene)
    osc = ob.OracleScene(prims, env=env)
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    _check(out, ref)
    c, st = osc.counters(), r.last_stats
    for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris"):
        assert st[k] == c[k], k
    # ONE shadow ray (+ at most one MIS ray) per vertex, whatever the number of lights: strategy "all" traces three
    assert st["any_rays"] <= st["closest_rays"]
    assert _abi.lib().dr_scene_sample_floats(scene._device().handle, _abi.DR_INTEGRATOR_DIRECT_ONE) == 14
    r_all = core.SamplerRenderer(core.LowDiscrepancySampler(cam, spp), cam, core.DirectLightingIntegrator(0, 5), core.EmissionIntegrator())
    assert not np.array_equal(r_all.render(scene).film, out.film)
    # serial reference stream -> host buffers
    rec = osc.render(ob.render_desc(r, sampler_mode=0), record=21 * 17 * spp, max_tail=200)
    assert rec["sample_vec"].shape[1] == 14
    r.sampler = core.HostBufferSampler(cam, spp, rec["pixel_xy"][::spp].copy(), rec["sample_vec"], rec["tail"])
    out2 = r.render(scene)
    assert np.array_equal(out2.film, rec["film"]) and np.array_equal(out2.rgb, rec["rgb"])


@pytest.mark.parametrize("ns", [(2, 1), (3, 5), (16, 1)])
def test_direct_lighting_with_several_samples_per_light(ob, gpu, ns):
    """UniformSampleAllLights with light.nSamples > 1 (integrator.dart:39-77): the LD sampler hands out
    roundSize(nSamples) entries per light slot (direct_lighting_integrator.dart:70-87); counter streams and the
    reference's serial stream replayed through host buffers."""
    e2 = scenes._quad((-9.9, -2, -2), (-9.9, 2, -2), (-9.9, 2, 2), (-9.9, -2, 2), (0.5, 0.5, 0.5),
                      core.DiffuseAreaLight((5.0, 9.0, 3.0), ns[1]))
    prims = scenes.cornell_prims(scenes.blob_prim(16, 8)) + [e2]
    next(gp for gp in prims if gp.areaLight is not None).areaLight.nSamples = ns[0]
    env = core.InfiniteAreaLight(scenes.SKY_TO_WORLD, (0.2, 0.3, 0.5), 2, None)
    film = core.ImageFilm(20, 16)
    cam = core.PerspectiveCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), 35.0, film)
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 8), cam, core.DirectLightingIntegrator(0, 5),
                             core.EmissionIntegrator())
    scene = scenes.make_scene(prims, env)
    out = r.render(scene)
    osc = ob.OracleScene(prims, env=env)
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    _check(out, ref)
    c, st = osc.counters(), r.last_stats
    for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris"):
        assert st[k] == c[k], k
    rp2 = [core.RoundUpPow2(n) for n in ns] + [2]
    nfloats = _abi.lib().dr_scene_sample_floats(scene._device().handle, _abi.DR_INTEGRATOR_DIRECT_ALL)
    assert nfloats == 5 + 2 + 6 * sum(rp2)
    # serial reference stream -> host buffers
    rec = osc.render(ob.render_desc(r, sampler_mode=0), record=21 * 17 * 8, max_tail=8)
    assert rec["sample_vec"].shape[1] == nfloats
    r.sampler = core.HostBufferSampler(cam, 8, rec["pixel_xy"][::8].copy(), rec["sample_vec"])
    out2 = r.render(scene)
    assert np.array_equal(out2.film, rec["film"]) and np.array_equal(out2.rgb, rec["rgb"])
    assert not np.array_equal(out2.film, out.film)


def test_task_and_tile_shards_sum_to_the_full_film(gpu):
    """GetSubWindow tasks (the reference's split) and round-robin tiles (the multi-GPU split): the sum of
    the shard films is bit-identical to the unsharded film (box filter => disjoint pixels)."""
    kw = dict(xres=70, yres=50, spp=8, blob=(24, 12))
    prims, mk = scenes.config("C2", **kw)
    scene = scenes.make_scene(prims)
    full = mk().render(scene).film
    for split in ("task", "tile"):
        acc = np.zeros_like(full)
        for i in range(4):
            extra = dict(taskNum=i, taskCount=4) if split == "task" else dict(tileRank=i, tileCount=4, tileSize=16)
            _, mk_i = scenes.config("C2", **kw, **extra)
            acc += mk_i().render(scene).film
        assert np.array_equal(acc, full), split


def test_render_is_deterministic_and_device_film_accumulates(gpu):
    import torch
    prims, mk = scenes.config("C2", xres=32, yres=32, spp=16, blob=(24, 12))
    scene = scenes.make_scene(prims)
    r = mk()
    a = r.render(scene)
    b = r.render(scene)
    assert np.array_equal(a.film, b.film)
    film = torch.zeros((32, 32, 4), dtype=torch.float32, device="cuda")
    rgb = torch.zeros((32, 32, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    r.render_device(scene, film.data_ptr(), stream)
    _abi.check(_abi.lib().dr_film_resolve_device(film.data_ptr(), 32 * 32, rgb.data_ptr(), stream))
    torch.cuda.synchronize()
    assert np.array_equal(film.cpu().numpy(), a.film) and np.array_equal(rgb.cpu().numpy(), a.rgb)


@pytest.mark.parametrize("impl", ["3", "5"])
def test_alternative_traversal_kernels_are_bit_exact_too(impl):
    """DARTRAY_TRACE_IMPL selects the sibling-pair kernels (3: k_trace3<0> + the 4-byte-entry any-hit kernel k_trace3a; 5: the
    closest-hit rays through k_trace3c, cold ray state in LDS) for A/B runs; all must reproduce the oracle's hits, visit counts
    and image exactly, like the default (2).  (Round 1's first kernels, round 3's 8-byte any-hit pair kernel and round 4's
    treelet-parked traversal -- measured negatives -- live in experiments/r06_*.diff.)"""
    extra = {}
    import subprocess
    import sys
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "import oracle.binding as ob\n"
        "from dartray_amd import _abi, scenes\n"
        "_abi.init(0)\n"
        "prims, mk = scenes.config('C2', xres=40, yres=40, spp=16, blob=(60, 30))\n"
        "r = mk(); out = r.render(scenes.make_scene(prims))\n"
        "osc = ob.OracleScene(prims); osc.counters(reset=True)\n"
        "ref = osc.render(ob.render_desc(r, sampler_mode=1)); c = osc.counters(); st = r.last_stats\n"
        "assert np.array_equal(out.film, ref['film'])\n"
        "assert all(st[k] == c[k] for k in ('closest_nodes', 'any_nodes', 'closest_tris', 'any_tris', 'closest_rays', 'any_rays'))\n"
        "print('OK')\n" % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ, DARTRAY_TRACE_IMPL=impl, **extra)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0 and "OK" in res.stdout, res.stderr[-2000:]


@pytest.mark.parametrize("kernels", [(2, 6), (3, 7), (5, 7), (2, 7)])
def test_any_hit_rays_far_child_first(ob, gpu, kernels):
    """BVHAccel.intersectP never modifies the ray (bvh_accel.dart:167-226): a leaf is reached iff its ancestors' slab tests pass, in
    whatever order the children are taken -- the boolean is the reference's in every order, only the work of a ray that FINDS an
    occluder changes.  Kernel ids 6 / 7 run the any-hit rays through k_trace<1> / k_trace3a with the far child first: the film must be
    the oracle's bit for bit, the closest-hit counters and the ray counts too, and the any-hit node visits / triangle tests must equal
    what the oracle counts for the SAME order (its order study walks every shadow ray of the render far child first as well)."""
    prims, mk = scenes.config("C2", xres=40, yres=40, spp=16, blob=(60, 30))
    r = mk()
    scene = scenes.make_scene(prims)
    dev = scene._device()
    assert dev.trace_kernels(kernels) == kernels
    _abi.check(_abi.lib().dr_set_option(b"TRACE_IMPL", b""))  # ("" hides the environment's value: the suite is also run with DARTRAY_TRACE_IMPL=5)
    try:
        _far_child_first_checks(ob, prims, r, scene, dev, kernels)
    finally:
        _abi.check(_abi.lib().dr_set_option(b"TRACE_IMPL", None))


def _far_child_first_checks(ob, prims, r, scene, dev, kernels):
    out = r.render(scene)
    info = dev.last_render_info()
    assert (info["closest_kernel"], info["any_hit_kernel"]) == kernels, info
    st = r.last_stats
    osc = ob.OracleScene(prims)
    osc.counters(reset=True)
    ob.order_study(True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    study = ob.order_study(False)
    c = osc.counters()
    assert np.array_equal(out.film, ref["film"]) and np.array_equal(out.rgb, ref["rgb"])
    for k in ("closest_rays", "any_rays", "closest_nodes", "closest_tris"):
        assert st[k] == c[k], k
    assert study["rays"] == c["any_rays"] and study["nodes_all"] == c["any_nodes"] and study["tris_all"] == c["any_tris"]  # the study's reference walk IS intersectP
    assert st["any_nodes"] == study["nodes_all"] - study["nodes_occ_ref"] + study["nodes_occ_far_first"]
    assert st["any_tris"] == study["tris_all"] - study["tris_occ_ref"] + study["tris_occ_far_first"]
    assert st["any_nodes"] != c["any_nodes"] and 0 < study["occluded"] < study["rays"]
    # Aggregate.intersectP on caller-supplied rays (dr_intersect): the same booleans in either order
    rng = np.random.default_rng(5)
    o = rng.uniform(-9, 9, (20000, 3)).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    far = dev.intersect(core.Ray(o, d, 1e-3, 25.0), any_hit=True)["prim"]
    dev.trace_kernels((kernels[0], 2 if kernels[1] == 6 else 3))
    assert np.array_equal(far, dev.intersect(core.Ray(o, d, 1e-3, 25.0), any_hit=True)["prim"]) and (far >= 0).any() and (far < 0).any()
    with pytest.raises(_abi.DartRayHipError):
        dev.trace_kernels((6, 2))   # the order is a property of any-hit rays only


def test_traversal_pilot_leaves_results_and_counters_untouched():
    """The first big render of a big scene times both traversal kernels on a sample of its own rays
    (dr_render_device's pilot) before rendering; forced here on a small scene: film and visit counters must equal
    the oracle's exactly, i.e. nothing of the pilot leaks into the film, the statistics or the sampler streams."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "import oracle.binding as ob\n"
        "from dartray_amd import _abi, scenes\n"
        "_abi.init(0)\n"
        "prims, mk = scenes.config('C2', xres=48, yres=40, spp=16, blob=(60, 30))\n"
        "r = mk(); scene = scenes.make_scene(prims); out = r.render(scene); out2 = r.render(scene)\n"
        "osc = ob.OracleScene(prims); osc.counters(reset=True)\n"
        "ref = osc.render(ob.render_desc(r, sampler_mode=1)); c = osc.counters(); st = r.last_stats\n"
        "assert np.array_equal(out.film, ref['film']) and np.array_equal(out2.film, ref['film'])\n"
        # (the pilot's first batch runs its any-hit rays far child first, and the scene may keep that order: the any-hit visits of rays that
        # find an occluder then differ from the reference order's, nothing else does)
        "assert all(st[k] == c[k] for k in ('closest_nodes', 'closest_tris', 'closest_rays', 'any_rays')), (st, c)\n"
        "far = scene._device().trace_kernels()[1] in (6, 7)\n"
        "scene._device().trace_kernels((scene._device().trace_kernels()[0], 2)); out3 = r.render(scene); st3 = r.last_stats\n"
        "assert np.array_equal(out3.film, ref['film']) and all(st3[k] == c[k] for k in ('closest_nodes', 'any_nodes', 'closest_tris', 'any_tris', 'closest_rays', 'any_rays')), (st3, c)\n"
        "print('OK')\n" % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ, DARTRAY_PILOT="force", DARTRAY_VERBOSE="1")
    env.pop("DARTRAY_TRACE_IMPL", None)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0 and "OK" in res.stdout, res.stderr[-2000:]
    assert res.stderr.count("traversal pilot") == 1, res.stderr[-2000:]   # once per scene


def test_both_state_layouts_render_the_same_film():
    """The kernels that touch the path state are built twice (LayoutOps, dr_api.hip): 64-slot runs, and sub-tiles of four slots
    (namespace sp4, picked by itself for plain-triangle scenes under an environment map).  DARTRAY_STATE_LAYOUT forces one: a
    closed scene, an open scene under a map, DirectLighting and a host-buffer replay give the same films either way."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from dartray_amd import core, scenes\n"
        "films = []\n"
        "prims, mk = scenes.config('C2', xres=96, yres=80, spp=64, blob=(40, 20))\n"
        "films.append(mk().render(scenes.make_scene(prims)).film)\n"
        "prims5, mk5 = scenes.config('C5', xres=96, yres=80, spp=64, yard=(6, 12), env_res=(128, 64))\n"
        "r5 = mk5(); films.append(r5.render(scenes.make_scene(prims5, r5.env)).film)\n"
        "prims1, mk1 = scenes.config('C1')\n"
        "r1 = mk1(); films.append(np.pad(r1.render(scenes.make_scene(prims1)).film, ((0, 16), (0, 32), (0, 0))))\n"
        "np.save(sys.argv[1], np.stack([f[:80, :96] for f in films]))\n" % ROOT)
    out = []
    for layout in ("64", "4"):
        path = os.path.join(ROOT, "gpurun_out", "film_layout%s.npy" % layout)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        res = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, DARTRAY_STATE_LAYOUT=layout), capture_output=True,
                             text=True, timeout=500)
        assert res.returncode == 0, res.stderr[-2000:]
        out.append(np.load(path))
        os.remove(path)
    assert np.array_equal(out[0], out[1]) and out[0].any()


def test_env_map_kernel_over_several_batches():
    """k_env (the environment-map work of a plain-triangle scene's path stages, with its own list per stage and batch) when
    a render needs several batches: the same film as the single-batch render."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from dartray_amd import scenes\n"
        "prims, mk = scenes.config('C5', xres=320, yres=256, spp=64, yard=(6, 12), env_res=(128, 64))\n"   # 5.2e6 samples
        "r = mk(); out = r.render(scenes.make_scene(prims, r.env))\n"
        "print('batches', r.last_stats['batches'])\n"
        "np.save(sys.argv[1], out.film)\n" % ROOT)
    films = []
    for bits in ("28", "20"):
        path = os.path.join(ROOT, "gpurun_out", "film_env_%s.npy" % bits)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        env = dict(os.environ, DARTRAY_BATCH_BITS=bits)
        res = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=500)
        assert res.returncode == 0, res.stderr[-2000:]
        if bits == "20":
            assert int(res.stdout.split("batches")[1].split()[0]) >= 4, res.stdout
        films.append(np.load(path))
        os.remove(path)
    assert np.array_equal(films[0], films[1])


def test_unread_sample_blocks_can_be_left_out():
    """The device sampler only produces the LD blocks some kernel reads (RenderParams.genMask: no time sample, no
    volume slots, no uComponents for single-lobe materials, no lens sample for a pinhole, no levels beyond maxDepth).
    Every (pixel, block) has its own keyed stream, so generating ALL blocks (DARTRAY_GEN_ALL_BLOCKS=1) must give the same
    film -- for a matte pinhole scene, a thin-lens camera, a mirror / glass scene and a depth-1 path."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from dartray_amd import core, scenes\n"
        "films = []\n"
        "prims, mk = scenes.config('C2', xres=96, yres=80, spp=64, blob=(60, 30))\n"
        "films.append(mk().render(scenes.make_scene(prims)).film)\n"
        "film = core.ImageFilm(96, 80)\n"
        "cam = core.PerspectiveCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), 35.0, film, lensradius=0.8, focaldistance=30.0)\n"
        "for depth in (5, 1):\n"
        "    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 64), cam, core.PathIntegrator(depth), core.EmissionIntegrator())\n"
        "    films.append(r.render(scenes.make_scene(prims)).film)\n"
        "prims2 = scenes.cornell_walls() + [scenes.emitter_quad(), core.GeometricPrimitive(scenes.blob_mesh(16, 8), core.MirrorMaterial((0.9, 0.9, 0.9)))]\n"
        "films.append(mk().render(scenes.make_scene(prims2)).film)\n"
        "prims3, mk3 = scenes.config('C2', xres=96, yres=80, spp=512, blob=(20, 10))\n"   # 512 spp: the two-wave sampler kernel
        "r3 = mk3(); r3.taskNum, r3.taskCount = 5, 60\n"
        "films.append(r3.render(scenes.make_scene(prims3)).film)\n"
        "np.save(sys.argv[1], np.stack([f[:80, :96] for f in films]))\n" % ROOT)
    out = []
    # third run: the device sampler draws its generator values in groups of plain steps and redoes a group the slow way
    # (Random.nextInt's retry loop) when a lane saw the one value in 2^32 that is redrawn; DARTRAY_GEN_SLOW_DRAWS=1 takes
    # that path for every group -- the streams must not change
    # fourth run: a stage's any-hit launch after its closest-hit launch instead of beside it (DARTRAY_OVERLAP_ANY=0); fifth: that and the slow draws
    for gen_all, slow, serial in ((False, False, False), (True, False, False), (False, True, False), (False, False, True), (False, True, True)):
        path = os.path.join(ROOT, "gpurun_out", "film_gen%d%d%d.npy" % (gen_all, slow, serial))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        env = dict(os.environ)
        for k in ("DARTRAY_GEN_ALL_BLOCKS", "DARTRAY_GEN_SLOW_DRAWS", "DARTRAY_OVERLAP_ANY"):
            env.pop(k, None)
        if gen_all:
            env["DARTRAY_GEN_ALL_BLOCKS"] = "1"
        if slow:
            env["DARTRAY_GEN_SLOW_DRAWS"] = "1"
        if serial:
            env["DARTRAY_OVERLAP_ANY"] = "0"
        res = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=500)
        assert res.returncode == 0, res.stderr[-2000:]
        out.append(np.load(path))
        os.remove(path)
    assert out[0].shape[0] == 5 and all(np.array_equal(out[0], o) for o in out[1:])
    assert out[0][..., :3].max() > 0


def test_invalid_arguments_raise(gpu):
    prims, mk = scenes.config("C1")
    scene = scenes.make_scene(prims)
    r = mk()
    r.sampler.samplesPerPixel = 3  # not a power of two
    with pytest.raises(_abi.DartRayHipError):
        r.render(scene)


@pytest.mark.parametrize("integ", [core.PathIntegrator(4), core.DirectLightingIntegrator(0, 5)])
def test_oren_nayar_matte(ob, gpu, integ):
    """MatteMaterial with sigma != 0 adds OrenNayar(Kd, sigma) instead of Lambertian (matte_material.dart:54-61,
    oren_nayar.dart); sigma is clamped to [0, 90] degrees."""
    prims = scenes.cornell_prims(scenes.blob_prim(16, 8))
    for gp, sig in zip(prims, [20.0, 0.0, 55.5, 120.0, 0.3, 5.0, 90.0, 33.0, 1.0]):
        gp.material = core.MatteMaterial(tuple(gp.material.Kd), sigma=sig)
    film = core.ImageFilm(32, 24)
    cam = core.PerspectiveCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), 35.0, film)
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 16), cam, integ, core.EmissionIntegrator())
    out = r.render(scenes.make_scene(prims))
    ref = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=1))
    _check(out, ref)
    lam = [core.GeometricPrimitive(gp.shape, core.MatteMaterial(tuple(gp.material.Kd)), gp.areaLight) for gp in prims]
    assert not np.array_equal(ob.OracleScene(lam).render(ob.render_desc(r, sampler_mode=1))["film"], ref["film"])


def test_full_size_c2_properties_and_sparse_parity(ob, gpu):
    """BASELINE config 1 at FULL size (1M triangles, 1024x1024, 256 spp; ~2.7e8 samples): the oracle cannot
    render it in reasonable time, so (a) size-independent properties: every film pixel's weightSum == spp
    up to the rare integral-imageX splats, the render is deterministic, the traversal counters are
    plausible; (b) sparse parity: 24 pixels spread over the image are rendered by the oracle with the same
    keyed streams and must match the GPU's pixels."""
    prims, mk = scenes.config("C2")
    scene = scenes.make_scene(prims)
    assert len(scene.aggregate.tri_idx) == 1000012
    r = mk()
    out = r.render(scene)
    st = r.last_stats
    w = out.film[..., 3]
    assert abs(float(w.sum()) - 1024 * 1024 * 256) <= 512 and np.mean(w == 256) > 0.9999
    assert st["film_samples"] == 1024 * 1024 * 256 and st["camera_samples"] == 1025 * 1025 * 256
    assert np.isfinite(out.rgb).all() and out.rgb.min() >= 0
    rng = np.random.Generator(np.random.PCG64(9))
    px = np.stack([rng.integers(0, 1024, 24), rng.integers(0, 1024, 24)], 1).astype(np.int32)
    px[:4] = [[512, 700], [300, 800], [700, 650], [512, 512]]  # on the blob and walls
    ref = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=1, pixels=px))
    for x, y in px:
        err = rel_err_image(out.rgb[y, x][None], ref["rgb"][y, x][None]).max()
        assert err <= REL_TOL, (x, y, out.rgb[y, x], ref["rgb"][y, x])
        assert np.array_equal(out.film[y, x], ref["film"][y, x])
    # The one-shot render above -- all a host of the reference ever does (Renderer.render once per task, dartray.dart:574) -- kept its
    # batches at 2^27 slots: a workspace of about a third of the steady state's; rendering the scene AGAIN grows it to one batch per image
    # (dr_scene_workspace_bytes; results do not depend on the batch size).
    if not any(os.environ.get(k) for k in ("DARTRAY_BATCH_BITS", "DARTRAY_PILOT", "DARTRAY_TRACE_IMPL", "DARTRAY_STATE_LAYOUT")):
        dev = scene._device()
        info, first_ws = dev.last_render_info(), dev.workspace_bytes()
        assert info["batches"] == 3 and info["pilot_batches"] in (3, 4) and st["batches"] >= 3
        assert 15e9 < first_ws < 30e9, first_ws
        out2 = r.render(scene)
        assert np.array_equal(out2.film, out.film)
        info2 = dev.last_render_info()
        assert info2["batches"] == 1 and info2["pilot_batches"] == 0 and r.last_stats["batches"] == 1
        assert 55e9 < dev.workspace_bytes() < 80e9 and dev.workspace_bytes() > 2.5 * first_ws


def test_state_layout_is_chosen_from_the_pilot_batch_densities():
    """the path-state layout of a scene comes from what its first pilot batch measured -- the share of
    slots still alive at the second bounce -- not from a property of its lights.  C2-like box: dense lists -> 64-slot runs; the
    same box with black surfaces (every path dies at its first vertex): four-slot sub-tiles; an open scene under an environment
    map: what its own density says.  The film is the same whichever layout runs (both forced through dr_set_option)."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "from dartray_amd import _abi, scenes, core\n"
        "_abi.init(0)\n"
        "lib = _abi.lib()\n"
        "def run(prims, mk, env=False):\n"
        "    r = mk(); scene = scenes.make_scene(prims, r.env if env else None)\n"
        "    out = r.render(scene); dev = scene._device(); lay, dens = dev.state_layout()\n"
        "    films = {}\n"
        "    for forced in (b'64', b'4'):\n"
        "        _abi.check(lib.dr_set_option(b'STATE_LAYOUT', forced)); films[forced] = r.render(scene).film\n"
        "    _abi.check(lib.dr_set_option(b'STATE_LAYOUT', None))\n"
        "    assert np.array_equal(films[b'64'], films[b'4']) and np.array_equal(out.film, films[b'4'])\n"
        "    assert dev.state_layout(64)[0] == 64 and dev.state_layout(0) == (0, -1.0)\n"
        "    return lay, dens\n"
        "prims, mk = scenes.config('C2', xres=48, yres=40, spp=16, blob=(60, 30))\n"
        "lay, dens = run(prims, mk); print('box', lay, dens); assert lay == 64 and dens > 0.5\n"
        "black = [core.GeometricPrimitive(p.shape, core.MatteMaterial((0.0, 0.0, 0.0)), p.areaLight) for p in prims]\n"
        "lay, dens = run(black, mk); print('black', lay, dens); assert lay == 4 and dens < 0.05\n"
        "prims, mk = scenes.config('C5', xres=48, yres=40, spp=16, yard=(4, 12), env_res=(64, 32))\n"
        "lay, dens = run(prims, mk, env=True); print('yard', lay, dens); assert lay == (4 if dens < 0.5 else 64) and 0.0 < dens < 1.0\n"
        "assert lib.dr_set_option(b'NO_SUCH_SWITCH', b'1') != 0\n"
        "print('OK')\n" % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ, DARTRAY_PILOT="force", DARTRAY_VERBOSE="1")
    for k in ("DARTRAY_TRACE_IMPL", "DARTRAY_STATE_LAYOUT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0 and "OK" in res.stdout, (res.stdout[-1000:], res.stderr[-2000:])
    assert res.stderr.count("state-layout pilot") == 3, res.stderr[-2000:]   # once per scene


def _serial_recording(ob, prims, r, spp, max_tail=40):
    """The reference's own sample sequence: ONE serial Random(taskNum) stream through the sampler and the integrator
    (sampler_renderer.dart:137), recorded by the oracle in tile order (the reference's default pixel order)."""
    r.sampler.pixelSampler = core.TilePixelSampler()
    fd = r.camera.film
    n = (fd.width + 1) * (fd.height + 1) * spp
    rec = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=0), record=n, max_tail=max_tail)
    return rec, n


def test_serial_stream_replay_across_several_batches(ob, gpu):
    """the host-buffer staging hand-over between batches (include/dartray_hip.h: the sample buffers of
    a batch are staged before the next batch reuses the area).  A recorded serial stream of 129 x 129 x 16 = 266 256 samples
    replayed with at most 2^16 slots per batch -- five batches -- equals the recording."""
    prims, mk = scenes.config("C2", xres=128, yres=128, spp=16, blob=(40, 20))
    r = mk()
    rec, n = _serial_recording(ob, prims, r, 16)
    assert n == 129 * 129 * 16
    lib = _abi.lib()
    r.sampler = core.HostBufferSampler(r.camera, 16, rec["pixel_xy"][::16], rec["sample_vec"], rec["tail"])
    scene = scenes.make_scene(prims)
    _abi.check(lib.dr_set_option(b"BATCH_BITS", b""))  # ("" hides the environment's value: the suite is also run with DARTRAY_BATCH_BITS=16)
    whole = r.render(scene)
    assert r.last_stats["batches"] == 1
    try:
        _abi.check(lib.dr_set_option(b"BATCH_BITS", b"16"))
        out = r.render(scene)
        assert r.last_stats["batches"] == 5
    finally:
        _abi.check(lib.dr_set_option(b"BATCH_BITS", None))
    assert np.array_equal(out.film, rec["film"]) and np.array_equal(out.rgb, rec["rgb"])
    assert np.array_equal(whole.film, rec["film"])
    # the PACKED form of the tail (DrRenderDesc.tail_offsets: only the values a sample drew cross the host link), whole and
    # across the same five batches -- a batch's runs are one contiguous piece of the packed array
    packed = core.HostBufferSampler(r.camera, 16, rec["pixel_xy"][::16], rec["sample_vec"], rec["tail"], rec["tail_count"])
    assert packed.tail.size == int(rec["tail_count"].sum()) < rec["tail"].size // 3 and packed.tail_offsets[-1] == packed.tail.size
    r.sampler = packed
    assert np.array_equal(r.render(scene).film, rec["film"])
    try:
        _abi.check(lib.dr_set_option(b"BATCH_BITS", b"16"))
        out = r.render(scene)
        assert r.last_stats["batches"] == 5
    finally:
        _abi.check(lib.dr_set_option(b"BATCH_BITS", None))
    assert np.array_equal(out.film, rec["film"]) and np.array_equal(out.rgb, rec["rgb"])
    # a run that is shorter than what the path draws reads 0.0 past its end, exactly like the zero fill of the fixed form
    cut = np.maximum(rec["tail_count"] - 3, 0)
    fixed = rec["tail"].copy()
    fixed[np.arange(fixed.shape[1])[None, :] >= cut[:, None]] = 0.0
    r.sampler = core.HostBufferSampler(r.camera, 16, rec["pixel_xy"][::16], rec["sample_vec"], fixed)
    a = r.render(scene).film
    r.sampler = core.HostBufferSampler(r.camera, 16, rec["pixel_xy"][::16], rec["sample_vec"], rec["tail"], cut)
    assert np.array_equal(r.render(scene).film, a) and not np.array_equal(a, rec["film"])


def test_serial_stream_replay_of_a_million_samples(ob, gpu):
    """"identical Sampler RNG seeds" at scale -- 129 x 129 x 64 = 1 065 024 samples of the C2-type
    scene (3 600-triangle blob) drawn from the reference's one serial generator, replayed through DR_SAMPLER_HOST_BUFFER in
    batches of 2^18: the film and the image equal the serial recording bit for bit."""
    prims, mk = scenes.config("C2", xres=128, yres=128, spp=64, blob=(60, 30))
    r = mk()
    rec, n = _serial_recording(ob, prims, r, 64)
    assert n == 1065024
    lib = _abi.lib()
    r.sampler = core.HostBufferSampler(r.camera, 64, rec["pixel_xy"][::64], rec["sample_vec"], rec["tail"], rec["tail_count"])  # packed tail
    try:
        _abi.check(lib.dr_set_option(b"BATCH_BITS", b"18"))
        out = r.render(scenes.make_scene(prims))
        assert r.last_stats["batches"] == 5
    finally:
        _abi.check(lib.dr_set_option(b"BATCH_BITS", None))
    assert np.array_equal(out.film, rec["film"]) and np.array_equal(out.rgb, rec["rgb"])
    assert (rec["tail_count"] > 0).mean() > 0.2  # bounces >= 3 and Russian roulette draw from the recorded tail
