"""Orthographic and environment cameras (lib/cameras/*.dart): oracle KATs and the host-side camera set-up.  CPU only;
tests/test_gpu_cameras.py renders through them."""
import ctypes as C
import math

import numpy as np
import pytest

from dartray_amd import core, pbrt, scenes


def _ray(ob, cam, x, y, lu=0.5, lv=0.5):
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 1), cam, core.PathIntegrator(1), core.EmissionIntegrator())
    rd = ob.render_desc(r, sampler_mode=1)
    out = np.zeros(6)
    ob.lib().orc_generate_ray(C.byref(rd), float(x), float(y), float(lu), float(lv), out.ctypes.data)
    return out[:3], out[3:]


def test_orthographic_rays_are_parallel_and_span_the_screen_window(ob):
    film = core.ImageFilm(40, 20)
    cam = core.OrthographicCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), film)
    # screen window [-2, 2] x [-1, 1] (frame = 2): raster (0, 0) is the top-left corner (x = -2, y = +1)
    for (x, y), want in (((0, 0), (-2, 1)), ((40, 20), (2, -1)), ((20, 10), (0, 0)), ((10, 15), (-1, -0.5))):
        o, d = _ray(ob, cam, x, y)
        assert np.allclose(d, (0, 0, 1), atol=1e-7)
        assert np.allclose(o, (want[0], want[1], -35), atol=1e-5), (o, want)
    # host matrices == the oracle's own composition (f32 products)
    r2c = np.zeros(16, np.float32)
    ob.lib().orc_camera_setup_ortho(40, 20, r2c.ctypes.data)
    assert np.allclose(cam.rasterToCamera.reshape(-1), r2c, rtol=2e-6, atol=1e-9)
    assert cam.cameraType == 1


def test_orthographic_depth_of_field_replaces_the_origin(ob):
    """orthographic_camera.dart:60-76 sets ray.origin = (lensU, lensV, 0): the raster point only survives in Pfocus."""
    film = core.ImageFilm(16, 16)
    cam = core.OrthographicCamera.lookAt((0, 0, 0), (0, 0, 1), (0, 1, 0), film, lensradius=0.5, focaldistance=4.0)
    o, d = _ray(ob, cam, 12, 4, lu=0.5, lv=0.5)          # ConcentricSampleDisk(0.5, 0.5) = (0, 0)
    assert np.allclose(o, (0, 0, 0), atol=1e-7)
    pf = np.array([0.5, 0.5, 4.0])                         # Pcamera (0.5, 0.5, 0) + 4 * (0, 0, 1)
    assert np.allclose(d, pf / np.linalg.norm(pf), atol=1e-6)
    o, d = _ray(ob, cam, 12, 4, lu=1.0, lv=0.5)           # lens point (+r, 0)
    assert np.allclose(o, (0.5, 0, 0), atol=1e-7)


def test_environment_camera_directions(ob):
    film = core.ImageFilm(64, 32)
    cam = core.EnvironmentCamera.lookAt((1, 2, 3), (1, 2, 4), (0, 1, 0), film)
    for x, y in ((0, 16), (16, 16), (32, 8), (5.5, 30.25)):
        theta, phi = math.pi * y / 32, 2 * math.pi * x / 64
        want = (math.sin(theta) * math.cos(phi), math.cos(theta), math.sin(theta) * math.sin(phi))
        o, d = _ray(ob, cam, x, y)
        assert np.allclose(o, (1, 2, 3), atol=1e-7)
        # lookAt along +z with up = +y: camera axes == world axes up to the LookAt convention (x -> left)
        c2w = cam.cameraToWorld[:3, :3].astype(np.float64)
        assert np.allclose(d, c2w @ np.array(want), atol=1e-6)
    assert cam.cameraType == 2 and cam.lensRadius == 0.0


def test_loader_creates_the_cameras():
    head = 'LookAt 0 0 -35 0 0 0 0 1 0\nFilm "image" "integer xresolution" [32] "integer yresolution" [16]\n'
    body = 'WorldBegin\nShape "sphere"\nWorldEnd\n'
    r = pbrt.loads(head + 'Camera "orthographic" "float screenwindow" [-3 3 -2 2] "float lensradius" [0.1]\n' + body).rendererObject
    assert isinstance(r.camera, core.OrthographicCamera) and r.camera.lensRadius == np.float32(0.1)
    want = core.OrthographicCamera(r.camera.cameraToWorld, [-3, 3, -2, 2], 0, 1, 0.1, 1e30, r.camera.film)
    assert np.array_equal(r.camera.rasterToCamera, want.rasterToCamera)
    r = pbrt.loads(head + 'Camera "environment"\n' + body).rendererObject
    assert isinstance(r.camera, core.EnvironmentCamera)
    assert isinstance(core.Plugin.get("camera", "orthographic"), type) and core.Plugin.get("camera", "environment") is core.EnvironmentCamera
