"""Procedural scenes of BASELINE.json's configs.

All geometry is synthetic and seeded; vertices are f32 world-space.  The Cornell
box follows web/scenes/cornell-path.pbrt:5-31 semantics: 20x20x20 box centred at
the origin, open towards -z, camera LookAt 0 0 -35 -> 0 0 0, fov 35, one quad
emitter L = 36 under the ceiling whose own surface is matte Kd 0.5
(graphics_state.dart:25, matte_material.dart:68).
"""
import math

import numpy as np

from .core import (BVHAccel, BoxFilter, DiffuseAreaLight, DirectLightingIntegrator, EmissionIntegrator,
                   GeometricPrimitive, ImageFilm, InfiniteAreaLight, LowDiscrepancySampler, MatteMaterial,
                   PathIntegrator, PerspectiveCamera, SamplerRenderer, Scene, TriangleMesh)

WHITE = (0.75, 0.75, 0.75)
RED = (0.48, 0.1125, 0.075)
GREEN = (0.1125, 0.375, 0.1125)


def _quad(p0, p1, p2, p3, Kd, light=None):
    """Quad p0 p1 p2 p3 as triangles (p0,p1,p2), (p0,p2,p3)."""
    P = np.array([p0, p1, p2, p3], dtype=np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)
    mesh = TriangleMesh(idx, P)
    return GeometricPrimitive(mesh, MatteMaterial(Kd), light)


def emitter_quad(L=(36.0, 36.0, 36.0), half=3.0, y=9.9):
    """6x6 quad facing down: area lights are one-sided (diffuse_area_light.dart:44-46) and the
    normal is normalize((p2-p1) x (p3-p1)), so the winding makes it (0,-1,0)."""
    h = half
    return _quad((-h, y, -h), (h, y, -h), (h, y, h), (-h, y, h), (0.5, 0.5, 0.5), DiffuseAreaLight(L, 1))


def floor_quad():
    return _quad((-10, -10, -10), (10, -10, -10), (10, -10, 10), (-10, -10, 10), WHITE)


def cornell_walls():
    s = 10.0
    return [
        floor_quad(),
        _quad((-s, s, -s), (-s, s, s), (s, s, s), (s, s, -s), WHITE),        # ceiling
        _quad((-s, -s, s), (s, -s, s), (s, s, s), (-s, s, s), WHITE),         # back wall (z = +10)
        _quad((-s, -s, -s), (-s, -s, s), (-s, s, s), (-s, s, -s), RED),       # left wall (x = -10)
        _quad((s, -s, -s), (s, s, -s), (s, s, s), (s, -s, s), GREEN),         # right wall (x = +10)
    ]


def blob_mesh(segments=1000, rows=500, radius=5.0, centre=(0.0, -5.0, 0.0)):
    """Displaced UV sphere, closed, no degenerate triangles:
    (rows-1) bands x segments x 2 + 2 polar fans x segments  ==  2 * rows * segments triangles
    (1000 x 500 -> exactly 1 000 000).  Radial displacement
    1 + 0.15 sin(7 theta) sin(5 phi) + 0.05 sin(31 theta + 1) sin(29 phi + 2)."""
    j = np.arange(rows, dtype=np.float64)
    theta = math.pi * (j + 0.5) / rows                  # polar angle, poles excluded
    i = np.arange(segments, dtype=np.float64)
    phi = 2.0 * math.pi * i / segments
    th, ph = np.meshgrid(theta, phi, indexing="ij")

    def disp(t, p):
        return 1.0 + 0.15 * np.sin(7 * t) * np.sin(5 * p) + 0.05 * np.sin(31 * t + 1) * np.sin(29 * p + 2)

    r = radius * disp(th, ph)
    x = r * np.sin(th) * np.cos(ph)
    y = r * np.cos(th)
    z = r * np.sin(th) * np.sin(ph)
    grid = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    top = np.array([[0.0, radius * disp(0.0, 0.0), 0.0]])
    bottom = np.array([[0.0, -radius * disp(math.pi, 0.0), 0.0]])
    P = (np.concatenate([grid, top, bottom]) + np.asarray(centre, dtype=np.float64)).astype(np.float32)
    nt, nb = rows * segments, rows * segments + 1
    a = (np.arange(rows - 1)[:, None] * segments + np.arange(segments)[None, :])
    b = (np.arange(rows - 1)[:, None] * segments + (np.arange(segments)[None, :] + 1) % segments)
    c, d = a + segments, b + segments
    bands = np.concatenate([np.stack([a, c, b], -1).reshape(-1, 3), np.stack([b, c, d], -1).reshape(-1, 3)])
    s0 = np.arange(segments)
    s1 = (s0 + 1) % segments
    fan_top = np.stack([np.full(segments, nt), s0, s1], -1)
    last = (rows - 1) * segments
    fan_bot = np.stack([np.full(segments, nb), last + s1, last + s0], -1)
    idx = np.concatenate([bands, fan_top, fan_bot]).astype(np.uint32)
    assert len(idx) == 2 * rows * segments
    return TriangleMesh(idx, P)


def hairball_mesh(strands=10000, segments=500, seed=0x9E3779B97F4A7C15, step=0.02, jitter=0.3, half_width=0.002,
                  scale=8.0):
    """Procedural hairball: each strand is a random walk from a point on the unit sphere, drawn as a
    ribbon of 2 triangles per segment (10 000 x 500 x 2 = 10 000 000 triangles), scaled so that it
    fits inside the Cornell box (max |coordinate| == `scale`)."""
    rng = np.random.Generator(np.random.PCG64(seed & 0xFFFFFFFFFFFFFFFF))
    v = rng.normal(size=(strands, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    pos = v.copy()
    dirn = v.copy()
    pts = np.empty((strands, segments + 1, 3), dtype=np.float64)
    pts[:, 0] = pos
    for s in range(segments):
        dirn = dirn + jitter * rng.normal(size=(strands, 3))
        dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
        pos = pos + step * dirn
        pts[:, s + 1] = pos
    side = np.cross(pts[:, 1:] - pts[:, :-1], pts[:, :-1])
    side = np.concatenate([side, side[:, -1:]], axis=1)
    side /= np.maximum(np.linalg.norm(side, axis=2, keepdims=True), 1e-12)
    # fit the ball inside the 20^3 box: the farthest strand point lands at |coordinate| == scale
    fit = scale / np.abs(pts).max()
    left = (pts - half_width * side) * fit
    right = (pts + half_width * side) * fit
    P = np.stack([left, right], axis=2).reshape(-1, 3).astype(np.float32)  # vertex (strand, k, side)
    base = (np.arange(strands)[:, None] * (segments + 1) + np.arange(segments)[None, :]) * 2
    a, b, c, d = base, base + 1, base + 2, base + 3
    idx = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([b, d, c], -1).reshape(-1, 3)])
    return TriangleMesh(idx.astype(np.uint32), P)


def cornell_c1_prims():
    """Config C1: floor quad (2 matte triangles) + the quad emitter."""
    return [floor_quad(), emitter_quad()]


def cornell_prims(extra=None):
    prims = cornell_walls() + [emitter_quad()]
    if extra is not None:
        prims.append(extra)
    return prims


def blob_prim(segments=1000, rows=500):
    return GeometricPrimitive(blob_mesh(segments, rows), MatteMaterial((0.48, 0.48, 0.48)))


def hairball_prim(strands=10000, segments=500):
    return GeometricPrimitive(hairball_mesh(strands, segments), MatteMaterial((0.48, 0.48, 0.48)))


def make_scene(prims, env=None):
    """Scene with one DiffuseAreaLight per emissive shape (dartray.dart:398-401) followed by the optional
    InfiniteAreaLight."""
    accel = BVHAccel(prims)
    return Scene(accel, accel.lights() + ([env] if env is not None else []))


# light space z (theta = 0) -> world +y: columns are the images of the light axes (x -> x, y -> -z, z -> y)
SKY_TO_WORLD = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)


def sky_env(width=1024, height=512, L=(1.0, 1.0, 1.0), nsamples=1):
    """Procedural lat-long radiance map: sky gradient over a dim ground plus a sun lobe (analytic, f32)."""
    v = (np.arange(height) + 0.5) / height * math.pi          # theta from the zenith
    u = (np.arange(width) + 0.5) / width * 2.0 * math.pi      # phi
    th, ph = np.meshgrid(v, u, indexing="ij")
    cz = np.cos(th)
    sky = np.stack([0.25 + 0.15 * cz, 0.45 + 0.2 * cz, 0.9 + 0.0 * cz], -1) * (0.35 + 0.65 * np.clip(cz, 0, 1))[..., None]
    ground = np.array([0.08, 0.07, 0.06])
    base = np.where((cz > 0)[..., None], sky, ground)
    sun_t, sun_p = math.radians(40.0), math.radians(200.0)
    sd = np.array([math.sin(sun_t) * math.cos(sun_p), math.sin(sun_t) * math.sin(sun_p), math.cos(sun_t)])
    d = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), cz], -1)
    lobe = np.exp(400.0 * (d @ sd - 1.0))[..., None] * np.array([40.0, 36.0, 30.0])
    return InfiniteAreaLight(SKY_TO_WORLD, L, nsamples, (base + lobe).astype(np.float32))


def courtyard_prims(patches=16, cells=125, size=100.0, seed=42):
    """San-Miguel-class synthetic courtyard (config C5): a patches x patches grid of displaced height-field
    patches (cells x cells x 2 triangles each; 16 x 16 x 125 x 125 x 2 = 8 000 000), one box column per patch,
    8 quad emitters (L = 20) and 12 matte Kd values drawn from a seeded generator."""
    rng = np.random.Generator(np.random.PCG64(seed))
    kds = 0.05 + 0.75 * rng.random((12, 3))
    pick = rng.integers(0, 12, (patches, patches))
    prims = []
    half = size / 2.0
    step = size / patches

    def height(x, z):
        return 0.6 * np.sin(0.35 * x) * np.cos(0.27 * z) + 0.25 * np.sin(1.7 * x + 0.3) * np.sin(1.3 * z + 1.1)

    g = np.arange(cells + 1) / cells
    ii, jj = np.meshgrid(np.arange(cells), np.arange(cells), indexing="ij")
    a = (ii * (cells + 1) + jj).reshape(-1)
    idx = np.concatenate([np.stack([a, a + 1, a + cells + 1], -1), np.stack([a + 1, a + cells + 2, a + cells + 1], -1)]).astype(np.uint32)
    for pi in range(patches):
        for pj in range(patches):
            x = -half + (pi + g) * step
            z = -half + (pj + g) * step
            X, Z = np.meshgrid(x, z, indexing="ij")
            P = np.stack([X, height(X, Z), Z], -1).reshape(-1, 3).astype(np.float32)
            prims.append(GeometricPrimitive(TriangleMesh(idx, P), MatteMaterial(kds[pick[pi, pj]])))
    # columns: one box (12 triangles) per patch corner
    bx = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=np.float64)
    bi = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6],
                   [3, 0, 4], [3, 4, 7]], dtype=np.uint32)
    for pi in range(patches):
        for pj in range(patches):
            cx, cz = -half + (pi + 0.5) * step, -half + (pj + 0.5) * step
            P = (bx * np.array([0.8, 10.0, 0.8]) + np.array([cx - 0.4, float(height(cx, cz)) - 0.5, cz - 0.4])).astype(np.float32)
            prims.append(GeometricPrimitive(TriangleMesh(bi, P), MatteMaterial(kds[pick[pi, pj] - 1])))
    # 8 quad emitters on a ring, facing down
    for k in range(8):
        ang = 2.0 * math.pi * k / 8
        cx, cz, h = 0.3 * size * math.cos(ang), 0.3 * size * math.sin(ang), 1.5
        prims.append(_quad((cx - h, 15.0, cz - h), (cx + h, 15.0, cz - h), (cx + h, 15.0, cz + h), (cx - h, 15.0, cz + h),
                           (0.5, 0.5, 0.5), DiffuseAreaLight((20.0, 20.0, 20.0), 1)))
    return prims


def cornell_camera(xres, yres):
    film = ImageFilm(xres, yres, BoxFilter(0.5, 0.5))
    return PerspectiveCamera.lookAt((0.0, 0.0, -35.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 35.0, film)


def config(name, xres=None, yres=None, spp=None, blob=(1000, 500), hair=(10000, 500), yard=(16, 125), env_res=(1024, 512),
           seed=5489, **renderer_kw):
    """(scene prims, renderer factory) for BASELINE.json's configs: 'C1', 'C2', 'C4', 'C5' (C3 is C2 at 4096^2 x
    1024 spp).  For C5 the InfiniteAreaLight is attached to the factory as `renderer.env` (make_scene(prims, env))."""
    if name == "C1":
        prims = cornell_c1_prims()
        xres, yres, spp = xres or 64, yres or 64, spp or 4
        integ = DirectLightingIntegrator(0, 5)
    elif name == "C2":
        prims = cornell_prims(blob_prim(*blob))
        xres, yres, spp = xres or 1024, yres or 1024, spp or 256
        integ = PathIntegrator(5)
    elif name == "C4":
        prims = cornell_prims(hairball_prim(*hair))
        xres, yres, spp = xres or 1024, yres or 1024, spp or 64
        integ = PathIntegrator(5)
    elif name == "C5":
        prims = courtyard_prims(*yard)
        xres, yres, spp = xres or 2048, yres or 2048, spp or 512
        integ = PathIntegrator(8)
    else:
        raise ValueError(name)
    if name == "C5":
        film = ImageFilm(xres, yres, BoxFilter(0.5, 0.5))
        cam = PerspectiveCamera.lookAt((0.0, 14.0, -72.0), (0.0, 3.0, 0.0), (0.0, 1.0, 0.0), 45.0, film)
        env = sky_env(*env_res)
    else:
        cam = cornell_camera(xres, yres)
        env = None

    def renderer():
        r = SamplerRenderer(LowDiscrepancySampler(cam, spp, seed), cam, integ, EmissionIntegrator(), **renderer_kw)
        r.env = env
        return r

    return prims, renderer
