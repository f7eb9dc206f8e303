"""Oracle checks for the quadric shapes (shapes/sphere.dart, shapes/disk.dart):
analytic known answers, clipping parameters, BVH against brute force, and the disk emitter's sampling /
pdf path against a finely tessellated triangle fan.  CPU only."""
import math

import numpy as np
import pytest

from dartray_amd import core, pbrt, scenes
from util import aggregate_test_rays, quadric_prims

I = pbrt.Transform()
MAT = core.MatteMaterial((0.5, 0.5, 0.5))


def _hits(ob, shape, o, d, tmin=0.0, tmax=np.inf, any_hit=False):
    osc = ob.OracleScene([core.GeometricPrimitive(shape, MAT)])
    o, d = np.atleast_2d(np.asarray(o, np.float32)), np.atleast_2d(np.asarray(d, np.float32))
    n = len(o)
    rays = ob.make_rays(o, d, np.full(n, tmin, np.float64), np.full(n, tmax, np.float64))
    return osc.intersect(rays, any_hit=any_hit)


def test_sphere_known_answers(ob):
    s = core.Sphere(I.m, I.mInv, False, 2.0)
    h = _hits(ob, s, [[0, 0, -5], [0, 0, 0], [0, 3, -5], [0, 0, -5], [0, 0, -5]],
              [[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, -1], [0, 1, 0]])
    assert h["prim"].tolist() == [0, 0, -1, -1, -1]
    assert h["t"][0] == 3.0 and h["t"][1] == 2.0          # entry from outside, exit from inside
    assert _hits(ob, s, [0, 0, -5], [0, 0, 1], tmax=2.9)["prim"][0] == -1
    assert _hits(ob, s, [0, 0, -5], [0, 0, 1], tmin=3.5)["t"][0] == 7.0      # t0 < tmin: the far root
    assert _hits(ob, s, [0, 0, -5], [0, 0, 1], tmin=3.5, tmax=6.9)["prim"][0] == -1
    assert _hits(ob, s, [0, 0, -5], [0, 0, 1], any_hit=True)["prim"][0] >= 0
    # unnormalised direction: t is in units of |d| (rays are not normalised by the traversal)
    assert _hits(ob, s, [0, 0, -5], [0, 0, 2])["t"][0] == 1.5


def test_transformed_sphere_hits_lie_on_the_ellipsoid(ob):
    t = pbrt.Transform.Translate(1, -2, 3) * pbrt.Transform.Rotate(33, 1, 2, 3) * pbrt.Transform.Scale(1, 2, 0.5)
    s = core.Sphere(t.m, t.mInv, False, 1.5)
    rng = np.random.Generator(np.random.PCG64(1))
    o = (rng.random((4000, 3)) * 16 - 8).astype(np.float32)
    d = (np.array([1, -2, 3]) - o + rng.normal(0, 0.7, (4000, 3))).astype(np.float32)
    h = _hits(ob, s, o, d)
    hit = h["prim"] >= 0
    assert hit.sum() > 1000
    p = o[hit].astype(np.float64) + d[hit].astype(np.float64) * h["t"][hit, None]
    q = core.transform_points(t.mInv, p).astype(np.float64)
    assert np.abs(np.linalg.norm(q, axis=1) - 1.5).max() < 1e-4
    lo, hi = s.worldBound()
    assert (p >= lo - 1e-4).all() and (p <= hi + 1e-4).all()


def test_clipped_sphere_falls_through_to_the_far_side(ob):
    s = core.Sphere(I.m, I.mInv, False, 2.0, z0=-1.0, z1=1.0)      # caps removed
    assert _hits(ob, s, [0, 0, -5], [0, 0, 1])["prim"][0] == -1       # through both holes
    h = _hits(ob, s, [0, 0, -5], [0, 0.35, 1])                        # enters through the hole, hits the inside
    assert h["prim"][0] == 0 and h["t"][0] > 5.0
    half = core.Sphere(I.m, I.mInv, False, 2.0, phiMax=180.0)         # y >= 0 half
    assert _hits(ob, half, [0, 5, 0], [0, -1, 0])["t"][0] == 3.0
    assert _hits(ob, half, [0, -5, 0], [0, 1, 0])["t"][0] == 7.0      # near side clipped away: far root
    assert _hits(ob, half, [1, -5, 0], [0, 1, 0], tmax=6.0)["prim"][0] == -1


def test_disk_known_answers(ob):
    d = core.Disk(I.m, I.mInv, False, height=1.0, radius=2.0, innerRadius=0.5, phiMax=270.0)
    o = [[1, 0.1, -3], [0.1, 0.1, -3], [3, 0, -3], [1, -1, -3], [1, 0.1, -3], [-1, 1, 5]]
    dirs = [[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [1, 0, 1e-8], [0, 0, -1]]
    h = _hits(ob, d, o, dirs)
    assert h["prim"].tolist() == [0, -1, -1, -1, -1, 0]   # inside; inner hole; outside; phi > phiMax; parallel; from above
    assert h["t"][0] == 4.0 and h["t"][5] == 4.0
    assert _hits(ob, d, [1, 0.1, -3], [0, 0, 1], tmax=3.9)["prim"][0] == -1
    assert _hits(ob, d, [1, 0.1, -3], [0, 0, 1], tmin=4.1)["prim"][0] == -1
    t = pbrt.Transform.Translate(0, 9.9, 0) * pbrt.Transform.Rotate(90, 1, 0, 0)
    lamp = core.Disk(t.m, t.mInv, False, 0.0, 3.0)
    assert abs(_hits(ob, lamp, [1, 0, 1], [0, 1, 0])["t"][0] - 9.9) < 1e-5
    lo, hi = lamp.worldBound()
    assert np.allclose(lo, [-3, 9.9, -3], atol=1e-5) and np.allclose(hi, [3, 9.9, 3], atol=1e-5)


def test_bvh_equals_brute_force_with_quadrics_in_the_leaves(ob):
    prims = quadric_prims()
    osc = ob.OracleScene(prims)
    nodes = osc.bvh()[0]
    assert len(nodes) >= 2 * (len(prims) // 4)
    o, d, tmin, tmax = aggregate_test_rays(nodes[0]["bmin"], nodes[0]["bmax"], 30000, seed=9)
    tmin[:] = 1e-3
    rays = ob.make_rays(o, d, tmin, tmax)
    a, b = osc.intersect(rays), osc.intersect(rays, brute=True)
    # ties between coincident surfaces may pick another primitive; the distance must agree
    assert np.array_equal(a["prim"] >= 0, b["prim"] >= 0)
    assert np.array_equal(a["t"], b["t"])
    assert np.array_equal(osc.intersect(rays, any_hit=True)["prim"] >= 0, a["prim"] >= 0)
    assert (a["prim"] >= 0).mean() > 0.05


def _render(ob, prims, integ, spp=64, res=24):
    film = core.ImageFilm(res, res)
    cam = core.PerspectiveCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), 35.0, film)
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, spp), cam, integ, core.EmissionIntegrator())
    return ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=1))["rgb"]


@pytest.mark.parametrize("integ", [core.DirectLightingIntegrator(0, 5), core.PathIntegrator(3)])
def test_disk_emitter_agrees_with_a_tessellated_disk(ob, integ):
    """Disk.sample / Shape.pdf2 / ShapeSet through EstimateDirect: the image under a disk light equals, up to
    Monte-Carlo noise, the image under the same disk cut into a 96-triangle fan (the triangle path is pinned
    by the other oracle tests)."""
    t = pbrt.Transform.Translate(0, 9.9, 0) * pbrt.Transform.Rotate(90, 1, 0, 0)
    L = (30.0, 30.0, 30.0)
    disk = core.GeometricPrimitive(core.Disk(t.m, t.mInv, False, 0.0, 3.0), MAT, core.DiffuseAreaLight(L, 1))
    n = 96
    ang = np.arange(n) * 2 * math.pi / n
    P = np.concatenate([[[0, 0, 0]], np.stack([3 * np.cos(ang), 3 * np.sin(ang), 0 * ang], 1)])
    idx = np.array([[0, 1 + k, 1 + (k + 1) % n] for k in range(n)], np.uint32)
    fan = core.GeometricPrimitive(core.TriangleMesh(idx, t.transformPoints(P.astype(np.float32))), MAT,
                                  core.DiffuseAreaLight(L, 1))
    walls = scenes.cornell_walls()
    a = _render(ob, walls + [disk], integ)
    b = _render(ob, walls + [fan], integ)
    assert a.mean() > 0.05
    assert abs(a.mean() / b.mean() - 1.0) < 0.02
    blur = lambda x: x.reshape(6, 4, 6, 4, 3).mean(axis=(1, 3))
    assert np.abs(blur(a) / blur(b) - 1.0).max() < 0.2


def test_reverse_orientation_flips_the_emitting_side(ob):
    t = pbrt.Transform.Translate(0, 5, 0) * pbrt.Transform.Rotate(90, 1, 0, 0)   # object +z -> world -y
    floor = scenes.floor_quad()
    imgs = []
    for ro in (False, True):
        lamp = core.GeometricPrimitive(core.Disk(t.m, t.mInv, ro, 0.0, 3.0), MAT, core.DiffuseAreaLight((20, 20, 20), 1))
        imgs.append(_render(ob, [floor, lamp], core.DirectLightingIntegrator(0, 5), spp=8, res=16))
    lit = [im[12:].mean() for im in imgs]   # bottom rows see the floor
    assert (lit[0] > 0.1) != (lit[1] > 0.1)


@pytest.mark.parametrize("integ", [core.DirectLightingIntegrator(0, 5), core.PathIntegrator(1)])
def test_spherical_emitter_irradiance_is_closed_form(ob, integ):
    """A Lambertian spherical emitter of radius R whose centre is at distance d straight above a matte floor
    point gives E = pi L (R/d)^2 there, so the floor's radiance is Kd L (R/d)^2: checks Sphere.sample2 /
    pdf2 (cone sampling) and the MIS combination."""
    R, d, L, kd = 1.0, 6.0, 50.0, 0.8
    t = pbrt.Transform.Translate(0, d, 0)
    lamp = core.GeometricPrimitive(core.Sphere(t.m, t.mInv, False, R), MAT, core.DiffuseAreaLight((L, L, L), 1))
    floor = scenes._quad((-60, 0, -60), (-60, 0, 60), (60, 0, 60), (60, 0, -60), (kd, kd, kd))
    film = core.ImageFilm(4, 4)
    cam = core.PerspectiveCamera.lookAt((0.0, 3.0, -0.01), (0, 0, 0), (0, 0, 1), 0.5, film)   # looks at the floor point below
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 256), cam, integ, core.EmissionIntegrator())
    img = ob.OracleScene([floor, lamp]).render(ob.render_desc(r, sampler_mode=1))["rgb"]
    expect = kd * L * (R / d) ** 2
    assert abs(img.mean() / expect - 1.0) < 0.02, (img.mean(), expect)
