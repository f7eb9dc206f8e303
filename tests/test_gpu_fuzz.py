"""Differential fuzzing of the whole path: random small scenes mixing every plugin on the path (triangle meshes with
and without N / S / uv, spheres, disks, matte / Oren-Nayar / plastic / mirror / glass, area lights on triangles,
disks and spheres, point / spot / distant / infinite lights, both integrators, odd resolutions and sample counts)
rendered on the GPU and by the oracle from the same inputs."""
import numpy as np
import pytest

from dartray_amd import core, pbrt, scenes
from test_oracle_shading import uv_sphere
from util import rel_err_image

pytestmark = pytest.mark.gpu


def random_scene(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    u = lambda a=0.0, b=1.0: float(a + (b - a) * rng.random())
    path = bool(rng.integers(0, 2))
    plastic = False

    def material():
        nonlocal plastic
        k = int(rng.integers(0, 6 if path else 4))
        col = tuple(0.15 + 0.7 * rng.random(3))
        if k == 0:
            return core.MatteMaterial(col)
        if k == 1:
            return core.MatteMaterial(col, sigma=u(0, 70))
        if k in (2, 3):
            plastic = True
            return core.PlasticMaterial(col, tuple(0.1 + 0.5 * rng.random(3)), u(0.02, 0.5))
        if k == 4:
            return core.MirrorMaterial(tuple(0.5 + 0.5 * rng.random(3)))
        return core.GlassMaterial(tuple(0.5 + 0.5 * rng.random(3)), tuple(0.5 + 0.5 * rng.random(3)), u(1.2, 2.0))

    def xform(scale=1.0):
        t = pbrt.Transform.Translate(u(-6, 6), u(-7, 2), u(-4, 6)) * pbrt.Transform.Rotate(u(0, 360), u(-1, 1), u(-1, 1), u(-1, 1) + 1e-3)
        if rng.random() < 0.5:
            t = t * pbrt.Transform.Scale(scale * u(0.6, 1.4), scale * u(0.6, 1.4), scale * u(0.6, 1.4))
        return t

    prims = scenes.cornell_walls()
    for gp in prims:
        if rng.random() < 0.5:
            gp.material = material()
    nlights = 0
    for _ in range(int(rng.integers(2, 6))):
        kind = int(rng.integers(0, 4))
        t = xform()
        ro = bool(rng.random() < 0.3)
        light = None
        if rng.random() < 0.3:
            light = core.DiffuseAreaLight(tuple(2 + 20 * rng.random(3)), int(rng.integers(1, 4)))
            nlights += 1
        if kind == 0:
            shape = core.Sphere(t.m, t.mInv, ro, u(1, 2.5), *((None, None, 360.0) if rng.random() < 0.6 else (u(-1, 0), u(0.2, 1), u(120, 360))))
        elif kind == 1:
            shape = core.Disk(t.m, t.mInv, ro, u(-0.5, 0.5), u(1, 3), u(0, 0.8) if rng.random() < 0.5 else 0.0, 360.0 if rng.random() < 0.6 else u(90, 300))
        elif kind == 2:
            shape = uv_sphere(u(1, 2.5), int(rng.integers(5, 12)), int(rng.integers(3, 8)), t, normals=bool(rng.integers(0, 2)),
                              uvs=bool(rng.integers(0, 2)), tangents=bool(rng.integers(0, 2)))
            shape.reverseOrientation = ro
        else:
            P = (rng.random((4, 3)) * 4 - 2).astype(np.float32)
            shape = core.TriangleMesh(np.array([[0, 1, 2], [0, 2, 3], [1, 3, 2]], np.uint32), t.transformPoints(P), ro)
        prims.append(core.GeometricPrimitive(shape, material(), light))
    if nlights == 0 or rng.random() < 0.5:
        prims.append(scenes.emitter_quad(L=tuple(10 + 30 * rng.random(3))))
    lights, points = [gp.areaLight for gp in prims if gp.areaLight is not None], []
    for _ in range(int(rng.integers(0, 3))):
        k = int(rng.integers(0, 3))
        if k == 0:
            pl = core.PointLight(pbrt.Transform.Translate(u(-8, 8), u(0, 9), u(-8, 8)).m, tuple(50 + 200 * rng.random(3)))
        elif k == 1:
            t = pbrt.Transform.Translate(u(-8, 8), u(4, 9), u(-8, 8)) * pbrt.Transform.Rotate(u(60, 120), 1, 0, u(-0.3, 0.3))
            pl = core.SpotLight(t.m, tuple(100 + 300 * rng.random(3)), u(20, 50), u(5, 19), t.mInv)
            pl.marshal_cosines = bool(seed & 1)  # DR_LIGHT_SPOT_COS: the two cosines a constructed SpotLight keeps
        else:
            pl = core.DistantLight(None, tuple(0.3 + rng.random(3)), (u(-1, 1), u(0.2, 1), u(-1, 1)))
        lights.append(pl)
        points.append((pl, None))
    env = None
    if rng.random() < 0.4:
        env = scenes.sky_env(32, 16, L=tuple(0.2 + 0.5 * rng.random(3))) if rng.random() < 0.5 else \
            core.InfiniteAreaLight(scenes.SKY_TO_WORLD, tuple(0.1 + 0.4 * rng.random(3)), int(rng.integers(1, 3)), None)
        lights.append(env)
    res = (int(rng.integers(9, 40)), int(rng.integers(9, 30)))
    film = core.ImageFilm(*res)
    cam = core.PerspectiveCamera.lookAt((u(-3, 3), u(-2, 3), -33.0), (0, 0, 0), (0, 1, 0), u(30, 45), film,
                                        lensradius=0.0 if rng.random() < 0.7 else u(0.1, 0.6), focaldistance=30.0)
    # (DirectLighting: strategy "all" on even seeds / 4, "one" on the others -- the seed decides, the generator's draws stay what they were)
    integ = core.PathIntegrator(int(rng.integers(0, 8))) if path else core.DirectLightingIntegrator((seed >> 2) & 1, 5)
    spp = int(2 ** rng.integers(0, 6))
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, spp, seed=int(rng.integers(1, 1 << 30))), cam, integ, core.EmissionIntegrator())
    return prims, lights, env, points, r, plastic


@pytest.mark.parametrize("seed", range(128))
def test_random_scene(ob, gpu, seed):
    prims, lights, env, points, r, plastic = random_scene(seed)
    scene = core.Scene(core.BVHAccel(prims), lights)
    out = r.render(scene)
    osc = ob.OracleScene(prims, env=env, points=points)   # delta lights after the area lights, the env light last
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    err = rel_err_image(out.rgb, ref["rgb"])
    assert err.max() <= 1e-4, (seed, err.max(), int((err > 1e-4).sum()))
    if plastic:   # pow() in the Blinn lobe: an ulp between maths libraries now and then
        assert np.allclose(out.film, ref["film"], rtol=5e-6, atol=1e-6)
    else:
        assert np.array_equal(out.film, ref["film"]), seed
    c, st = osc.counters(), r.last_stats
    for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris"):
        assert st[k] == c[k], (seed, k)
