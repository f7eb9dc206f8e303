"""White furnace on the device: closed box, every wall emits Le and reflects rho.  Known answers for the whole
integrators, independent of the oracle: PathIntegrator L = Le * (1 + rho + ... + rho^(maxDepth+1)) (emission at the
camera vertex + one light estimate per vertex, throughput rho^i, Russian roulette after bounce 3 keeps the
expectation), DirectLighting L = Le * (1 + rho)."""
import numpy as np
import pytest

from dartray_amd import core, scenes
from test_oracle_render import _furnace_prims

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("integ,rho", [(core.PathIntegrator(0), 0.5), (core.PathIntegrator(3), 0.5), (core.PathIntegrator(8), 0.8),
                                       (core.DirectLightingIntegrator(0, 5), 0.6), (core.DirectLightingIntegrator(1, 5), 0.6)],
                         ids=["path0", "path3", "path8rr", "direct", "direct-one"])
def test_white_furnace(gpu, integ, rho):
    Le = 1.25
    prims = _furnace_prims(rho, Le)
    film = core.ImageFilm(48, 48)
    cam = core.PerspectiveCamera.lookAt((0.1, -0.2, 0.05), (0.3, 0.1, 1.0), (0, 1, 0), 70.0, film)
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 1024), cam, integ, core.EmissionIntegrator())
    out = r.render(core.Scene(core.BVHAccel(prims), [gp.areaLight for gp in prims]))
    if isinstance(integ, core.PathIntegrator):
        expect = Le * sum(rho ** j for j in range(integ.maxDepth + 2))
    else:
        expect = Le * (1 + rho)
    assert out.rgb.mean() == pytest.approx(expect, rel=2e-3), (out.rgb.mean(), expect)
    assert np.abs(out.rgb - expect).max() < 0.1 * expect       # per pixel at 1024 spp


def _sphere_scene(material, env):
    eye = np.eye(4, dtype=np.float32)
    gp = core.GeometricPrimitive(core.Sphere(eye, eye, False, 1.0), material)
    film = core.ImageFilm(24, 24)
    cam = core.PerspectiveCamera.lookAt((0, 0, -5), (0, 0, 0), (0, 1, 0), 12.0, film)  # every pixel sees the sphere
    return [gp], cam, core.Scene(core.BVHAccel([gp]), [env])


@pytest.mark.parametrize("which", ["matte", "mirror", "glass"])
def test_uniform_environment_known_answers(ob, gpu, which):
    """A convex object in a uniform environment Lenv: a Lambertian sphere shows rho * Lenv (one light estimate at the
    camera vertex with both MIS halves over the env map's Distribution2D; the continuation leaves the scene), a mirror
    sphere Kr * Lenv (the specular-bounce escape term, path_integrator.dart:107-111), a clear glass sphere Lenv up to
    the paths cut at maxDepth (Fresnel-weighted choice of reflection / refraction, the radiance scaling (eta_i/eta_t)^2
    on the way in undone on the way out).  Oracle and device both, and against each other."""
    Lenv = 0.8
    env = core.InfiniteAreaLight(scenes.SKY_TO_WORLD, (Lenv,) * 3, 1, None)
    mat, depth, expect, tol = {
        "matte": (core.MatteMaterial((0.6,) * 3), 4, 0.6 * Lenv, 5e-3),
        "mirror": (core.MirrorMaterial((0.9,) * 3), 4, 0.9 * Lenv, 1e-5),
        "glass": (core.GlassMaterial((1.0,) * 3, (1.0,) * 3, 1.5), 24, Lenv, 2e-2),
    }[which]
    prims, cam, scene = _sphere_scene(mat, env)
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 256), cam, core.PathIntegrator(depth), core.EmissionIntegrator())
    out = r.render(scene)
    ref = ob.OracleScene(prims, env=env).render(ob.render_desc(r, sampler_mode=1))
    assert np.array_equal(out.film, ref["film"])
    for img in (out.rgb, ref["rgb"]):
        assert img.mean() == pytest.approx(expect, rel=tol), (which, img.mean(), expect)
