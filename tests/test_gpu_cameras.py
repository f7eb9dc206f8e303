"""Renders through the orthographic (with and without depth of field) and environment cameras: k_raygen against the
oracle's generateRay, bit for bit like the perspective camera."""
import numpy as np
import pytest

from dartray_amd import core, scenes
from util import rel_err_image

pytestmark = pytest.mark.gpu


def _cams(film):
    return {
        "ortho": core.OrthographicCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), film, screenWindow=[-11, 11, -8, 8]),
        "ortho_dof": core.OrthographicCamera.lookAt((0.5, 1, -30), (0, 0, 0), (0, 1, 0), film, lensradius=0.8,
                                                    focaldistance=28.0, screenWindow=[-11, 11, -8, 8]),
        "env": core.EnvironmentCamera.lookAt((1, 6, -8), (0, 5, 1), (0, 1, 0), film),
    }


@pytest.mark.parametrize("which", ["ortho", "ortho_dof", "env"])
@pytest.mark.parametrize("integ", ["path", "direct"])
def test_camera_render_matches_oracle(ob, gpu, which, integ):
    prims = scenes.cornell_prims(scenes.blob_prim(16, 8))
    film = core.ImageFilm(44, 32)
    cam = _cams(film)[which]
    si = core.PathIntegrator(3) if integ == "path" else core.DirectLightingIntegrator(0, 5)
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, 8), cam, si, core.EmissionIntegrator())
    out = r.render(scenes.make_scene(prims))
    osc = ob.OracleScene(prims)
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    assert rel_err_image(out.rgb, ref["rgb"]).max() <= 1e-4
    if which == "env":   # sin / cos of the device maths library: an ulp now and then
        assert np.allclose(out.film, ref["film"], rtol=2e-5, atol=1e-6)
    else:
        assert np.array_equal(out.film, ref["film"])
        c, st = osc.counters(), r.last_stats
        for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris"):
            assert st[k] == c[k], k
    assert out.rgb.mean() > 0.005
