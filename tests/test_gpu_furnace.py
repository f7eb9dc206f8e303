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
                                       (core.DirectLightingIntegrator(0, 5), 0.6)], ids=["path0", "path3", "path8rr", "direct"])
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
