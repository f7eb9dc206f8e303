"""GPU parity of triangle meshes with per-vertex normals / tangents / uvs (Triangle.getUVs, Triangle.getShadingGeometry;
triangle.dart:247-263, 271-364): bit-exact against the oracle."""
import numpy as np
import pytest

from dartray_amd import core, pbrt, scenes
from test_oracle_shading import uv_sphere
from util import rel_err_image

pytestmark = pytest.mark.gpu


def _parity(ob, prims, integ, res=(32, 24), spp=16):
    film = core.ImageFilm(*res)
    cam = core.PerspectiveCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), 35.0, film)
    r = core.SamplerRenderer(core.LowDiscrepancySampler(cam, spp), cam, integ, core.EmissionIntegrator())
    out = r.render(scenes.make_scene(prims))
    osc = ob.OracleScene(prims)
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    assert rel_err_image(out.rgb, ref["rgb"]).max() <= 1e-4
    assert np.array_equal(out.film, ref["film"])
    c, st = osc.counters(), r.last_stats
    for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris"):
        assert st[k] == c[k], k
    return out


@pytest.mark.parametrize("integ", [core.PathIntegrator(4), core.DirectLightingIntegrator(0, 5)])
@pytest.mark.parametrize("kw", [dict(normals=True), dict(normals=False, uvs=True), dict(normals=False, tangents=True),
                                dict(normals=True, tangents=True, uvs=True)])
def test_meshes_with_shading_data(ob, gpu, integ, kw):
    xf = pbrt.Transform.Translate(-3, -4, 1) * pbrt.Transform.Rotate(-70, 1, 0.2, 0) * pbrt.Transform.Scale(1, 1.3, 0.8)
    ball = core.GeometricPrimitive(uv_sphere(3.5, 20, 10, xf, **kw), core.MatteMaterial((0.6, 0.5, 0.4)))
    xf2 = pbrt.Transform.Translate(5, -6, 2)
    rev = uv_sphere(2.5, 12, 6, xf2, **kw)
    rev.reverseOrientation = True
    prims = scenes.cornell_prims(scenes.blob_prim(12, 6)) + [ball, core.GeometricPrimitive(rev, core.MatteMaterial((0.3, 0.5, 0.7), sigma=25.0))]
    out = _parity(ob, prims, integ)
    assert out.rgb.mean() > 0.05


def test_emitter_with_uvs_and_glass_with_normals(ob, gpu):
    """uvs on an area light's mesh decide its geometric normal (dpdu x dpdv) and so the side Le is seen from;
    a glass mesh with vertex normals refracts through the shading frame but decides reflection vs transmission
    with the geometric normal (bsdf.dart:193-199)."""
    P = np.array([[-3, 9.9, -3], [3, 9.9, -3], [3, 9.9, 3], [-3, 9.9, 3]], np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    for uv in ([[0, 0], [1, 0], [1, 1], [0, 1]], [[0, 0], [0, 1], [1, 1], [1, 0]], [[0.2, 0.2]] * 4):
        lamp = core.GeometricPrimitive(core.TriangleMesh(idx, P, uvs=np.array(uv, np.float32)), core.MatteMaterial(),
                                       core.DiffuseAreaLight((30, 30, 30), 1))
        xf = pbrt.Transform.Translate(0, -5, 0) * pbrt.Transform.Rotate(-90, 1, 0, 0)
        glass = core.GeometricPrimitive(uv_sphere(4.0, 16, 8, xf, normals=True, uvs=True), core.GlassMaterial(index=1.5))
        prims = scenes.cornell_walls() + [lamp, glass]
        _parity(ob, prims, core.PathIntegrator(6))
