"""GPU parity of PlasticMaterial (Lambertian + Blinn microfacet lobe, plastic_material.dart / microfacet.dart /
blinn.dart) and PointLight (point_light.dart) -- the plugins the bundled teapot-area-light.pbrt adds to the path."""
import numpy as np
import pytest

from dartray_amd import pbrt
from test_oracle_shading import uv_sphere
from util import rel_err_image

pytestmark = pytest.mark.gpu

SCENE = '''
Film "image" "integer xresolution" [40] "integer yresolution" [30]
SurfaceIntegrator "{integ}" "integer maxdepth" [4]
Sampler "lowdiscrepancy" "integer pixelsamples" [16]
LookAt 0 0 -35 0 0 0 0 1 0
Camera "perspective" "float fov" [35]
WorldBegin
AttributeBegin
  AreaLightSource "area" "color L" [20 20 20] "integer nsamples" [{ns}]
  Translate 0 9.9 0
  Rotate 90 1 0 0
  Shape "disk" "float radius" [3]
AttributeEnd
AttributeBegin
  CoordSysTransform "camera"
  LightSource "point" "color I" [400 380 350]
AttributeEnd
LightSource "point" "color I" [60 60 90] "point from" [6 -6 -4] "color scale" [0.5 0.5 0.5]
{extra}
AttributeBegin
  Material "plastic" "color Kd" [0.5 0.3 0.8] "color Ks" [0.2 0.2 0.2] "float roughness" [0.1]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [10 -10 -10 -10 -10 -10 -10 -10 10 10 -10 10]
  Material "plastic" "color Kd" [0 0 0] "color Ks" [0.8 0.8 0.8] "float roughness" [0.02]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [10 -10 10 -10 -10 10 -10 10 10 10 10 10]
  Material "matte" "color Kd" [0.48 0.1125 0.075]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-10 -10 10 -10 -10 -10 -10 10 -10 -10 10 10]
  Material "plastic" "float roughness" [1e-6]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [10 -10 -10 10 -10 10 10 10 10 10 10 -10]
AttributeEnd
AttributeBegin
  Material "plastic" "color Kd" [.6 .5 .2] "color Ks" [.3 .3 .3] "float roughness" [.05]
  Translate -3 -5 0
  Shape "sphere" "float radius" 3
AttributeEnd
WorldEnd
'''


EXTRA = '''
AttributeBegin
  Rotate 20 0 0 1
  LightSource "spot" "color I" [300 200 100] "point from" [-6 8 -6] "point to" [0 -8 2] "float coneangle" [25] "float conedeltaangle" [8]
AttributeEnd
LightSource "distant" "color L" [0.6 0.7 1.0] "point from" [2 9 -8] "point to" [0 0 0]
'''


@pytest.mark.parametrize("integ,ns", [("path", 1), ("directlighting", 1), ("directlighting", 4)])
def test_plastic_and_point_lights_render_like_the_oracle(ob, gpu, integ, ns):
    api = pbrt.loads(SCENE.format(integ=integ, ns=ns, extra=EXTRA if ns == 1 else ""), render=True)
    out, r = api.outputImage, api.rendererObject
    assert [type(l).__name__ for l in api.sceneLights][:3] == ["DiffuseAreaLight", "PointLight", "PointLight"]
    assert len(api.sceneLights) == (5 if ns == 1 else 3)
    osc = ob.OracleScene(api.scenePrimitives, points=api.pointLights())
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    err = rel_err_image(out.rgb, ref["rgb"])
    assert err.max() <= 1e-4, (err.max(), (err > 1e-4).sum())
    c, st = osc.counters(), r.last_stats
    for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris"):
        assert st[k] == c[k], k
    # pow() differs between the host and device maths libraries by an ulp now and then: the film is compared to
    # 1e-6 instead of bit for bit where the Blinn lobe is involved
    assert np.allclose(out.film, ref["film"], rtol=2e-6, atol=1e-7)
    assert out.rgb.mean() > 0.05


def test_smooth_plastic_mesh_serial_stream(ob, gpu):
    """The structure of the bundled teapot-area-light.pbrt: a smooth-shaded (per-vertex N) plastic mesh under a disk
    emitter and a point light at the eye, DirectLighting; also through the reference's serial RNG stream."""
    from dartray_amd import core
    txt = SCENE.format(integ="directlighting", ns=2, extra="").replace('[40]', '[24]').replace('[30]', '[18]')
    api = pbrt.loads(txt)
    xf = pbrt.Transform.Translate(4, -6, 2) * pbrt.Transform.Rotate(-90, 1, 0, 0)
    ball = core.GeometricPrimitive(uv_sphere(3.0, 16, 8, xf, normals=True), core.PlasticMaterial((.5, .3, .8), (.2, .2, .2), .1))
    prims = api.scenePrimitives + [ball]
    scene = core.Scene(core.BVHAccel(prims), api.sceneLights)
    r = api.rendererObject
    out = r.render(scene)
    osc = ob.OracleScene(prims, points=api.pointLights())
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    assert rel_err_image(out.rgb, ref["rgb"]).max() <= 1e-4
    rec = osc.render(ob.render_desc(r, sampler_mode=0), record=25 * 19 * 16, max_tail=8)
    r.sampler = core.HostBufferSampler(r.camera, 16, rec["pixel_xy"][::16].copy(), rec["sample_vec"])
    out2 = r.render(scene)
    assert rel_err_image(out2.rgb, rec["rgb"]).max() <= 1e-4
