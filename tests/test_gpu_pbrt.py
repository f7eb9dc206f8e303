"""GPU parity of scenes loaded through the PBRT front end (row f3): the device render of the parsed scene
equals the oracle's render of the same primitive / light lists, including the light ORDER of the file."""
import numpy as np
import pytest

from dartray_amd import pbrt
from util import rel_err_image

pytestmark = pytest.mark.gpu

SCENE = '''
Film "image" "integer xresolution" [40] "integer yresolution" [32]
SurfaceIntegrator "{integ}" "integer maxdepth" [4]
Sampler "lowdiscrepancy" "integer pixelsamples" [16]
LookAt 0 2 -34  0 0 0  0 1 0
Camera "perspective" "float fov" [38]
WorldBegin
{env_first}
AttributeBegin
  AreaLightSource "area" "color L" [30 28 24] "integer nsamples" [1]
  Translate 0 9.9 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-3 0 -3  3 0 -3  3 0 3  -3 0 3]
AttributeEnd
{env_mid}
AttributeBegin
  Material "matte" "color Kd" [0.75 0.75 0.75]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [10 -10 -10 -10 -10 -10 -10 -10 10 10 -10 10]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [10 -10 10 -10 -10 10 -10 10 10 10 10 10]
  Material "matte" "color Kd" [0.48 0.1125 0.075]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-10 -10 10 -10 -10 -10 -10 10 -10 -10 10 10]
AttributeEnd
AttributeBegin
  AreaLightSource "area" "color L" [4 6 9] "integer nsamples" [2]
  ReverseOrientation
  Translate 4 -7 4  Scale 0.3 0.4 0.3  Rotate 30 0 1 0
  Material "matte" "color Kd" [0.48 0.48 0.48]
  Shape "trianglemesh" "integer indices" [0 2 1 0 3 2] "point P" [10 -10 -10 -10 -10 -10 -10 -10 10 10 -10 10]
  Shape "trianglemesh" "integer indices" [0 2 1 0 3 2] "point P" [10 10 -10 10 10 10 -10 10 10 -10 10 -10]
AttributeEnd
{env_last}
WorldEnd
'''
ENV = 'AttributeBegin Rotate -90 1 0 0 LightSource "infinite" "color L" [0.3 0.4 0.6] "integer nsamples" [1] AttributeEnd'


@pytest.mark.parametrize("integ", ["path", "directlighting"])
@pytest.mark.parametrize("where", ["env_first", "env_mid", "env_last", "none"])
def test_loaded_scene_renders_like_the_oracle(ob, gpu, integ, where):
    slots = dict(env_first="", env_mid="", env_last="")
    if where != "none":
        slots[where] = ENV
    api = pbrt.loads(SCENE.format(integ=integ, **slots), render=True)
    out, r = api.outputImage, api.rendererObject
    env, before = api.envLight()
    assert (env is None) == (where == "none")
    osc = ob.OracleScene(api.scenePrimitives, env=env, env_before=before)
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    assert rel_err_image(out.rgb, ref["rgb"]).max() <= 1e-4
    assert np.array_equal(out.film, ref["film"])
    c, st = osc.counters(), r.last_stats
    for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris"):
        assert st[k] == c[k], k
    assert out.rgb.mean() > 0.01


def test_light_order_changes_the_image(ob, gpu):
    """Sanity of the test above: the position of the infinite light in Scene.lights is observable."""
    a = pbrt.loads(SCENE.format(integ="path", env_first=ENV, env_mid="", env_last=""), render=True).outputImage
    b = pbrt.loads(SCENE.format(integ="path", env_first="", env_mid="", env_last=ENV), render=True).outputImage
    assert not np.array_equal(a.film, b.film)
