"""GPU parity of the quadric shapes (Sphere, Disk: SURVEY.md section 8 row f4) inside the BVH hot path:
hit records and traversal counters against the oracle, renders of scenes with a disk emitter and matte
spheres (the structure of the bundled cornell-path.pbrt), and the loader's Shape "sphere" / "disk"."""
import numpy as np
import pytest

from dartray_amd import core, pbrt, scenes
from util import aggregate_test_rays, quadric_prims, rel_err_image

pytestmark = pytest.mark.gpu


def test_hit_records_and_counters_match_the_oracle(ob, gpu):
    prims = quadric_prims()
    scene = scenes.make_scene(prims)
    osc = ob.OracleScene(prims)
    nodes = osc.bvh()[0]
    assert np.array_equal(nodes["bmin"], scene.aggregate.nodes["bmin"]) and np.array_equal(nodes["bmax"], scene.aggregate.nodes["bmax"])
    assert np.array_equal(nodes["offset"], scene.aggregate.nodes["offset"])
    o, d, tmin, tmax = aggregate_test_rays(nodes[0]["bmin"], nodes[0]["bmax"], 200000, seed=5)
    rays, orays = core.Ray(o, d, tmin, tmax), ob.make_rays(o, d, tmin, tmax)
    h = scene.intersect(rays)
    st = scene.aggregate.stats()
    osc.counters(reset=True)
    ho = osc.intersect(orays)
    c = osc.counters()
    for k in ("prim", "t", "b1", "b2"):
        assert np.array_equal(h[k], ho[k]), k
    assert (st["closest_rays"], st["closest_nodes"], st["closest_tris"]) == (c["closest_rays"], c["closest_nodes"], c["closest_tris"])
    quad_rows = np.nonzero(scene.aggregate.tri_idx[:, 0] == 0xFFFFFFFF)[0]
    assert np.isin(h["prim"], quad_rows).sum() > 1000  # the quadrics are actually being hit
    hp = scene.intersectP(rays)
    st = scene.aggregate.stats()
    osc.counters(reset=True)
    hpo = osc.intersect(orays, any_hit=True)["prim"] >= 0
    c = osc.counters()
    assert np.array_equal(hp, hpo)
    assert (st["any_rays"], st["any_nodes"], st["any_tris"]) == (c["any_rays"], c["any_nodes"], c["any_tris"])


SCENE = '''
Film "image" "integer xresolution" [48] "integer yresolution" [36]
SurfaceIntegrator "{integ}" "integer maxdepth" [5]
Sampler "lowdiscrepancy" "integer pixelsamples" [16]
LookAt 0 0 -35 0 0 0 0 1 0
Camera "perspective" "float fov" [35]
WorldBegin
AttributeBegin
  AreaLightSource "area" "color L" [36 36 36] "integer nsamples" [{ns}]
  Translate 0 9.9 0
  Rotate 90 1 0 0
  Shape "disk" "float radius" [3]
AttributeEnd
AttributeBegin
  Material "matte" "color Kd" [0.75 0.75 0.75]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [10 -10 -10 -10 -10 -10 -10 -10 10 10 -10 10]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [10 10 -10 10 10 10 -10 10 10 -10 10 -10]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [10 -10 10 -10 -10 10 -10 10 10 10 10 10]
  Material "matte" "color Kd" [0.48 0.1125 0.075]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-10 -10 10 -10 -10 -10 -10 10 -10 -10 10 10]
  Material "matte" "color Kd" [0.1125 0.375 0.1125]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [10 -10 -10 10 -10 10 10 10 10 10 10 -10]
AttributeEnd
AttributeBegin
  Material "matte" "color Kd" [0.48 0.48 0.48]
  Translate -4 -4 0
  Shape "sphere" "float radius" 3
AttributeEnd
AttributeBegin
  Material "matte" "color Kd" [0.3 0.4 0.6]
  Translate 4 -6 2  Rotate 40 0 1 0  Scale 1 1.5 1
  Shape "sphere" "float radius" 2.5 "float zmin" [-1.5] "float zmax" [2] "float phimax" [300]
AttributeEnd
AttributeBegin
  AreaLightSource "area" "color L" [3 5 8] "integer nsamples" [1]
  ReverseOrientation
  Translate 6 2 6  Rotate -60 0 1 0
  Shape "disk" "float radius" [2] "float innerradius" [0.8] "float phimax" [270] "float height" [0.5]
AttributeEnd
WorldEnd
'''


@pytest.mark.parametrize("integ,ns", [("path", 1), ("directlighting", 1), ("directlighting", 4)])
def test_disk_lights_and_spheres_render_like_the_oracle(ob, gpu, integ, ns):
    api = pbrt.loads(SCENE.format(integ=integ, ns=ns), render=True)
    out, r = api.outputImage, api.rendererObject
    osc = ob.OracleScene(api.scenePrimitives)
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    err = rel_err_image(out.rgb, ref["rgb"])
    assert err.max() <= 1e-4, (err.max(), (err > 1e-4).sum())
    assert np.array_equal(out.film, ref["film"])
    c, st = osc.counters(), r.last_stats
    for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris"):
        assert st[k] == c[k], k
    assert out.rgb.mean() > 0.05


SPHERE_LIGHTS = '''
Film "image" "integer xresolution" [40] "integer yresolution" [30]
SurfaceIntegrator "{integ}" "integer maxdepth" [4]
Sampler "lowdiscrepancy" "integer pixelsamples" [16]
LookAt 0 0 -35 0 0 0 0 1 0
Camera "perspective" "float fov" [35]
WorldBegin
AttributeBegin
  AreaLightSource "area" "color L" [20 18 15] "integer nsamples" [{ns}]
  Translate 3 5 0
  Shape "sphere" "float radius" [1.5]
AttributeEnd
AttributeBegin
  AreaLightSource "area" "color L" [4 8 16]
  ReverseOrientation
  Translate -5 -6 3  Rotate 35 1 1 0  Scale 1 1.4 0.8
  Shape "sphere" "float radius" [2.5]
AttributeEnd
AttributeBegin
  Material "matte" "color Kd" [0.75 0.75 0.75]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [10 -10 -10 -10 -10 -10 -10 -10 10 10 -10 10]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [10 -10 10 -10 -10 10 -10 10 10 10 10 10]
  Material "matte" "color Kd" [0.48 0.1125 0.075]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-10 -10 10 -10 -10 -10 -10 10 -10 -10 10 10]
  Translate 0 -8 -3
  Shape "sphere" "float radius" [2]
AttributeEnd
WorldEnd
'''


@pytest.mark.parametrize("integ,ns", [("path", 1), ("directlighting", 1), ("directlighting", 2)])
def test_sphere_area_lights_render_like_the_oracle(ob, gpu, integ, ns):
    """Sphere.sample2 (cone sampling from outside, uniform sphere sampling from inside the reversed emitter's
    hull is not reached here) and Sphere.pdf2 (sphere.dart:269-326) on the device."""
    api = pbrt.loads(SPHERE_LIGHTS.format(integ=integ, ns=ns), render=True)
    out, r = api.outputImage, api.rendererObject
    osc = ob.OracleScene(api.scenePrimitives)
    osc.counters(reset=True)
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    err = rel_err_image(out.rgb, ref["rgb"])
    assert err.max() <= 1e-4, (err.max(), (err > 1e-4).sum())
    assert np.array_equal(out.film, ref["film"])
    c, st = osc.counters(), r.last_stats
    for k in ("closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris"):
        assert st[k] == c[k], k
    assert out.rgb.mean() > 0.05
