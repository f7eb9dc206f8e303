"""GPU parity of BVHAccel.intersect / intersectP (k_trace / k_intersect) against the oracle, through
the C ABI (dr_intersect): hit records bit-exact, traversal counters equal."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from dartray_amd import core, scenes
from test_oracle_bvh import SCENES, soup_prims
from util import aggregate_test_rays

pytestmark = pytest.mark.gpu


def _compare(ob, prims, n, seed):
    scene = scenes.make_scene(prims)
    osc = ob.OracleScene(prims)
    nodes, tri, _, _ = osc.bvh()
    surf = osc.verts()[tri].astype(np.float64).mean(axis=1)
    o, d, tmin, tmax = aggregate_test_rays(nodes[0]["bmin"], nodes[0]["bmax"], n, seed=seed, hits=surf)
    rays = core.Ray(o, d, tmin, tmax)
    orays = ob.make_rays(o, d, tmin, tmax)
    h = scene.intersect(rays)
    st = scene.aggregate.stats()
    osc.counters(reset=True)
    ho = osc.intersect(orays)
    c = osc.counters()
    for k in ("prim", "t", "b1", "b2"):
        assert np.array_equal(h[k], ho[k]), k
    assert (st["closest_rays"], st["closest_nodes"], st["closest_tris"]) == (c["closest_rays"], c["closest_nodes"], c["closest_tris"])
    hp = scene.intersectP(rays)
    st = scene.aggregate.stats()
    osc.counters(reset=True)
    hpo = osc.intersect(orays, any_hit=True)["prim"] >= 0
    c = osc.counters()
    assert np.array_equal(hp, hpo)
    assert (st["any_rays"], st["any_nodes"], st["any_tris"]) == (c["any_rays"], c["any_nodes"], c["any_tris"])
    return h


@pytest.mark.parametrize("name", list(SCENES))
def test_hit_records_match_oracle(ob, gpu, name):
    h = _compare(ob, SCENES[name](), 50000, seed=21)
    assert (h["prim"] >= 0).sum() > 100


def test_medium_mesh_100k_triangles(ob, gpu):
    _compare(ob, scenes.cornell_prims(scenes.blob_prim(320, 160)), 300000, seed=22)


def test_golden_hit_records(gpu):
    g = np.load(os.path.join(GOLDEN, "c2small_hits.npz"))
    prims, _ = scenes.config("C2", xres=16, yres=16, spp=8, blob=(32, 16))
    scene = scenes.make_scene(prims)
    assert scene.aggregate.nodes.tobytes() == g["nodes"].tobytes() and np.array_equal(scene.aggregate.tri_idx, g["tri"])
    rays = core.Ray(g["o"], g["d"], g["tmin"], g["tmax"])
    h = scene.intersect(rays)
    assert np.array_equal(h["prim"], g["prim"]) and np.array_equal(h["t"], g["t"])
    assert np.array_equal(h["b1"], g["b1"]) and np.array_equal(h["b2"], g["b2"])
    assert np.array_equal(scene.intersectP(rays), g["occluded"])


def test_deep_tree_uses_the_spill_stack(ob, gpu):
    """A geometric progression of triangles makes the SAH tree a long chain: deeper than the 32 LDS stack
    entries per lane, exercising the global-memory spill (the reference's todo stack holds 64)."""
    n = 60
    s = 4.0 ** np.arange(n)
    P = np.zeros((n, 3, 3), np.float32)
    P[:, 0] = np.stack([s, 0 * s, 0 * s], 1)
    P[:, 1] = np.stack([s, 0.3 * s, 0 * s], 1)
    P[:, 2] = np.stack([s, 0 * s, 0.3 * s], 1)
    prims = [core.GeometricPrimitive(core.TriangleMesh(np.arange(3 * n).reshape(-1, 3), P.reshape(-1, 3)),
                                     core.MatteMaterial((0.5, 0.5, 0.5)))]
    scene = scenes.make_scene(prims)
    assert scene.aggregate.depth > 32
    osc = ob.OracleScene(prims)
    rng = np.random.Generator(np.random.PCG64(5))
    m = 20000
    # rays from beyond the largest triangle back towards the origin cross many boxes
    # rays from beyond the k-th triangle back towards the origin cross many nested boxes
    k = rng.integers(10, n, m)
    o = np.stack([s[k] * 2, rng.random(m) * 0.02 * s[k], rng.random(m) * 0.02 * s[k]], 1).astype(np.float32)
    tgt = np.stack([np.zeros(m), rng.random(m) * 0.05, rng.random(m) * 0.05], 1)
    # ... and rays from the origin outwards visit the small boxes first, stacking one far sibling per level
    o2 = np.stack([np.full(m, -1.0), rng.random(m) * 0.05, rng.random(m) * 0.05], 1).astype(np.float32)
    tgt = np.concatenate([tgt, o.astype(np.float64)])
    o = np.concatenate([o, o2])
    d = (tgt - o.astype(np.float64))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    h = scene.intersect(core.Ray(o, d))
    osc.counters(reset=True)
    ho = osc.intersect(ob.make_rays(o, d))
    assert osc.counters()["max_stack"] > 32  # the rays really do overflow the LDS part of the stack
    assert np.array_equal(h["prim"], ho["prim"]) and np.array_equal(h["t"], ho["t"])
    assert (h["prim"] >= 0).sum() > 1000
    assert np.array_equal(scene.intersectP(core.Ray(o, d)), osc.intersect(ob.make_rays(o, d), any_hit=True)["prim"] >= 0)


def test_edge_cases(ob, gpu):
    prims = scenes.cornell_c1_prims()
    scene = scenes.make_scene(prims)
    osc = ob.OracleScene(prims)
    # empty batch
    assert len(scene.intersect(core.Ray(np.zeros((0, 3)), np.zeros((0, 3))))) == 0
    # axis-parallel rays on slab planes (0 * inf = NaN in the slab test), -0.0 components, rays along the floor
    o = np.array([[0, 5, 0], [10, 5, 0], [-10, 5, 0], [0, 5, 10], [0, -10, 0], [3, 20, 3], [0, 0, 0], [0, 0, 0]], np.float32)
    d = np.array([[0, -1, 0], [0, -1, 0], [0, -1, 0], [0, -1, 0], [1, 0, 0], [-0.0, -1, -0.0], [0, 1, 0], [0, 0, 0]], np.float32)
    h = scene.intersect(core.Ray(o, d))
    ho = osc.intersect(ob.make_rays(o, d))
    assert np.array_equal(h["prim"], ho["prim"]) and np.array_equal(h["t"], ho["t"])
    # Triangle.intersect accepts t == tmax (triangle.dart:96) but the leaf's box test is strict
    # (`tmin < ray.maxDistance`, bvh_accel.dart:471): through the BVH a hit at exactly tmax is lost
    o3, d3 = [[0.5, 5, 0.25]] * 4, [[0, -1, 0]] * 4
    tmin3, tmax3 = [0.0, 0.0, 15.5, 0.0], [15.0, 14.999, np.inf, 15.001]
    h = scene.intersect(core.Ray(o3, d3, tmin3, tmax3))
    ho = osc.intersect(ob.make_rays(o3, d3, np.array(tmin3), np.array(tmax3)))
    assert np.array_equal(h["prim"], ho["prim"]) and np.array_equal(h["t"], ho["t"])
    assert list(h["prim"] >= 0) == [False, False, False, True] and h["t"][3] == 15.0
    # an empty scene never hits
    empty = core.Scene(core.BVHAccel([]), [])
    assert np.all(empty.intersect(core.Ray(o, d))["prim"] == -1) and not empty.intersectP(core.Ray(o, d)).any()
