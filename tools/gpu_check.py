"""Ad-hoc GPU bring-up check (not a test): intersect + render parity against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ctypes as C
import numpy as np
from dartray_amd import _abi, scenes, core
import oracle.binding as ob
from util import aggregate_test_rays, rel_err_image

_abi.init(0)
g = C.c_double()
_abi.check(_abi.lib().dr_copy_bandwidth(1 << 30, 5, C.byref(g)))
print("copy bandwidth GB/s:", g.value)


def check_intersect(name, prims, n=200000):
    sc = scenes.make_scene(prims)
    osc = ob.OracleScene(prims)
    nodes, tri, _, _ = osc.bvh()
    print(name, "nodes equal", nodes.tobytes() == sc.aggregate.nodes.tobytes(), "tris equal", np.array_equal(tri, sc.aggregate.tri_idx),
          "depth", sc.aggregate.depth, osc.depth)
    bmin, bmax = sc.aggregate.worldBound()
    o, d, tmin, tmax = aggregate_test_rays(bmin, bmax, n, seed=7)
    rays = core.Ray(o, d, tmin, tmax)
    t0 = time.time(); h = sc.intersect(rays); t1 = time.time()
    orays = ob.make_rays(o, d, tmin, tmax)
    osc.counters(reset=True)
    ho = osc.intersect(orays)
    c = osc.counters()
    st = sc.aggregate.stats()
    print("  closest: prim eq", np.array_equal(h["prim"], ho["prim"]), "t eq", np.array_equal(h["t"], ho["t"]),
          "b eq", np.array_equal(h["b1"], ho["b1"]) and np.array_equal(h["b2"], ho["b2"]), "hits", (h["prim"] >= 0).sum(),
          "gpu s %.3f" % (t1 - t0))
    print("  counters gpu nodes/tris", st["closest_nodes"], st["closest_tris"], "oracle", c["closest_nodes"], c["closest_tris"])
    bad = np.nonzero(h["prim"] != ho["prim"])[0]
    if len(bad):
        print("  mismatches:", len(bad), bad[:5], h[bad[:5]], ho[bad[:5]])
    hp = sc.intersectP(rays)
    osc.counters(reset=True)
    hpo = osc.intersect(orays, any_hit=True)["prim"] >= 0
    c = osc.counters()
    st = sc.aggregate.stats()
    print("  any: eq", np.array_equal(hp, hpo), hp.sum(), "counters gpu", st["any_nodes"], st["any_tris"], "oracle", c["any_nodes"], c["any_tris"])
    return sc, osc


def check_render(name, prims, renderer):
    scene = scenes.make_scene(prims)
    t0 = time.time(); out = renderer.render(scene); t1 = time.time()
    osc = ob.OracleScene(prims)
    rd = ob.render_desc(renderer, sampler_mode=1)
    osc.counters(reset=True)
    t2 = time.time(); ref = osc.render(rd); t3 = time.time()
    err = rel_err_image(out.rgb, ref["rgb"])
    st = renderer.last_stats
    c = osc.counters()
    print(name, "max rel err", err.max(), "n>1e-4:", (err > 1e-4).sum(), "film exact:", np.array_equal(out.film, ref["film"]),
          "gpu s %.3f oracle s %.3f" % (t1 - t0, t3 - t2))
    print("  mean", out.rgb.mean(), ref["rgb"].mean())
    print("  gpu stats", {k: st[k] for k in ("camera_samples", "closest_rays", "any_rays", "closest_nodes", "any_nodes", "closest_tris", "any_tris", "trace_ms", "total_ms", "batches")})
    print("  oracle  ", c)
    if err.max() > 0:
        bad = np.argwhere(err > 1e-4)
        print("  bad pixels", bad[:10].tolist())
        for y, x in bad[:3]:
            print("   ", out.rgb[y, x], ref["rgb"][y, x])
    return out, ref


prims, mk = scenes.config("C1")
check_intersect("C1", prims, 100000)
check_render("C1", prims, mk())
prims, mk = scenes.config("C2", xres=64, yres=64, spp=16, blob=(40, 20))
check_intersect("C2-small", prims, 200000)
check_render("C2-small path", prims, mk())
prims, mk = scenes.config("C2", xres=32, yres=32, spp=256, blob=(100, 50))
check_render("C2-10k path 256spp", prims, mk())
