"""is the visit-order freedom of BVHAccel.intersectP worth anything?  Counters only, on the CPU oracle.

intersectP never modifies the ray (bvh_accel.dart:167-226): its boolean does not depend on the order the children are visited in;
only the work of a ray that FINDS an occluder does.  For the shadow / MIS rays of a strided pixel subset of a config this prints the
share of any-hit node visits made by rays that end occluded, and for those rays the visits under the reference order, far child
first, larger-surface-area child first, and a lower bound (the shallowest occluding leaf).  Stop line: the best alternative saves
< 15 % of ALL any-hit algorithmic bytes on both C2 and C4 -> close it.

    python tools/r06_anyhit_order.py C2 96          (config, side of the strided pixel grid)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DARTRAY_BVH_BUILDER", "host")

import oracle.binding as ob  # noqa: E402
from dartray_amd import scenes  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    grid = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    prims, mk = scenes.config(cfg)
    r = mk()
    fd = r.camera.film
    H, W, spp = fd.height, fd.width, r.sampler.samplesPerPixel
    t0 = time.time()
    osc = ob.OracleScene(prims, env=getattr(r, "env", None)) if getattr(r, "env", None) is not None else ob.OracleScene(prims)
    t_build = time.time() - t0
    ys = (np.arange(grid) * (H // grid) + (H // grid) // 2).astype(np.int32)
    xs = (np.arange(grid) * (W // grid) + (W // grid) // 2).astype(np.int32)
    px = np.stack(np.meshgrid(xs, ys), axis=-1).reshape(-1, 2)
    rd = ob.render_desc(r, sampler_mode=1, pixels=px)
    ob.order_study(True)
    t0 = time.time()
    osc.render(rd, want_film=False)
    dt = time.time() - t0
    s = ob.order_study(False)
    b = lambda n, t: 32.0 * n + 48.0 * t
    all_b = b(s["nodes_all"], s["tris_all"])
    orders = ("ref", "far_first", "larger_area_first", "longer_interval_first", "leaf_then_far_first", "smaller_area_first")
    occ = {k: b(s["nodes_occ_" + k], s["tris_occ_" + k]) for k in orders}
    # a cost in node visits: the f64 triangle test weighs ~16 node visits (fitted on the GPU's any-hit times of C2, far child first
    # against the reference order: profiles/r06_far_first_ab.txt)
    W = 16.0
    cost_all = s["nodes_all"] + W * s["tris_all"]
    cost = {k: s["nodes_occ_" + k] + W * s["tris_occ_" + k] for k in orders}
    out = {"config": cfg, "pixels": int(len(px)), "spp": spp, "oracle_build_s": round(t_build, 1), "render_s": round(dt, 1), **s,
           "occluded_share_of_rays": round(s["occluded"] / max(1, s["rays"]), 4),
           "occluded_share_of_any_hit_node_visits": round(s["nodes_occ_ref"] / max(1, s["nodes_all"]), 4),
           "occluded_share_of_any_hit_alg_bytes": round(occ["ref"] / max(1.0, all_b), 4),
           "node_visits_per_occluded_ray": {k: round(s["nodes_occ_" + k] / max(1, s["occluded"]), 2) for k in orders},
           "triangle_tests_per_occluded_ray": {k: round(s["tris_occ_" + k] / max(1, s["occluded"]), 3) for k in orders},
           "saving_of_all_any_hit_cost_at_16_visits_per_triangle_test": {k: round((cost["ref"] - cost[k]) / max(1.0, cost_all), 4) for k in orders[1:]},
           "lower_bound_node_visits_per_occluded_ray": round(s["ideal_nodes_occ"] / max(1, s["occluded"]), 2),
           # what an order would save of ALL any-hit algorithmic bytes (rays that find nothing cost the same in every order)
           "saving_of_all_any_hit_alg_bytes": {k: round((occ["ref"] - occ[k]) / max(1.0, all_b), 4) for k in orders[1:]},
           "saving_of_all_any_hit_node_visits_at_the_lower_bound": round((s["nodes_occ_ref"] - s["ideal_nodes_occ"]) / max(1, s["nodes_all"]), 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
