"""algorithmic bytes per camera sample, measured by the ORACLE's counting mode (the probes of
bvh_accel.dart:106-163) on a fixed strided pixel subset -- every 16th pixel in x and y; all spp for C1, 16 spp
otherwise -- and committed as roofline/<config>.json.  The device counts the same quantities during every render
(bench.py "per_sample", asserted equal to the oracle's in tests/test_gpu_*.py); this file is the CPU-side record.

    python tools/make_roofline.py C1 C2 [C4 C5]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import oracle.binding as ob  # noqa: E402
from dartray_amd import scenes  # noqa: E402


def run(name):
    spp = None if name == "C1" else 16
    prims, mk = scenes.config(name, spp=spp) if spp else scenes.config(name)
    r = mk()
    film = r.camera.film
    xs = np.arange(0, film.xResolution, 16, dtype=np.int32)
    ys = np.arange(0, film.yResolution, 16, dtype=np.int32)
    px = np.stack(np.meshgrid(xs, ys), axis=-1).reshape(-1, 2)
    t0 = time.time()
    env = getattr(r, "env", None)
    osc = ob.OracleScene(prims, env=env) if env is not None else ob.OracleScene(prims)
    t1 = time.time()
    osc.counters(reset=True)
    osc.render(ob.render_desc(r, sampler_mode=1, pixels=px), want_film=False)
    c = osc.counters()
    t2 = time.time()
    n = len(px) * r.sampler.samplesPerPixel
    rays = c["closest_rays"] + c["any_rays"]
    nodes = c["closest_nodes"] + c["any_nodes"]
    tris = c["closest_tris"] + c["any_tris"]
    nfloats = osc.sample_floats(1 if name != "C1" else 0, r.surfaceIntegrator.maxDepth)
    out = {
        "config": name,
        "subset": "every 16th pixel in x and y (%d pixels) x %d spp = %d camera samples" % (len(px), r.sampler.samplesPerPixel, n),
        "mean_rays_per_sample": rays / n,
        "mean_nodes_per_ray": nodes / rays,
        "mean_tris_per_ray": tris / rays,
        "mean_nodes_per_sample": nodes / n,
        "mean_tris_per_sample": tris / n,
        "sample_vector_floats": nfloats,
        "B_alg": (32.0 * nodes + 48.0 * tris) / n + 4 * nfloats + 32,
        "B_alg_formula": "sum over rays (32 B x node visits + 48 B x triangle tests) + 4 B x sample-vector floats + 32 B film RMW",
        "oracle_seconds": {"scene": round(t1 - t0, 1), "render": round(t2 - t1, 1)},
    }
    json.dump(out, open(os.path.join(ROOT, "roofline", name + ".json"), "w"), indent=1)
    print(name, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items() if k.startswith("mean") or k == "B_alg"})


if __name__ == "__main__":
    for nm in sys.argv[1:] or ["C1", "C2"]:
        run(nm)
