"""Generates the golden fixtures under tests/golden/ with the CPU oracle.

The reference (Dart) cannot be executed here and ships no golden vectors of its own
(test/spectrum_test.dart is an empty function), so these vectors pin the ORACLE: they guard
the restatement against accidental change and are replayed against the HIP path on the GPU
box.  Re-run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import oracle.binding as ob  # noqa: E402
from dartray_amd import scenes  # noqa: E402
from util import aggregate_test_rays  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    # (1) C1 in the reference's own serial mode: one DartRandom(taskNum) through sampler and integrator.
    prims, mk = scenes.config("C1")
    r = mk()
    osc = ob.OracleScene(prims)
    rec = osc.render(ob.render_desc(r, sampler_mode=0), record=65 * 65 * 4, max_tail=8)
    np.savez_compressed(os.path.join(OUT, "c1_serial.npz"), rgb=rec["rgb"], film=rec["film"],
                        pixel_xy=rec["pixel_xy"][::4].copy(), sample_vec=rec["sample_vec"], Ls=rec["Ls"],
                        tail_count=rec["tail_count"])
    # (2) hit records on a small C2-class scene
    prims, mk = scenes.config("C2", xres=16, yres=16, spp=8, blob=(32, 16))
    osc = ob.OracleScene(prims)
    nodes, tri, _, _ = osc.bvh()
    o, d, tmin, tmax = aggregate_test_rays(nodes[0]["bmin"], nodes[0]["bmax"], 4000, seed=11)
    rays = ob.make_rays(o, d, tmin, tmax)
    h = osc.intersect(rays)
    hp = osc.intersect(rays, any_hit=True)
    np.savez_compressed(os.path.join(OUT, "c2small_hits.npz"), o=o, d=d, tmin=tmin, tmax=tmax, prim=h["prim"], t=h["t"],
                        b1=h["b1"], b2=h["b2"], occluded=(hp["prim"] >= 0), nodes=nodes, tri=tri)
    # (3) the same scene path traced in serial mode, with the in-Li RNG draws recorded
    r = mk()
    rec = osc.render(ob.render_desc(r, sampler_mode=0), record=17 * 17 * 8, max_tail=40)
    np.savez_compressed(os.path.join(OUT, "c2small_path_serial.npz"), rgb=rec["rgb"], film=rec["film"],
                        pixel_xy=rec["pixel_xy"][::8].copy(), sample_vec=rec["sample_vec"], tail=rec["tail"],
                        tail_count=rec["tail_count"], Ls=rec["Ls"])
    # (4) counter-mode image (the device sampler's mode) of the same scene
    ref = osc.render(ob.render_desc(r, sampler_mode=1))
    np.savez_compressed(os.path.join(OUT, "c2small_path_counter.npz"), rgb=ref["rgb"], film=ref["film"])
    # (5) mirror + glass under the path integrator, serial mode (the stream tests/golden/make_restatement_fixtures.py
    #     replays through the independent Python restatement)
    import make_restatement_fixtures as mrf
    prims, mk = mrf.spec_case()
    r = mk()
    rec = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=0), record=17 * 17 * 8, max_tail=40)
    np.savez_compressed(os.path.join(OUT, "cspec_path_serial.npz"), rgb=rec["rgb"], film=rec["film"],
                        pixel_xy=rec["pixel_xy"][::8].copy(), sample_vec=rec["sample_vec"], tail=rec["tail"],
                        tail_count=rec["tail_count"], Ls=rec["Ls"])
    # (6) an InfiniteAreaLight scene, serial mode (same purpose as (5))
    prims, mk = mrf.env_case()
    r = mk()
    rec = ob.OracleScene(prims, env=r.env).render(ob.render_desc(r, sampler_mode=0), record=17 * 17 * 8, max_tail=40)
    np.savez_compressed(os.path.join(OUT, "cenv_path_serial.npz"), rgb=rec["rgb"], film=rec["film"],
                        pixel_xy=rec["pixel_xy"][::8].copy(), sample_vec=rec["sample_vec"], tail=rec["tail"],
                        tail_count=rec["tail_count"], Ls=rec["Ls"])
    # (7) mirror + glass under DirectLighting (SpecularReflect / SpecularTransmit recursion), serial mode
    prims, mk = mrf.dlspec_case()
    r = mk()
    rec = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=0), record=17 * 17 * 4, max_tail=200)
    np.savez_compressed(os.path.join(OUT, "cdlspec_direct_serial.npz"), rgb=rec["rgb"], film=rec["film"],
                        pixel_xy=rec["pixel_xy"][::4].copy(), sample_vec=rec["sample_vec"], tail=rec["tail"],
                        tail_count=rec["tail_count"], Ls=rec["Ls"])
    # (8) thin-lens camera, serial mode
    prims, mk = mrf.lens_case()
    r = mk()
    rec = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=0), record=17 * 17 * 4, max_tail=40)
    np.savez_compressed(os.path.join(OUT, "clens_path_serial.npz"), rgb=rec["rgb"], film=rec["film"],
                        pixel_xy=rec["pixel_xy"][::4].copy(), sample_vec=rec["sample_vec"], tail=rec["tail"],
                        tail_count=rec["tail_count"], Ls=rec["Ls"])
    # (9) DirectLighting, two lights with several samples each, serial mode
    prims, mk = mrf.dl2_case()
    r = mk()
    rec = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=0), record=17 * 17 * 4, max_tail=8)
    np.savez_compressed(os.path.join(OUT, "cdl2_direct_serial.npz"), rgb=rec["rgb"], film=rec["film"],
                        pixel_xy=rec["pixel_xy"][::4].copy(), sample_vec=rec["sample_vec"], tail=rec["tail"],
                        tail_count=rec["tail_count"], Ls=rec["Ls"])
    # (10) spheres and disks (BVH primitives and emitters), path and direct, serial mode
    for direct, name, spp, mt in ((False, "cquad_path_serial.npz", 8, 40), (True, "cquaddl_direct_serial.npz", 4, 200)):
        prims, mk = mrf.quad_case(direct=direct)
        r = mk()
        rec = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=0), record=17 * 17 * spp, max_tail=mt)
        np.savez_compressed(os.path.join(OUT, name), rgb=rec["rgb"], film=rec["film"],
                            pixel_xy=rec["pixel_xy"][::spp].copy(), sample_vec=rec["sample_vec"], tail=rec["tail"],
                            tail_count=rec["tail_count"], Ls=rec["Ls"])
    # (11) DirectLighting strategy "one": two lights, a matte and a mirror blob, serial mode
    prims, mk = mrf.dlone_case()
    r = mk()
    rec = ob.OracleScene(prims).render(ob.render_desc(r, sampler_mode=0), record=17 * 17 * 4, max_tail=200)
    np.savez_compressed(os.path.join(OUT, "cdlone_direct_serial.npz"), rgb=rec["rgb"], film=rec["film"],
                        pixel_xy=rec["pixel_xy"][::4].copy(), sample_vec=rec["sample_vec"], tail=rec["tail"],
                        tail_count=rec["tail_count"], Ls=rec["Ls"])
    # (12) the environment-map scene under a 24 x 10 map: MIPMap.texture's resampling to 32 x 16, serial mode
    prims, mk = mrf.env_case(np2=True)
    r = mk()
    rec = ob.OracleScene(prims, env=r.env).render(ob.render_desc(r, sampler_mode=0), record=17 * 17 * 8, max_tail=40)
    np.savez_compressed(os.path.join(OUT, "cenvnp2_path_serial.npz"), rgb=rec["rgb"], film=rec["film"],
                        pixel_xy=rec["pixel_xy"][::8].copy(), sample_vec=rec["sample_vec"], tail=rec["tail"],
                        tail_count=rec["tail_count"], Ls=rec["Ls"])
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
