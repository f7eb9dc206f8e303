#!/usr/bin/env python3
"""Round 5, step 0 of "regenerate camera samples into dead slots": the BOUND, before anything is built.

What regeneration could return is the difference between what a stage costs per list entry / per ray when its list is
THIN (C5: 0.22 of a batch's slots are alive at the second bounce) and what the same work costs when the list is DENSE.
Measured per stage with the library's own diagnostics (DARTRAY_STAGE_COUNTS=2: list lengths, kernel times from the
batch's HIP events, node visits and triangle tests per stage; DARTRAY_OVERLAP_ANY=0: one kernel at a time) on
  * C5 as it is (open courtyard under an environment map: most bounce rays leave), and
  * the same courtyard inside a closed matte box ("dome": nothing leaves, lists stay dense until Russian roulette) --
    same geometry under the camera, same lights, same kernels (the environment light is still sampled and always occluded).
      (Matte, not emissive: an emissive dome would be a 12-triangle area light that ShapeSet.sample / pdf walk three times
      per vertex -- the per-entry cost of the comparator would change for a reason that is not density.)

usage: r05_regen_bound.py [--dome] [--res R] [--spp S] [--layout 4|64] [--kernels 5,3] [--config C5|C2]
prints one JSON line: per stage {entries, shade_ms, env_ms, ps_per_entry, closest rays / nodes / ms, any ...}."""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dome_prims(lo=(-130.0, -8.0, -130.0), hi=(130.0, 90.0, 130.0), kd=(0.5, 0.5, 0.5)):
    from dartray_amd import scenes
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    q = scenes._quad
    return [q((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1), kd), q((x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0), kd),
            q((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0), kd), q((x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1), kd),
            q((x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0), kd), q((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1), kd)]


def child(args):
    import torch
    from dartray_amd import _abi, scenes
    _abi.init(0)
    lib = _abi.lib()
    kw = dict(xres=args.res, yres=args.res, spp=args.spp)
    prims, mk = scenes.config(args.config, **{k: v for k, v in kw.items() if v})
    if args.dome:
        prims = prims + dome_prims()
    r = mk()
    scene = scenes.make_scene(prims, r.env)
    dev = scene._device()
    if args.kernels:
        dev.trace_kernels(tuple(int(x) for x in args.kernels.split(",")))
    if args.layout:
        dev.state_layout(int(args.layout))
    fd = r.camera.film
    film = torch.zeros((fd.height, fd.width, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for rep in range(2):  # first render: workspace, pilot (if any); second: measured
        if rep == 1:
            _abi.check(lib.dr_set_option(b"STAGE_COUNTS", b"2"))
            dev.reset_stats()
        film.zero_()
        r.render_device(scene, film.data_ptr(), stream)
        torch.cuda.synchronize()
    st = dev.stats()
    print("RESULT " + json.dumps({"stats": {k: st[k] for k in ("closest_ms", "any_ms", "shade_ms", "gen_ms", "film_ms", "total_ms", "camera_samples",
                                                                "shade_items", "shade_vertices", "batches")},
                                  "info": dev.last_render_info(), "mean_film": float(film[..., :3].mean())}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dome", action="store_true")
    ap.add_argument("--config", default="C5")
    ap.add_argument("--res", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--layout", default=None)
    ap.add_argument("--kernels", default=None)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    env = dict(os.environ, DARTRAY_OVERLAP_ANY="0")
    res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + sys.argv[1:], env=env, capture_output=True, text=True, timeout=900)
    if res.returncode != 0:
        sys.stderr.write(res.stderr[-3000:])
        raise SystemExit(res.returncode)
    result = json.loads(next(l for l in res.stdout.splitlines() if l.startswith("RESULT "))[7:])
    stages = {}
    for line in res.stderr.splitlines():
        m = re.match(r"stage_counts batch (\d+) stage (-?\d+): in (\d+) active_out (\d+) closest (\d+) any (\d+) env (\d+)", line)
        if m:
            b = int(m.group(2))
            s = stages.setdefault(b, {"entries": 0, "closest_q": 0, "any_q": 0, "env_q": 0, "shade_ms": 0.0, "env_ms": 0.0, "closest_ms": 0.0, "any_ms": 0.0,
                                      "closest_rays": 0, "closest_nodes": 0, "closest_tris": 0, "any_rays": 0, "any_nodes": 0, "any_tris": 0, "batches": 0})
            s["entries"] += int(m.group(3)); s["closest_q"] += int(m.group(5)); s["any_q"] += int(m.group(6)); s["env_q"] += int(m.group(7)); s["batches"] += 1
            continue
        m = re.match(r"stage_times batch (\d+) stage (-?\d+): shade ([\d.]+) env ([\d.]+) closest ([\d.]+) any ([\d.]+) ms(.*)", line)
        if m:
            b = int(m.group(2))
            s = stages.setdefault(b, {"entries": 0, "closest_q": 0, "any_q": 0, "env_q": 0, "shade_ms": 0.0, "env_ms": 0.0, "closest_ms": 0.0, "any_ms": 0.0,
                                      "closest_rays": 0, "closest_nodes": 0, "closest_tris": 0, "any_rays": 0, "any_nodes": 0, "any_tris": 0, "batches": 0})
            s["shade_ms"] += float(m.group(3)); s["env_ms"] += float(m.group(4)); s["closest_ms"] += float(m.group(5)); s["any_ms"] += float(m.group(6))
            mm = re.search(r"closest rays (\d+) nodes (\d+) tris (\d+) any rays (\d+) nodes (\d+) tris (\d+)", m.group(7))
            if mm:
                for k, v in zip(("closest_rays", "closest_nodes", "closest_tris", "any_rays", "any_nodes", "any_tris"), mm.groups()):
                    s[k] += int(v)
    slots = result["stats"]["camera_samples"]
    rows = []
    for b in sorted(stages):
        s = stages[b]
        row = {"stage": b, "entries": s["entries"], "density": round(s["entries"] / slots, 4) if b >= 0 else None,
               "shade_ms": round(s["shade_ms"], 3), "env_ms": round(s["env_ms"], 3),
               "shade_ps_per_entry": round((s["shade_ms"] + s["env_ms"]) * 1e9 / s["entries"], 1) if s["entries"] else None,
               "closest_ms": round(s["closest_ms"], 3), "closest_rays": s["closest_rays"], "closest_nodes": s["closest_nodes"],
               "closest_ps_per_node": round(s["closest_ms"] * 1e9 / s["closest_nodes"], 2) if s["closest_nodes"] else None,
               "closest_ps_per_ray": round(s["closest_ms"] * 1e9 / s["closest_rays"], 1) if s["closest_rays"] else None,
               "any_ms": round(s["any_ms"], 3), "any_rays": s["any_rays"], "any_nodes": s["any_nodes"],
               "any_ps_per_node": round(s["any_ms"] * 1e9 / s["any_nodes"], 2) if s["any_nodes"] else None,
               "any_ps_per_ray": round(s["any_ms"] * 1e9 / s["any_rays"], 1) if s["any_rays"] else None}
        rows.append(row)
    print(json.dumps({"scene": args.config + ("+dome" if args.dome else ""), "args": sys.argv[1:], "slots": slots, "result": result, "stages": rows}))


if __name__ == "__main__":
    main()
