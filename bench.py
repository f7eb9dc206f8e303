#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its named config.

    python bench.py --gpus N --steps K --warmup W

metric  : Msamples/s (primary + path rays), film pixels x spp per second, whole job over N GPUs
workload: configs[1] -- Cornell box + 1M-triangle displaced blob, PathIntegrator maxdepth 5,
          1024x1024, 256 spp (2.68e8 camera samples per GPU and step).  One "step" = one full pass
          of the hot path over that batch.  At N > 1 the image's 32x32 tiles are dealt round-robin over
          the ranks (north_star; the reference's task split, render_manager.dart:100-141), every rank
          accumulates a full-frame (X, Y, Z, weight) film and ONE RCCL reduce per step sums them on rank 0.
          --scaling weak (default): the image has N times the pixels (side 1024 * sqrt(N), same scene and
          camera, towards configs[2]'s 4096x4096), so every GPU keeps 2.68e8 samples per step;
          --scaling strong: the 1024x1024 image itself is split, each GPU's launches shrink with N;
          --scaling samples: no tiles -- every rank adds its own 256 spp (seed + rank) to the same image.
Inputs (scene, BVH) are resident in HBM before the timed region; samples are generated on the
device.  Synthetic procedural scene, no files.  Before the W warm-up steps one priming render allocates the
path-state workspace and lets the library pick its traversal kernel for the scene (set-up, like the BVH build).

Extra objects on the JSON line: "roofline" for the dominant kernel k_trace<0> (closest-hit BVH
traversal): algorithmic bytes (32 B per node visit + 48 B per triangle test, counted on the
device) / the kernel's summed HIP-event time; "cpu_baseline": the CPU oracle (a C++ port of the
reference path) timed on one host core on a strided pixel subset of the same image; "cpu_baseline_threads": the
same oracle on up to 64 host threads (one serial loop per thread).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", choices=["C2", "C4", "C5"])
    ap.add_argument("--res", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipelines", type=int, default=1, choices=[1, 2],
                    help="2: odd batches run on a second stream / workspace (kernel tails and memory-bound shading overlap "
                         "the ALU-bound traversal: +10 %% on C2) -- per-kernel event times then overlap, so the roofline "
                         "object is only meaningful with 1")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong", "samples"],
                    help="N > 1: weak = tiles of an image with N x the pixels (per-GPU work fixed); strong = tiles of the same "
                         "image; samples = every rank renders spp samples of the whole image with its own seed, films summed")
    ap.add_argument("--cpu-pixels", type=int, default=96, help="cpu baseline: side of the strided pixel grid")
    args = ap.parse_args()

    os.environ["DARTRAY_PIPELINES"] = str(args.pipelines)  # read once by the library
    import numpy as np
    import torch
    from dartray_amd import _abi, scenes, dist as drdist

    rank, world, local = drdist.init_process_group()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local)
    _abi.init(local)

    spp = args.spp or {"C2": 256, "C4": 64, "C5": 512}[args.config]
    args.res = args.res or {"C2": 1024, "C4": 1024, "C5": 2048}[args.config]
    if args.scaling == "weak" and world > 1:
        args.res = int(round(args.res * world ** 0.5 / 32.0)) * 32  # N x the pixels, whole tiles
    prims, mk = scenes.config(args.config, xres=args.res, yres=args.res, spp=spp)
    renderer = mk()
    if args.scaling == "samples":
        renderer = drdist.sample_set(renderer, rank)  # independent sample sets of the same image
    else:
        renderer = drdist.shard(renderer, rank, world)  # round-robin 32 x 32 tiles
    scene = scenes.make_scene(prims, renderer.env)  # every rank builds + uploads its own copy (render_isolate.dart:31-41)
    film_desc = renderer.camera.film
    H, W = film_desc.height, film_desc.width
    film = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    dev = scene._device()
    lib = _abi.lib()

    def step():
        film.zero_()
        renderer.render_device(scene, film.data_ptr(), stream)
        drdist.reduce_film(film, 0)
        if rank == 0:
            _abi.check(lib.dr_film_resolve_device(film.data_ptr(), H * W, rgb.data_ptr(), stream))

    # Set-up, not a step: the first render of a scene allocates the path-state workspace (56 GB for C2) and runs the
    # traversal-kernel pilot (DESIGN.md section 5 row j) -- like the BVH build and the scene upload, outside the W + K steps.
    step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dev.reset_stats()
    drdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    drdist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    st = dev.stats()

    if rank == 0:
        samples_per_step = H * W * spp * (world if args.scaling == "samples" else 1)
        value = samples_per_step * args.steps / dt / 1e6
        alg_bytes = 32.0 * st["closest_nodes"] + 48.0 * st["closest_tris"]
        launches = max(1, st["closest_launches"])
        achieved = alg_bytes / max(st["closest_ms"] * 1e-3, 1e-12) / 1e9
        peak = 8000.0  # HBM3E spec GB/s (MI355X_MICROARCH.md chip table)
        all_alg = alg_bytes + 32.0 * st["any_nodes"] + 48.0 * st["any_tris"]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_k_traffic.json")
        if (args.config == "C2" and args.res == 1024 and spp == 256 and world == 1 and args.pipelines == 1
                and not os.environ.get("DARTRAY_BATCH_BITS") and os.path.exists(tpath)):
            # HBM-side bytes per launch of k_trace<0> from the PMC passes of this same command (see the file)
            traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
        out = {
            "metric": "Msamples/sec (primary+path rays)",
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.scaling == "strong" else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: %s, PathIntegrator maxdepth=%d, %dx%d, %d spp, LD sampler (device, counter streams), box filter"
                       % (args.config, {"C2": "Cornell box + 1M-triangle displaced blob", "C4": "Cornell box + 10M-triangle hairball",
                                        "C5": "8M-triangle courtyard, 8 area lights + env map"}[args.config],
                          renderer.surfaceIntegrator.maxDepth, args.res, args.res, spp),
                       "triangles": int(len(scene.aggregate.tri_idx)), "bvh_nodes": int(len(scene.aggregate.nodes)),
                       "samples_per_step": samples_per_step, "parallelism": ("spp-sets x%d, film reduce" if args.scaling == "samples" else "tiles32 x%d, film reduce") % world, "pipelines": args.pipelines},
            "roofline": {"bound": "hbm", "kernel": "k_trace<0> (closest-hit BVH traversal)",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "GB/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic,
                         "alg_bytes_per_launch": round(alg_bytes / launches, 1),
                         "avg_launch_ms": round(st["closest_ms"] / launches, 4), "launches": int(st["closest_launches"]),
                         "rank0_job_alg_GBps": round(all_alg / dt / 1e9, 2),
                         "rank0_trace_share_of_time": round(st["trace_ms"] * 1e-3 / dt, 4)},
            "kernel_ms_per_step": {k: round(st[k] / args.steps, 2) for k in ("closest_ms", "any_ms", "shade_ms", "gen_ms", "film_ms", "total_ms")},
            "per_sample": {"rays": round((st["closest_rays"] + st["any_rays"]) / max(1, st["camera_samples"]), 3),
                           "nodes": round((st["closest_nodes"] + st["any_nodes"]) / max(1, st["camera_samples"]), 2),
                           "tris": round((st["closest_tris"] + st["any_tris"]) / max(1, st["camera_samples"]), 3),
                           "alg_bytes": round(all_alg / max(1, st["camera_samples"]) + 148 + 32, 1)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(prims, renderer, args.cpu_pixels, H, W, spp)
            out["cpu_baseline_threads"] = cpu_baseline_threads(prims, renderer, 2 * args.cpu_pixels, H, W, spp)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def cpu_baseline(prims, renderer, grid, H, W, spp):
    """The oracle (C++ port of the reference CPU path, -O2, no fast-math) on ONE host core, on a
    strided grid x grid pixel subset of the same image with the same per-pixel sample streams."""
    import numpy as np
    import oracle.binding as ob
    osc = ob.OracleScene(prims)
    ys = (np.arange(grid) * (H // grid) + (H // grid) // 2).astype(np.int32)
    xs = (np.arange(grid) * (W // grid) + (W // grid) // 2).astype(np.int32)
    px = np.stack(np.meshgrid(xs, ys), axis=-1).reshape(-1, 2)
    rd = ob.render_desc(renderer, sampler_mode=1, pixels=px)
    t0 = time.perf_counter()
    osc.render(rd, want_film=False)
    dt = time.perf_counter() - t0
    n = len(px) * spp
    return {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%dx%d pixels on a stride-%d grid of the %dx%d image x %d spp = %d samples in %.1f s; "
                      "C++ restatement of the Dart reference path (oracle/), single thread"
                      % (grid, grid, H // grid, W, H, spp, n, dt),
            "host_cores": os.cpu_count()}


def cpu_baseline_threads(prims, renderer, grid, H, W, spp):
    """the same oracle on min(host cores, 64) OS threads, each running the unmodified serial
    loop on its own rows of a strided grid x grid pixel subset -- the reference's one-isolate-per-task model
    (dartray_web/render_manager.dart:100-141) with a shared read-only scene.  ctypes releases the GIL inside the call."""
    import threading
    import numpy as np
    import oracle.binding as ob
    osc = ob.OracleScene(prims)
    nthreads = max(1, min(os.cpu_count() or 1, 64))
    ys = (np.arange(grid) * (H // grid) + (H // grid) // 2).astype(np.int32)
    xs = (np.arange(grid) * (W // grid) + (W // grid) // 2).astype(np.int32)
    px = np.stack(np.meshgrid(xs, ys), axis=-1).reshape(-1, 2)
    parts = [p for p in np.array_split(px, nthreads) if len(p)]
    descs = [ob.render_desc(renderer, sampler_mode=1, pixels=p) for p in parts]
    # a first threaded pass on one pixel per thread pays the process's one-off costs (thread stacks, malloc arenas, TLS)
    warm = [threading.Thread(target=osc.render, args=(ob.render_desc(renderer, sampler_mode=1, pixels=p[:1]),),
                             kwargs={"want_film": False}) for p in parts]
    for t in warm:
        t.start()
    for t in warm:
        t.join()
    threads = [threading.Thread(target=osc.render, args=(d,), kwargs={"want_film": False}) for d in descs]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    n = len(px) * spp
    return {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": len(parts), "kind": "port",
            "sample": "%dx%d pixels on a strided grid of the %dx%d image x %d spp = %d samples in %.1f s on %d threads; "
                      "C++ restatement of the Dart reference path (oracle/), one serial loop per thread over a shared scene"
                      % (grid, grid, W, H, spp, n, dt, len(parts)),
            "host_cores": os.cpu_count()}


if __name__ == "__main__":
    main()
