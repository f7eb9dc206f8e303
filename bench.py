#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its named configs.

    python bench.py --gpus N --steps K --warmup W

metric  : Msamples/s (primary + path rays), film pixels x spp per second, whole job over N GPUs
workload: N = 1 -- configs[1] (C2): Cornell box + 1M-triangle displaced blob, PathIntegrator maxdepth 5,
          1024x1024, 256 spp (2.68e8 camera samples per step).
          N > 1 -- BASELINE configs[2] verbatim (C3; round 5: the N-rank HEADLINE, it used to be an extra): the C2 scene at
          4096x4096, 1024 spp (1.72e10 samples per step), its 32x32 tiles dealt round-robin over the N ranks (north_star;
          the reference's task split, render_manager.dart:100-141), every rank accumulating a full-frame (X, Y, Z,
          weight) film and ONE RCCL reduce of the 268 MB film per step (dr_film_reduce of the C ABI: ncclReduce over
          xGMI) summing them on rank 0; total work is the same for every N ("scaling": "strong"), --steps / --warmup are
          honoured as given.  Expected run time (one GPU renders C3 in 27.6 s per step, 623 Msamples/s: profiles/r06_c3_one_gpu.json): a step is
          ~28 s / N -- N = 8: 3.6 s, N = 4: 7.1 s, N = 2: 14.3 s -- so the driver's `--steps 20 --warmup 5` (+ the first
          render) is ~1.6 min at N = 8, ~3.1 min at N = 4 and ~6.2 min at N = 2, plus ~15 s of scene build per process.
          The same C2 scene per GPU as the N = 1 line is appended as "extra_configs"[0] ("scaling": "weak": the C2 view at
          256 spp on a square image with N x the pixels, side 1024 sqrt(N) in whole 32-pixel tiles, 2 timed steps of
          ~0.45 s; --no-extra skips it), so that a per-GPU-like-for-like figure sits next to the N = 1 line's.
          Options: --scaling weak (that image as the headline), --scaling strong-c2 (the 1024x1024 image itself
          split), --scaling samples (no tiles: every rank adds its own 256 spp of the C2 image, seed + rank).
          One "step" = one full pass of the hot path over the image.  Launch: `python bench.py --gpus N` starts
          the N ranks itself (torch.distributed.run, one process per GPU) unless it already runs under torchrun.
Inputs (scene, BVH) are resident in HBM before the timed region; samples are generated on the device.  Synthetic
procedural scene, no files.  The FIRST render of a scene allocates the path-state workspace (batches of 2^27 slots: what a one-shot
host pays) and runs the library's traversal-kernel pilot: it is timed separately ("first_render_ms", "pilot_ms") and is not one of the
W + K steps.  The SECOND render grows the workspace to one batch per image ("second_render_ms"): it is the first warm-up step.
The printed line carries numbers only (< 8 000 bytes); the full result is written to the sidecar file the line names ("detail").

Extra objects: "roofline" for the dominant kernel (the per-lane closest-hit BVH traversal): "achieved" = ALGORITHMIC bytes (32 B per
node visit + 48 B per triangle test, counted on the device) / the kernel's summed HIP-event time, "frac" = that against the 8 TB/s
spec.  Algorithmic bytes are mostly served by L1 / L2 / the Infinity Cache, so next to it, from the committed rocprofv3 --pmc passes
of the same command AND ONLY when they were taken with the library that is running (kernel-source hashes; otherwise
"stale_profile"): "traffic" (memory-side bytes per launch), "frac_physical_of_copy", "binding" / "frac_of_binding_ceiling" (the
throughput ceilings); "roofline_others" likewise for the camera kernel, shading and the sampler; "cpu_baseline": the CPU oracle (a C++
port of the reference path) on one host core on a strided pixel subset; "cpu_baseline_threads": the same oracle, one OS thread per
GetSubWindow task rectangle (the reference's isolate-per-task model); "extra_configs": short C4 and C5 runs (3 steps each, N = 1
only, after the headline's timed region).  Glossary of every key: profiles/README.md.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NAMES = {"C2": "Cornell box + 1M-triangle displaced blob", "C3": "Cornell box + 1M-triangle displaced blob",
         "C4": "Cornell box + 10M-triangle hairball", "C5": "8M-triangle courtyard, 8 area lights + env map"}
PEAK_GBPS = 8000.0  # HBM3E spec (MI355X_MICROARCH.md chip table)


def rank_command(gpus, argv, port):
    """The command line that starts one rank per GPU (the driver's own form: torch.distributed.run on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child job (before anything touches the GPU)
    and exit with its code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    raise SystemExit(subprocess.call(rank_command(args.gpus, sys.argv[1:], port)))


def weak_resolution(res, world):
    """Side of the square image with `world` x the pixels of a res x res one, in whole 32-pixel tiles."""
    return int(round(res * world ** 0.5 / 32.0)) * 32


def plan(world, scaling=None, config=None, res=0, spp=0, no_extra=False):
    """What a run renders, as data (tested on the CPU: tests/test_bench_launch.py): the headline Run's (config, resolution,
    spp, mode) and the extra runs appended after its timed region as (config, resolution, spp, mode, steps, warmup).
    N = 1: configs[1] (C2) + short C4 / C5 runs.  N > 1: configs[2] (C3: 4096^2, 1024 spp, tiles over the N ranks, total
    work fixed) + the weak C2 image (the N = 1 workload per GPU), short."""
    mode = scaling or ("c3" if world > 1 else "weak")
    cfg = config or ("C3" if world > 1 and mode == "c3" else "C2")
    spp_ = spp or {"C2": 256, "C3": 1024, "C4": 64, "C5": 512}[cfg]
    res_ = res or {"C2": 1024, "C3": 4096, "C4": 1024, "C5": 2048}[cfg]
    if mode == "weak" and world > 1:
        res_ = weak_resolution(res_, world)
    extras = []
    default_shape = not (no_extra or config or res or spp)
    if world == 1 and default_shape:
        extras = [("C4", 1024, 64, mode, 3, 2), ("C5", 2048, 512, mode, 3, 2)]  # (three timed steps behind two warm-up steps: two were noisy, 252 - 268 on C4)
    elif world > 1 and default_shape and mode == "c3":
        extras = [("C2", weak_resolution(1024, world), 256, "weak", 2, 1)]
    return {"headline": (cfg, res_, spp_, mode), "extras": extras}


def check_world(gpus, world_env):
    """The launch contract: one rank per GPU.  Returns "launch" when `python bench.py --gpus N` has to start the ranks
    itself, None when this process is a rank (or the single-GPU run); anything else is refused."""
    if gpus > 1 and world_env == 0:
        return "launch"
    if world_env not in (0, gpus) or (world_env == 0 and gpus != 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%s: launch one rank per GPU" % (gpus, os.environ.get("WORLD_SIZE")))
    return None


def shade_alg_bytes(st):
    """Algorithmic bytes of the shading stages: what one vertex step of PathIntegrator.Li
    must read and write of the per-sample path state and the scene, from the device's own counts."""
    items, verts = st["shade_items"], st["shade_vertices"]
    later = max(0, items - st["camera_samples"])  # entries of stages > 0: they carry a pending light estimate
    return (items * 28.0                  # flags, hprim, L in; flags out (+ the entry of the active list, below)
            + later * 20.0                # active-list entry, shOcc, the finished light term (Ld1)
            + (st["camera_samples"] + st["shade_shadow"]) * 12.0   # L out: the camera stage initialises it, a light term changes it
            + st["shade_mis"] * 120.0     # a MIS ray: misD, Ld2, misLight, betaNee out + queue entry; misPrim, misLight, Ld2, misD, betaNee, 32-B shading record in
            + verts * 101.0               # ht, ro, rd, beta, 32-B shading record (round 3; 48-B vertex record before), 5 sample-index bytes in; ro, rtmin out
            + st["shade_cont"] * 28.0     # rd, beta + queue entry
            + st["shade_shadow"] * 36.0)  # shD, shTmax, Ld1 + queue entry


def gen_blocks(renderer, scene):
    """LD blocks of a pixel sample the device sampler generates for a path render (dr_api.hip: rp.genMask): the image
    sample, the lens sample of a thin-lens camera and, per SAMPLE_DEPTH level b <= min(2, maxDepth), the light number, the
    light component, the light position, the BSDF and path directions (+ the two uComponent slots where a material has
    more than one lobe).  Never generated: the time sample, the volume integrator's slots."""
    d = scene._device()
    general = bool(getattr(d, "general", False))
    n = 1 + (1 if getattr(renderer.camera, "lensRadius", 0.0) > 0.0 else 0)
    for b in range(3):
        if b <= renderer.surfaceIntegrator.maxDepth:
            n += 2 + (2 if general else 0) + 3
    return n


def gen_alg_bytes_per_sample(nblocks, spp):
    """What the sampler and the ray generator must move per camera sample (compact sample form): one
    permuted-index entry per generated LD block (1 byte up to 256 spp, 2 above) + the block's two scramble words once per
    pixel, written by k_gen_samples_*; k_raygen reads the image (+ lens) entries back and writes origin, direction and
    minDistance (32 B)."""
    idx = 1 if spp <= 256 else 2
    return nblocks * idx + 8.0 * nblocks / spp + 2 * idx + 32.0


# ---------------------------------------------------------------------------------------------------------------------
# Figures derived from the committed rocprofv3 passes (profiles/r0N_<cfg>_traffic.json, built by tools/make_traffic.py).
# They belong to the LIBRARY that was profiled: since round 6 a traffic file carries the sha256 of that library's kernel
# sources, and nothing below is printed unless they equal the running library's -- otherwise the objects say
# "stale_profile" and carry nulls (round 5 matched profiles to kernels by NAME only: change a kernel without retaking the
# passes and the line printed last round's counters next to this round's time).
# ---------------------------------------------------------------------------------------------------------------------
KERNEL_SOURCES = ("dr_trace.hip", "dr_kernels.hip", "dr_device.h", "dr_kernels.h", "dr_wave.h", "dr_rng.h")
CLOCK_HZ, SIMDS, CUS = 2.4e9, 1024, 256


def library_hashes():
    """sha256 of the running library's kernel sources (dartray_amd/libdartray_hip.buildinfo.json, written by build(); _abi.lib()
    has already checked that file against the sources and the .so)."""
    try:
        have = json.load(open(os.path.join(ROOT, "dartray_amd", "libdartray_hip.buildinfo.json"))).get("sources", {})
    except (OSError, ValueError):
        return {}
    return {k: have[k] for k in KERNEL_SOURCES if k in have}


def traffic_files(tag):
    """The committed traffic files of this config, newest round first."""
    pat = re.compile(r"r\d\d_%s_traffic\.json$" % tag.lower())
    try:
        return sorted((n for n in os.listdir(os.path.join(ROOT, "profiles")) if pat.match(n)), reverse=True)
    except OSError:
        return []


def load_profile(tag, hashes=None, directory=None):
    """The newest committed traffic file of this config: (name, contents, fresh).  fresh = the file is stamped with the kernel-source
    hashes of the library that is running; (None, None, False) when the config has no file."""
    directory = directory or os.path.join(ROOT, "profiles")
    hashes = library_hashes() if hashes is None else hashes
    for name in traffic_files(tag) if directory == os.path.join(ROOT, "profiles") else sorted(os.listdir(directory), reverse=True):
        try:
            d = json.load(open(os.path.join(directory, name)))
        except (OSError, ValueError):
            continue
        if "kernels" not in d:
            continue
        stamp = d.get("library_sources")
        return name, d, bool(stamp) and bool(hashes) and stamp == hashes
    return None, None, False


def _valu_quad_cycles(v):
    """SQ_ACTIVE_INST_VALU per launch of a traffic-file entry (files of rounds 3-5 kept it as a share of the launch time)."""
    if v.get("valu_active_quad_cycles_per_launch") is not None:
        return v["valu_active_quad_cycles_per_launch"]
    if v.get("valu_issue_share_at_4_cycles_per_instruction") and v.get("avg_launch_ms_kernel_trace"):
        return v["valu_issue_share_at_4_cycles_per_instruction"] * v["avg_launch_ms_kernel_trace"] * 1e-3 * CLOCK_HZ * SIMDS / 4.0
    return None


def _find_kernel(d, prefix, exact=False):
    for n, v in d.get("kernels", {}).items():
        base = n.split("::")[-1]
        if base == prefix or (not exact and base.startswith(prefix)):
            return v
    return None


def machine_rates():
    """Measured peak rates of the shared resources (tools/gather_rate.hip on an MI355X: a property of the chip, not of the library)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r05_gather_rates.json")))
    except (OSError, ValueError):
        return None


def ceilings(k, scale, measured_ms, copy_gbps, kind, extra=None):
    """Throughput ceilings of one kernel, in ms: the time each shared resource ALONE would need for the work, from what the profiled
    launches asked of it (per-launch counters of traffic-file entry `k`, times `scale` = this run's work over the profiled work) at the
    resource's measured peak rate.  The kernel is under each of them; `binding` is the largest, `frac_of_binding_ceiling` = that over
    the measured time: what a perfectly overlapped, latency-free version would reach.
      valu_issue       SQ_ACTIVE_INST_VALU quad-cycles x 4 / (1024 SIMDs x clock): f64 / transcendental instructions occupy the pipe longer and
                       are counted at what they occupy.  Traversal kernels (many waves interleaved: 2 cycles per wave64 f32 instruction): x 2.
      lds_array        SQ_LDS_IDX_ACTIVE (all CUs) / (256 CUs x clock)
      l2_requests      TCC_REQ / the measured L2 request rate
      memory_lines     memory-side bytes / this run's measured copy rate (streaming kernels) or read requests / the measured random-line rate
    """
    rates = machine_rates()
    c = {}
    q = _valu_quad_cycles(k)
    if q is not None:
        cyc = 2.0 if kind == "trace" else 4.0
        c["valu_issue_ms"] = q * cyc / (SIMDS * CLOCK_HZ) * 1e3 * scale
    if k.get("lds_array_cycles_per_launch"):
        c["lds_array_ms"] = k["lds_array_cycles_per_launch"] / (CUS * CLOCK_HZ) * 1e3 * scale
    if k.get("l2_requests_per_launch") and rates:
        c["l2_requests_ms"] = k["l2_requests_per_launch"] / rates["l2_hit_records_per_s"] * 1e3 * scale
    if kind == "trace":
        if k.get("read_requests_per_launch") and rates:
            c["lines_beyond_l2_ms"] = k["read_requests_per_launch"] / rates["infinity_cache_lines_per_s"] * 1e3 * scale
    elif k.get("hbm_bytes_per_launch") and copy_gbps:
        c["memory_lines_ms"] = k["hbm_bytes_per_launch"] / (copy_gbps * 1e9) * 1e3 * scale
    for name, v in (extra or {}).items():
        c[name] = v
    if not c or not measured_ms:
        return None
    binding = max(c, key=c.get)
    return {"measured_ms": round(measured_ms, 3), "ceilings_ms": {n: round(v, 3) for n, v in c.items()}, "binding": binding,
            "frac_of_binding_ceiling": round(c[binding] / measured_ms, 3)}


def _attach_physical(obj, k, copy_gbps):
    """The physical picture of one kernel from its (fresh) traffic-file entry."""
    if not k or not k.get("hbm_side_GBps"):
        return
    obj["physical_GBps"] = round(k["hbm_side_GBps"], 1)
    obj["frac_physical_of_spec"] = round(k["hbm_side_GBps"] / PEAK_GBPS, 4)
    obj["frac_physical_of_copy"] = round(k["hbm_side_GBps"] / copy_gbps, 4) if copy_gbps else None
    for src_key, dst_key in (("valu_lane_utilisation", "lane_utilisation"), ("wait_any_share_of_wave_cycles", "wait_share"), ("l2_hit_rate", "l2_hit_rate")):
        if k.get(src_key) is not None:
            obj[dst_key] = round(k[src_key], 4)


def roofline_objects(st, dt_total, copy_gbps, steps, closest_kernel="k_trace<0>", tag="C2", gen_bytes_per_sample=None, kernels_forced=False, pk=None,
                     profile=None):
    """The roofline objects of a run: the dominant kernel (per-lane closest-hit traversal), the camera rays' coherent kernel, shading,
    the sampler.  `achieved` = ALGORITHMIC bytes / the kernel's HIP-event time against the 8 TB/s spec;
    next to it, when the committed profile belongs to this library: memory-side traffic, lane utilisation, and the ceilings.
    profile: load_profile(tag) (the tests hand in their own)."""
    pk = pk or {"rays": 0, "nodes": 0, "tris": 0, "launches": 0, "ms": 0.0}
    alg_all_closest = 32.0 * st["closest_nodes"] + 48.0 * st["closest_tris"]
    alg_pk = 32.0 * pk["nodes"] + 48.0 * pk["tris"]
    alg = alg_all_closest - alg_pk  # the per-lane kernel alone: the camera rays' k_trace_pk gets its own object
    launches = max(1, st["closest_launches"] - pk["launches"])
    lane_ms = st["closest_ms"] - pk["ms"]
    achieved = alg / max(lane_ms * 1e-3, 1e-12) / 1e9
    all_alg = alg_all_closest + 32.0 * st["any_nodes"] + 48.0 * st["any_tris"]
    pname, prof, fresh = profile if profile is not None else load_profile(tag)
    # (the passes profiled ONE closest-hit kernel: when this run's pilot picked the other of a near tie, the dominant kernel has no profile;
    # shading, sampler and camera kernel are the same kernels either way)
    usable = fresh
    usable_dom = fresh and _find_kernel(prof, closest_kernel, exact=True) is not None
    stale = None if (usable or not pname) else "profiles/%s: taken with another library (kernel sources differ)" % pname
    stale_dom = stale if not usable else (None if usable_dom else "profiles/%s: another traversal kernel was profiled (%s)" % (pname, prof.get("kernel", "?").split("::")[-1]))
    roof = {"bound": "hbm", "kernel": closest_kernel, "achieved": round(achieved, 2), "peak": PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / PEAK_GBPS, 4), "saturated": bool(achieved / PEAK_GBPS > 1.0), "traffic": None,
            "hbm_copy_GBps": round(copy_gbps, 1) if copy_gbps else None,
            "alg_bytes_per_launch": round(alg / launches, 1), "avg_launch_ms": round(lane_ms / launches, 4), "launches": int(launches),
            "job_alg_GBps": round(all_alg / dt_total / 1e9, 2), "job_alg_GBps_without_camera_kernel": round((all_alg - alg_pk) / dt_total / 1e9, 2),
            "trace_share_of_time": round(st["trace_ms"] * 1e-3 / dt_total, 4), "stale_profile": stale_dom}
    sb = shade_alg_bytes(st)
    sa = sb / max(st["shade_ms"] * 1e-3, 1e-12) / 1e9
    shade = {"bound": "hbm", "kernel": "k_shade_path (+ k_env under an environment map)", "achieved": round(sa, 2), "peak": PEAK_GBPS, "unit": "GB/s",
             "frac": round(sa / PEAK_GBPS, 4), "traffic": None, "alg_bytes_per_item": round(sb / max(1, st["shade_items"]), 1),
             "items": int(st["shade_items"]), "vertices": int(st["shade_vertices"]), "cont_rays": int(st["shade_cont"]), "mis_rays": int(st["shade_mis"]),
             "shadow_rays": int(st["shade_shadow"]), "resolve_only_entries": int(st["shade_items"] - st["camera_samples"] - st["shade_cont"]),
             "stale_profile": stale}
    gen = None
    if gen_bytes_per_sample and st.get("gen_ms"):
        gb = gen_bytes_per_sample * st["camera_samples"]
        ga = gb / max(st["gen_ms"] * 1e-3, 1e-12) / 1e9
        gen = {"bound": "hbm", "kernel": "k_gen_samples_* + k_raygen", "achieved": round(ga, 2), "peak": PEAK_GBPS, "unit": "GB/s", "frac": round(ga / PEAK_GBPS, 4),
               "traffic": None, "alg_bytes_per_sample": round(gen_bytes_per_sample, 2), "stale_profile": stale}
    cam = None
    if pk["launches"]:
        ca = alg_pk / max(pk["ms"] * 1e-3, 1e-12) / 1e9
        cam = {"bound": "hbm", "kernel": "k_trace_pk", "achieved": round(ca, 2), "peak": PEAK_GBPS, "unit": "GB/s", "frac": round(ca / PEAK_GBPS, 4),
               "saturated": bool(ca / PEAK_GBPS > 1.0), "traffic": None, "alg_bytes_per_launch": round(alg_pk / pk["launches"], 1),
               "avg_launch_ms": round(pk["ms"] / pk["launches"], 4), "launches": int(pk["launches"]), "rays": int(pk["rays"]),
               "node_visits_per_ray": round(pk["nodes"] / max(1, pk["rays"]), 2), "stale_profile": stale}
    if usable:
        # the dominant kernel: memory-side bytes per launch, the observed bound, the ceilings
        k = _find_kernel(prof, closest_kernel, exact=True) if usable_dom else None
        if k:
            _attach_physical(roof, k, copy_gbps)
            roof["traffic"] = float(k["hbm_bytes_per_launch"]) if k.get("hbm_bytes_per_launch") else None
            if prof.get("alg_bytes_per_launch") and roof["traffic"] and not kernels_forced_mismatch(prof, closest_kernel):
                roof["traffic_over_algorithmic"] = round(roof["traffic"] / prof["alg_bytes_per_launch"], 3)
                scale = (alg / launches) / prof["alg_bytes_per_launch"]
                extra = {}
                rates = machine_rates()
                if rates:  # the per-CU L1's divergent-address path: two 16-byte lane loads per node visit, three per triangle test
                    nodes, tris = st["closest_nodes"] - pk["nodes"], st["closest_tris"] - pk["tris"]
                    extra["l1_divergent_loads_ms"] = (2.0 * nodes + 3.0 * tris) / launches / (rates["l1_hit_lane_loads_per_cu_cycle"] * CUS * CLOCK_HZ) * 1e3
                roof["ceilings"] = ceilings(k, scale, lane_ms / launches, copy_gbps, "trace", extra)
                if roof["ceilings"]:
                    roof["bound"] = "latency" if roof["ceilings"]["frac_of_binding_ceiling"] < 0.8 else roof["ceilings"]["binding"].replace("_ms", "")
            roof["profile"] = "profiles/" + pname
        # shading and the sampler: every launch of the profiled run's kernels of that kind, per render, against this run's time per step
        # (same config, same workload; the shading demand is scaled by the algorithmic bytes)
        renders = float(prof.get("pmc_renders", 2.0))
        ks = _find_kernel(prof, "k_shade_path")
        if ks:
            _attach_physical(shade, ks, copy_gbps)
            sh = prof.get("shade") or {}
            if sh.get("traffic_over_algorithmic"):
                shade["traffic_over_algorithmic"] = round(sh["traffic_over_algorithmic"], 3)
            tot, n_shade = per_render_totals(prof, ("k_shade_path", "k_env"), renders, count=("k_shade_path",))
            scale = (sb / steps) / (sh["alg_bytes_per_launch"] * n_shade) if sh.get("alg_bytes_per_launch") and n_shade else 1.0
            shade["ceilings"] = ceilings(tot, scale, st["shade_ms"] / steps, copy_gbps, "shade")
            if shade["ceilings"]:
                shade["ceilings"]["per"] = "step (all shading launches)"
                shade["bound"] = shade["ceilings"]["binding"].replace("_ms", "") if shade["ceilings"]["frac_of_binding_ceiling"] >= 0.5 else "latency"
        kg = _find_kernel(prof, "k_gen_samples")
        if gen and kg:
            _attach_physical(gen, kg, copy_gbps)
            tot, _ = per_render_totals(prof, ("k_gen_", "k_raygen", "k_mark_alive", "k_sum_alive"), renders)
            gen["ceilings"] = ceilings(tot, 1.0, st["gen_ms"] / steps, copy_gbps, "gen")
            if gen["ceilings"]:
                gen["ceilings"]["per"] = "step (all sampler / raygen launches)"
                gen["bound"] = gen["ceilings"]["binding"].replace("_ms", "") if gen["ceilings"]["frac_of_binding_ceiling"] >= 0.5 else "lds-latency (the shuffle's serial chain)"
        if cam:
            kc = _find_kernel(prof, "k_trace_pk")
            if kc:
                _attach_physical(cam, kc, copy_gbps)
                cam["traffic"] = float(kc["hbm_bytes_per_launch"]) if kc.get("hbm_bytes_per_launch") else None
                cam["ceilings"] = ceilings(kc, 1.0, pk["ms"] / pk["launches"], copy_gbps, "shade")  # (full lane width, scalar loads: VALU at what it occupies)
                if cam["ceilings"]:
                    cam["bound"] = cam["ceilings"]["binding"].replace("_ms", "") if cam["ceilings"]["frac_of_binding_ceiling"] >= 0.5 else "scalar-load latency"
    return roof, shade, gen, all_alg, cam, alg_pk


def per_render_totals(prof, prefixes, renders, count=None):
    """Counters of every profiled kernel whose name starts with one of `prefixes`, summed over the pass's launches, per render.
    -> (totals keyed like a traffic-file entry, launches per render of the `count` prefixes)."""
    keys = ("valu_active_quad_cycles_per_launch", "l2_requests_per_launch", "hbm_bytes_per_launch", "lds_array_cycles_per_launch", "read_requests_per_launch")
    tot, n = {}, 0.0
    for name, v in prof.get("kernels", {}).items():
        base = name.split("::")[-1]
        if not base.startswith(tuple(prefixes)):
            continue
        for key in keys:
            val = _valu_quad_cycles(v) if key == "valu_active_quad_cycles_per_launch" else v.get(key)
            if val is not None:
                tot[key] = tot.get(key, 0.0) + val * v["dispatches_in_pmc_pass"] / renders
        if count is None or base.startswith(tuple(count)):
            n += v["dispatches_in_pmc_pass"] / renders
    return tot, n


def kernels_forced_mismatch(prof, closest_kernel):
    """The profile's dominant kernel is another one than this run's (forced kernels, another pilot pick)."""
    return bool(prof) and bool(prof.get("kernel")) and prof["kernel"].split("::")[-1] != closest_kernel


def kernel_profile(tag, names, profile=None):
    """Per traversal kernel of the line: lane utilisation, wait share, L2 hit rate, memory-side rate -- from the committed PMC passes
    when they belong to this library; {"profiled": False} / {"stale_profile": file} otherwise."""
    pname, prof, fresh = profile if profile is not None else load_profile(tag)
    out = {}
    for role, kname in names.items():
        k = _find_kernel(prof, kname, exact=True) if prof else None
        if not k:
            out[role] = {"kernel": kname, "profiled": False}
        elif not fresh:
            out[role] = {"kernel": kname, "profiled": True, "stale_profile": "profiles/" + pname}
        else:
            out[role] = {"kernel": kname, "profiled": True, "source": "profiles/" + pname,
                         "lane_utilisation": round(k["valu_lane_utilisation"], 4) if k.get("valu_lane_utilisation") is not None else None,
                         "wait_share": round(k["wait_any_share_of_wave_cycles"], 4) if k.get("wait_any_share_of_wave_cycles") is not None else None,
                         "l2_hit_rate": round(k["l2_hit_rate"], 4) if k.get("l2_hit_rate") is not None else None,
                         "physical_GBps": round(k["hbm_side_GBps"], 1) if k.get("hbm_side_GBps") else None,
                         "avg_launch_ms_serial": round(k["avg_launch_ms_kernel_trace"], 3) if k.get("avg_launch_ms_kernel_trace") else None}
    return out


LINE_LIMIT = 8000  # bytes: the driver keeps a bounded window of stdout (round 5's 26 KB line was not parsed)
HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "saturated", "traffic", "traffic_over_algorithmic", "frac_physical_of_copy",
                 "frac_of_binding_ceiling", "binding", "avg_launch_ms", "launches", "alg_bytes_per_launch", "hbm_copy_GBps", "lane_utilisation", "stale_profile")


def _short_kernel(name):
    """`k_trace<0> (closest-hit BVH traversal)` -> `k_trace<0>`."""
    return name.split(" (")[0] if isinstance(name, str) else name


def _roofline_numbers(r):
    """The numbers of one roofline object (glossary: profiles/README.md); prose stays in the sidecar."""
    if not r:
        return None
    o = {k: r[k] for k in ROOFLINE_KEYS if k in r}
    if "kernel" in o:
        o["kernel"] = _short_kernel(o["kernel"])
    c = r.get("ceilings")
    if c:
        o["binding"] = c.get("binding")
        o["frac_of_binding_ceiling"] = c.get("frac_of_binding_ceiling")
    for k in ("traffic", "traffic_over_algorithmic", "frac_physical_of_copy", "frac_of_binding_ceiling"):
        o.setdefault(k, None)
    return o


def _workload_tag(workload):
    return workload.split(":")[0].strip()


def compact_line(d, detail_path=None):
    """The ONE JSON line of a run, numbers only and under LINE_LIMIT bytes, from the full result `d` (which goes to the sidecar
    file the line names).  Pure function of `d`: tests/test_bench_line.py builds it from canned results without a GPU."""
    line = {k: d[k] for k in HEADLINE_KEYS if k in d}
    c = d.get("config", {})
    tk = c.get("trace_kernels", {})
    lay = c.get("state_layout", "")
    line["config"] = {"workload": c.get("workload"), "triangles": c.get("triangles"), "bvh_nodes": c.get("bvh_nodes"),
                      "samples_per_step": c.get("samples_per_step"), "parallelism": c.get("parallelism"),
                      "kernels": {k: tk.get(k) for k in ("closest", "any_hit", "camera_rays")}, "kernel_ids": tk.get("ids"),
                      "any_hit_far_child_first": tk.get("any_hit_far_child_first"),
                      "kernels_forced": bool(str(tk.get("picked_by", "")).startswith("--trace-kernels")),
                      "state_layout": 4 if str(lay).startswith("four-slot") else 64,
                      "sampler": str(c.get("sampler_mode", "")).split(" ")[0] or None}
    line["roofline"] = _roofline_numbers(d.get("roofline"))
    others = {}
    for key, tag in (("roofline_camera", "camera"), ("roofline_shade", "shade"), ("roofline_gen", "gen")):
        r = _roofline_numbers(d.get(key))
        if r:
            others[tag] = {k: r.get(k) for k in ("kernel", "bound", "frac", "saturated", "frac_physical_of_copy", "binding", "frac_of_binding_ceiling") if r.get(k) is not None}
    if others:
        line["roofline_others"] = others
    cb = d.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "host_cores")}
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:120]
    ct = d.get("cpu_baseline_threads")
    if ct:
        line["cpu_baseline_threads"] = {k: ct.get(k) for k in ("value", "unit", "cores", "kind")}
    if d.get("kernel_ms_per_step"):
        line["kernel_ms_per_step"] = {k: v for k, v in d["kernel_ms_per_step"].items() if not isinstance(v, str)}
    if d.get("per_sample"):
        line["per_sample"] = d["per_sample"]
    for k in ("first_render_ms", "second_render_ms", "pilot_ms", "workspace_bytes", "first_render_workspace_bytes", "replay_Msamples_s", "rccl_world", "reduce_ms", "film_bytes_reduced_per_step"):
        if d.get(k) is not None:
            line[k] = d[k]
    if d.get("one_shot_ms"):
        line["one_shot_ms"] = d["one_shot_ms"].get("total")
    if d.get("replay"):
        line["replay_bit_exact"] = d["replay"].get("bit_exact_vs_serial_oracle")
    if d.get("n_gpus", 1) > 1:
        col = str(d.get("collective", ""))
        line["collective"] = ("gloo-rehearsal" if d.get("rehearsal") else "torch-rccl-fallback" if col.startswith("FALLBACK") else "ncclReduce")
        line["rehearsal"] = bool(d.get("rehearsal"))
        prs = d.get("per_rank_step_ms") or {}
        line["per_rank_step_ms"] = {k: prs.get(k) for k in ("min", "max")}
        ref = d.get("one_gpu_same_workload") or {}
        line["one_gpu_same_workload"] = {"value": ref.get("value"), "source": ref.get("source")}
        if d.get("trace_kernels_per_rank"):
            line["trace_kernels_per_rank"] = d["trace_kernels_per_rank"]
    ex = []
    for e in d.get("extra_configs") or []:
        r = e.get("roofline") or {}
        ex.append({"workload": _workload_tag(e.get("config", {}).get("workload", "?")), "value": e.get("value"), "ms_per_step": e.get("ms_per_step"),
                   "steps": e.get("steps"), "kernel": _short_kernel(r.get("kernel")), "roofline_frac": r.get("frac"),
                   "frac_physical_of_copy": r.get("frac_physical_of_copy"),
                   "frac_of_binding_ceiling": (r.get("ceilings") or {}).get("frac_of_binding_ceiling")})
        if e.get("scaling"):
            ex[-1]["scaling"] = e["scaling"]
    if ex:
        line["extra_configs"] = ex
    line["detail"] = detail_path
    line["glossary"] = "profiles/README.md"
    # the size is enforced, not hoped for: drop the optional objects, largest first, until the line fits
    for k in ("roofline_others", "per_sample", "cpu_baseline_threads", "extra_configs", "kernel_ms_per_step"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(k, None)
    if len(json.dumps(line)) >= LINE_LIMIT:
        raise AssertionError("bench line is %d bytes (limit %d)" % (len(json.dumps(line)), LINE_LIMIT))
    return line


def write_detail(d):
    """The full result (every roofline object, ceilings, the occupancy model, the pilot table, replay, one-shot costs, all prose)
    as a sidecar file; returns its path relative to the repo (None when no directory is writable)."""
    tag = _workload_tag(d.get("config", {}).get("workload", "run")).split(" ")[0].lower()
    name = "bench_detail_%s_n%d.json" % (tag, d.get("n_gpus", 1))
    for base in (os.environ.get("DARTRAY_BENCH_DETAIL_DIR"), os.path.join(ROOT, "gpurun_out"), ROOT, "/tmp"):
        if not base:
            continue
        try:
            os.makedirs(base, exist_ok=True)
            path = os.path.join(base, name)
            with open(path, "w") as fh:
                json.dump(d, fh, indent=1)
            return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
        except OSError:
            continue
    return None


def load_detail(path):
    """A full result: a sidecar file (one JSON document) or a log whose last line is one (the round 1-5 lines under profiles/)."""
    text = open(path).read().strip()
    try:
        return json.loads(text)
    except ValueError:
        return json.loads(text.splitlines()[-1])


def emit(d):
    """Sidecar first, then the line -- the LAST thing this process writes to stdout."""
    path = write_detail(d)
    sys.stdout.flush()
    print(json.dumps(compact_line(d, path)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--res", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short C4 / C5 runs appended at N = 1")
    ap.add_argument("--scaling", default=None, choices=["c3", "weak", "strong-c2", "samples"],
                    help="N > 1: c3 (default) = BASELINE configs[2] verbatim, tiles of the 4096^2 x 1024 spp image over the N ranks; weak = "
                         "tiles of the C2 image with N x the pixels (the N = 1 workload per GPU); strong-c2 = tiles of the 1024^2 image; "
                         "samples = every rank renders 256 spp of the whole C2 image with its own seed, films summed")
    ap.add_argument("--cpu-pixels", type=int, default=96, help="cpu baseline: side of the strided pixel grid")
    ap.add_argument("--blob", default=None, help="diagnostics only (NOT a BASELINE config): C2 / C3 with a blob of SEGMENTS,ROWS (2 x SEGMENTS x ROWS triangles) "
                                                 "instead of 1000,500 -- e.g. 180,90 = 32 400 triangles, a tree that fits one XCD's L2; the line says so")
    ap.add_argument("--from-detail", default=None, help="no run: print the line of a committed / sidecar detail file (CPU)")
    ap.add_argument("--trace-kernels", default=None, help="A/B runs: force the traversal kernels (closest,any), e.g. 2,3 (default: the scene's pilot)")
    args = ap.parse_args()

    if args.from_detail:
        print(json.dumps(compact_line(load_detail(args.from_detail), os.path.relpath(os.path.abspath(args.from_detail), ROOT))), flush=True)
        return

    if check_world(args.gpus, int(os.environ.get("WORLD_SIZE", "0"))) == "launch":
        launch_ranks(args)

    import torch
    from dartray_amd import _abi, scenes, dist as drdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    local = drdist.local_device(int(os.environ.get("LOCAL_RANK", "0")))  # (= LOCAL_RANK; DARTRAY_COMM_REHEARSAL shares GPUs between ranks)
    torch.cuda.set_device(local)
    os.environ.setdefault("DARTRAY_COMM_FALLBACK", "1")  # N > 1: see dartray_amd/dist.py (_fallback_group); reported on the line
    rank, world, local = drdist.init_process_group()  # gloo control plane + dr_comm_init (RCCL) on every rank
    local = drdist.local_device(local)
    lib = _abi.lib()
    _abi.init(local)  # before the scenes are built: BVHAccel then takes the device builder (dr_bvh_build_device)

    pl = plan(world, args.scaling, args.config, args.res, args.spp, args.no_extra)
    cfg, res, spp, mode = pl["headline"]
    run = Run(cfg, res, spp, rank, world, mode, args)
    out = run.headline(args.steps, args.warmup)

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(run.prims, run.renderer, args.cpu_pixels, run.H, run.W, spp)
            out["cpu_baseline_threads"] = cpu_baseline_threads(run.prims, run.renderer, run.H, run.W, spp)
            if cfg == "C2":
                out["replay"] = replay_leg(run)
                out["replay_Msamples_s"] = out["replay"]["value"]
        if world == 1 and pl["extras"]:
            del run
            torch.cuda.empty_cache()
            extra = []
            for (c, eres, espp, emode, esteps, ewarm) in pl["extras"]:
                r = Run(c, eres, espp, 0, 1, emode, args)
                e = r.headline(esteps, ewarm)
                extra.append(e)
                del r
                torch.cuda.empty_cache()
            out["extra_configs"] = extra
    if world > 1 and pl["extras"]:
        # next to the N-rank headline (configs[2]): the N = 1 line's workload per GPU -- the C2 view at 256 spp on a square image
        # with N x the pixels, tiles over the N ranks, one reduce per step ("weak") -- every rank takes part; after the
        # headline's timed region, like C4 / C5 at N = 1; first render + one warm-up + 2 timed steps of ~0.45 s.
        del run
        torch.cuda.empty_cache()
        extra = []
        for (c, eres, espp, emode, esteps, ewarm) in pl["extras"]:
            rx = Run(c, eres, espp, rank, world, emode, args)
            e = rx.headline(esteps, ewarm)
            if rank == 0:
                extra.append(e)
            del rx
            torch.cuda.empty_cache()
        if rank == 0:
            out["extra_configs"] = extra
    drdist.barrier()
    drdist.comm_destroy()
    if world > 1:
        torch.distributed.destroy_process_group()
    if rank == 0:
        emit(out)


class Run:
    """One configuration resident on this rank's GPU: scene, renderer, device film."""

    def __init__(self, cfg, res, spp, rank, world, mode, args):
        import torch
        from dartray_amd import _abi, scenes, dist as drdist
        self.cfg, self.res, self.spp, self.rank, self.world, self.mode, self.args = cfg, res, spp, rank, world, mode, args
        skw = {"blob": tuple(int(x) for x in args.blob.split(","))} if getattr(args, "blob", None) and cfg in ("C2", "C3") else {}
        self.prims, mk = scenes.config("C2" if cfg == "C3" else cfg, xres=res, yres=res, spp=spp, **skw)
        r = mk()
        if mode == "samples" and world > 1:
            r = drdist.sample_set(r, rank)  # independent sample sets of the same image
        else:
            r = drdist.shard(r, rank, world)  # round-robin 32 x 32 tiles
        self.renderer = r
        self.scene = scenes.make_scene(self.prims, r.env)  # every rank builds + uploads its own copy (render_isolate.dart:31-41)
        fd = r.camera.film
        self.H, self.W = fd.height, fd.width
        self.film = torch.zeros((self.H, self.W, 4), dtype=torch.float32, device="cuda")
        self.rgb = torch.zeros((self.H, self.W, 3), dtype=torch.float32, device="cuda")
        self.stream = torch.cuda.current_stream().cuda_stream
        t0 = time.perf_counter()
        self.dev = self.scene._device()  # dr_scene_create: validation, pair records, uploads (host pointers in)
        self.create_ms = (time.perf_counter() - t0) * 1e3
        self.lib = _abi.lib()
        if getattr(args, "trace_kernels", None):
            self.dev.trace_kernels(tuple(int(x) for x in args.trace_kernels.split(",")))

    def step(self, first=False):
        import torch
        from dartray_amd import _abi, dist as drdist
        self.film.zero_()
        if first and self.world > 1:
            # the scene's first render carries the pilot: rank 0 renders first, its picks (kernel pair, state layout) go to the other
            # ranks over the control plane, and they render without calibration batches of their own (dartray_amd/dist.py: share_pilot)
            if self.rank == 0:
                self.renderer.render_device(self.scene, self.film.data_ptr(), self.stream)
                torch.cuda.synchronize()
            self.picks = drdist.share_pilot(self.dev)
            if self.rank != 0:
                self.renderer.render_device(self.scene, self.film.data_ptr(), self.stream)
        else:
            self.renderer.render_device(self.scene, self.film.data_ptr(), self.stream)
        drdist.reduce_film(self.film, 0, self.stream)
        if self.rank == 0:
            _abi.check(self.lib.dr_film_resolve_device(self.film.data_ptr(), self.H * self.W, self.rgb.data_ptr(), self.stream))

    def headline(self, steps, warmup):
        import ctypes as C
        import torch
        from dartray_amd import _abi, dist as drdist
        args, world, rank = self.args, self.world, self.rank
        # The first render of a scene: workspace allocation + the traversal-kernel pilot (MEASUREMENTS.md) -- what a
        # one-shot render pays on top of a steady-state step; timed on its own, not one of the W + K steps.
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.step(first=True)
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) * 1e3
        picks_per_rank = drdist.gather_picks(self.dev)  # (closest, any_hit, layout) of every rank's scene after its first render
        if rank == 0:
            self.rgb.cpu()  # the image back on the host: what Renderer.render hands to its caller
        one_shot_render_ms = (time.perf_counter() - t0) * 1e3
        pilot_ms = self.dev.stats()["pilot_ms"]
        first_workspace = self.dev.workspace_bytes()
        # the SECOND render of a scene: the library kept the first one's batches at 2^27 slots (what a one-shot host pays for) and now grows the
        # workspace to one batch per image -- the steady state of a frame loop; timed on its own.  It is the first of the W warm-up steps
        # (an extra one when W = 0: the timed steps must run on the grown workspace), never one of the K timed steps.
        t0 = time.perf_counter()
        self.step()
        torch.cuda.synchronize()
        second_ms = (time.perf_counter() - t0) * 1e3
        warmup_left = max(0, warmup - 1)
        for _ in range(warmup_left):
            self.step()
        torch.cuda.synchronize()
        self.dev.reset_stats()
        drdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        dt_own = time.perf_counter() - t0  # this rank's own share (BEFORE the barrier: after it every rank reads the same clock)
        drdist.barrier()
        torch.cuda.synchronize()
        dt_rank = time.perf_counter() - t0
        dt = drdist.max_over_ranks(dt_rank)
        own_max = drdist.max_over_ranks(dt_own)
        own_min = -drdist.max_over_ranks(-dt_own)
        st = self.dev.stats()
        reduce_ms = None
        if world > 1:  # the collective alone (after the timed region): three reduces of a scratch film, barrier-bracketed
            scratch = torch.zeros_like(self.film)
            torch.cuda.synchronize()
            drdist.barrier()
            t1 = time.perf_counter()
            for _ in range(3):
                drdist.reduce_film(scratch, 0, self.stream)
            torch.cuda.synchronize()
            reduce_ms = drdist.max_over_ranks(time.perf_counter() - t1) / 3 * 1e3
            del scratch
        if rank != 0:
            return None
        copy = C.c_double(0.0)
        _abi.check(self.lib.dr_copy_bandwidth(1 << 30, 10, C.byref(copy)))  # measured HBM denominator (float4 copy, 2 GiB moved per pass)
        samples_per_step = self.H * self.W * self.spp * (world if (self.mode == "samples" and world > 1) else 1)
        value = samples_per_step * steps / dt / 1e6
        picked = self.dev.trace_kernels()  # (closest, any): 2 = k_trace, 3 = k_trace3 / k_trace3a, 5 = k_trace3c (the pilot's choice for this scene)
        # (lazy generation shuffles bounce b's blocks only for pixel groups alive at bounce b: the blocks that count are the ones generated)
        ss = self.dev.sampler_stats()
        gen_share = ss["generated"] / ss["named"] if ss["named"] else 1.0
        gbs = gen_alg_bytes_per_sample(gen_blocks(self.renderer, self.scene) * gen_share, self.spp)
        forced = bool(getattr(args, "trace_kernels", None)) or bool(os.environ.get("DARTRAY_TRACE_IMPL"))
        knames = {"closest": {2: "k_trace<0>", 3: "k_trace3<0>", 5: "k_trace3c"}.get(picked[0], str(picked[0])),
                  "any_hit": {2: "k_trace<1>", 3: "k_trace3a", 6: "k_trace<1>", 7: "k_trace3a"}.get(picked[1], str(picked[1]))}
        far_first = picked[1] in (6, 7)  # the any-hit rays take the far child first (the boolean of intersectP does not depend on the order)
        pk = self.dev.coherent_stats()
        profile = load_profile(self.cfg)
        if os.environ.get("DARTRAY_BATCH_BITS") or os.environ.get("DARTRAY_TRACE_WG_PER_CU"):
            profile = (profile[0], profile[1], False)  # (launches of another size / occupancy than the profiled ones)
        roof, shade, gen, all_alg, cam, alg_pk = roofline_objects(st, dt, copy.value, steps, knames["closest"], self.cfg, gbs, kernels_forced=forced, pk=pk, profile=profile)
        if gen:
            gen["blocks_generated_of_named"] = round(gen_share, 4)
        pilot = self.dev.pilot()
        near = []  # picks the calibration batches decided by less than 2 %
        pc, pa = pilot["closest"], pilot["any_hit"]
        if not forced and pc[2] > 0 and pc[3] > 0:
            cands = {k: v for k, v in pc.items() if v > 0}
            best = min(cands.values())
            near += ["closest: kernel %d within 2 %% of kernel %d" % (k, picked[0]) for k, v in cands.items() if k != picked[0] and v < 1.02 * best]
            if pa[2] > 0 and pa[3] > 0 and abs(pa[2] - pa[3]) < 0.02 * min(pa[2], pa[3]):
                near.append("any_hit: kernels 2 and 3 within 2 %")
        agg = self.scene.aggregate
        red = ("film summed through host memory by gloo (REHEARSAL on shared GPUs)" if drdist.rehearsal() else
               "RCCL film reduce (world %d)" % self.lib.dr_comm_world())
        if world == 1:
            par = "1 GPU"
        elif self.mode == "samples":
            par = "spp-sets x%d, %s" % (world, red)
        else:
            par = "tiles32 x%d, %s" % (world, red)
        out = {
            "metric": "Msamples/sec (primary+path rays)",
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",  # (N = 1, and the per-GPU-fixed modes; the N-rank C3 / strong-c2 lines overwrite it below)
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: %s, PathIntegrator maxdepth=%d, %dx%d, %d spp, LD sampler (device, counter streams), box filter"
                       % (self.cfg + (" VARIANT (--blob %s: not a BASELINE config)" % args.blob if getattr(args, "blob", None) else ""),
                          NAMES[self.cfg], self.renderer.surfaceIntegrator.maxDepth, self.res, self.res, self.spp),
                       "triangles": int(len(agg.tri_idx)), "bvh_nodes": int(len(agg.nodes)),
                       "bvh_builder": "%s (dr_bvh_build_%s), %.0f ms" % (agg.builder, "device" if agg.builder == "device" else "mixed", agg.build_ms),
                       "trace_kernels": {"closest": knames["closest"], "any_hit": knames["any_hit"], "any_hit_far_child_first": far_first,
                                         "ids": [int(picked[0]), int(picked[1])],
                                         "pilot_far_first_over_reference_order_per_ray": round(pilot.get("far_first", 0.0), 4),
                                         "camera_rays": "k_trace_pk" if pk["launches"] else knames["closest"],
                                         "picked_by": ("--trace-kernels / DARTRAY_TRACE_IMPL" if forced else
                                                       "the scene's pilot batches (closest-hit: best time per algorithmic byte, a pair kernel needs 5 %; any-hit: "
                                                       "the closest-hit kernel's family unless the other wins its own batch by more than 15 %)"),
                                         "pilot_ms_per_alg_GB": {"closest": {str(k): round(v, 4) for k, v in pc.items()},
                                                                 "any_hit": {str(k): round(v, 4) for k, v in pa.items()}},
                                         "near_ties": near},
                       "samples_per_step": samples_per_step, "parallelism": par,
                       "sampler_mode": "DR_SAMPLER_COUNTER (keyed per-pixel streams, bit-exact vs the oracle's same mode); the reference's "
                                       "single serial Random(taskNum) stream is replayed bit-exactly through DR_SAMPLER_HOST_BUFFER in the "
                                       "tests but is not a throughput mode (one serial stream; 148 B + RNG tail per sample over PCIe)"},
            "roofline": roof,
            "roofline_shade": shade,
            "roofline_gen": gen,
            "roofline_camera": cam,
            # per traversal kernel of this line, from the committed PMC passes of the same command (not this run): where the lanes and
            # the wave-cycles go -- the figures that move when a traversal kernel gets better or worse
            "trace_kernels_profiled": kernel_profile(self.cfg, knames, profile),
            "kernel_ms_per_step": dict({k: round(st[k] / steps, 2) for k in ("closest_ms", "any_ms", "shade_ms", "gen_ms", "film_ms", "total_ms")},
                                       note="a stage's any-hit launch runs beside its closest-hit launch (second stream): any_ms is its time "
                                            "after the closest-hit launch ended" if os.environ.get("DARTRAY_OVERLAP_ANY", "1") != "0" else "one kernel at a time"),
            "per_sample": {"rays": round((st["closest_rays"] + st["any_rays"]) / max(1, st["camera_samples"]), 3),
                           "nodes": round((st["closest_nodes"] + st["any_nodes"]) / max(1, st["camera_samples"]), 2),
                           "tris": round((st["closest_tris"] + st["any_tris"]) / max(1, st["camera_samples"]), 3),
                           # (148 B of sample vector + 32 B of film per sample on top of the traversal bytes)  the reference-count figure prices every
                           # node visit of every ray at 32 B; k_trace_pk loads a node once per WAVE: the second figure leaves its never-requested bytes out
                           "alg_bytes": round(all_alg / max(1, st["camera_samples"]) + 148 + 32, 1),
                           "alg_bytes_without_camera_kernel": round((all_alg - alg_pk) / max(1, st["camera_samples"]) + 148 + 32, 1)},
            "first_render_ms": round(first_ms, 1),
            "second_render_ms": round(second_ms, 1),
            "pilot_ms": round(pilot_ms, 1),
            "workspace_bytes": int(self.dev.workspace_bytes()), "first_render_workspace_bytes": int(first_workspace),
            # what a one-shot host pays from host pointers to the image on the host (Renderer.render once per task,
            # lib/dartray/dartray.dart:574): the BVHAccel constructor, the scene upload, the first render incl. resolve and copy out
            "one_shot_ms": {"bvh_build": round(agg.build_ms, 1), "scene_create": round(self.create_ms, 1),
                            "first_render_resolve_copy_out": round(one_shot_render_ms, 1),
                            "total": round(agg.build_ms + self.create_ms + one_shot_render_ms, 1),
                            "steady_step": round(dt / steps * 1e3, 1)},
        }
        lay, dens = self.dev.state_layout()
        out["config"]["state_layout"] = ("%s, picked from the first pilot batch: %.2f of its slots alive at the second bounce" % (
            "four-slot line-grouped sub-tiles (sp4)" if lay == 4 else "64-slot runs", dens)) if lay else "64-slot runs (no pilot: small render)"
        if world > 1:
            out["scaling"] = {"c3": "strong", "strong-c2": "strong", "weak": "weak", "samples": "weak"}[self.mode]
            out["rccl_world"] = int(self.lib.dr_comm_world())
            out["collective"] = ("dr_film_reduce (C ABI: ncclReduce(sum, f32) over xGMI, librccl bound at run time)" if drdist.comm_error is None else
                                 (drdist.comm_error if drdist.rehearsal() else "FALLBACK torch.distributed RCCL group -- dr_comm_init failed: %s" % drdist.comm_error))
            if drdist.rehearsal():
                out["rehearsal"] = "NOT a measurement: %d ranks on %d GPU(s)" % (world, torch.cuda.device_count())
            # what every rank runs (rank 0's pilot, handed on): printed on the line only where the ranks differ
            out["trace_kernels_all_ranks"] = [list(p) for p in picks_per_rank]
            if len(set(picks_per_rank)) > 1:
                out["trace_kernels_per_rank"] = [list(p) for p in picks_per_rank]
            out["per_rank_step_ms"] = {"min": round(own_min / steps * 1e3, 3), "max": round(own_max / steps * 1e3, 3),
                                       "note": "each rank's own clock before the closing barrier (rank 0's includes waiting for the reduce)"}
            out["reduce_ms"] = round(reduce_ms, 3)
            out["film_bytes_reduced_per_step"] = int(self.H * self.W * 16)
            out["one_gpu_same_workload"] = one_gpu_reference(self.mode, self.cfg)
        return out


def one_gpu_reference(mode, cfg):
    """The single-GPU rate a scaling figure of this N-rank line should be read against, from a committed single-GPU run
    of the same per-GPU (weak) or the same total (strong) workload -- labelled with its source, not measured now."""
    if mode in ("weak", "samples", "strong-c2"):
        finals = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.startswith("r") and n.endswith("_bench_final.json"))
        src = "profiles/" + finals[-1] if finals else None  # the newest committed single-GPU line
        key = "C2, 1024x1024, 256 spp on one GPU: the per-GPU workload of the weak modes, the total one of strong-c2"
    else:  # the newest committed whole-C3-on-one-GPU line (`bench.py --config C3 --steps 1 --warmup 0 --no-extra --no-cpu-baseline`)
        c3 = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.startswith("r") and n.endswith("_c3_one_gpu.json"))
        src = "profiles/" + c3[-1] if c3 else None
        key = "C3 (4096x4096, 1024 spp, the whole image) on one GPU" + ("" if src else ": 602 Msamples/s, 28.5 s per step (MEASUREMENTS.md, round 2)")
    ref = {"workload": key, "source": src}
    if src and os.path.exists(os.path.join(ROOT, src)):
        try:
            ref["value"] = json.loads(open(os.path.join(ROOT, src)).read().strip().splitlines()[-1])["value"]
            ref["unit"] = "Msamples/s"
        except (ValueError, KeyError):
            pass
    elif mode == "c3":
        ref["value"], ref["unit"], ref["source"] = 602.0, "Msamples/s", "MEASUREMENTS.md (round 2)"
    return ref


def replay_leg(run, res=128, spp=64):
    """The reference-exact sampler mode as a number.  The reference draws every sample of a task from ONE serial generator
    threaded through the sampler and the integrator (lib/renderers/sampler_renderer.dart:137,
    lib/samplers/low_discrepancy_sampler.dart:64-88, lib/core/montecarlo.dart:407-473), so "identical Sampler RNG seeds" means
    replaying that stream: the CPU oracle renders a res x res x spp window of the bench scene in its serial mode (timed: a CPU
    baseline in the reference's own mode) and records every sample's inputs; the device renders the recording through
    DR_SAMPLER_HOST_BUFFER from host pointers (dr_render: sample vectors and the RNG tail over PCIe in, film and image out)
    and must return the oracle's film bit for bit."""
    import numpy as np
    import oracle.binding as ob
    from dartray_amd import core, scenes
    _, mk = scenes.config("C2", xres=res, yres=res, spp=spp, blob=(2, 2))  # (the renderer only: same camera, a res x res film)
    r = mk()
    n = (res + 1) * (res + 1) * spp
    osc = ob.OracleScene(run.prims)
    t0 = time.perf_counter()
    rec = osc.render(ob.render_desc(r, sampler_mode=0), record=n, max_tail=40)
    t_cpu = time.perf_counter() - t0
    # (round 5: the PACKED tail -- DrRenderDesc.tail_offsets -- only the values a sample drew cross the host link)
    r.sampler = core.HostBufferSampler(r.camera, spp, rec["pixel_xy"][::spp], rec["sample_vec"], rec["tail"], rec["tail_count"])
    fixed_tail_bytes = rec["tail"].nbytes
    out = r.render(run.scene)  # warm-up: the workspace of this shape
    exact = bool(np.array_equal(out.film, rec["film"]) and np.array_equal(out.rgb, rec["rgb"]))
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        out = r.render(run.scene)
    dt = (time.perf_counter() - t0) / reps
    film_samples = res * res * spp
    bytes_in = rec["sample_vec"].nbytes + r.sampler.tail.nbytes + r.sampler.tail_offsets.nbytes + rec["pixel_xy"][::spp].nbytes
    return {"value": round(film_samples / dt / 1e6, 3), "unit": "Msamples/s", "bit_exact_vs_serial_oracle": exact,
            "samples_traced": n, "ms_per_render": round(dt * 1e3, 2), "host_bytes_in_per_render": int(bytes_in),
            "cpu_serial_oracle_Msamples_s": round(film_samples / t_cpu / 1e6, 4),
            "host_bytes_in_with_the_fixed_tail_form": int(bytes_in - r.sampler.tail.nbytes - r.sampler.tail_offsets.nbytes + fixed_tail_bytes),
            "what": "DR_SAMPLER_HOST_BUFFER replay of the reference's ONE serial Random(taskNum) stream (recorded by the CPU oracle in its "
                    "serial mode): %dx%d film window of the bench scene x %d spp, dr_render from host pointers -- %d B of sample vector + "
                    "%.0f B of RNG tail per sample (packed: the values a sample drew + one 8-byte offset; the fixed form is %d B) over PCIe in, "
                    "film + image out; the price of 'identical Sampler RNG seeds' against the keyed per-pixel streams of the headline"
                    % (res, res, spp, rec["sample_vec"].shape[1] * 4, (r.sampler.tail.nbytes + r.sampler.tail_offsets.nbytes) / n, rec["tail"].shape[1] * 8)}


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
            "sample": "%dx%d px, stride %d, of %dx%d x %d spp = %d samples, %.1f s (oracle/, 1 thread)" % (grid, grid, H // grid, W, H, spp, n, dt),
            "host_cores": os.cpu_count()}


def cpu_baseline_threads(prims, renderer, H, W, spp, budget_samples=3.0e7):
    """the reference's multi-worker model -- one isolate per task, task t tracing its
    GetSubWindow rectangle (lib/core/common.dart:52-73, dartray_web/render_manager.dart:100-141) -- as one OS thread
    per task running the unmodified serial oracle loop over a shared read-only scene, task count = host cores.  Bounded:
    every task traces a strided subset of ITS rectangle (the same stride for all, so the load imbalance between
    rectangles is kept).  ctypes releases the GIL inside the call."""
    import copy
    import threading
    import numpy as np
    import oracle.binding as ob
    osc = ob.OracleScene(prims)
    ntasks = max(1, os.cpu_count() or 1)
    stride = 1
    while (H // stride) * (W // stride) * spp > budget_samples:
        stride += 1
    parts = []
    for t in range(ntasks):
        r = copy.copy(renderer)
        r.taskNum, r.taskCount, r.tileRank, r.tileCount = t, ntasks, 0, 1
        px = r.pixels()  # this task's GetSubWindow rectangle (dr_enumerate_pixels, host-only)
        px = px[(px[:, 0] % stride == stride // 2) & (px[:, 1] % stride == stride // 2) & (px[:, 0] < W) & (px[:, 1] < H)]
        if len(px):
            parts.append(px)
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
    n = sum(len(p) for p in parts) * spp
    return {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": len(parts), "kind": "port",
            "sample": "%d GetSubWindow tasks of %dx%d (one thread each), stride %d x %d spp = %d samples, %.1f s (oracle/)" % (len(parts), W, H, stride, spp, n, dt),
            "host_cores": os.cpu_count()}


if __name__ == "__main__":
    main()
