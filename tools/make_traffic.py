"""profiles/r0N_<cfg>_traffic.json from the PMC summaries of tools/profile_r0N.sh.

usage: make_traffic.py SRC cfg OUT [round]
  SRC = the script's output directory (pmc_<cfg>_rdreq.txt, <cfg>_kernel_stats_serial.csv, <cfg>_bench.json), or
  SRC = profiles with the committed names (<round>_pmc_<cfg>_rdreq.txt, <round>_<cfg>_kernel_stats_serial.csv,
        <round>_<cfg>_bench_under_rocprof.json) -- so that every traffic file can be rebuilt from what is committed.

HBM-side bytes per launch of every kernel of the library, exactly as MI355X_MICROARCH.md prescribes (separate --pmc
passes; reads = TCC_EA0_RDREQ_128B * 128 + _64B * 64 + _32B * 32 -- every wide read of these kernels is a 128-B request,
so FETCH_SIZE would under-report 2x; writes = WRITE_SIZE KB * 1024).  A counter block WITHOUT a kernel header is an
error (round 3: such a block -- the builder kernels, whose names an over-eager strip had emptied -- was filed under
the kernel printed above it; files written by that version are read with the block kept apart as `<unnamed>`)."""
import csv
import json
import os
import re
import sys

# The sources whose text decides what a kernel does (bench.py compares their hashes with the running library's before it prints a
# figure derived from a traffic file: profiles of another library are "stale_profile", not numbers).
KERNEL_SOURCES = ("dr_trace.hip", "dr_kernels.hip", "dr_device.h", "dr_kernels.h", "dr_wave.h", "dr_rng.h")


def library_sources(src, rnd):
    """sha256 of the kernel sources of the library the passes profiled: OUTDIR/buildinfo.json (copied there by the profile script)
    or profiles/<round>_buildinfo.json.  None for rounds that did not record it (r01-r05)."""
    for name in ("buildinfo.json", "%s_buildinfo.json" % rnd):
        p = os.path.join(src, name)
        if os.path.exists(p):
            have = json.load(open(p)).get("sources", {})
            return {k: have[k] for k in KERNEL_SOURCES if k in have}
    return None


def parse_pmc(path):
    """{kernel: {"dispatches": n, counter: per-dispatch value}} of one pmc_summary.py text."""
    d, k = {}, None
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(.*?) ?dispatches=(\d+)\s*$", line)
        if m and not line.startswith("   "):
            k = m.group(1).strip() or "<unnamed>"
            if k in d:
                raise ValueError("%s:%d: kernel %r listed twice" % (path, ln, k))
            d[k] = {"dispatches": int(m.group(2))}
            continue
        m = re.match(r"^\s+(\S+)\s+(\S+)\s+per-dispatch\s+(\S+)", line)
        if m:
            if k is None:
                raise ValueError("%s:%d: counter line before any kernel header" % (path, ln))
            d[k][m.group(1)] = float(m.group(3))
        elif line.strip():
            raise ValueError("%s:%d: unrecognised line %r" % (path, ln, line))
    return d


def is_ours(name):
    base = name.split("::")[-1]
    return base.startswith("k_") or base.startswith("kb_")


def build(src, cfg, rnd):
    def first(*names):
        for n in names:
            p = os.path.join(src, n)
            if os.path.exists(p):
                return p
        return None

    def pmc(kind):
        p = first("pmc_%s_%s.txt" % (cfg, kind), "%s_pmc_%s_%s.txt" % (rnd, cfg, kind))
        return (parse_pmc(p), p) if p else ({}, None)

    (rd, rdp), (wr, wrp), (sq, sqp), (tcc, tccp), (lds, ldsp) = pmc("rdreq"), pmc("wrreq"), pmc("sq"), pmc("tcc"), pmc("lds")
    if not rd or not wr:
        raise SystemExit("missing pmc_%s_rdreq / wrreq under %s" % (cfg, src))
    statsp = first("%s_kernel_stats_serial.csv" % cfg, "%s_%s_kernel_stats_serial.csv" % (rnd, cfg), "%s_kernel_stats.csv" % cfg,
                   "%s_%s_kernel_stats.csv" % (rnd, cfg))
    stats = {r["Name"]: r for r in csv.DictReader(open(statsp))}
    benchp = first("%s_bench.json" % cfg, "%s_%s_bench_under_rocprof.json" % (rnd, cfg))
    text = open(benchp).read().strip()
    try:
        bench = json.loads(text)  # round 6: the full result (sidecar), one JSON document
    except ValueError:
        bench = json.loads(text.splitlines()[-1])  # rounds 1-5: the last line of the log
    res = {"workload": bench["config"]["workload"], "bench_value_under_rocprof": bench["value"], "kernels": {}}
    srcs = library_sources(src, rnd)
    if srcs:
        res["library_sources"] = srcs
    res["pmc_renders"] = 3.0 if rnd >= "r06" else 2.0  # renders of a PMC pass (`--steps 1 --warmup 0`): the first, (r06: the second,) one step
    for name in rd:
        if not is_ours(name):
            continue
        r, w, s, t, l = rd[name], wr.get(name, {}), sq.get(name, {}), tcc.get(name, {}), lds.get(name, {})
        reads = r.get("TCC_EA0_RDREQ_128B", 0) * 128 + r.get("TCC_EA0_RDREQ_64B", 0) * 64 + r.get("TCC_EA0_RDREQ_32B", 0) * 32
        writes = w.get("WRITE_SIZE", 0) * 1024
        # the stats file prints full signatures: match on the name in front of the argument list
        full = [v for k, v in stats.items() if k.replace("void ", "").startswith(name + "(") or k.replace("void ", "") == name]
        avg_ms = float(full[0]["AverageNs"]) * 1e-6 if full else None
        e = {"dispatches_in_pmc_pass": r["dispatches"], "hbm_bytes_per_launch": reads + writes, "reads_bytes_per_launch": reads,
             "writes_bytes_per_launch": writes, "read_requests_per_launch": r.get("TCC_EA0_RDREQ"),
             "avg_launch_ms_kernel_trace": avg_ms,
             "hbm_side_GBps": (reads + writes) / (avg_ms * 1e-3) / 1e9 if avg_ms else None}
        if s and "SQ_INSTS_VALU" in s and s["SQ_INSTS_VALU"] > 0:
            # SQ_ACTIVE_INST_VALU counts quad-cycles per wave instruction (4 cycles each, the cost of a wave issuing ALONE);
            # with other waves interleaved a wave64 instruction occupies the SIMD for 2 cycles (MI355X_MICROARCH.md), so the
            # pipe's real occupancy is about half of this figure (A/B runs: MEASUREMENTS.md, round 2).
            cycles = s["SQ_ACTIVE_INST_VALU"] * 4 / 1024.0  # quad-cycles -> cycles per SIMD
            e["valu_issue_share_at_4_cycles_per_instruction"] = cycles / (avg_ms * 1e-3 * 2.4e9) if avg_ms else None
            e["valu_lane_utilisation"] = s["SQ_THREAD_CYCLES_VALU"] / (s["SQ_INSTS_VALU"] * 64)
            e["wait_any_share_of_wave_cycles"] = s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"] if s.get("SQ_WAVE_CYCLES") else None
        if s and s.get("SQ_ACTIVE_INST_VALU") is not None:
            e["valu_active_quad_cycles_per_launch"] = s["SQ_ACTIVE_INST_VALU"]  # x 4 cycles / (1024 SIMDs x clock) = the VALU's own time
            e["valu_instructions_per_launch"] = s.get("SQ_INSTS_VALU")
        if l and l.get("SQ_INSTS_LDS") is not None:
            # LDS: instructions, issue quad-cycles, array cycles (all CUs summed), conflict cycles (MI355X_MICROARCH.md: SQ_LDS_IDX_ACTIVE = all
            # LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra ones)
            e["lds_instructions_per_launch"] = l["SQ_INSTS_LDS"]
            e["lds_active_quad_cycles_per_launch"] = l.get("SQ_ACTIVE_INST_LDS")
            e["lds_array_cycles_per_launch"] = l.get("SQ_LDS_IDX_ACTIVE")
            e["lds_bank_conflict_cycles_per_launch"] = l.get("SQ_LDS_BANK_CONFLICT")
            e["lds_issue_stall_quad_cycles_per_launch"] = l.get("SQ_WAIT_INST_LDS")
            e["waves_per_launch"] = l.get("SQ_WAVES")
        if t and t.get("TCC_REQ_sum"):
            e["l2_requests_per_launch"] = t["TCC_REQ_sum"]
            e["l2_hit_rate"] = t["TCC_HIT_sum"] / max(1.0, t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
        res["kernels"][name] = e
    base = lambda k: k.split("::")[-1]
    dom = next((k for k in res["kernels"] if base(k) in ("k_trace3<0>", "k_trace3c")), None) or next(k for k in res["kernels"] if base(k) == "k_trace<0>")
    res["kernel"] = dom
    res["hbm_bytes_per_launch"] = res["kernels"][dom]["hbm_bytes_per_launch"]
    res["alg_bytes_per_launch"] = bench["roofline"]["alg_bytes_per_launch"]
    res["traffic_over_algorithmic"] = res["hbm_bytes_per_launch"] / bench["roofline"]["alg_bytes_per_launch"]
    # the shading stages: algorithmic bytes per launch from the bench line's own roofline_shade (items x bytes per item over
    # the timed steps) against k_shade_path + k_env of the PMC pass
    rs = bench.get("roofline_shade")
    if rs and rs.get("items"):
        launches = sum(v["dispatches_in_pmc_pass"] for k, v in res["kernels"].items() if base(k).startswith("k_shade_path"))
        # the PMC pass: --steps 1 --warmup 0 = the first render + one step (since round 6: + the second render bench.py times on its own)
        steps, pmc_renders = bench["steps"], (3.0 if rnd >= "r06" else 2.0)
        per_render_launches = launches / pmc_renders
        alg = rs["items"] * rs["alg_bytes_per_item"] / steps / max(1.0, per_render_launches)
        phys = sum(v["hbm_bytes_per_launch"] * v["dispatches_in_pmc_pass"] for k, v in res["kernels"].items()
                   if base(k).startswith("k_shade_path") or base(k).startswith("k_env")) / max(1, launches)
        res["shade"] = {"alg_bytes_per_launch": alg, "hbm_bytes_per_launch_incl_k_env": phys, "traffic_over_algorithmic": phys / alg}
    res["method"] = ("tools/profile_%s.sh: rocprofv3 --pmc passes (counters only) on `DARTRAY_TRACE_IMPL=%s python3 bench.py %s--steps 1 --warmup 0 "
                     "--no-cpu-baseline --no-extra` (two full renders, every launch full size); launch times from the --kernel-trace --stats "
                     "run of the same script.  Counters sit on the L2's memory side, so Infinity-Cache hits are included: an upper bound on "
                     "HBM bytes." % (rnd, "3" if "k_trace3" in dom else "2", "" if cfg == "c2" else "--config %s " % cfg.upper()))
    tk = bench.get("config", {}).get("trace_kernels")
    if rnd >= "r05" and tk:
        res["trace_kernels"] = {"closest": tk["closest"], "any_hit": tk["any_hit"]}
        res["method"] = ("tools/profile_%s.sh: a dry run of the bench command reads what this config's pilot picks on the box (closest-hit %s, any-hit %s), "
                         "then rocprofv3 --pmc passes (counters only, one group per pass) on `DARTRAY_PILOT=0 DARTRAY_STATE_LAYOUT=<picked> [r06: DARTRAY_BATCH_BITS=28] python3 bench.py %s"
                         "--trace-kernels <picked> --steps 1 --warmup 0 --no-cpu-baseline --no-extra` (two -- r06: three -- full renders, every launch a full-size launch of "
                         "those kernels); launch times from the --kernel-trace --stats run of the same script (DARTRAY_OVERLAP_ANY=0).  Counters sit on the "
                         "L2's memory side, so Infinity-Cache hits are included: an upper bound on HBM bytes."
                         % (rnd, tk["closest"], tk["any_hit"], "" if cfg == "c2" else "--config %s " % cfg.upper()))
    res["sources"] = ["profiles/%s_pmc_%s_%s.txt" % (rnd, cfg, k) for k, p in (("rdreq", rdp), ("wrreq", wrp), ("sq", sqp), ("tcc", tccp), ("lds", ldsp)) if p] + \
                     ["profiles/%s_%s" % (rnd, os.path.basename(statsp).replace(rnd + "_", ""))]
    return res


if __name__ == "__main__":
    src, cfg, out = sys.argv[1], sys.argv[2], sys.argv[3]
    rnd = sys.argv[4] if len(sys.argv) > 4 else "r04"
    res = build(src, cfg, rnd)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:2500])
