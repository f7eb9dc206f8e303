"""profiles/r0N_<cfg>_traffic.json from the PMC summaries of tools/profile_r0N.sh (usage: make_traffic.py SRC cfg OUT [round]): HBM-side bytes per launch of the
dominant traversal kernel and of k_shade_path, exactly as MI355X_MICROARCH.md prescribes (separate --pmc passes;
reads = TCC_EA0_RDREQ_128B * 128 + _64B * 64 + _32B * 32 -- every wide read of these kernels is a 128-B request, so
FETCH_SIZE would under-report 2x; writes = WRITE_SIZE KB * 1024)."""
import csv, json, re, sys

src, cfg, out = sys.argv[1], sys.argv[2], sys.argv[3]
rnd = sys.argv[4] if len(sys.argv) > 4 else "r02"


def pmc(path):
    d, k = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*) dispatches=(\d+)", line)
        if m:
            k = m.group(1)
            d[k] = {"dispatches": int(m.group(2))}
            continue
        m = re.match(r"^\s+(\S+)\s+(\S+)\s+per-dispatch\s+(\S+)", line)
        if m and k:
            d[k][m.group(1)] = float(m.group(3))
    return d


rd, wr, sq = pmc("%s/pmc_%s_rdreq.txt" % (src, cfg)), pmc("%s/pmc_%s_wrreq.txt" % (src, cfg)), pmc("%s/pmc_%s_sq.txt" % (src, cfg))
import os
_serial = "%s/%s_kernel_stats_serial.csv" % (src, cfg)  # one kernel at a time (profile_r02.sh): per-kernel durations
stats = {r["Name"]: r for r in csv.DictReader(open(_serial if os.path.exists(_serial) else "%s/%s_kernel_stats.csv" % (src, cfg)))}
bench = json.load(open("%s/%s_bench.json" % (src, cfg)))
res = {"workload": bench["config"]["workload"], "bench_value_under_rocprof": bench["value"], "kernels": {}}
for name in rd:
    base = name.split("::")[-1]  # (kernels of the four-slot state layout live in namespace sp4)
    if not (base.startswith("k_trace") or base.startswith("k_shade_path") or base.startswith("k_env")):
        continue
    r, w, s = rd[name], wr.get(name, {}), sq.get(name, {})
    reads = r.get("TCC_EA0_RDREQ_128B", 0) * 128 + r.get("TCC_EA0_RDREQ_64B", 0) * 64 + r.get("TCC_EA0_RDREQ_32B", 0) * 32
    writes = w.get("WRITE_SIZE", 0) * 1024
    full = [v for k, v in stats.items() if name in k]
    avg_ms = float(full[0]["AverageNs"]) * 1e-6 if full else None
    e = {"hbm_bytes_per_launch": reads + writes, "reads_bytes_per_launch": reads, "writes_bytes_per_launch": writes,
         "avg_launch_ms_kernel_trace": avg_ms,
         "hbm_side_GBps": (reads + writes) / (avg_ms * 1e-3) / 1e9 if avg_ms else None}
    if s:
        # SQ_ACTIVE_INST_VALU counts quad-cycles per wave instruction (4 cycles each, the cost of a wave issuing ALONE);
        # with other waves interleaved a wave64 instruction occupies the SIMD for 2 cycles (MI355X_MICROARCH.md), so the
        # pipe's real occupancy is about half of this figure.  The A/B runs agree: 9 % fewer VALU instructions in
        # k_trace changed its time by < 1 %.
        cycles = s["SQ_ACTIVE_INST_VALU"] * 4 / 1024.0  # quad-cycles -> cycles per SIMD
        e["valu_issue_share_at_4_cycles_per_instruction"] = cycles / (avg_ms * 1e-3 * 2.4e9) if avg_ms else None
        e["valu_lane_utilisation"] = s["SQ_THREAD_CYCLES_VALU"] / (s["SQ_INSTS_VALU"] * 64)
        e["wait_any_share_of_wave_cycles"] = s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"]
    res["kernels"][name] = e
dom = next((k for k in res["kernels"] if k.split("::")[-1] == "k_trace3<0>"), None) or next(k for k in res["kernels"] if k.split("::")[-1] == "k_trace<0>")
res["kernel"] = dom
res["hbm_bytes_per_launch"] = res["kernels"][dom]["hbm_bytes_per_launch"]
res["alg_bytes_per_launch"] = bench["roofline"]["alg_bytes_per_launch"]
res["traffic_over_algorithmic"] = res["hbm_bytes_per_launch"] / bench["roofline"]["alg_bytes_per_launch"]
res["method"] = ("tools/profile_%s.sh: rocprofv3 --pmc passes (counters only) on `DARTRAY_TRACE_IMPL=%s python3 bench.py %s--steps 1 --warmup 0 "
                 "--no-cpu-baseline --no-extra` (two full renders, every launch full size); launch times from the --kernel-trace --stats "
                 "run of the same script.  Counters sit on the L2's memory side, so Infinity-Cache hits are included: an upper bound on "
                 "HBM bytes." % (rnd, "3" if "k_trace3" in dom else "2", "" if cfg == "c2" else "--config %s " % cfg.upper()))
res["sources"] = ["profiles/%s_pmc_%s_rdreq.txt" % (rnd, cfg), "profiles/%s_pmc_%s_wrreq.txt" % (rnd, cfg), "profiles/%s_pmc_%s_sq.txt" % (rnd, cfg),
                  "profiles/%s_%s_kernel_stats_serial.csv" % (rnd, cfg)]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:1800])
