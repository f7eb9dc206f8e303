#!/bin/bash
# Round 5, review item 3: what are the occupancy-independent 96 ms of C2's closest-hit traversal (t = 96 + 839 / w ms, w = workgroups per CU)?
# (a) the workgroups-per-CU sweep of k_trace<0> / k_trace<1> on the C2 scene with the 1M-triangle blob AND with a 32 400-triangle blob whose
#     whole tree (2 MB of nodes + 1.5 MB of triangles) fits one XCD's 4 MB L2, the memory-side read requests and the L2 hit rate beside
#     three of the five points: does the intercept follow the bytes beyond L2?
# usage (repo root, GPU box): tools/r05_c2_intercept.sh OUTDIR
set -u
ulimit -c 0
out="${1:-gpurun_out/r05i}"
root="$PWD"
export TMPDIR=/tmp
mkdir -p "$root/$out"
X="--no-cpu-baseline --no-extra --trace-kernels 2,2"
RD="TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"
TCC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
pmc() {  # name, wg, blob-args, counters
  local name="$1" wg="$2" bargs="$3" ctrs="$4" d="$root/$out/$1"
  rm -rf "$d"; mkdir -p "$d"
  (cd /tmp && export DARTRAY_TRACE_WG_PER_CU=$wg DARTRAY_LAYOUT_PILOT=0 DARTRAY_OVERLAP_ANY=0 && timeout -s KILL 400 rocprofv3 --pmc $ctrs --output-format csv -d "$d" -o run -- python3 "$root/bench.py" $bargs --steps 1 --warmup 0 $X > "$d.log" 2>&1)
  python3 "$root/tools/pmc_summary.py" "$d" > "$root/$out/pmc_$name.txt" 2>&1
  [ -s "$root/$out/pmc_$name.txt" ] || tail -c 2000 "$d.log" > "$root/$out/pmc_$name.err"
  rm -rf "$d" "$d.log"
}
for scene in big small; do
  [ $scene = small ] && B="--blob 180,90" || B=""
  for wg in 3 4 5 6 7; do
    ( export DARTRAY_TRACE_WG_PER_CU=$wg DARTRAY_LAYOUT_PILOT=0 DARTRAY_OVERLAP_ANY=0; timeout 400 python3 bench.py $B --steps 3 --warmup 1 $X > "$out/time_${scene}_w$wg.json" 2> "$out/time_${scene}_w$wg.err" )
    python3 - "$out/time_${scene}_w$wg.json" $scene $wg <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernel_ms_per_step"]
    print("time", sys.argv[2], "w", sys.argv[3], "closest_ms", k["closest_ms"], "any_ms", k["any_ms"], "shade_ms", k["shade_ms"], "value", d["value"], "nodes/sample", d["per_sample"]["nodes"], "alg_bytes_per_launch", d["roofline"]["alg_bytes_per_launch"])
except Exception as e:
    print("time", sys.argv[2], "w", sys.argv[3], "FAILED", e)
PY
  done
  for wg in 3 5 7; do
    pmc ${scene}_w${wg}_rdreq $wg "$B" "$RD"
    pmc ${scene}_w${wg}_tcc $wg "$B" "$TCC"
  done
done 2>&1 | tee "$out/sweep.txt"
ls "$out"
