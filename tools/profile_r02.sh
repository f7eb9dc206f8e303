#!/bin/bash
# Round-2 evidence, one gpurun call: rocprofv3 --kernel-trace --stats of the bench command, then PMC passes (counters
# only) for C2 and C4.  Run from the repo root on the GPU box:  tools/profile_r02.sh gpurun_out/r02p
set -u
out="${1:-gpurun_out/r02p}"
root="$PWD"
export TMPDIR=/tmp
mkdir -p "$root/$out"
stats() {  # name, env assignment, bench args
  local name="$1" envs="$2" bargs="$3" d="$root/$out/$1"
  rm -rf "$d"; mkdir -p "$d"
  (cd /tmp && export $envs && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o run -- python3 "$root/bench.py" $bargs > "$d/bench.json" 2> "$d/bench.err")
  f=$(find "$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$root/$out/${name}_kernel_stats.csv"
  tail -1 "$d/bench.json" > "$root/$out/${name}_bench.json"
  rm -rf "$d"
}
pmc() {  # name, env, bench args, counters
  local name="$1" envs="$2" bargs="$3" ctrs="$4" d="$root/$out/$1"
  rm -rf "$d"; mkdir -p "$d"
  (cd /tmp && export $envs && timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d "$d" -o run -- python3 "$root/bench.py" $bargs > "$d.log" 2>&1)
  python3 "$root/tools/pmc_summary.py" "$d" > "$root/$out/pmc_$name.txt" 2>&1
  rm -rf "$d" "$d.log"
}
B="--steps 3 --warmup 1 --no-cpu-baseline --no-extra"
P="--steps 1 --warmup 0 --no-cpu-baseline --no-extra"
RD="TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU"
stats c2 "DARTRAY_TRACE_IMPL=2" "$B"
stats c4 "DARTRAY_TRACE_IMPL=3" "--config C4 $B"
stats c5 "DARTRAY_TRACE_IMPL=2" "--config C5 --steps 2 --warmup 1 --no-cpu-baseline --no-extra"
# the same commands with one kernel at a time (a stage's any-hit launch otherwise runs beside its closest-hit launch and
# its entry in the table is the span from submission to end): per-kernel durations for the traffic files
for cfg in c2 c4 c5; do
  for f in kernel_stats bench; do [ -f "$root/$out/${cfg}_$f.csv" ] && mv "$root/$out/${cfg}_$f.csv" "$root/$out/${cfg}_${f}_sbs.csv"; done
  mv "$root/$out/${cfg}_bench.json" "$root/$out/${cfg}_bench_sbs.json"
done
stats c2 "DARTRAY_TRACE_IMPL=2 DARTRAY_OVERLAP_ANY=0" "$B"
stats c4 "DARTRAY_TRACE_IMPL=3 DARTRAY_OVERLAP_ANY=0" "--config C4 $B"
stats c5 "DARTRAY_TRACE_IMPL=2 DARTRAY_OVERLAP_ANY=0" "--config C5 --steps 2 --warmup 1 --no-cpu-baseline --no-extra"
for cfg in c2 c4 c5; do
  mv "$root/$out/${cfg}_kernel_stats.csv" "$root/$out/${cfg}_kernel_stats_serial.csv"
  mv "$root/$out/${cfg}_bench.json" "$root/$out/${cfg}_bench_serial.json"
  mv "$root/$out/${cfg}_kernel_stats_sbs.csv" "$root/$out/${cfg}_kernel_stats.csv"
  mv "$root/$out/${cfg}_bench_sbs.json" "$root/$out/${cfg}_bench.json"
done
for cfg in c2 c4; do
  if [ $cfg = c2 ]; then e="DARTRAY_TRACE_IMPL=2"; a="$P"; else e="DARTRAY_TRACE_IMPL=3"; a="--config C4 $P"; fi
  pmc ${cfg}_rdreq "$e" "$a" "$RD"
  pmc ${cfg}_wrreq "$e" "$a" "WRITE_SIZE"
  pmc ${cfg}_sq "$e" "$a" "$SQ"
done
ls -la "$root/$out"
