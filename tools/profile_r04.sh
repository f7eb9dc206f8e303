#!/bin/bash
# Round-4 evidence: per config, rocprofv3 --kernel-trace --stats of the bench command (kernels side by side and one at
# a time) and PMC passes (counters only, one group per pass).  Run from the repo root on the GPU box:
#   tools/profile_r04.sh OUTDIR "c2 c4 c5" [stats|pmc|all]
set -u
ulimit -c 0
out="${1:-gpurun_out/r04p}"
cfgs="${2:-c2 c4 c5}"
what="${3:-all}"
root="$PWD"
export TMPDIR=/tmp
mkdir -p "$root/$out"
stats() {  # name, env assignment, bench args
  local name="$1" envs="$2" bargs="$3" d="$root/$out/$1"
  rm -rf "$d"; mkdir -p "$d"
  (cd /tmp && export $envs && timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o run -- python3 "$root/bench.py" $bargs > "$d/bench.json" 2> "$d/bench.err")
  f=$(find "$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$root/$out/${name}_kernel_stats.csv"
  tail -1 "$d/bench.json" > "$root/$out/${name}_bench.json"
  [ -s "$root/$out/${name}_bench.json" ] || tail -c 2000 "$d/bench.err" > "$root/$out/${name}_bench.err"
  rm -rf "$d"
}
pmc() {  # name, env, bench args, counters
  local name="$1" envs="$2" bargs="$3" ctrs="$4" d="$root/$out/$1"
  rm -rf "$d"; mkdir -p "$d"
  (cd /tmp && export $envs && timeout -s KILL 500 rocprofv3 --pmc $ctrs --output-format csv -d "$d" -o run -- python3 "$root/bench.py" $bargs > "$d.log" 2>&1)
  python3 "$root/tools/pmc_summary.py" "$d" > "$root/$out/pmc_$name.txt" 2>&1
  [ -s "$root/$out/pmc_$name.txt" ] || tail -c 2000 "$d.log" > "$root/$out/pmc_$name.err"
  rm -rf "$d" "$d.log"
}
RD="TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU"
TCC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
for cfg in $cfgs; do
  case $cfg in
    # (the kernels are pinned to what the pilots pick -- DARTRAY_TRACE_IMPL, and DARTRAY_LAYOUT_PILOT=0 = the state layout by rule: sp4 for C5,
    # 64-slot runs for C2 / C4, what the density pilot chooses too -- so that every launch of a pass is a full-size one)
    c2) e="DARTRAY_TRACE_IMPL=2 DARTRAY_LAYOUT_PILOT=0"; a=""; s="--steps 3 --warmup 1";;
    c4) e="DARTRAY_TRACE_IMPL=3 DARTRAY_LAYOUT_PILOT=0"; a="--config C4"; s="--steps 3 --warmup 1";;
    c5) e="DARTRAY_PILOT=0 DARTRAY_LAYOUT_PILOT=0"; a="--config C5 --trace-kernels 5,3"; s="--steps 2 --warmup 1";;  # (what C5's pilot picks since k_trace3c)
  esac
  X="--no-cpu-baseline --no-extra"
  if [ "$what" != pmc ]; then
    stats ${cfg} "$e" "$a $s $X"
    mv "$root/$out/${cfg}_kernel_stats.csv" "$root/$out/${cfg}_kernel_stats_sbs.csv"; mv "$root/$out/${cfg}_bench.json" "$root/$out/${cfg}_bench_sbs.json"
    stats ${cfg} "$e DARTRAY_OVERLAP_ANY=0" "$a $s $X"
    mv "$root/$out/${cfg}_kernel_stats.csv" "$root/$out/${cfg}_kernel_stats_serial.csv"; mv "$root/$out/${cfg}_bench.json" "$root/$out/${cfg}_bench_serial.json"
    mv "$root/$out/${cfg}_kernel_stats_sbs.csv" "$root/$out/${cfg}_kernel_stats.csv"; mv "$root/$out/${cfg}_bench_sbs.json" "$root/$out/${cfg}_bench.json"
  fi
  if [ "$what" != stats ]; then
    P="$a --steps 1 --warmup 0 $X"
    pmc ${cfg}_rdreq "$e" "$P" "$RD"
    pmc ${cfg}_wrreq "$e" "$P" "WRITE_SIZE"
    pmc ${cfg}_sq "$e" "$P" "$SQ"
    pmc ${cfg}_tcc "$e" "$P" "$TCC"
  fi
done
ls -la "$root/$out"
