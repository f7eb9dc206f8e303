#!/bin/bash
# Round-6 evidence: per config, rocprofv3 --kernel-trace --stats of the bench command (kernels side by side and one at a time) and
# PMC passes (counters only, one group per pass) -- of the kernels the config's pilot ACTUALLY picks on this box: a dry run of the
# bench command (no profiler) reads the sidecar's config.trace_kernels / config.state_layout, and the profiled runs pin exactly those
# (--trace-kernels, DARTRAY_STATE_LAYOUT), so that every launch of a pass is a full-size launch of a kernel the driver's line runs.
# New in round 6: the library's buildinfo (sha256 of every source the .so was built from) is copied next to the passes, and
# tools/make_traffic.py stamps the traffic file with the kernel sources' hashes -- bench.py prints profile-derived figures only for
# the library they were taken with; an "lds" pass (LDS instructions / array cycles / conflicts) for the sampler's ceiling.
#   tools/profile_r06.sh OUTDIR "c2 c4 c5" [stats|pmc|all]
set -u
ulimit -c 0
out="${1:-gpurun_out/r06p}"
cfgs="${2:-c2 c4 c5}"
what="${3:-all}"
root="$PWD"
export TMPDIR=/tmp
mkdir -p "$root/$out"
cp "$root/dartray_amd/libdartray_hip.buildinfo.json" "$root/$out/buildinfo.json"
stats() {  # name, env assignment, bench args
  local name="$1" envs="$2" bargs="$3" d="$root/$out/$1"
  rm -rf "$d"; mkdir -p "$d"
  (cd /tmp && export $envs DARTRAY_BENCH_DETAIL_DIR="$d" && timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o run -- python3 "$root/bench.py" $bargs > "$d/bench.json" 2> "$d/bench.err")
  f=$(find "$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$root/$out/${name}_kernel_stats.csv"
  f=$(ls "$d"/bench_detail_*.json 2>/dev/null | head -1)   # the FULL result (the printed line carries numbers only)
  [ -n "$f" ] && cp "$f" "$root/$out/${name}_bench.json"
  tail -1 "$d/bench.json" > "$root/$out/${name}_line.json"
  [ -s "$root/$out/${name}_bench.json" ] || tail -c 2000 "$d/bench.err" > "$root/$out/${name}_bench.err"
  rm -rf "$d"
}
pmc() {  # name, env, bench args, counters
  local name="$1" envs="$2" bargs="$3" ctrs="$4" d="$root/$out/$1"
  rm -rf "$d"; mkdir -p "$d"
  (cd /tmp && export $envs DARTRAY_BENCH_DETAIL_DIR="$d" && timeout -s KILL 500 rocprofv3 --pmc $ctrs --output-format csv -d "$d" -o run -- python3 "$root/bench.py" $bargs > "$d.log" 2>&1)
  python3 "$root/tools/pmc_summary.py" "$d" > "$root/$out/pmc_$name.txt" 2>&1
  [ -s "$root/$out/pmc_$name.txt" ] || tail -c 2000 "$d.log" > "$root/$out/pmc_$name.err"
  rm -rf "$d" "$d.log"
}
RD="TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU"
TCC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
LDS="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"
X="--no-cpu-baseline --no-extra"
for cfg in $cfgs; do
  case $cfg in
    c2) a=""; s="--steps 3 --warmup 1";;
    c4) a="--config C4"; s="--steps 3 --warmup 1";;
    c5) a="--config C5"; s="--steps 2 --warmup 1";;
  esac
  # dry run: what does this config's pilot pick here?
  mkdir -p "$out/${cfg}_dry"
  ( export DARTRAY_VERBOSE=1 DARTRAY_BENCH_DETAIL_DIR="$root/$out/${cfg}_dry"; timeout 500 python3 bench.py $a --steps 1 --warmup 0 $X > "$out/${cfg}_dry.json" 2> "$out/${cfg}_dry.err" )
  pick=$(python3 - "$out/${cfg}_dry.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c, a = d["config"]["kernel_ids"]
print("%d,%d %d" % (c, a, d["config"]["state_layout"]))
PY
)
  rm -rf "$out/${cfg}_dry"
  [ -n "$pick" ] || { echo "$cfg: dry run failed"; tail -5 "$out/${cfg}_dry.err"; continue; }
  kern="${pick%% *}"; lay="${pick##* }"
  echo "$cfg: pilot picked kernels $kern, state layout $lay" | tee "$out/${cfg}_picked.txt"
  grep -h "traversal pilot\|state-layout pilot" "$out/${cfg}_dry.err" >> "$out/${cfg}_picked.txt"
  # (DARTRAY_BATCH_BITS=28: the scene's first render too goes in full-size batches, so that every launch of a pass is one)
  e="DARTRAY_PILOT=0 DARTRAY_STATE_LAYOUT=$lay DARTRAY_BATCH_BITS=28"
  a="$a --trace-kernels $kern"
  if [ "$what" != pmc ]; then
    stats ${cfg} "$e" "$a $s $X"
    for k in kernel_stats.csv bench.json line.json; do mv "$root/$out/${cfg}_$k" "$root/$out/${cfg}_sbs_$k" 2>/dev/null; done
    stats ${cfg} "$e DARTRAY_OVERLAP_ANY=0" "$a $s $X"
    mv "$root/$out/${cfg}_kernel_stats.csv" "$root/$out/${cfg}_kernel_stats_serial.csv"; mv "$root/$out/${cfg}_bench.json" "$root/$out/${cfg}_bench_serial.json"; mv "$root/$out/${cfg}_line.json" "$root/$out/${cfg}_line_serial.json"
    mv "$root/$out/${cfg}_sbs_kernel_stats.csv" "$root/$out/${cfg}_kernel_stats.csv"; mv "$root/$out/${cfg}_sbs_bench.json" "$root/$out/${cfg}_bench.json"; mv "$root/$out/${cfg}_sbs_line.json" "$root/$out/${cfg}_line.json"
  fi
  if [ "$what" != stats ]; then
    P="$a --steps 1 --warmup 0 $X"
    pmc ${cfg}_rdreq "$e" "$P" "$RD"
    pmc ${cfg}_wrreq "$e" "$P" "WRITE_SIZE"
    pmc ${cfg}_sq "$e" "$P" "$SQ"
    pmc ${cfg}_tcc "$e" "$P" "$TCC"
    pmc ${cfg}_lds "$e" "$P" "$LDS"
  fi
done
ls -la "$root/$out"
