#!/bin/bash
# round 4, GPU run 3: treelet-parked traversal with XCD shards -- timings, per-round breakdown, PMC (TCC hit rate, read requests)
cd "$(dirname "$0")/.."
out=gpurun_out/r04c; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "alternative_traversal" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -3 $out/pytest.log
run() {  # name, env...
  local name=$1; shift
  ( export "$@"; timeout 400 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $out/$name.json 2> $out/$name.err )
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["kernel_ms_per_step"]["closest_ms"], d["kernel_ms_per_step"]["any_ms"], d["kernel_ms_per_step"]["total_ms"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
  grep "treelet-parked" $out/$name.err | tail -2
}
B="DARTRAY_OVERLAP_ANY=0 DARTRAY_VERBOSE=1"
run order12 $B DARTRAY_TRACE_IMPL=3 DARTRAY_PAIR_ORDER=top:12
for T in 10 12 14; do run tl8_T${T} $B DARTRAY_TRACE_IMPL=4 DARTRAY_PAIR_ORDER=top:$T DARTRAY_TREELET_SHARDS=8; done
run tl1_T12 $B DARTRAY_TRACE_IMPL=4 DARTRAY_PAIR_ORDER=top:12 DARTRAY_TREELET_SHARDS=1
run tl8_T12_r2 $B DARTRAY_TRACE_IMPL=4 DARTRAY_PAIR_ORDER=top:12 DARTRAY_TREELET_ROUNDS=2
( export DARTRAY_OVERLAP_ANY=0 DARTRAY_VERBOSE=2 DARTRAY_TRACE_IMPL=4 DARTRAY_PAIR_ORDER=top:12; timeout 400 python bench.py --config C4 --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $out/rounds_T12.err )
grep "treelets" $out/rounds_T12.err | tail -30
RD="TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"
TCC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
( export DARTRAY_OVERLAP_ANY=0 DARTRAY_TRACE_IMPL=3 DARTRAY_PAIR_ORDER=top:12; PMC_TIMEOUT=300 tools/pmc_run.sh $out/pmc_v3 "--config C4 --steps 1 --warmup 0 --no-cpu-baseline --no-extra" rdreq "$RD" tcc "$TCC" wr "WRITE_SIZE" > /dev/null 2>&1 )
( export DARTRAY_OVERLAP_ANY=0 DARTRAY_TRACE_IMPL=4 DARTRAY_PAIR_ORDER=top:12; PMC_TIMEOUT=300 tools/pmc_run.sh $out/pmc_tl "--config C4 --steps 1 --warmup 0 --no-cpu-baseline --no-extra" rdreq "$RD" tcc "$TCC" wr "WRITE_SIZE" > /dev/null 2>&1 )
for d in pmc_v3 pmc_tl; do for f in rdreq tcc wr; do echo "== $d $f"; grep -A6 "k_trace" $out/$d/pmc_$f.txt | head -40; done; done
