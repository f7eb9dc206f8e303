#!/bin/bash
# round 4, GPU run 2: treelet-parked traversal -- correctness on the small scene, then C4 timings
cd "$(dirname "$0")/.."
out=gpurun_out/r04b; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "alternative_traversal" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -15 $out/pytest.log
run() {  # name, env...
  local name=$1; shift
  ( export "$@" DARTRAY_VERBOSE=1; timeout 400 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $out/$name.json 2> $out/$name.err )
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["kernel_ms_per_step"]["closest_ms"], d["kernel_ms_per_step"]["any_ms"], d["kernel_ms_per_step"]["total_ms"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
  grep "treelet-parked" $out/$name.err | tail -2
}
run base3 DARTRAY_TRACE_IMPL=3 DARTRAY_OVERLAP_ANY=0
run order12 DARTRAY_TRACE_IMPL=3 DARTRAY_OVERLAP_ANY=0 DARTRAY_PAIR_ORDER=top:12
for T in 8 12 16; do run tl_T${T}_r1 DARTRAY_TRACE_IMPL=4 DARTRAY_PAIR_ORDER=top:$T DARTRAY_TREELET_ROUNDS=1; done
run tl_T12_r2 DARTRAY_TRACE_IMPL=4 DARTRAY_PAIR_ORDER=top:12 DARTRAY_TREELET_ROUNDS=2
run tl_T12_r0 DARTRAY_TRACE_IMPL=4 DARTRAY_PAIR_ORDER=top:12 DARTRAY_TREELET_ROUNDS=0
