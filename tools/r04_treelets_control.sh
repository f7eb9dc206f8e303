#!/bin/bash
# round 4: treelet-parked traversal -- batched parking, T = 12 / 6, per-round timings.
# (The "unsorted" control legs were removed in round 5: DARTRAY_TREELET_SORT never shipped -- that build's 1-bit radix-sort path
# faulted after a few stages and the switch was taken out again -- so those legs ran the sorted path a second time.  The control
# figures quoted in MEASUREMENTS.md come from profiles/r04_tl_c4_timings.txt, printed by the build that still had the switch.)
cd "$(dirname "$0")/.."
out=gpurun_out/r04d; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "alternative_traversal" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -3 $out/pytest.log
run() {  # name, env...
  local name=$1; shift
  ( export "$@"; timeout 400 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $out/$name.json 2> $out/$name.err )
  python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["kernel_ms_per_step"]["closest_ms"], d["kernel_ms_per_step"]["any_ms"], d["kernel_ms_per_step"]["total_ms"], d["per_sample"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
  grep "treelet-parked" $out/$name.err | tail -2
}
B="DARTRAY_OVERLAP_ANY=0 DARTRAY_VERBOSE=1"
run tl_T12 $B DARTRAY_TRACE_IMPL=4 DARTRAY_PAIR_ORDER=top:12
run tl_T6 $B DARTRAY_TRACE_IMPL=4 DARTRAY_PAIR_ORDER=top:6
( export DARTRAY_OVERLAP_ANY=0 DARTRAY_VERBOSE=2 DARTRAY_TRACE_IMPL=4 DARTRAY_PAIR_ORDER=top:12; timeout 400 python bench.py --config C4 --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $out/rounds_T12.err )
grep "treelets" $out/rounds_T12.err | tail -26 | head -12
