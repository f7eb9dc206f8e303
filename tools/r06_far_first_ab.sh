#!/bin/bash
# Any-hit rays far child first (kernel ids 6 / 7) against the reference order, per config, kernels forced; then what each pilot picks.
#   tools/r06_far_first_ab.sh OUTDIR
out="${1:-gpurun_out/r06ff}"
mkdir -p "$out"
X="--no-cpu-baseline --no-extra"
run() {  # name, args
  local name="$1"; shift
  ( export DARTRAY_BENCH_DETAIL_DIR="$PWD/$out/$name.d" DARTRAY_VERBOSE=1; mkdir -p "$DARTRAY_BENCH_DETAIL_DIR"; timeout 600 python3 bench.py "$@" $X > "$out/$name.json" 2> "$out/$name.err" )
  python3 - "$out/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernel_ms_per_step"]
    print("%-14s %8.1f Msamples/s  step %8.1f ms  closest %7.1f any %7.1f shade %7.1f  kernels %s far_first %s" % (
        sys.argv[2], d["value"], d["ms_per_step"], k["closest_ms"], k["any_ms"], k["shade_ms"], d["config"]["kernel_ids"], d["config"]["any_hit_far_child_first"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  grep -h "traversal pilot" "$out/$name.err" | tail -1
}
run c2_ref --trace-kernels 2,2 --steps 4 --warmup 1
run c2_far --trace-kernels 2,6 --steps 4 --warmup 1
run c4_ref --config C4 --trace-kernels 5,3 --steps 4 --warmup 1
run c4_far --config C4 --trace-kernels 5,7 --steps 4 --warmup 1
run c5_ref --config C5 --trace-kernels 5,3 --steps 2 --warmup 1
run c5_far --config C5 --trace-kernels 5,7 --steps 2 --warmup 1
run c2_pilot --steps 3 --warmup 1
run c4_pilot --config C4 --steps 3 --warmup 1
run c5_pilot --config C5 --steps 2 --warmup 1
