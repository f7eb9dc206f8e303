#!/bin/bash
# round 5: the committed evidence of the round -- profiles of what the pilots pick (stats + PMC), the batch-size / workspace trade on C2, the final line.
cd "$(dirname "$0")/.."
out=gpurun_out/r05d; mkdir -p $out
tools/profile_r05.sh $out/prof "c2 c4 c5" all > $out/profile.log 2>&1; tail -5 $out/profile.log; cat $out/prof/*_picked.txt
for bits in 28 27 26 25; do
  ( export DARTRAY_BATCH_BITS=$bits DARTRAY_VERBOSE=1; timeout 400 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $out/bits_$bits.json 2> $out/bits_$bits.err )
  python3 - $out/bits_$bits.json $bits <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kernel_ms_per_step"]; print("batch_bits", sys.argv[2], d["value"], "closest", k["closest_ms"], "any", k["any_ms"], "shade", k["shade_ms"], "first_render_ms", d["first_render_ms"])
except Exception as e:
    print("batch_bits", sys.argv[2], "FAILED", e)
PY
  grep -h "workspace for" $out/bits_$bits.err | tail -1
done 2>&1 | tee $out/batch_bits.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench_final.json 2> $out/bench_final.err; tail -c 600 $out/bench_final.json
