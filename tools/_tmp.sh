#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r05_pk3; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -5 $out/pytest.log
for cfg in C2 C4 C5; do for sh in 0 1; do
  a="--config $cfg"; [ $cfg = C2 ] && a=""
  k=$([ $cfg = C2 ] && echo 2,2 || ([ $cfg = C4 ] && echo 3,3 || echo 5,3))
  ( export DARTRAY_COHERENT_SHADOW=$sh DARTRAY_PILOT=0 DARTRAY_STAGE_COUNTS=1; timeout 500 python3 bench.py $a --steps 2 --warmup 1 --no-cpu-baseline --no-extra --trace-kernels $k > $out/${cfg}_s$sh.json 2> $out/${cfg}_s$sh.err )
  python3 - $out/${cfg}_s$sh.json "$cfg coherent-shadow $sh" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kernel_ms_per_step"]
    print(sys.argv[2], d["value"], "closest", k["closest_ms"], "any", k["any_ms"], "shade", k["shade_ms"], "total", k["total_ms"], d["per_sample"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  grep "stage_times batch .* stage 0" $out/${cfg}_s$sh.err | tail -1
done; done 2>&1 | tee $out/shadow.txt
