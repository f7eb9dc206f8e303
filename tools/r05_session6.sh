#!/bin/bash
# round 5: the two-wave shuffle kernel with 32-pixel groups (twice the groups per CU) against 64-pixel groups, at 512 and 1024 spp
cd "$(dirname "$0")/.."
out=gpurun_out/r05g; mkdir -p $out
for g in 64 32; do
  ( export DARTRAY_GEN_GROUP=$g; timeout 600 python -m pytest tests/test_gpu_render.py -m gpu -q -x -k "sample_counts or unread_sample" > $out/pytest_g$g.log 2>&1; echo "group $g pytest rc $?"; tail -2 $out/pytest_g$g.log )
done
for cfg in "C5:--config C5" "C3x1024:--config C3 --res 1024" "C5x1024spp:--config C5 --res 1024 --spp 1024"; do
  name="${cfg%%:*}"; a="${cfg#*:}"
  for g in 64 32; do
    ( export DARTRAY_GEN_GROUP=$g; timeout 500 python3 bench.py $a --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $out/${name}_g$g.json 2> $out/${name}_g$g.err )
    python3 - $out/${name}_g$g.json $name $g <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kernel_ms_per_step"]; print(sys.argv[2], "group", sys.argv[3], d["value"], "gen", k["gen_ms"], "total", k["total_ms"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done 2>&1 | tee $out/gen_group.txt
