#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r05k; mkdir -p $out
for rep in 1 2; do
for v in base padv1 padv2; do
  lib="$PWD/dartray_amd/libdartray_hip_$v.so"; [ $v = base ] && lib="$PWD/dartray_amd/libdartray_hip.so"
  ( export DARTRAY_LIB="$lib" DARTRAY_OVERLAP_ANY=0; timeout 400 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --trace-kernels 2,2 > $out/pad_${v}_$rep.json 2> $out/pad_${v}_$rep.err )
  python3 - $out/pad_${v}_$rep.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kernel_ms_per_step"]; print("pad", sys.argv[2], d["value"], "closest", k["closest_ms"], "any", k["any_ms"], "shade", k["shade_ms"])
except Exception as e:
    print("pad", sys.argv[2], "FAILED", e)
PY
done
done 2>&1 | tee $out/pads_vmem.txt
