#!/bin/bash
# round 5, third GPU session: k_trace3c at seven workgroups per CU as the compiler builds it (72 VGPRs + 28 B of scratch, 7 stack rows),
# against the shipped six (76 VGPRs, 9 rows) and two controls (six workgroups with 8 / 7 rows): what do the seventh workgroup and the
# shallower LDS stack each cost or return?
cd "$(dirname "$0")/.."
out=gpurun_out/r05c; mkdir -p $out
for cfg in C5 C4 C2; do
for v in base c77 c68 c67; do
  lib="$PWD/dartray_amd/libdartray_hip_$v.so"; [ $v = base ] && lib="$PWD/dartray_amd/libdartray_hip.so"
  a="--config $cfg"; [ $cfg = C2 ] && a=""
  ( export DARTRAY_LIB="$lib" DARTRAY_OVERLAP_ANY=0; timeout 400 python3 bench.py $a --steps 2 --warmup 1 --no-cpu-baseline --no-extra --trace-kernels 5,3 > $out/${cfg}_$v.json 2> $out/${cfg}_$v.err )
  python3 - $out/${cfg}_$v.json $cfg $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kernel_ms_per_step"]; print(sys.argv[2], sys.argv[3], d["value"], "closest", k["closest_ms"], "any", k["any_ms"], "shade", k["shade_ms"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
done
done 2>&1 | tee $out/c7.txt
