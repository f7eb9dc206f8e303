#!/bin/bash
# round 5, second GPU session: (1) the divergent-load rate of the vector memory path, (2) what one more instruction per node visit
# costs k_trace at today's occupancy (VALU / LDS pads; variants built by tools/variants.sh), (3) the GPU suite.
cd "$(dirname "$0")/.."
out=gpurun_out/r05b; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o /tmp/gather_rate 2> /dev/null && timeout 300 /tmp/gather_rate > $out/gather_rate.txt 2>&1; cat $out/gather_rate.txt
for rep in 1 2; do
for v in base pad8 pad16 pad32 padlds4; do
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
done 2>&1 | tee $out/pads.txt
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
