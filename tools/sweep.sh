#!/bin/bash
# Build A/B variants of libdartray_hip.so with different -D flags and bench each on the GPU box.
# usage: tools/sweep.sh "name1:-DX=1 -DY=2" "name2:..."   (run via gpurun)
cd "$(dirname "$0")/../dartray_amd/csrc"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function $flags \
    -o ../libdartray_hip_$name.so dr_kernels.hip dr_trace.hip dr_api.hip dr_bvh_build.cpp dr_comm.cpp -ldl 2>/dev/null || { echo "$name: build failed"; continue; }
  DARTRAY_LIB=$PWD/../libdartray_hip_$name.so timeout 300 python ../../bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra ${BENCH_ARGS} > /tmp/sweep_$name.log 2>&1
  python - "$name" <<'PY'
import json,sys
name=sys.argv[1]
try:
    d=json.loads(open("/tmp/sweep_%s.log"%name).read().strip().splitlines()[-1])
    print(name, d["value"], d["kernel_ms_per_step"])
except Exception as e:
    print(name, "FAILED", e); print(open("/tmp/sweep_%s.log"%name).read()[-500:])
PY
done
