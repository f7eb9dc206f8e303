#!/bin/bash
# Like sweep.sh, but times `python -m dartray_amd.pbrt <scene>` (the general shading kernels) and bench.py --config C5.
# usage: tools/sweep_scene.sh scene.pbrt "name1:-DX=1" ...   (run via gpurun)
scene="$(realpath "$1")"; shift
cd "$(dirname "$0")/../dartray_amd/csrc"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function $flags \
    -o ../libdartray_hip_$name.so dr_kernels.hip dr_trace.hip dr_api.hip dr_bvh_build.cpp 2>/dev/null || { echo "$name: build failed"; continue; }
  export DARTRAY_LIB=$PWD/../libdartray_hip_$name.so
  (cd ../.. && timeout 300 python -m dartray_amd.pbrt "$scene" -o /tmp/out_$name.npy | tail -1 | sed "s/^/$name scene: /")
  (cd ../.. && timeout 600 python bench.py --config C5 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name C5:', d['value'], d['kernel_ms_per_step'])")
done
