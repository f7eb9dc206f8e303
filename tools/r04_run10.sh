#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04i; mkdir -p $out
for cfg in C4 C2 C5; do
  DARTRAY_LIB=$PWD/dartray_amd/libdartray_hip_sprof.so DARTRAY_TRACE_IMPL=3 timeout 400 python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $out/sprof_$cfg.err
  echo "== $cfg"; grep stack_prof $out/sprof_$cfg.err | tail -2
done
