#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04q; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "split_stage or state_layout" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -12 $out/pytest.log
for cfg in C5 C2 C4; do st=3; [ $cfg = C5 ] && st=2
  for sp in 0 1; do
    DARTRAY_SPLIT=$sp BENCH_ARGS="--config $cfg" STEPS=$st tools/bench_variants.sh base 2>&1 | sed "s/^/$cfg split=$sp /" | cut -c1-230 >> $out/v.txt
  done
done
cat $out/v.txt
