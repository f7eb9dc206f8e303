#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04j; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_intersect.py -m gpu -x -q -k "alternative_traversal or intersect" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
DARTRAY_CLOSEST_COLD=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c4" > $out/pytest_c4.log 2>&1; echo "pytest rc $?" >> $out/pytest_c4.log; tail -3 $out/pytest_c4.log
BENCH_ARGS="--config C4" STEPS=3 tools/bench_variants.sh base > $out/v.txt 2>&1
DARTRAY_CLOSEST_COLD=1 BENCH_ARGS="--config C4" STEPS=3 tools/bench_variants.sh base c8 >> $out/v.txt 2>&1
for cfg in C2 C5; do st=3; [ $cfg = C5 ] && st=2
  BENCH_ARGS="--config $cfg --trace-kernels 3,2" STEPS=$st tools/bench_variants.sh base 2>&1 | sed "s/^/$cfg v3,2 /" >> $out/v.txt
  DARTRAY_CLOSEST_COLD=1 BENCH_ARGS="--config $cfg --trace-kernels 3,2" STEPS=$st tools/bench_variants.sh base 2>&1 | sed "s/^/$cfg v3c,2 /" >> $out/v.txt
done
cut -c1-200 $out/v.txt
