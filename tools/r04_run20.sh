#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04s; mkdir -p $out; rm -f $out/v.txt
for v in base ra32 ra16; do
  BENCH_ARGS="--config C4 --trace-kernels 3,3" STEPS=3 tools/bench_variants.sh $v 2>&1 | sed "s/^/C4 /" | cut -c1-150 >> $out/v.txt
  BENCH_ARGS="--config C2 --trace-kernels 2,3" STEPS=3 tools/bench_variants.sh $v 2>&1 | sed "s/^/C2 /" | cut -c1-150 >> $out/v.txt
done
sort $out/v.txt
