#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04r; mkdir -p $out
for v in base l8 l16 r8 r24 w128; do
  BENCH_ARGS="--config C5 --trace-kernels 5,3" STEPS=2 tools/bench_variants.sh $v 2>&1 | sed "s/^/C5 /" | cut -c1-150 >> $out/v.txt
  BENCH_ARGS="--config C4 --trace-kernels 3,3" STEPS=3 tools/bench_variants.sh $v 2>&1 | sed "s/^/C4 /" | cut -c1-150 >> $out/v.txt
done
sort $out/v.txt
