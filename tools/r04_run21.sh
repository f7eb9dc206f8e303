#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04t; mkdir -p $out; rm -f $out/v.txt
for v in base a8; do
  BENCH_ARGS="--config C2 --trace-kernels 2,3" STEPS=3 tools/bench_variants.sh $v 2>&1 | sed "s/^/C2 2,3 /" | cut -c1-150 >> $out/v.txt
  BENCH_ARGS="--config C5 --trace-kernels 5,3" STEPS=2 tools/bench_variants.sh $v 2>&1 | sed "s/^/C5 5,3 /" | cut -c1-150 >> $out/v.txt
done
BENCH_ARGS="--config C2 --trace-kernels 2,2" STEPS=3 tools/bench_variants.sh base 2>&1 | sed "s/^/C2 2,2 /" | cut -c1-150 >> $out/v.txt
sort $out/v.txt
