#!/bin/bash
# round 4, GPU run 8: the 4-byte any-hit pair kernel on the cache-resident configs (C2, C5)
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04h; mkdir -p $out
for cfg in C2 C5; do
  st=3; [ $cfg = C5 ] && st=2
  for v in "base:2,2" "base:2,3" "a7:2,3"; do
    name=${v%%:*}; tk=${v#*:}
    BENCH_ARGS="--config $cfg --trace-kernels $tk" STEPS=$st tools/bench_variants.sh $name 2>&1 | sed "s/^/$cfg $tk /" | cut -c1-260 >> $out/any3a.txt
  done
done
cat $out/any3a.txt
