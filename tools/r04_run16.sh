#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04n; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
for cfg in C2 C5; do st=3; [ $cfg = C5 ] && st=2
  BENCH_ARGS="--config $cfg" STEPS=$st tools/bench_variants.sh base 2>&1 | sed "s/^/$cfg prepass /" | cut -c1-230 >> $out/v.txt
  DARTRAY_GEN_PREPASS=0 BENCH_ARGS="--config $cfg" STEPS=$st tools/bench_variants.sh base 2>&1 | sed "s/^/$cfg no-prepass /" | cut -c1-230 >> $out/v.txt
done
BENCH_ARGS="--config C3 --res 1024" STEPS=2 tools/bench_variants.sh base 2>&1 | sed "s/^/C3-class(1024^2x1024spp) prepass /" | cut -c1-250 >> $out/v.txt
DARTRAY_GEN_PREPASS=0 BENCH_ARGS="--config C3 --res 1024" STEPS=2 tools/bench_variants.sh base 2>&1 | sed "s/^/C3-class no-prepass /" | cut -c1-250 >> $out/v.txt
cat $out/v.txt
