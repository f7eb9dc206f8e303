#!/bin/bash
# round 4, GPU run 1: suite, bench line, any-hit variants on C4, set-up times
cd "$(dirname "$0")/.."
out=gpurun_out/r04a; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.err
BENCH_ARGS="--config C4" STEPS=3 tools/bench_variants.sh base a7 a8 > $out/variants_c4.txt 2>&1
DARTRAY_ANY8=1 BENCH_ARGS="--config C4" STEPS=3 tools/bench_variants.sh base >> $out/variants_c4.txt 2>&1
cat $out/variants_c4.txt
for c in C2 C4; do timeout 300 python tools/setup_times.py $c >> $out/setup_times.txt 2>&1; done; cat $out/setup_times.txt
