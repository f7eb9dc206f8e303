#!/bin/bash
# round 4, GPU run 5: suite, bench line with layout pilot / one_shot_ms / top:12 order, shade at 4 waves per SIMD
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04e; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -8 $out/pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.err
STEPS=3 tools/bench_variants.sh base sw4 > $out/variants_c2.txt 2>&1; cat $out/variants_c2.txt
