#!/bin/bash
# round 4, GPU run 6: device-side scene preparation
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04f; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_scene_prep.py -m gpu -x -q -s > $out/pytest_prep.log 2>&1; echo "pytest rc $?" >> $out/pytest_prep.log; tail -15 $out/pytest_prep.log
for c in C2 C4 C5; do timeout 300 python tools/setup_times.py $c >> $out/setup_times.txt 2>&1; done; cat $out/setup_times.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -5 $out/pytest.log
