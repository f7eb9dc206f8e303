#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04u; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_scene_prep.py tests/test_gpu_env.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -12 $out/pytest.log
