#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r05i; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_options.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -15 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -3 $out/smoke.log
