#!/bin/bash
# the whole GPU suite with the new kernels forced everywhere (the pilot only picks them on big scenes)
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04m; mkdir -p $out
DARTRAY_TRACE_IMPL=5 timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_impl5.log 2>&1; echo "rc $?" >> $out/pytest_impl5.log; tail -6 $out/pytest_impl5.log
DARTRAY_TRACE_IMPL=5 DARTRAY_STATE_LAYOUT=4 timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_impl5_sp4.log 2>&1; echo "rc $?" >> $out/pytest_impl5_sp4.log; tail -6 $out/pytest_impl5_sp4.log
