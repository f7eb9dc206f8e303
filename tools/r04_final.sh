#!/bin/bash
# round 4: the suite and the driver's bench command line
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04z; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -3 $out/smoke.log
DARTRAY_VERBOSE=1 timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_final.json 2> $out/bench_final.err; grep -E "pilot|workspace" $out/bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04z/bench_final.json').read().strip().splitlines()[-1])
def p(e): print(e['config']['workload'][:3], e['value'], e['ms_per_step'], {k:v for k,v in e['kernel_ms_per_step'].items() if k!='note'}, e['config']['trace_kernels']['closest'], e['config']['trace_kernels']['any_hit'], e['one_shot_ms'])
p(d)
for e in d['extra_configs']: p(e)
print(d['replay']['value'], d['cpu_baseline']['value'], d['cpu_baseline_threads']['value'])
print({k: d['roofline'].get(k) for k in ('bound','achieved','frac','physical_GBps','traffic_over_algorithmic')})
PY
