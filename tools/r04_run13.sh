#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04l; mkdir -p $out
for i in 1 2; do
DARTRAY_VERBOSE=1 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench$i.json 2> $out/bench$i.err; grep "traversal pilot" $out/bench$i.err
python - $out/bench$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def p(e): print(e['config']['workload'][:3], e['value'], {k:v for k,v in e['kernel_ms_per_step'].items() if k!='note'}, e['config']['trace_kernels']['closest'], e['config']['trace_kernels']['any_hit'], e['first_render_ms'], e['pilot_ms'])
p(d)
for e in d['extra_configs']: p(e)
PY
done
