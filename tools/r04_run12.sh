#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04k; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
DARTRAY_VERBOSE=1 timeout 900 python bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err; grep "pilot" $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04k/bench.json').read().strip().splitlines()[-1])
def p(e): print(e['config']['workload'][:3], e['value'], {k:v for k,v in e['kernel_ms_per_step'].items() if k!='note'}, e['config']['trace_kernels'], e['first_render_ms'], e['pilot_ms'])
p(d)
for e in d['extra_configs']: p(e)
print(d['roofline_shade']['resolve_only_entries'], d['roofline_shade']['items'], d['roofline_shade']['vertices'], d['roofline_shade']['cont_rays'], d['roofline_shade']['mis_rays'])
PY
