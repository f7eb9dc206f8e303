#!/bin/bash
# round 4, GPU run 7: replay tests + the bench line with the replay leg
cd "$(dirname "$0")/.."
ulimit -c 0
out=gpurun_out/r04g; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "replay or serial" > $out/pytest_replay.log 2>&1; echo "pytest rc $?" >> $out/pytest_replay.log; tail -6 $out/pytest_replay.log
timeout 600 python bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04g/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d.get('replay')); print(d['one_shot_ms'])
for e in d['extra_configs']: print(e['config']['workload'][:3], e['value'], e['one_shot_ms'])
PY
