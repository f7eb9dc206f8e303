#!/bin/bash
# round 5: thin launches (dynamic work chunks in the traversal kernels, shorter list chunks in k_shade_path): the suite, the per-stage table, the line
cd "$(dirname "$0")/.."
out=gpurun_out/r05h; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout 600 python tools/r05_regen_bound.py --kernels 5,3 --layout 4 > $out/stages_c5.json 2> $out/stages_c5.err; echo "c5 rc $?"
timeout 600 python tools/r05_regen_bound.py --config C2 --kernels 2,2 --layout 64 > $out/stages_c2.json 2> $out/stages_c2.err; echo "c2 rc $?"
timeout 900 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python3 - $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["kernel_ms_per_step"]); print([(e["value"], e["kernel_ms_per_step"]) for e in d["extra_configs"]])
PY
