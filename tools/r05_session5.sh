#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r05e; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -5 $out/pytest.log
timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench_final.json 2> $out/bench_final.err; python3 - $out/bench_final.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["kernel_ms_per_step"]); print(d["replay"]); print([(e["value"]) for e in d["extra_configs"]])
PY
