#!/bin/bash
# C4 with the sibling-pair kernel for both ray kinds, several pair orders
for o in ${ORDERS:-dfs sib veb:1:2 veb:1:3 veb:1:4 veb:12:1 veb:12:3 veb:16:4}; do
  if [ $o = dfs ]; then unset DARTRAY_PAIR_ORDER; else export DARTRAY_PAIR_ORDER=$o; fi
  DARTRAY_TRACE_IMPL=3 timeout 300 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /tmp/o.json 2>/tmp/o.err
  python - "$o" <<'PY'
import json,sys
try:
    d=json.loads(open("/tmp/o.json").read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]
    print(sys.argv[1], d["value"], "closest", k["closest_ms"], "any", k["any_ms"], "shade", k["shade_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", open("/tmp/o.err").read()[-300:])
PY
done
