#!/bin/bash
O=gpurun_out/r05lazy; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_options.py -m gpu -x -q > $O/tests2.log 2>&1; tail -3 $O/tests2.log
for lz in 1 0; do
  DARTRAY_LAZY_GEN=$lz timeout 600 python bench.py --config C5 --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $O/C5b_lazy${lz}.json 2> $O/C5b_lazy${lz}.err
  python - <<PY
import json
l=[x for x in open("$O/C5b_lazy${lz}.json") if x.startswith("{")]
j=json.loads(l[-1]); print("C5 lazy=$lz", j["value"], j["ms_per_step"], j["kernel_ms_per_step"], j["roofline_gen"]["achieved"], j["roofline_gen"].get("blocks_generated_of_named"))
PY
done
