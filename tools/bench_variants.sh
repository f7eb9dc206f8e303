#!/bin/bash
# Time prebuilt variants (tools/variants.sh) on the GPU box: tools/bench_variants.sh name1 name2 ...   (BENCH_ARGS extra)
cd "$(dirname "$0")/.."
for name in "$@"; do
  lib="$PWD/dartray_amd/libdartray_hip_$name.so"
  [ "$name" = "base" ] && lib="$PWD/dartray_amd/libdartray_hip.so"
  DARTRAY_LIB="$lib" timeout 300 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-extra ${BENCH_ARGS} > /tmp/var_$name.log 2>&1
  python - "$name" <<'PY'
import json,sys
name=sys.argv[1]
try:
    d=json.loads(open("/tmp/var_%s.log"%name).read().strip().splitlines()[-1])
    print(name, d["value"], d["kernel_ms_per_step"])
except Exception as e:
    print(name, "FAILED", e); print(open("/tmp/var_%s.log"%name).read()[-500:])
PY
done
