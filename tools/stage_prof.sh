#!/bin/bash
# per-stage kernel times + list lengths of one config: stage_prof.sh OUT "bench args"
out="$1"; shift
root="$PWD"; export TMPDIR=/tmp
mkdir -p "$root/$out"
d="$root/$out/trace"; rm -rf "$d"; mkdir -p "$d"
(cd /tmp && DARTRAY_TRACE_IMPL=${IMPL:-2} DARTRAY_OVERLAP_ANY=0 DARTRAY_STAGE_COUNTS=1 timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d "$d" -o run -- python3 "$root/bench.py" $@ --steps 1 --warmup 0 --no-cpu-baseline --no-extra > "$root/$out/bench.json" 2> "$root/$out/bench.err")
python3 "$root/tools/per_stage.py" "$d" > "$root/$out/per_stage.txt" 2>&1
grep stage_counts "$root/$out/bench.err" | tail -40 > "$root/$out/stage_counts.txt"
rm -rf "$d"
