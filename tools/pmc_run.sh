#!/bin/bash
# Collect rocprofv3 PMC passes (one counter group per pass, counters only -- never combined with a trace domain that
# gpurun refuses) for a bench.py command line and print per-kernel sums.
# usage (on the GPU box, from the repo root):  tools/pmc_run.sh OUTDIR "bench args" group1 "C1 C2 C3" group2 "C4 ..." ...
set -u
out="$1"; shift
bargs="$1"; shift
export TMPDIR=/tmp
mkdir -p "$out"
root="$PWD"
while [ $# -ge 2 ]; do
  name="$1"; ctrs="$2"; shift 2
  d="$root/$out/$name"
  rm -rf "$d"; mkdir -p "$d"
  (cd /tmp && timeout -s KILL ${PMC_TIMEOUT:-240} rocprofv3 --pmc $ctrs --output-format csv -d "$d" -o run -- python3 "$root/bench.py" $bargs > "$d.log" 2>&1)
  python3 "$root/tools/pmc_summary.py" "$d" > "$root/$out/pmc_$name.txt" 2>&1
  rm -rf "$d"
  head -c 2500 "$root/$out/pmc_$name.txt"
done
