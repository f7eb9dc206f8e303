#!/bin/bash
# round 5, first GPU session: the changed library through the GPU suite, then the measurements that decide the round's kernel work.
cd "$(dirname "$0")/.."
out=gpurun_out/r05a; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
# (1) the regeneration bound: C5 as it is, and inside a closed box (dense lists), sp4 and 64-slot layouts
for lay in 4 64; do
  timeout 600 python tools/r05_regen_bound.py --kernels 5,3 --layout $lay > $out/bound_c5_l$lay.json 2> $out/bound_c5_l$lay.err; echo "bound c5 l$lay rc $?"
  timeout 600 python tools/r05_regen_bound.py --kernels 5,3 --layout $lay --dome > $out/bound_c5dome_l$lay.json 2> $out/bound_c5dome_l$lay.err; echo "bound c5dome l$lay rc $?"
done
timeout 600 python tools/r05_regen_bound.py --config C2 --kernels 2,2 --layout 64 > $out/bound_c2.json 2> $out/bound_c2.err; echo "bound c2 rc $?"
# (2) C2's occupancy-independent 96 ms
tools/r05_c2_intercept.sh $out/intercept > $out/intercept.log 2>&1; tail -12 $out/intercept.log
# (3) DR_NSHARD=8 in today's regime, with counters
for v in base nshard8; do
  lib="$PWD/dartray_amd/libdartray_hip_$v.so"; [ $v = base ] && lib="$PWD/dartray_amd/libdartray_hip.so"
  ( export DARTRAY_LIB="$lib" DARTRAY_OVERLAP_ANY=0; timeout 400 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --trace-kernels 2,2 > $out/nshard_$v.json 2> $out/nshard_$v.err )
  python3 - $out/nshard_$v.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("nshard", sys.argv[2], d["value"], d["kernel_ms_per_step"])
except Exception as e:
    print("nshard", sys.argv[2], "FAILED", e)
PY
done
export TMPDIR=/tmp
root="$PWD"
for v in base nshard8; do
  lib="$root/dartray_amd/libdartray_hip_$v.so"; [ $v = base ] && lib="$root/dartray_amd/libdartray_hip.so"
  for grp in rdreq tcc; do
    [ $grp = rdreq ] && ctrs="TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B" || ctrs="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
    d="$root/$out/pmc_nshard_${v}_$grp"; rm -rf "$d"; mkdir -p "$d"
    (cd /tmp && export DARTRAY_LIB="$lib" DARTRAY_OVERLAP_ANY=0 && timeout -s KILL 400 rocprofv3 --pmc $ctrs --output-format csv -d "$d" -o run -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra --trace-kernels 2,2 > "$d.log" 2>&1)
    python3 tools/pmc_summary.py "$d" > "$out/pmc_nshard_${v}_$grp.txt" 2>&1; rm -rf "$d" "$d.log"
  done
done
ls $out
