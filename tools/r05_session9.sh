#!/bin/bash
# round 5: where a wave iteration of k_trace goes (s_memtime stamps, -DDR_TRACE_PROF): the node-visit phase split into
# fetch issue -> data arrived / slab filter / push-pop, on C2 with the 1 M-triangle and the 32 K-triangle blob, at 3 / 5 / 7 workgroups per CU
cd "$(dirname "$0")/.."
out=gpurun_out/r05j; mkdir -p $out
for scene in big small; do
  [ $scene = small ] && B="--blob 180,90" || B=""
  for wg in 7 5 3; do
    ( export DARTRAY_LIB="$PWD/dartray_amd/libdartray_hip_tprof.so" DARTRAY_TRACE_WG_PER_CU=$wg DARTRAY_OVERLAP_ANY=0 DARTRAY_LAYOUT_PILOT=0; timeout 400 python3 bench.py $B --steps 1 --warmup 0 --no-cpu-baseline --no-extra --trace-kernels 2,2 > $out/prof_${scene}_w$wg.json 2> $out/prof_${scene}_w$wg.err )
    echo "== $scene w=$wg"; grep "trace_prof" $out/prof_${scene}_w$wg.err | tail -34
  done
done 2>&1 | tee $out/trace_prof.txt
