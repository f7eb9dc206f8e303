#!/bin/bash
# Round 5: how the committed evidence under profiles/r05_* was produced (each section is what one gpurun call ran on the MI355X box;
# variants are built HERE first with tools/variants.sh, their .so files travel with the snapshot).  usage: tools/r05_evidence.sh SECTION
#   bound      regeneration bound: per-stage tables of C5, C5 in a closed box (both layouts), C2      -> r05_regen_bound_*.json
#   intercept  C2 workgroups-per-CU sweep, 1 M / 32 K triangles, PMC beside it                        -> r05_c2_intercept_sweep.txt, r05_intercept_pmc_*
#   nshard     -DDR_NSHARD=8 against the shipped library, time + PMC (variants.sh "nshard8:-DDR_NSHARD=8") -> r05_nshard_pmc_*
#   gather     tools/gather_rate.hip                                                                   -> r05_gather_rate.txt
#   pads       per-visit instruction pads (variants pad8/16/32, padlds4, padv1/2)                     -> r05_c2_instruction_pads.txt, r05_c2_lane_load_pads.txt
#   c7         k_trace3c at seven workgroups (variants c77, c68, c67)                                 -> r05_trace3c_seven_workgroups.txt
#   phases     -DDR_TRACE_PROF phase profile (variant tprof)                                          -> r05_c2_trace_phase_profile.txt
#   stages     per-stage tables after the thin-launch change                                          -> r05_stages_c{5,2}_thin_launches.json
#   final      profiles of what the pilots pick (tools/profile_r05.sh), batch bits, the driver's line -> r05_c{2,4,5}_*, r05_pmc_*, r05_c2_batch_bits.txt, r05_bench_final.json
#   alloc      tools/alloc_probe.py                                                                   -> r05_alloc_probe.txt
# Afterwards here: tools/make_traffic.py (traffic files), tools/fit_occupancy.py (occupancy model), copies into profiles/.
cd "$(dirname "$0")/.."
out=gpurun_out/r05_$1; mkdir -p $out
X="--no-cpu-baseline --no-extra"
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["kernel_ms_per_step"]
    print(sys.argv[2], d["value"], "closest", k["closest_ms"], "any", k["any_ms"], "shade", k["shade_ms"], "gen", k["gen_ms"], "total", k["total_ms"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
variants() {  # tag, bench args, variant names...: each with DARTRAY_OVERLAP_ANY=0, twice
  local tag="$1" args="$2"; shift 2
  for rep in 1 2; do for v in "$@"; do
    lib="$PWD/dartray_amd/libdartray_hip_$v.so"; [ $v = base ] && lib="$PWD/dartray_amd/libdartray_hip.so"
    ( export DARTRAY_LIB="$lib" DARTRAY_OVERLAP_ANY=0; timeout 400 python3 bench.py $args --steps 3 --warmup 1 $X > $out/${tag}_${v}_$rep.json 2> $out/${tag}_${v}_$rep.err )
    line $out/${tag}_${v}_$rep.json "$tag $v"
  done; done
}
case "$1" in
  bound)
    for lay in 4 64; do for dome in "" "--dome"; do
      timeout 600 python tools/r05_regen_bound.py --kernels 5,3 --layout $lay $dome > $out/bound_c5${dome#--}_l$lay.json 2> $out/bound_c5${dome#--}_l$lay.err
    done; done
    timeout 600 python tools/r05_regen_bound.py --config C2 --kernels 2,2 --layout 64 > $out/bound_c2.json 2> $out/bound_c2.err;;
  intercept) tools/r05_c2_intercept.sh $out;;
  nshard)
    variants nshard "--trace-kernels 2,2" base nshard8
    export TMPDIR=/tmp; root="$PWD"
    for v in base nshard8; do
      lib="$root/dartray_amd/libdartray_hip_$v.so"; [ $v = base ] && lib="$root/dartray_amd/libdartray_hip.so"
      for grp in rdreq tcc; do
        [ $grp = rdreq ] && ctrs="TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B" || ctrs="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
        d="$root/$out/pmc_${v}_$grp"; rm -rf "$d"; mkdir -p "$d"
        (cd /tmp && export DARTRAY_LIB="$lib" DARTRAY_OVERLAP_ANY=0 && timeout -s KILL 400 rocprofv3 --pmc $ctrs --output-format csv -d "$d" -o run -- python3 "$root/bench.py" --steps 1 --warmup 0 $X --trace-kernels 2,2 > "$d.log" 2>&1)
        python3 tools/pmc_summary.py "$d" > "$out/pmc_nshard_${v}_$grp.txt" 2>&1; rm -rf "$d" "$d.log"
      done
    done;;
  gather) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o /tmp/gather_rate 2> /dev/null && timeout 300 /tmp/gather_rate | tee $out/gather_rate.txt;;
  pads) variants pad "--trace-kernels 2,2" base pad8 pad16 pad32 padlds4 padv1 padv2 | tee $out/pads.txt;;
  c7) for cfg in C5 C4 C2; do a="--config $cfg"; [ $cfg = C2 ] && a=""; variants $cfg "$a --trace-kernels 5,3" base c77 c68 c67; done | tee $out/c7.txt;;
  phases)
    for scene in big small; do [ $scene = small ] && B="--blob 180,90" || B=""; for wg in 7 5 3; do
      ( export DARTRAY_LIB="$PWD/dartray_amd/libdartray_hip_tprof.so" DARTRAY_TRACE_WG_PER_CU=$wg DARTRAY_OVERLAP_ANY=0 DARTRAY_LAYOUT_PILOT=0; timeout 400 python3 bench.py $B --steps 1 --warmup 0 $X --trace-kernels 2,2 > $out/prof_${scene}_w$wg.json 2> $out/prof_${scene}_w$wg.err )
      echo "== $scene w=$wg"; grep "trace_prof" $out/prof_${scene}_w$wg.err | tail -34
    done; done | tee $out/trace_prof.txt;;
  stages)
    timeout 600 python tools/r05_regen_bound.py --kernels 5,3 --layout 4 > $out/stages_c5.json 2> $out/stages_c5.err
    timeout 600 python tools/r05_regen_bound.py --config C2 --kernels 2,2 --layout 64 > $out/stages_c2.json 2> $out/stages_c2.err;;
  final)
    tools/profile_r05.sh $out/prof "c2 c4 c5" all > $out/profile.log 2>&1; cat $out/prof/*_picked.txt
    for bits in 28 27 26 25; do
      ( export DARTRAY_BATCH_BITS=$bits DARTRAY_VERBOSE=1; timeout 400 python3 bench.py --steps 3 --warmup 1 $X > $out/bits_$bits.json 2> $out/bits_$bits.err )
      line $out/bits_$bits.json "batch_bits $bits"; grep -h "workspace for" $out/bits_$bits.err | tail -1
    done | tee $out/batch_bits.txt
    timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench_final.json 2> $out/bench_final.err; tail -c 400 $out/bench_final.json;;
  alloc) python3 tools/alloc_probe.py | tee $out/alloc_probe.txt;;
  *) sed -n 2,16p "$0";;
esac
