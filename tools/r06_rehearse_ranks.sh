#!/bin/bash
# Rehearsal of the N-rank bench line on a ONE-GPU box (DARTRAY_COMM_REHEARSAL=1: ranks share the GPU, gloo sums the film through host
# memory -- NOT a measurement and NOT the product's collective; dartray_amd/dist.py).  The driver's own launch form.
#   gpurun -- 'bash tools/r06_rehearse_ranks.sh 2'   -> gpurun_out/r06rehearsal/ranks<N>.json (the line) + the sidecar
N=${1:-2}; O=gpurun_out/r06rehearsal; mkdir -p $O
export DARTRAY_COMM_REHEARSAL=1 DARTRAY_BENCH_DETAIL_DIR="$PWD/$O" DARTRAY_VERBOSE=1
timeout 2400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N --steps 1 --warmup 0 > $O/ranks$N.json 2> $O/ranks$N.err
echo "rc=$?"; tail -c 2500 $O/ranks$N.json; grep -h "traversal pilot\|state-layout pilot" $O/ranks$N.err | head; tail -3 $O/ranks$N.err
