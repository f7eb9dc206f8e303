#!/bin/bash
# Rehearsal of the N-rank bench line on a ONE-GPU box (DARTRAY_COMM_REHEARSAL=1: ranks share the GPU, gloo sums the film through host
# memory -- NOT a measurement and NOT the product's collective; dartray_amd/dist.py).  The driver's own launch form.
#   gpurun -- 'bash tools/r05_rehearse_ranks.sh 2'   -> gpurun_out/r05rehearsal/ranks<N>.json
N=${1:-2}; O=gpurun_out/r05rehearsal; mkdir -p $O
export DARTRAY_COMM_REHEARSAL=1
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N --steps 1 --warmup 0 > $O/ranks$N.json 2> $O/ranks$N.err
echo "rc=$?"; tail -c 1500 $O/ranks$N.json; tail -5 $O/ranks$N.err
