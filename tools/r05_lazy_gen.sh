#!/bin/bash
# Round 5, MEASUREMENTS.md 5.10: lazy sample generation on / off (DARTRAY_LAZY_GEN), two runs each of C5 / C2 / C4.
#   gpurun -- 'bash tools/r05_lazy_gen.sh'   -> gpurun_out/r05lazy/*.json (summarised into profiles/r05_lazy_gen.txt)
O=gpurun_out/r05lazy; mkdir -p $O
for c in C5 C2 C4; do
  for lz in 1 0; do
    for rep in 1 2; do
      DARTRAY_LAZY_GEN=$lz timeout 600 python bench.py --config $c --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $O/${c}_lazy${lz}_$rep.json 2> $O/${c}_lazy${lz}_$rep.err
    done
  done
done
