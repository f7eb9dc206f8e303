#!/bin/bash
cd "$(dirname "$0")/.."
tools/profile_r04.sh gpurun_out/r04p "c2 c4 c5" all > gpurun_out/r04p_log.txt 2>&1
for c in c2 c4 c5; do python tools/make_traffic.py gpurun_out/r04p $c gpurun_out/r04p/${c}_traffic.json r04 > /dev/null 2>> gpurun_out/r04p_log.txt; done
ls gpurun_out/r04p | head -60
