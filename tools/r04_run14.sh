#!/bin/bash
cd "$(dirname "$0")/.."
tools/profile_r04.sh gpurun_out/r04p "c5" all > gpurun_out/r04p_log.txt 2>&1
python tools/make_traffic.py gpurun_out/r04p c5 gpurun_out/r04p/c5_traffic.json r04 > /dev/null 2>> gpurun_out/r04p_log.txt
tail -3 gpurun_out/r04p_log.txt
