#!/bin/bash
# Per-kernel register / LDS / scratch usage of the library's kernels (hipcc -Rpass-analysis=kernel-resource-usage), both state layouts.
#   tools/kernel_resources.sh [OUT]     (CPU only: hipcc cross-compiles gfx950)
set -u
root="$(cd "$(dirname "$0")/.." && pwd)"
out="${1:-/dev/stdout}"
cd "$root/dartray_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -x hip -c --cuda-device-only -Rpass-analysis=kernel-resource-usage -o /dev/null"
{
for src in dr_trace.hip dr_kernels.hip; do
  for lay in "" "-DDR_SUB=4 -DDR_NS=sp4 -DDR_STATE_WORDS_K=48 -DDR_GROUPED=1"; do
    /opt/rocm/bin/hipcc $F $lay $src 2>&1 | python3 -c '
import re, sys
name = None
row = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); row = {}
        continue
    for key, tag in (("VGPRs:", "vgpr"), ("AGPRs", "agpr"), ("SGPRs:", "sgpr"), ("ScratchSize", "scratch"), ("Occupancy", "occ"), ("LDS Size", "lds")):
        m = re.search(re.escape(key) + r".*?(\d+)", line)
        if m and name:
            row[tag] = int(m.group(1))
            if tag == "lds":
                print("%-90s vgpr %3d sgpr %3d scratch %4d occ %2d lds %6d" % (name, row.get("vgpr", -1), row.get("sgpr", -1), row.get("scratch", -1), row.get("occ", -1), row["lds"]))
                name = None
' | while read -r l; do n=$(echo "$l" | awk '{print $1}' | c++filt | sed 's/(.*//'); echo "$n | $(echo "$l" | cut -d' ' -f2- | sed 's/^ *//')"; done
  done
done
} | sort > "$out"
