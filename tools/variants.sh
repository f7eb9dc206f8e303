#!/bin/bash
# A/B variants of ONE source file, built HERE (hipcc cross-compiles) and linked against the other objects of the
# regular build; the .so files travel with the gpurun snapshot and tools/bench_variants.sh times them on the GPU.
# usage: tools/variants.sh dr_trace.hip "name1:-DX=1" "name2:-DY=2 -DZ=3" ...
set -e
cd "$(dirname "$0")/../dartray_amd/csrc"
src="$1"; shift
python ../../__graft_entry__.py > /dev/null
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -pthread $flags \
      -x hip -c "$src" -o "_obj/var_$name.o" 2>/dev/null
    objs=""
    for s in dr_kernels.hip dr_trace.hip dr_api.hip dr_bvh_build.cpp dr_comm.cpp dr_bvh_device.hip dr_scene_prep.hip; do
      if [ "$s" = "$src" ]; then objs="$objs _obj/var_$name.o"; else objs="$objs _obj/$s.o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "../libdartray_hip_$name.so" $objs _obj/dr_kernels.hip.sp4.o _obj/dr_trace.hip.sp4.o -ldl
    rm -f "_obj/var_$name.o"; echo "built $name" ) &
done
wait
