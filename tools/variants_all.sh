#!/bin/bash
# A/B variants that need EVERY device source rebuilt with the same flags (layout switches such as -DDR_SUB=16).
# usage: tools/variants_all.sh "name1:-DX=1" "name2:-DY=2" ...   ->  dartray_amd/libdartray_hip_<name>.so
set -e
cd "$(dirname "$0")/../dartray_amd/csrc"
python ../../__graft_entry__.py > /dev/null
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( objs=""
    for s in dr_kernels.hip dr_trace.hip dr_api.hip; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -pthread $flags \
        -x hip -c "$s" -o "_obj/va_${name}_$s.o" 2>/dev/null &
      objs="$objs _obj/va_${name}_$s.o"
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "../libdartray_hip_$name.so" $objs _obj/dr_bvh_build.cpp.o _obj/dr_comm.cpp.o _obj/dr_bvh_device.hip.o _obj/dr_scene_prep.hip.o _obj/dr_kernels.hip.sp4.o _obj/dr_trace.hip.sp4.o -ldl
    rm -f $objs; echo "built $name" ) &
done
wait
