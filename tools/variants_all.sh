#!/bin/bash
# A/B variants that need EVERY device source rebuilt with the same flags (layout switches such as -DDR_SUB=16, experiments in headers):
# both layouts' objects (the 64-slot one and the line-grouped sp4 one) are compiled with the flags.
# usage: tools/variants_all.sh "name1:-DX=1" "name2:-DY=2" ...   ->  dartray_amd/libdartray_hip_<name>.so
set -e
cd "$(dirname "$0")/../dartray_amd/csrc"
python ../../__graft_entry__.py > /dev/null
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -pthread"
SP4="-DDR_SUB=4 -DDR_NS=sp4 -DDR_STATE_WORDS_K=48 -DDR_GROUPED=1"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( objs=""
    for s in dr_kernels.hip dr_trace.hip dr_api.hip; do
      $CC $flags -x hip -c "$s" -o "_obj/va_${name}_$s.o" 2>/dev/null &
      objs="$objs _obj/va_${name}_$s.o"
    done
    for s in dr_kernels.hip dr_trace.hip; do
      $CC $flags $SP4 -x hip -c "$s" -o "_obj/va_${name}_$s.sp4.o" 2>/dev/null &
      objs="$objs _obj/va_${name}_$s.sp4.o"
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "../libdartray_hip_$name.so" $objs _obj/dr_bvh_build.cpp.o _obj/dr_comm.cpp.o _obj/dr_bvh_device.hip.o _obj/dr_scene_prep.hip.o -ldl
    rm -f $objs; echo "built $name" ) &
done
wait
