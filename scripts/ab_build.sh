#!/bin/bash
# Variant builds of libilqr_hip.so for A/B runs with scripts/ab_libs.py (run from ilqr_planner_amd/csrc after `make`; outputs under build/exp/,
# which is git-ignored but travels to the GPU box):
#   scripts/ab_build.sh sed  NAME FILE.hip 'sed args...'     FILE.hip edited by sed (line-addressed edits of ONE kernel file), rest of the library as built
#   scripts/ab_build.sh flag NAME FILE.hip extra-flags...    FILE.hip compiled with extra compiler flags (e.g. -mllvm -amdgpu-sched-strategy=max-ilp)
#   scripts/ab_build.sh capi NAME 'sed args...'              ilqr_capi.cpp edited by sed (launch schedule experiments)
# The ablation tables in profiles/r02_forward_ablation.txt were made this way (variants that drop a part of a kernel give wrong results on
# purpose; only their time is read).
set -e
mode=$1; name=$2; shift 2
mkdir -p build/exp
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -I."
case $mode in
  sed)  src=$1; shift; sed "$@" $src > build/exp/${name}_$src; hipcc $FLAGS -c build/exp/${name}_$src -o build/exp/${name}.o; skip=build/${src%.hip}.o ;;
  flag) src=$1; shift; hipcc $FLAGS "$@" -c $src -o build/exp/${name}.o; skip=build/${src%.hip}.o ;;
  capi) sed "$@" ilqr_capi.cpp > build/exp/${name}_capi.cpp; hipcc $FLAGS -x hip -c build/exp/${name}_capi.cpp -o build/exp/${name}.o; skip=build/ilqr_capi.o ;;
  *) echo "mode: sed | flag | capi"; exit 1 ;;
esac
objs=$(ls build/*.o | grep -v "$skip")
hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/lib_${name}.so $objs build/exp/${name}.o
echo built build/exp/lib_${name}.so
