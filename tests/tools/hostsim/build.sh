#!/bin/sh
# Sanitizer harness: the v1 (lane-per-instance) kernels + the C-ABI orchestration as host C++ under ASan/UBSan.  See hip/hip_runtime.h.
set -e
here=$(cd "$(dirname "$0")" && pwd)
src=$here/../../../ilqr_planner_amd/csrc
mkdir -p "$here/_build"
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -shared \
    -I"$here" -I"$src" -Wno-unused-result \
    -x c++ "$src/ilqr_kernels.hip" "$src/ilqr_capi.cpp" "$src/urdf_chain.cpp" "$here/stubs.cpp" \
    -o "$here/_build/libilqr_hostsim.so"
echo "$here/_build/libilqr_hostsim.so"
