#!/bin/bash
# build_variant.sh NAME [-DACF_X=.. ...]: a second build of the library with other compile-time constants, as
# acf_amd/libacf_hip_NAME.so (git-ignored; travels to the GPU box).  Select it with ACF_HIP_LIB=acf_amd/libacf_hip_NAME.so.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden \
  -Wno-unused-function -Wno-pass-failed "$@" -x hip acf_amd/csrc/acf_hip.hip -x hip acf_amd/csrc/host_plan.cpp \
  -o acf_amd/libacf_hip_$name.so -Wl,--version-script=acf_amd/csrc/exports.map
echo built acf_amd/libacf_hip_$name.so
