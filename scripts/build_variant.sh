#!/bin/bash
# build a variant of the library into build_variants/libdexsim_<name>.so:  bash scripts/build_variant.sh <name> [-Dflags...]
NAME=$1; shift
R=/root/repo
mkdir -p $R/build_variants
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared -std=c++17 "$@" -o $R/build_variants/libdexsim_$NAME.so $R/dexrobot_isaac_amd/csrc/dexsim.hip
