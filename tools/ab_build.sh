#!/bin/bash
# usage: tools/ab_build.sh "<sed expr on dw_hip.hip>" [extra hipcc flags]  -> rebuilds the in-tree lib from a patched copy
set -e
cd $(dirname $0)/../isaacgymdyros_amd/csrc
sed "$1" dw_hip.hip > _ab.hip
hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared -fno-strict-aliasing -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp $2 -o ../libdyroswalk_hip.so _ab.hip
rm -f _ab.hip
