#!/bin/bash
# usage: tools/ab_lib.sh <name> [extra hipcc flags for the quad TU, e.g. -DDQ_VARIANT_X]
# builds a complete library (both translation units, as isaacgymdyros_amd/build.py does) into isaacgymdyros_amd/_ab/<name>.so
# for tools/ab_time.py; tools only -- the product always loads the in-tree library.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p isaacgymdyros_amd/_ab
F="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -fno-strict-aliasing"
SLP=${DW_AB_SLP:--fno-slp-vectorize}
[ -f isaacgymdyros_amd/_ab/base_a.o ] || hipcc $F -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -c -o isaacgymdyros_amd/_ab/base_a.o isaacgymdyros_amd/csrc/dw_hip.hip
hipcc $F $SLP "$@" -Rpass-analysis=kernel-resource-usage -c -o isaacgymdyros_amd/_ab/${NAME}_b.o isaacgymdyros_amd/csrc/dw_quad_kernels.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o isaacgymdyros_amd/_ab/${NAME}.so isaacgymdyros_amd/_ab/base_a.o isaacgymdyros_amd/_ab/${NAME}_b.o
echo isaacgymdyros_amd/_ab/${NAME}.so
