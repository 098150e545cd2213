#!/bin/bash
# usage: tools/lane_lib.sh <name> [extra hipcc flags for dw_lane_kernels.hip, e.g. -DDL_STAMPS]
# builds a complete library with a variant of the lane translation unit into isaacgymdyros_amd/_ab/<name>.so (the other units:
# the objects of the in-tree build, isaacgymdyros_amd/_obj).  Tools only -- the product always loads the in-tree library.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p isaacgymdyros_amd/_ab
[ -f isaacgymdyros_amd/_obj/dw_hip.o ] || python -c "from isaacgymdyros_amd import build; build.build(force=True)"
hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -fno-strict-aliasing -fno-slp-vectorize ${DL_BASEFLAGS:-} "$@" -Rpass-analysis=kernel-resource-usage -c -o isaacgymdyros_amd/_ab/${NAME}_lane.o isaacgymdyros_amd/csrc/dw_lane_kernels.hip 2>&1 | grep -E "error|Function Name|ScratchSize|VGPRs:|AGPRs" | sed 's/\[-Rpass.*//' | paste - - - - | sed 's/dw_lane_kernels.hip:[0-9]*:1: remark: //g' || true
hipcc --offload-arch=gfx950 -shared -fPIC -o isaacgymdyros_amd/_ab/${NAME}.so isaacgymdyros_amd/_obj/dw_hip.o isaacgymdyros_amd/_obj/dw_quad_kernels.o isaacgymdyros_amd/_obj/dw_oct_kernels.o isaacgymdyros_amd/_obj/dw_amp.o isaacgymdyros_amd/_ab/${NAME}_lane.o
echo isaacgymdyros_amd/_ab/${NAME}.so
