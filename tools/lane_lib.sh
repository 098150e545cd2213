#!/bin/bash
# usage: tools/lane_lib.sh <name> [extra hipcc flags for dw_lane_kernels.hip, e.g. -DDL_STAMPS]
# a variant of the lane translation unit in a complete library, isaacgymdyros_amd/_ab/<name>.so (tools/tu_lib.sh: flags and the
# other units as build.py has them).  Tools only -- the product always loads the in-tree library.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
tools/tu_lib.sh "$NAME" dw_lane_kernels.hip ${DL_BASEFLAGS:-} "$@"
