#!/bin/bash
# builds variants of the library with phases of the octet physics compiled out (-DOCT_ABL_*: instruction-count and register
# experiments, never shipped) into isaacgymdyros_amd/_ab/libdw_<tag>.so; the other two translation units come from _obj/
set -e
cd "$(dirname "$0")/.."
mkdir -p isaacgymdyros_amd/_ab
F="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -fno-strict-aliasing -fno-slp-vectorize"
for tag in "$@"; do
  hipcc $F -D$tag -c -o isaacgymdyros_amd/_ab/abl_$tag.o isaacgymdyros_amd/csrc/dw_oct_kernels.hip
  hipcc --offload-arch=gfx950 -shared -fPIC -o isaacgymdyros_amd/_ab/libdw_$tag.so isaacgymdyros_amd/_obj/dw_hip.o isaacgymdyros_amd/_obj/dw_quad_kernels.o isaacgymdyros_amd/_obj/dw_amp.o isaacgymdyros_amd/_ab/abl_$tag.o
done
