#!/bin/bash
# builds a -DDQ_STAMPS copy of the library into isaacgymdyros_amd/_ab/ (here) ; run tools/phase_stamps.py with DW_LIB set on the GPU box
set -e
cd "$(dirname "$0")/.."
mkdir -p isaacgymdyros_amd/_ab
F="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -fno-strict-aliasing -fno-slp-vectorize -DDQ_STAMPS"
hipcc $F -mllvm -amdgpu-sched-strategy=iterative-ilp -c -o isaacgymdyros_amd/_ab/st_a.o isaacgymdyros_amd/csrc/dw_hip.hip
hipcc $F -c -o isaacgymdyros_amd/_ab/st_b.o isaacgymdyros_amd/csrc/dw_quad_kernels.hip
hipcc $F -c -o isaacgymdyros_amd/_ab/st_c.o isaacgymdyros_amd/csrc/dw_oct_kernels.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o isaacgymdyros_amd/_ab/libdw_stamps.so isaacgymdyros_amd/_ab/st_a.o isaacgymdyros_amd/_ab/st_b.o isaacgymdyros_amd/_ab/st_c.o isaacgymdyros_amd/_obj/dw_amp.o
