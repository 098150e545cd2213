#!/bin/bash
# builds a -DDQ_STAMPS copy of the library into isaacgymdyros_amd/_ab/ (here) ; run tools/phase_stamps.py with DW_LIB set on the GPU box
set -e
cd "$(dirname "$0")/.."
mkdir -p isaacgymdyros_amd/_ab
hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared -fno-strict-aliasing -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -DDQ_STAMPS \
  -o isaacgymdyros_amd/_ab/libdw_stamps.so isaacgymdyros_amd/csrc/dw_hip.hip
