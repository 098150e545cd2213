#!/bin/bash
# builds a -DDQ_STAMPS copy of the octet unit into isaacgymdyros_amd/_ab/libdw_stamps.so (flags and the other units as build.py has
# them, tools/tu_lib.sh); run tools/phase_stamps.py with DW_LIB set on the GPU box
set -e
cd "$(dirname "$0")/.."
tools/tu_lib.sh libdw_stamps dw_oct_kernels.hip -DDQ_STAMPS
