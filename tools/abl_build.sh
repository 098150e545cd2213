#!/bin/bash
# builds variants of the library with phases of the octet physics compiled out (-DOCT_ABL_*: instruction-count and register
# experiments, never shipped) into isaacgymdyros_amd/_ab/libdw_<tag>.so; flags and the other translation units as build.py has them
# (tools/tu_lib.sh).  usage: tools/abl_build.sh OCT_ABL_INWARD OCT_ABL_GEOM ...
set -e
cd "$(dirname "$0")/.."
for tag in "$@"; do tools/tu_lib.sh libdw_$tag dw_oct_kernels.hip -D$tag; done
