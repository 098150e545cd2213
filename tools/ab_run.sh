#!/bin/bash
# A/B timing of library variants on ONE GPU box: tools/pipe_time.py (octet pipeline, 4096 and 16384 envs, three rounds each) once per
# library named on the command line (files under isaacgymdyros_amd/_ab/; list the baseline first AND last to see the box's drift).
# usage (GPU box): bash tools/ab_run.sh base.so variant.so base.so
mkdir -p gpurun_out
for lib in "$@"; do
  echo "== $lib"
  DW_LIB=isaacgymdyros_amd/_ab/$lib python tools/pipe_time.py --pipes 3 --envs 4096,16384 --rounds 3 2>&1 | grep "N="
done
