#!/bin/bash
# A/B of library variants at 1024 / 4096 / 8192 envs (the one-wave-per-SIMD builds: hex up to 4096, the octet KEEP build at 8192) on ONE GPU
# box, baseline first and last.  usage (GPU box): bash tools/ab_mid.sh <tag> base.so v1.so ...   (files under isaacgymdyros_amd/_ab/)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_abm.txt
mkdir -p $ROOT/gpurun_out
: > $OUT
for lib in "$@" "$1"; do
  echo "== $lib" >> $OUT
  DW_LIB=$ROOT/isaacgymdyros_amd/_ab/$lib python $ROOT/tools/pipe_time.py --pipes 3 --envs 1024,4096,8192 --rounds 3 --steps 400 2>&1 | grep "N=" >> $OUT
done
cat $OUT
