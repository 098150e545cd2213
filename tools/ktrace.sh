#!/bin/bash
# usage (on the GPU box): tools/ktrace.sh <tag> [bench args...]  -- rocprofv3 kernel trace of bench.py, per-kernel summary into gpurun_out/<tag>_kernels.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o $TAG -- python3 $ROOT/bench.py "$@" > $ROOT/gpurun_out/${TAG}_bench.json 2> $ROOT/gpurun_out/${TAG}_bench.err
cd $ROOT
python3 tools/ktrace_summary.py gpurun_out/prof_$TAG > gpurun_out/${TAG}_kernels.txt
cat gpurun_out/${TAG}_kernels.txt
