#!/bin/bash
# usage (on the GPU box): tools/ppo_ktrace.sh <tag>  -- rocprofv3 kernel trace of two PPO epochs at 4096 envs (tools/ppo_prof.py), per-kernel summary
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o $TAG -- python3 $ROOT/tools/ppo_prof.py > $ROOT/gpurun_out/${TAG}_ppo.log 2>&1
cd $ROOT
python3 - gpurun_out/prof_$TAG > gpurun_out/${TAG}_kernels.txt <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:110]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values())
print("total kernel time %.1f ms" % (tot / 1e6))
for k in sorted(acc, key=lambda k: -sum(acc[k]))[:40]:
    v = acc[k]
    print("%-110s %7d %9.1f us %6.1f%%" % (k, len(v), sum(v) / len(v) / 1e3, 100.0 * sum(v) / tot))
PY
rm -rf gpurun_out/prof_$TAG
tail -5 gpurun_out/${TAG}_ppo.log
cat gpurun_out/${TAG}_kernels.txt
