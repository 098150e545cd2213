#!/bin/bash
# one rocprofv3 --pmc pass over two PPO epochs at 4096 envs with the fused update: per-wave counters of k_mlp.  usage (GPU box): tools/ppo_pmc.sh <tag>
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PPO_FUSED=1
d=$ROOT/gpurun_out/pmc_$TAG
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $d -- python3 $ROOT/tools/ppo_prof.py > $d.log 2>&1
python3 - "$d" > $ROOT/gpurun_out/${TAG}_pmc.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_mlp" in k or "k_grad" in k or "k_adam" in k: acc[k[:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, m in acc.items():
    mm = {c: sum(v) / len(v) for c, v in m.items()}
    w = mm.get("SQ_WAVES", 1) or 1
    print(k, {c: round(v / w, 1) for c, v in mm.items()})
PY
rm -rf $d
cat $ROOT/gpurun_out/${TAG}_pmc.txt
