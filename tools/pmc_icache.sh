#!/bin/bash
# usage (GPU box): tools/pmc_icache.sh <outdir> <N> <steps> -- instruction-cache counters of the step kernels
OUT=$1; N=$2; STEPS=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES" "SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/pass$i -- python3 $ROOT/tools/prof_step.py $N $STEPS > $ROOT/$OUT/pass$i.log 2>&1 || echo "pass $i failed" >> $ROOT/$OUT/fail.log
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "dw_k" in k:
            acc[k.split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
