#!/bin/bash
# VALU instruction classes of the step kernel (rocprofv3 --pmc, one pass per counter group; GPU box).  usage: tools/pmc_valu_classes.sh <outdir> <N>
# run twice: as is, and with DW_FREEZE=1 (physics frozen: what is left is the task code)
OUT=$1; N=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_WAVES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  DW_PIPE=3 timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/pass$i -- python3 $ROOT/tools/prof_step.py $N 12 > $ROOT/$OUT/pass$i.log 2>&1 || echo "pass $i failed" >> $ROOT/$OUT/fail.log
done
cd $ROOT
python3 - "$OUT" <<PY
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "dw_k_step_oct" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m.get("SQ_WAVES", 1) or 1
print("per wave:", {k: round(v / w, 1) for k, v in sorted(m.items())})
PY
