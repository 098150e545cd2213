#!/bin/bash
# one rocprofv3 --pmc pass (instruction counts) per library variant; run on the GPU box.  usage: tools/pmc_valu.sh <outdir> <N> tag...
OUT=$1; N=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  lib=$ROOT/isaacgymdyros_amd/_ab/libdw_$tag.so
  [ "$tag" = base ] && lib=$ROOT/isaacgymdyros_amd/libdyroswalk_hip.so
  [ "$tag" = freeze ] && lib=$ROOT/isaacgymdyros_amd/libdyroswalk_hip.so
  fr=""; [ "$tag" = freeze ] && fr=1
  DW_FREEZE=$fr DW_LIB=$lib DW_PIPE=3 timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $ROOT/$OUT/$tag -- python3 $ROOT/tools/prof_step.py $N 12 > $ROOT/$OUT/$tag.log 2>&1
done
cd $ROOT
python3 - "$OUT" "$@" <<PY
import csv, glob, sys, collections
out=sys.argv[1]
for tag in sys.argv[2:]:
    acc=collections.defaultdict(list)
    for f in glob.glob(out+"/"+tag+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "dw_k_step_oct" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    m={k: sum(v)/len(v) for k,v in acc.items()}
    w=m.get("SQ_WAVES",1) or 1
    print(tag, {k: round(v/w,1) for k,v in m.items()})
PY
