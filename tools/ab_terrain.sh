#!/bin/bash
# A/B of library variants on the height field (terrain_cfg defaults, trimesh + curriculum) on ONE GPU box: step time at 16384 envs, plane
# beside it, and one PMC pass per variant.  usage (GPU box): bash tools/ab_terrain.sh <tag> a.so b.so ...   (isaacgymdyros_amd/_ab/)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_abt.txt
mkdir -p $ROOT/gpurun_out
: > $OUT
for lib in "$@" "$1"; do
  echo "== $lib" >> $OUT
  DW_LIB=$ROOT/isaacgymdyros_amd/_ab/$lib python $ROOT/tools/pipe_time.py --pipes 3 --envs 16384 --rounds 3 --terrain 2>&1 | grep "N=" | sed 's/^/terrain /' >> $OUT
  DW_LIB=$ROOT/isaacgymdyros_amd/_ab/$lib python $ROOT/tools/pipe_time.py --pipes 3 --envs 16384 --rounds 2 2>&1 | grep "N=" | sed 's/^/plane   /' >> $OUT
done
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  d=$ROOT/gpurun_out/${TAG}_pmct_${lib%.so}
  DW_TERRAIN=1 DW_LIB=$ROOT/isaacgymdyros_amd/_ab/$lib DW_PIPE=3 timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $d -- python3 $ROOT/tools/prof_step.py 16384 60 > $d.log 2>&1
  python3 - "$d" "$lib" >> $OUT <<PY
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "dw_k_step_oct" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v[-20:]) / len(v[-20:]) for k, v in acc.items()}
w = m.get("SQ_WAVES", 1) or 1
print("pmc(last 20 launches)", sys.argv[2], {k: round(v / w, 1) for k, v in m.items()})
PY
  rm -rf $d
done
cat $OUT
