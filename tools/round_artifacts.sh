#!/bin/bash
# Produces the judged artefacts of a round on the GPU box: bench line, rocprofv3 kernel stats of the same command,
# PMC passes (incl. HBM traffic), a 1-rank torchrun of the distributed path.  usage: tools/round_artifacts.sh <tag>
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
# PMC passes first: bench.py reports the HBM traffic of THIS build (profiles/pmc_traffic.json) in its roofline object
tools/pmc_passes.sh gpurun_out/$TAG/pmc 16384 40 > /dev/null 2>&1
python tools/collect_profiles.py $TAG --traffic-only
timeout -k 10 800 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cat $OUT/bench.json | cut -c1-400
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 128 --warmup 16 --no-cpu-baseline --no-also-4096 --no-config5 --no-terrain --no-ppo --no-amp > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err
echo "torchrun rc=$?"; cut -c1-200 $OUT/bench_torchrun1.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py --no-cpu-baseline --no-ppo --no-amp --no-also-4096 --no-config5 --no-terrain > $OUT/bench_under_rocprof.json 2>/dev/null
cd $ROOT
head -3 $(find $OUT/prof -name "*kernel_stats.csv" | head -1) | cut -c1-160
cat $OUT/pmc/summary.txt | head -40
