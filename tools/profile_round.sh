#!/bin/bash
# One profiling pass of the headline bench on the GPU box (run through gpurun): kernel-trace statistics, the
# bench line itself and the PMC traffic passes.  Outputs under gpurun_out/; copy what is to be kept into profiles/.
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
mkdir -p $OUT/stats
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
bash $ROOT/tools/pmc.sh > $OUT/pmc.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/pmc $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1
ls $OUT/stats | head
tail -1 $OUT/bench.json | cut -c 1-400
