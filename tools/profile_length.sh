#!/bin/bash
# Kernel statistics and memory-side traffic of ONE frame length (default 44102: the four-step route with register rows):
#   bash tools/profile_length.sh 44102 <tag>      (through gpurun) -> gpurun_out/<tag>_l<length>/
set -u
L=${1:-44102}
TAG=${2:-len}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/${TAG}_l$L
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --frames 16 --length $L --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $ROOT/bench.py --frames 16 --length $L --steps 4 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1; echo "stats rc=$?"
PMC_CMD="python3 $ROOT/bench.py --frames 8 --length $L --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc/$name -o $name --output-format csv -- $PMC_CMD > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT/pmc $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
cat $OUT/pmc_traffic.json
