#!/bin/bash
# One profiling pass on the GPU box (run through gpurun): `bash tools/profile_round.sh r02_a`
#   - the headline bench line (with the CPU baseline), then the same command under rocprofv3 --kernel-trace --stats
#   - PMC passes of the same command: FETCH_SIZE, WRITE_SIZE (separate passes, MI355X_MICROARCH.md section HBM) and the
#     fp64 instruction counters -> <tag>_pmc_summary.txt, <tag>_pmc_traffic.json, <tag>_pmc_flops.json
#   - one bench line + kernel statistics per BASELINE configuration C2..C5
#   - the C5 stream chain (tools/bench_stream.py)
# Outputs under gpurun_out/<tag>/; copy what is to be kept into profiles/.
set -u
TAG=${1:-r02}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (PARTS: which parts to run - "metric", "configs" (CFGS, default c2 c3 c4 c5), "stream"; default all.  One gpurun call holds
#  at most twenty minutes: `PARTS="metric" ...`, then `PARTS="configs stream" CFGS="c2 c3" ...`, ...)
PARTS=${PARTS:-metric configs stream}
CFGS=${CFGS:-c2 c3 c4 c5}
if [[ " $PARTS " == *" metric "* ]]; then
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats.log 2>&1; echo "stats rc=$?"
PMC_CMD="python3 $ROOT/bench.py --steps 2 --warmup 1 --frames 16 --no-cpu-baseline --no-kernel-events"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "flops SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  set -- $pass; name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc/$name -o $name --output-format csv -- $PMC_CMD > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT/pmc $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1
python3 $ROOT/tools/flops_per_pair.py $OUT/pmc_flops.json $OUT/pmc_traffic.json 16 > $OUT/fp64_flops_per_pair.json
fi
if [[ " $PARTS " == *" configs "* ]]; then
for cfg in $CFGS; do
  python3 $ROOT/bench.py --config $cfg --steps 10 --warmup 3 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; echo "bench $cfg rc=$?"
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$cfg -o stats --output-format csv -- python3 $ROOT/bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline > $OUT/stats_$cfg.log 2>&1
  # memory-side traffic and fp64 operations of the configuration's kernels: the same three counter passes as the metric run
  CFG_CMD="python3 $ROOT/bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events"
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "flops SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES"; do
    set -- $pass; name=$1; shift
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$cfg/$name -o $name --output-format csv -- $CFG_CMD > $OUT/pmc_${cfg}_$name.log 2>&1 || echo "pass $cfg $name failed"
  done
  python3 $ROOT/tools/pmc_summary.py $OUT/pmc_$cfg $OUT/pmc_traffic_$cfg.json > $OUT/pmc_summary_$cfg.txt 2>&1
done
fi
if [[ " $PARTS " == *" stream "* ]]; then
python3 $ROOT/tools/bench_stream.py 128 2 > $OUT/stream.json 2> $OUT/stream.err; echo "stream rc=$?"
fi
find $OUT -name "*.db" -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
ls $OUT | head -40
[ -f $OUT/bench.json ] && tail -c 400 $OUT/bench.json
