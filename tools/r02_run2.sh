#!/bin/bash
# Round 2, second GPU pass: GPU tests with the fused column pass (per-block pivots) on by default, the metric bench with
# it on and off, C4, the C5 stream chain, and the fp64 instruction counters of the metric run.
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $ROOT/bench.py --steps 10 --warmup 3 > $OUT/bench_metric.json 2> $OUT/bench_metric.err; echo "metric rc=$?"
PAL_FUSED=0 timeout -k 10 300 python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_metric_unfused.json 2> $OUT/bench_metric_unfused.err; echo "unfused rc=$?"
timeout -k 10 300 python3 $ROOT/bench.py --config c4 --steps 5 --warmup 2 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?"
timeout -k 10 400 python3 $ROOT/tools/bench_stream.py 64 2 > $OUT/stream.json 2> $OUT/stream.err; echo "stream rc=$?"
tail -c 1500 $OUT/stream.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES -d $OUT/flops -o flops --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --frames 16 --no-cpu-baseline --no-kernel-events > $OUT/flops.log 2>&1
echo "flops rc=$?"
python3 $ROOT/tools/pmc_summary.py $OUT/flops > $OUT/flops_summary.txt 2>&1
rm -rf $OUT/flops/*/*.db
grep -A7 "k_pfa_cols_stats\|k_pfa_rows_rader\|k_peak_finish" $OUT/flops_summary.txt | head -60
