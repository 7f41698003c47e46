#!/bin/bash
# Round 2, first GPU pass: the whole GPU test suite, then one bench line per BASELINE configuration (metric workload with
# the fused column pass off and on), and the list of fp64 instruction counters rocprofv3 knows on this part.
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for cfg in metric c2 c3 c4 c5; do
  timeout -k 10 300 python3 $ROOT/bench.py --config $cfg --steps 5 --warmup 2 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  echo "bench $cfg rc=$?"
  tail -c 600 $OUT/bench_$cfg.json | head -c 300; echo
done
PAL_FUSED=1 timeout -k 10 300 python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_metric_fused.json 2> $OUT/bench_metric_fused.err
echo "bench fused rc=$?"
rocprofv3 -L 2>/dev/null | grep -i -E "F64|FLOPS|VALU_(ADD|MUL|FMA|TRANS)" | head -40 > $OUT/counters_f64.txt
wc -l $OUT/counters_f64.txt
