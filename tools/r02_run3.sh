#!/bin/bash
# quick pass: fused-path tests, then the metric bench with the fallback counters printed
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02c
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
PAL_DEBUG_FALLBACK=1 timeout -k 10 300 python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_metric.json 2> $OUT/bench_metric.err; echo "metric rc=$?"
tail -3 $OUT/bench_metric.err
PAL_FUSED=0 timeout -k 10 300 python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_metric_unfused.json 2> $OUT/bench_metric_unfused.err; echo "unfused rc=$?"
timeout -k 10 300 python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --max-expected-delay -1 > $OUT/bench_metric_nowin.json 2> $OUT/bench_metric_nowin.err; echo "nowin rc=$?"
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02c/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], d['ms_per_step'], d.get('kernels_alone_us'))
    except Exception as e: print(f, 'ERR', e)
PY
