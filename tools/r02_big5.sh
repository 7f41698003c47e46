#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02_big5
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
run() {
  local label=$1; shift
  timeout -k 10 200 python3 $ROOT/bench.py "$@" --no-cpu-baseline > $OUT/$label.json 2>> $OUT/err.txt
  python3 - $OUT/$label.json $label <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), d['kernels_alone_us'], d['config']['workload'].split(';')[-1][:70])
PY
}
run metric --steps 20 --warmup 5
run c2 --config c2 --steps 8 --warmup 3
PAL_PFA=0 run c2_four --config c2 --steps 8 --warmup 3
run c3 --config c3 --steps 8 --warmup 3
run c5 --config c5 --steps 8 --warmup 3
for L in 44107 44108 44109 44111 44103; do run l$L --frames 8 --length $L --steps 5 --warmup 2; done
