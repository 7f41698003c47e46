#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02_reg4
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" $OUT/pytest.log | tail -8
cd /tmp && export TMPDIR=/tmp
run() {
  local label=$1; shift
  timeout -k 10 200 python3 $ROOT/bench.py "$@" --no-cpu-baseline > $OUT/$label.json 2>> $OUT/err.txt
  python3 - $OUT/$label.json $label <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), d['kernels_alone_us'])
PY
}
run l44102 --frames 8 --length 44102 --steps 5 --warmup 2
run l44105 --frames 8 --length 44105 --steps 5 --warmup 2
run l44300 --frames 8 --length 44300 --steps 5 --warmup 2
