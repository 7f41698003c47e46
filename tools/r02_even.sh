#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02_even
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" $OUT/pytest.log | tail -8
cd /tmp && export TMPDIR=/tmp
run() {
  local label=$1; shift
  PAL_DEBUG_FALLBACK=1 timeout -k 10 200 python3 $ROOT/bench.py "$@" --no-cpu-baseline > $OUT/$label.json 2> $OUT/$label.err
  grep "row(s)" $OUT/$label.err | sort | uniq -c | tail -3
  python3 - $OUT/$label.json $label <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), d['kernels_alone_us'])
PY
}
run l44300 --frames 8 --length 44300 --steps 5 --warmup 2
run metric --steps 20 --warmup 5
run c3 --config c3 --steps 8 --warmup 3
run c2 --config c2 --steps 8 --warmup 3
run l44103 --frames 8 --length 44103 --steps 5 --warmup 2
