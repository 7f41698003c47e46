#!/bin/bash
# bench under several environment settings on one box: tools/exp_env.sh <tag> "VAR=1 VAR2=x" "..." ...
set -u
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting timeout -k 10 200 python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events ${BENCH_ARGS:-} > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  echo "== [$setting] rc=$? $(python3 -c "
import json
d=json.loads(open('$OUT/bench_$i.json').read().strip().splitlines()[-1]); print(round(d['value']))" 2>&1)" | tee -a $OUT/summary.txt
done
