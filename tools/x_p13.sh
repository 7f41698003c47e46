#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
LIB=$ROOT/pyaudiolocalization_amd/libpal_hip.so
cp $LIB /tmp/keep.so
cd /tmp && export TMPDIR=/tmp
run() { timeout -k 10 200 python3 $ROOT/bench.py "$@" --no-cpu-baseline > /tmp/o.json 2>/tmp/e.txt; python3 -c "
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('   ', round(d['value']), {k:v for k,v in d['kernels_alone_us'].items() if 'rows' in k})"; }
for rep in 1 2; do for v in plan_new plan_13old; do
  cp $ROOT/ab/$v.so $LIB; echo "== $v"
  run --frames 8 --length 44103 --steps 5 --warmup 2
  run --frames 8 --length 44102 --steps 5 --warmup 2
done; done
cp /tmp/keep.so $LIB
