#!/bin/bash
# A/B of builds of libpal_hip.so on one GPU box: tools/r02_ab.sh "<bench args>" lib1 lib2 ... (libraries under ab/)
set -u
ROOT=$GRAFT_REPO_ROOT
ARGS=$1; shift
LIB=$ROOT/pyaudiolocalization_amd/libpal_hip.so
cp $LIB /tmp/keep.so
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for v in "$@"; do
    cp $ROOT/ab/$v.so $LIB
    timeout -k 10 200 python3 $ROOT/bench.py $ARGS --no-cpu-baseline > /tmp/out.json 2>/tmp/err.txt
    python3 -c "
import json
d=json.loads(open('/tmp/out.json').read().strip().splitlines()[-1])
print('$v', round(d['value']), d['kernels_alone_us'])"
  done
done
cp /tmp/keep.so $LIB
