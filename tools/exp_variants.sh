#!/bin/bash
# bench the in-tree library and A/B variants under tools/bin on one box: tools/exp_variants.sh <tag> name1 name2 ...
set -u
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v = base ]; then unset PAL_LIB_PATH; else export PAL_LIB_PATH=$ROOT/tools/bin/libpal_$v.so; fi
  timeout -k 10 200 python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  echo "== $v rc=$?"
  python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1])
print('   value', round(d['value']), 'alone', d['kernels_alone_us'])
print('   live', {k:(round(v['ms']/v['launches']*1e3,1)) for k,v in d['kernels_ms'].items()})" 2>&1 | tee -a $OUT/summary.txt
done
