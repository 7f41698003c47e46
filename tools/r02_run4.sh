#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02f
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "prefilter or filter or stream or c1_ or c2_ or c5 or localize or position or batched" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $ROOT/tools/bench_stream.py 128 2 > $OUT/stream.json 2> $OUT/stream.err; echo "stream rc=$?"
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02f/stream.json').read().strip().splitlines()[-1])
print({k:v for k,v in d.items() if k!='kernels_ms'})
k=d['kernels_ms']
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms'])[:10]: print('  ',n,v)
PY
