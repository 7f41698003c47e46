#!/bin/bash
# experiment: sub-groups of the row / column pass small enough for Y to stay in the XCDs' L2 (PAL_PFA_SUB), separate launches
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sub in 0 8 16 24 32 48; do
  PAL_FUSED=0 PAL_PFA_SUB=$sub timeout -k 10 200 python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --frames 32 > $OUT/sub_$sub.json 2> $OUT/sub_$sub.err
  python3 - $OUT/sub_$sub.json $sub <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('sub',sys.argv[2], d['value'], {k:v for k,v in d['kernels_alone_us'].items() if 'pfa_rows' in k or 'pfa_cols' in k})
PY
done
