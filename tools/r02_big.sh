#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02h
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "prime_factor_route or fused_column or c3 or small_and_odd or phat_correlation" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for big in 1 0; do
  PAL_PFA_BIG=$big timeout -k 10 200 python3 $ROOT/bench.py --config c3 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/c3_big$big.json 2> $OUT/c3_big$big.err
  python3 - $OUT/c3_big$big.json $big <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('c3 big',sys.argv[2], d['value'], d['kernels_alone_us'])
PY
done
