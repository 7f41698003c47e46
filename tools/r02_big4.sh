#!/bin/bash
# A/B of the register-resident row tiles (pfa_big.h): points per lane, workgroups per CU.
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02_big4
mkdir -p $OUT
cd $ROOT
LIB=pyaudiolocalization_amd/libpal_hip.so
cp ab/new.so $LIB
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
run() {  # lib, label, bench args
  cp $ROOT/ab/$1.so $ROOT/$LIB; shift; local label=$1; shift
  timeout -k 10 200 python3 $ROOT/bench.py "$@" --no-cpu-baseline > $OUT/$label.json 2>> $OUT/err.txt
  python3 - $OUT/$label.json $label <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), {k:v for k,v in d['kernels_alone_us'].items() if 'rows' in k or 'cols_stats' in k}, d['config']['workload'].split(';')[-1][:70])
PY
}
for v in base new; do run $v c3_$v --config c3 --steps 8 --warmup 3; done
for v in base new; do run $v c2_$v --config c2 --steps 8 --warmup 3; done
for L in 44108 44116; do for v in base new; do run $v l${L}_$v --frames 8 --length $L --steps 5 --warmup 2; done; done
for L in 44103 44118; do for v in base new b13two b13p32 b13p32two; do run $v l${L}_$v --frames 8 --length $L --steps 5 --warmup 2; done; done
cp $ROOT/ab/new.so $ROOT/$LIB
