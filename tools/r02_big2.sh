#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
 for big in 1 0; do
  PAL_PFA_BIG=$big timeout -k 10 200 python3 $ROOT/bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $OUT/c3_big${big}_$rep.json 2> $OUT/c3.err
  python3 - $OUT/c3_big${big}_$rep.json $big <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('c3 big',sys.argv[2], d['value'])
PY
 done
done
# a length with N1 = 5 and a big tile: L = 20224 -> n = 40447 = 11 x 3677?  use the plan printed
for L in 16385 24005 12008; do
 for big in 1 0; do
  PAL_PFA_BIG=$big timeout -k 10 200 python3 $ROOT/bench.py --config c3 --length $L --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/l${L}_big$big.json 2> $OUT/l.err
  python3 - $OUT/l${L}_big$big.json $big $L <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('L',sys.argv[3],'big',sys.argv[2], d['value'], d['config']['workload'].split(';')[-1])
PY
 done
done
