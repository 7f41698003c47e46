#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in c3 c5; do
 for ch in 0 48 96 160; do
  if [ $ch = 0 ]; then unset PAL_CHUNK; else export PAL_CHUNK=$ch; fi
  timeout -k 10 200 python3 $ROOT/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${cfg}_$ch.json 2> $OUT/${cfg}_$ch.err
  python3 - $OUT/${cfg}_$ch.json $cfg $ch <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2],'chunk',sys.argv[3], d['value'])
PY
 done
done
unset PAL_CHUNK
for ov in 0 1 3; do
  PAL_OVERLAP=$ov timeout -k 10 200 python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --frames 32 > $OUT/ov_$ov.json 2> $OUT/ov_$ov.err
  python3 - $OUT/ov_$ov.json $ov <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('metric overlap',sys.argv[2], d['value'])
PY
done
for ch in 120 192 320; do
  PAL_CHUNK=$ch timeout -k 10 200 python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --frames 32 > $OUT/mch_$ch.json 2> $OUT/mch_$ch.err
  python3 - $OUT/mch_$ch.json $ch <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('metric chunk',sys.argv[2], d['value'])
PY
done
