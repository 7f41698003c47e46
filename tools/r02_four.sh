#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ov in 3 1 0 2; do
 for ch in 0 64 120; do
  if [ $ch = 0 ]; then unset PAL_CHUNK; else export PAL_CHUNK=$ch; fi
  PAL_OVERLAP=$ov timeout -k 10 200 python3 $ROOT/bench.py --frames 8 --length 44104 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/o${ov}_c$ch.json 2> $OUT/err.txt
  python3 - $OUT/o${ov}_c$ch.json $ov $ch <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('overlap',sys.argv[2],'chunk',sys.argv[3], d['value'], d['config']['workload'].split(';')[-1][:60])
PY
 done
done
