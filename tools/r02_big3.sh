#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02j
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for cfg in c5 c3 c2; do
 for pfa in 1 0; do
  PAL_PFA=$pfa timeout -k 10 200 python3 $ROOT/bench.py --config $cfg --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-events > $OUT/${cfg}_pfa$pfa.json 2> $OUT/err.txt
  python3 - $OUT/${cfg}_pfa$pfa.json $cfg $pfa <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2],'pfa',sys.argv[3], d['value'], d['config']['workload'].split(';')[-1][:110])
PY
 done
done
for L in 44103 44116 44126 48003 96003; do
 for pfa in 1 0; do
  PAL_PFA=$pfa timeout -k 10 200 python3 $ROOT/bench.py --frames 8 --length $L --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/l${L}_pfa$pfa.json 2> $OUT/err.txt
  python3 - $OUT/l${L}_pfa$pfa.json $L $pfa <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('L',sys.argv[2],'pfa',sys.argv[3], d['value'], d['config']['workload'].split(';')[-1][:110])
PY
 done
done
