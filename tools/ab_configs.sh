#!/bin/bash
# A/B of builds under ab/ over several bench configurations on one box: tools/r02_ab2.sh lib1 lib2 ...
set -u
ROOT=$GRAFT_REPO_ROOT
LIB=$ROOT/pyaudiolocalization_amd/libpal_hip.so
cp $LIB /tmp/keep.so
cd /tmp && export TMPDIR=/tmp
run() {
  timeout -k 10 200 python3 $ROOT/bench.py "$@" --no-cpu-baseline > /tmp/out.json 2>/tmp/err.txt
  python3 -c "
import json
d=json.loads(open('/tmp/out.json').read().strip().splitlines()[-1])
print('   ', round(d['value']), {k:v for k,v in d['kernels_alone_us'].items() if 'rows' in k})"
}
for v in "$@"; do
  cp $ROOT/ab/$v.so $LIB
  echo "== $v"
  echo " c2 four-step"; PAL_PFA=0 run --config c2 --steps 8 --warmup 3
  echo " c4"; run --config c4 --steps 6 --warmup 2
  echo " c3"; run --config c3 --steps 8 --warmup 3
  echo " L=44103 (8192-point tiles)"; run --frames 8 --length 44103 --steps 5 --warmup 2
  echo " L=44102 (four-step)"; run --frames 8 --length 44102 --steps 5 --warmup 2
done
cp /tmp/keep.so $LIB
