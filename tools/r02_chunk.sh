#!/bin/bash
# Four-step route: does a launch group small enough for its workspace to stay in the 256 MiB Infinity Cache pay?
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02_chunk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c4; do for ch in 0 16 24 32 48 64 120; do
  PAL_CHUNK=$ch python3 $ROOT/bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline > $OUT/b_${cfg}_$ch.json 2>> $OUT/err.log
  python3 -c "import json;d=json.load(open('$OUT/b_${cfg}_$ch.json'));print('$cfg chunk=$ch',d['value'],d['kernels_alone_us'])"
done; done
for ch in 16 32; do
  PAL_OVERLAP=0 PAL_CHUNK=$ch python3 $ROOT/bench.py --config c2 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/b1_c2_$ch.json 2>> $OUT/err.log
  python3 -c "import json;d=json.load(open('$OUT/b1_c2_$ch.json'));print('c2 one stream chunk=$ch',d['value'])"
done
