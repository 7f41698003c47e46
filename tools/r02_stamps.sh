#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PAL_OVERLAP=0 PAL_DEBUG_STAMPS=1 timeout -k 10 200 python3 $ROOT/bench.py --steps 1 --warmup 1 --frames 4 --no-cpu-baseline --no-kernel-events > $OUT/fused.json 2> $OUT/fused.err
PAL_OVERLAP=0 PAL_FUSED=0 PAL_DEBUG_STAMPS=1 timeout -k 10 200 python3 $ROOT/bench.py --steps 1 --warmup 1 --frames 4 --no-cpu-baseline --no-kernel-events > $OUT/unfused.json 2> $OUT/unfused.err
echo fused; grep "k_peak_finish" $OUT/fused.err | tail -8
echo unfused; grep "k_peak_finish" $OUT/unfused.err | tail -8
