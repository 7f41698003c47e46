#!/bin/bash
# The parity suite under every diagnostic toggle of the engine (each run: a fresh process with the variable set globally).
#   bash tools/toggle_matrix.sh        (through gpurun; about 40 s per toggle)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/toggles
mkdir -p $OUT
cd $ROOT
for t in "PAL_FUSED=0" "PAL_PFA=0" "PAL_FOUR_REG=0" "PAL_FOUR_REG=13" "PAL_XCD_ROWS=0" "PAL_PFA_BIG=0" "PAL_OVERLAP=0" "PAL_RADER=0" "PAL_PFA_FWD=0" "PAL_RADIX3=0" "PAL_FIN=0" "PAL_R89=0" "PAL_FIN_HIST=1" "PAL_FIN_DENSE=1" "PAL_FIN_DENSE=0" "PAL_FIN_FOUR=1" "PAL_FIN_WIDE=1" "PAL_FIN_SERIAL=1" "PAL_LEAN_STORE=0" "PAL_ROWS_LEAN=0"; do
  name=$(echo $t | tr '=' '_')
  env $t timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -m gpu -q -p no:cacheprovider > $OUT/$name.log 2>&1
  echo "$t rc=$? $(grep -E "passed|failed" $OUT/$name.log | tail -1)"; grep -E "^FAILED" $OUT/$name.log | cut -c1-150
done
