#!/bin/bash
# A/B of two builds of libpal_hip.so on ONE GPU box (box-to-box spread is ~4 %, run-to-run on a box < 1 %):
#   tools/ab_bench.sh ab/base.so ab/new.so [bench args]      (through gpurun; the in-tree library is swapped per run)
set -u
A=$1; B=$2; shift 2
LIB=pyaudiolocalization_amd/libpal_hip.so
cp $LIB /tmp/keep.so
for rep in 1 2 3; do
  for v in $A $B; do
    cp $v $LIB
    echo -n "$v  "
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"]), d["roofline"]["kernel"], d["roofline"]["avg_launch_us"])'
  done
done
cp /tmp/keep.so $LIB
