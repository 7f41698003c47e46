#!/bin/bash
# A/B build: tools/build_variant.sh <name> [-DFLAG ...]  ->  tools/bin/libpal_<name>.so (pfa.hip and peaks.hip recompiled with the
# flags, the other objects taken from the in-tree build).  Run a variant with PAL_LIB_PATH=tools/bin/libpal_<name>.so.
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/pyaudiolocalization_amd/csrc
OUT=$ROOT/tools/bin
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -I$ROOT/include -I$SRC"
make -s -C $SRC -j4
hipcc $FLAGS -fno-signed-zeros "$@" -c $SRC/pfa.hip -o $OUT/pfa_$NAME.o &
hipcc $FLAGS "$@" -c $SRC/peaks.hip -o $OUT/peaks_$NAME.o &
wait
hipcc -shared -fPIC --offload-arch=gfx950 $SRC/bluestein.o $OUT/pfa_$NAME.o $OUT/peaks_$NAME.o $SRC/sim.o $SRC/images.o $SRC/pal_api.o -o $OUT/libpal_$NAME.so -ldl
echo "built $OUT/libpal_$NAME.so"
