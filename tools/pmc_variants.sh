#!/bin/bash
# dynamic instruction counts of the finishing column pass under A/B builds: bash tools/pmc_variants.sh <tag> name1 name2 ... (base = in-tree)
set -u
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export STAMPS=0
for v in "$@"; do
  if [ $v = base ]; then unset PAL_LIB_PATH; else export PAL_LIB_PATH=$ROOT/tools/bin/libpal_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/$v -o p --output-format csv -- python3 $ROOT/tools/stamps_fin.py > $OUT/$v.log 2>&1 || echo "$v failed"
done
python3 - <<PY
import csv, glob, collections
for v in "$*".split():
    agg = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % v, recursive=True):
        for row in csv.DictReader(open(path)):
            if "k_pfa_cols_fin" not in row["Kernel_Name"] or int(row.get("Grid_Size", "0")) < 200000: continue
            c = agg[row["Counter_Name"]]; c[0] += float(row["Counter_Value"]); c[1] += 1
    w = agg["SQ_WAVES"][0] or 1
    print(v, {k: round(x[0] / w, 1) for k, x in sorted(agg.items())})
PY
