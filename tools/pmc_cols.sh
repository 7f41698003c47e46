#!/bin/bash
# instruction / cycle counters of the column pass, finishing (PAL_FIN=1) against stored rows (PAL_FIN=0): bash tools/pmc_cols.sh <tag>
set -u
TAG=${1:-pmc_cols}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export STAMPS=0
for fin in ${FINS:-1 0}; do
  export PAL_FIN=$fin
  for pass in "a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "c SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F64 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU"; do
    set -- $pass; name=$1; shift
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/fin$fin/$name -o $name --output-format csv -- python3 $ROOT/tools/stamps_fin.py > $OUT/fin${fin}_$name.log 2>&1 || echo "pass $fin $name failed"
  done
done
python3 - <<PY
import csv, glob, collections
for fin in [int(v) for v in "${FINS:-1 0}".split()]:
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob("$OUT/fin%d/*/*counter_collection.csv" % fin):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if "k_pfa_cols" not in k and "k_peak_finish" not in k and "k_pfa_rows" not in k: continue
            if int(row.get("Grid_Size", "0")) < 200000 and "finish" not in k: continue      # full launch groups only
            k = k.split("(")[0][-60:]
            c = agg[k][row["Counter_Name"]]; c[0] += float(row["Counter_Value"]); c[1] += 1
    print("== PAL_FIN=%d" % fin)
    for k in agg:
        print(" ", k)
        for name, (v, cnt) in sorted(agg[k].items()):
            print("     %-26s per launch %14.0f   (%d launches)" % (name, v / cnt, cnt))
PY
find $OUT -name "*.db" -delete
