#!/bin/bash
# A/B of the XCD-aware workgroup order of the row passes (PAL_XCD_ROWS): parity, throughput, fetch traffic of the row pass.
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02_xcd
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pfa.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for x in 1 0 1 0; do
  PAL_XCD_ROWS=$x python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_x$x.json 2>> $OUT/bench.err
  python3 -c "import json;d=json.load(open('$OUT/bench_x$x.json'));print('xcd=$x',d['value'],d['kernels_alone_us'])"
done
for cfg in c3 c5; do for x in 1 0; do
  PAL_XCD_ROWS=$x python3 $ROOT/bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_${cfg}_x$x.json 2>> $OUT/bench.err
  python3 -c "import json;d=json.load(open('$OUT/bench_${cfg}_x$x.json'));print('$cfg xcd=$x',d['value'],d['kernels_alone_us'])"
done; done
PMC_CMD="python3 $ROOT/bench.py --steps 2 --warmup 1 --frames 16 --no-cpu-baseline --no-kernel-events"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc/$name -o $name --output-format csv -- $PMC_CMD > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT/pmc $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_traffic.json
find $OUT -name "*.db" -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
