#!/bin/bash
# Counter passes for the hot kernels (run on the GPU box through gpurun).  One rocprofv3 run per
# counter group, kernel-trace only (no sys/hip traces), outputs under gpurun_out/pmc/<group>/.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name --output-format csv -- $CMD > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE
ls -R $OUT | head -40
