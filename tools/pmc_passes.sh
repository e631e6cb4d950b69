#!/bin/bash
# Collect PMC counters for the bench kernels in separate passes (each pass = its own run, counters
# only: no --kernel-trace/--sys-trace combined with --pmc).  Usage: tools/pmc_passes.sh <outdir> [bench args]
set -e
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${@:-"--steps 3 --warmup 1 --no-cpu-baseline --no-secondary"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64
run sq3 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_WR
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
