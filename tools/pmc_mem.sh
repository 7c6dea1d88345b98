#!/bin/bash
# Memory-pipeline counters (texture addresser, vector cache, data return) in separate passes.
# usage: tools/pmc_mem.sh <tag> <bench args...>
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
cd /tmp
BARGS=("$@")
run() { # name counters...
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -o p -- python3 $R/bench.py "${BARGS[@]}" > $R/gpurun_out/pmc_${tag}_$name.out 2>&1 || { echo "pass $name failed"; tail -3 $R/gpurun_out/pmc_${tag}_$name.out; return 1; }
}
run m1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES &&
run m2 TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT &&
run m3 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAVES SQ_WAVE_CYCLES
cd $R
python3 tools/pmc_summarize.py $tag
