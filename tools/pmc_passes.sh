#!/bin/bash
# Collects PMC counters in separate passes (one rocprofv3 run per pass).
# usage: tools/pmc_passes.sh <tag> <bench args...>
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
cd /tmp
run() { # name counters...
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -o p -- python3 $R/bench.py "${BARGS[@]}" > $R/gpurun_out/pmc_${tag}_$name.out 2>&1 || { echo "pass $name failed"; tail -3 $R/gpurun_out/pmc_${tag}_$name.out; return 1; }
}
BARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT &&
run sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM &&
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE GRBM_GUI_ACTIVE
cd $R
wl=teapot3_8k16; for ((i=0;i<${#BARGS[@]};i++)); do [ "${BARGS[$i]}" = "--workload" ] && wl=${BARGS[$((i+1))]}; done
python3 tools/pmc_summarize.py $tag $wl
