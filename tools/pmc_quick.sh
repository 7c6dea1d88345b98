#!/bin/bash
# Two counter passes that tell which unit limits the traversal kernels (VALU, vector memory, LDS).
# usage: tools/pmc_quick.sh <tag> <bench args...>
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
cd /tmp
BARGS=("$@")
run() { # name counters...
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -o p -- python3 $R/bench.py "${BARGS[@]}" > $R/gpurun_out/pmc_${tag}_$name.out 2>&1 || { echo "pass $name failed"; tail -3 $R/gpurun_out/pmc_${tag}_$name.out; return 1; }
}
run q1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE &&
run q2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE TA_TOTAL_WAVEFRONTS_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
cd $R
python3 tools/pmc_summarize.py $tag > /dev/null
python3 - $tag <<'PY'
import json, sys
j = json.load(open("gpurun_out/pmc_%s.json" % sys.argv[1]))
for k, e in j.items():
    if k == "_meta":
        continue
    c = e["counters"]
    if "closest" not in k and "shadow" not in k and "lightcam" not in k:
        continue
    gui = c["GRBM_GUI_ACTIVE"] / 2 / 8           # two passes; per XCD
    cu = 256.0
    print("%-24s n=%d  cycles/XCD %.3g | VALU busy %.2f | TA busy %.2f TD busy %.2f (TD waiting %.2f) | VMEM instr/wave-cycle... VMEM_RD %.3g WR %.3g LDS %.3g | LDS active %.2f bank-conflict %.3g"
          % (k, e["dispatches"], gui, c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / gui, c["TA_TA_BUSY_sum"] / cu / gui,
             c["TD_TD_BUSY_sum"] / cu / gui, c["TD_TC_STALL_sum"] / cu / gui, c["SQ_INSTS_VMEM_RD"], c["SQ_INSTS_VMEM_WR"],
             c["SQ_INSTS_LDS"], c["SQ_ACTIVE_INST_LDS"] * 4 / 1024 / gui, c["SQ_LDS_BANK_CONFLICT"]))
PY
