#!/bin/bash
# counter passes of one workload: tools/r3_pmc.sh <tag> <workload>
tag=$1; wl=$2
out=gpurun_out/r3; mkdir -p $out
tools/pmc_passes.sh $tag --workload $wl --steps 1 --warmup 1 --no-cpu-baseline > $out/pmc_$tag.log 2>&1 || { tail -5 $out/pmc_$tag.log; exit 1; }
python3 - $tag <<'PY'
import json, sys
j = json.load(open("gpurun_out/pmc_%s.json" % sys.argv[1]))
for k, e in sorted(j.items()):
    if k == "_meta": continue
    c = e["counters"]
    if c.get("SQ_INSTS_VALU", 0) < 1e6: continue
    n = e["dispatches"]
    fp64 = c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_FMA_F64", 0)
    print("%-26s n=%3d VALU/launch %8.1fM (fp64 %4.1f%% trans %4.2f%%) SALU/VALU %.2f LDS/VALU %.3f VMEM rd %6.2fM wr %6.2fM | wait %.2f wait_inst %.2f active %.2f | lanes %.2f | HBM rd %.2f GB wr %.2f GB L2hit %.2f"
          % (k, n, c["SQ_INSTS_VALU"] / n / 1e6, 100 * fp64 / c["SQ_INSTS_VALU"], 100 * c.get("SQ_INSTS_VALU_TRANS_F64", 0) / c["SQ_INSTS_VALU"],
             c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], c["SQ_INSTS_LDS"] / c["SQ_INSTS_VALU"], c["SQ_INSTS_VMEM_RD"] / n / 1e6, c["SQ_INSTS_VMEM_WR"] / n / 1e6,
             c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
             c["SQ_THREAD_CYCLES_VALU"] / 64.0 / c["SQ_ACTIVE_INST_VALU"], e.get("hbm_read_bytes_per_launch", 0) / 1e9, e.get("hbm_write_bytes_per_launch", 0) / 1e9, e.get("l2_hit_rate", 0)))
PY
