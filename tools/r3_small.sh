#!/bin/bash
out=gpurun_out/r3; mkdir -p $out
for sc in teapot-3 balls refract; do
  EXP_SCENE=$sc EXP_RX=512 EXP_RY=512 EXP_N=1 timeout -k 10 300 python -u tools/exp_opt.py grid_min_samples 4194304 0 2>&1 | grep -v amdgpu.ids | sed "s/^/$sc /"
done
EXP_SCENE=glass EXP_RX=900 EXP_RY=900 EXP_N=1 timeout -k 10 300 python -u tools/exp_opt.py grid_min_samples 4194304 0 2>&1 | grep -v amdgpu.ids | sed "s/^/glass /"
