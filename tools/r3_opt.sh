#!/bin/bash
out=gpurun_out/r3; mkdir -p $out
timeout -k 10 600 python -u tools/exp_opt.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a $out/opt.txt
