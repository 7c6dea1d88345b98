#!/bin/bash
# Clock stamps per wave on the 8K frame from a -DRT_STAMP build (tools/build_variant.sh stamp -DRT_STAMP): proportions only, the stamps slow the passes down.
out=gpurun_out/r3; mkdir -p $out
RAYTRACER_AMD_LIB=build_variants/lib_stamp.so timeout -k 10 300 python -u tools/exp_stamps_big.py "$@" 2>&1 | grep -v amdgpu.ids | tee $out/stamps.txt
