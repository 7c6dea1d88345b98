#!/bin/bash
# The view-grid tests on a build that leaves entries out of the lists (tools/build_variant.sh drop -DVG_TEST_DROP=11): they must fail.
out=gpurun_out/r3; mkdir -p $out
RAYTRACER_AMD_LIB=build_variants/lib_drop.so timeout -k 10 600 python -u -m pytest tests/test_gpu_view_grids.py -m gpu -q --timeout 300 --timeout-method thread > $out/pytest_drop.log 2>&1
echo "variant that leaves one bin in eleven out of every list: exit $? (1 expected)"; tail -3 $out/pytest_drop.log
