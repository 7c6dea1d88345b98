#!/bin/bash
out=gpurun_out/r3; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_view_grids.py tests/test_gpu_device_build.py -x -q --timeout 300 --timeout-method thread > $out/pytest_dev.log 2>&1
rc=$?
tail -3 $out/pytest_dev.log
[ $rc -ne 0 ] && { grep -E "^E|Error|assert" $out/pytest_dev.log | head -30; exit $rc; }
timeout -k 10 200 python -u tools/exp_build2.py 400 100000 2>&1 | grep -v amdgpu.ids | tee $out/exp_build2.txt && \
timeout -k 10 500 python -u tools/exp_build.py 2>&1 | grep -v amdgpu.ids | tee $out/exp_build.txt
