#!/bin/bash
# device-built hierarchies: their tests alone, output kept
out=gpurun_out/r3; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_device_build.py -x -q -s --timeout 300 --timeout-method thread > $out/pytest_dev.log 2>&1
rc=$?
tail -5 $out/pytest_dev.log
grep -E "upload of|^E|Error|assert" $out/pytest_dev.log | head -30
exit $rc
