#!/bin/bash
out=gpurun_out/r3
mkdir -p $out
timeout -k 10 600 python -u -m pytest "$@" -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest_one.log 2>&1
rc=$?
tail -5 $out/pytest_one.log
[ $rc -ne 0 ] && grep -E "^E|Error|assert|^tests.*(FAILED|ERROR)" $out/pytest_one.log | head -30
exit $rc
