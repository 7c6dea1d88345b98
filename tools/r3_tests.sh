#!/bin/bash
# The whole GPU suite; on failure the first errors.
out=gpurun_out/r3
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1
rc=$?
tail -5 $out/pytest.log
[ $rc -ne 0 ] && grep -E "^E|Error|assert|^tests.*(FAILED|ERROR)" $out/pytest.log | head -30
exit $rc
