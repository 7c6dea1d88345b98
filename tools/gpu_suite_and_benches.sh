#!/bin/bash
# whole GPU suite, then the 8K frame and the other bench workloads in brief
out=gpurun_out/r3; mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1
rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { grep -E "^E|Error|assert|^tests.*(FAILED|ERROR)" $out/pytest.log | head -30; exit $rc; }
timeout -k 10 300 python -u tools/exp_variants.py product 2>&1 | grep -v amdgpu.ids
for w in ${@:-teapot3 balls refract glass gears}; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $out/b_$w.json 2> $out/b_$w.err || { tail -5 $out/b_$w.err; exit 1; }
  python - "$w" $out/b_$w.json <<'PY'
import json,sys
w,p=sys.argv[1],sys.argv[2]
j=json.loads(open(p).read().strip().splitlines()[-1])
print("%-14s %9.1f Mrays/s  %9.3f ms/step  pinned frame: %s" % (w, j["value"], j["ms_per_step"], j["frame_matches_pinned"]))
PY
done
[ -n "$R3_BUILD_EXP" ] && timeout -k 10 400 python -u tools/exp_build.py 2>&1 | grep -v amdgpu.ids
exit 0
