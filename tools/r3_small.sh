#!/bin/bash
# small-frame benches (and the tests that cover the one-launch kernels), option A/B
out=gpurun_out/r3; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_random_scenes.py -x -q --timeout 300 --timeout-method thread > $out/pytest_small.log 2>&1 || { tail -20 $out/pytest_small.log; exit 1; }
tail -1 $out/pytest_small.log
for w in ${@:-teapot3 balls refract glass gears}; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $out/b_$w.json 2> $out/b_$w.err || { tail -5 $out/b_$w.err; exit 1; }
  python - "$w" $out/b_$w.json <<'PY'
import json,sys
w,p=sys.argv[1],sys.argv[2]
j=json.loads(open(p).read().strip().splitlines()[-1])
print("%-14s %9.1f Mrays/s  %9.3f ms/step  pinned frame: %s" % (w, j["value"], j["ms_per_step"], j["frame_matches_pinned"]))
PY
done
