#!/bin/bash
# One GPU iteration: parity tests, then the benchmark workloads in brief form.
# usage: tools/gpu_iter.sh [tag] [workloads...]
tag=${1:-iter}; shift
wl=${@:-teapot3 balls refract teapot3_8k16}
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -4 gpurun_out/pytest_$tag.log
if [ $rc -ne 0 ]; then grep -E "^E|Error|assert" gpurun_out/pytest_$tag.log | tail -20; exit $rc; fi
for w in $wl; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_${tag}_$w.json 2> gpurun_out/bench_${tag}_$w.err || { tail -5 gpurun_out/bench_${tag}_$w.err; exit 1; }
  python - "$w" gpurun_out/bench_${tag}_$w.json <<'PY'
import json,sys
w,p=sys.argv[1],sys.argv[2]
j=json.loads(open(p).read().strip().splitlines()[-1])
print("%-14s %9.1f Mrays/s  %9.3f ms/step  rays/frame %d" % (w, j["value"], j["ms_per_step"], j["config"]["rays_per_frame"]))
PY
done
