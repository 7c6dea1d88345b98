#!/bin/bash
# First GPU call of the round: the whole GPU suite, the poisoned-ring test against a build without the repair (must fail), baseline bench lines.
out=gpurun_out/r3
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest_first.log 2>&1
rc=$?
tail -5 $out/pytest_first.log
[ $rc -ne 0 ] && { grep -E "^E|Error|assert" $out/pytest_first.log | tail -20; exit $rc; }
RAYTRACER_AMD_LIB=build_variants/lib_stale_ring.so timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 --timeout-method thread -k stale_ring > $out/pytest_stale_variant.log 2>&1
echo "stale-ring variant exit code (1 expected: the test must fail without the repair): $?"
tail -3 $out/pytest_stale_variant.log
for w in teapot3_8k16 teapot3 balls refract glass; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $out/base_$w.json 2> $out/base_$w.err || { tail -5 $out/base_$w.err; exit 1; }
  python - "$w" $out/base_$w.json <<'PY'
import json,sys
w,p=sys.argv[1],sys.argv[2]
j=json.loads(open(p).read().strip().splitlines()[-1])
print("%-14s %9.1f Mrays/s  %9.3f ms/step  %s" % (w, j["value"], j["ms_per_step"], {k: round(v, 2) for k, v in j["roofline"]["kernel_ms_per_frame"].items()}))
PY
done
