#!/bin/bash
# The ring_poison test on the product and on a build without the repair (tools/build_variant.sh stale_ring -DRT_TEST_STALE_RING): the second must fail.
out=gpurun_out/r3
mkdir -p $out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 --timeout-method thread -k "stale_ring or units_drawn" > $out/pytest_ring.log 2>&1
echo "product build: exit $? (0 expected)"; tail -3 $out/pytest_ring.log
RAYTRACER_AMD_LIB=build_variants/lib_stale_ring.so timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 --timeout-method thread -k stale_ring > $out/pytest_stale_variant.log 2>&1
echo "build without the repair: exit $? (1 expected: the test must fail)"; grep -E "^E|passed|failed" $out/pytest_stale_variant.log | tail -5
