#!/bin/bash
# light-camera tests (both device builders, host builder), then the set-up workload
out=gpurun_out/r3; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_random_scenes.py tests/test_gpu_edge_cases.py -x -q -k "light or glass or camera or pairs" --timeout 300 --timeout-method thread > $out/pytest_lc.log 2>&1
rc=$?
tail -3 $out/pytest_lc.log
[ $rc -ne 0 ] && { grep -E "^E|Error|assert|rt\]" $out/pytest_lc.log | head -30; exit $rc; }
timeout -k 10 120 python3 -u - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, time
sys.path.insert(0, ".")
import raytracer_amd as rt
s = rt.Scene.load("glass"); g = rt.Renderer(0)
g.upload(s); g.render(s.camera())
for k in range(3):
    t0 = time.perf_counter(); g.upload(s); t1 = time.perf_counter(); r = g.render(s.camera()); t2 = time.perf_counter()
    print("upload %.2f ms, frame call %.2f ms (gpu %.2f ms)" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), r["stats"]["gpu_ms"]))
g.set_option("verbose", 1)
g.upload(s); g.render(s.camera())
g.set_option("lightcam_levels", 1)
g.upload(s); g.render(s.camera())
PY
timeout -k 10 300 python bench.py --workload glass_setup --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('glass_setup', j['ms_per_step'], 'ms/step')"
