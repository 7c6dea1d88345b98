#!/bin/bash
# where a scene set-up with light cameras goes: kernel trace of the glass_setup workload, and the upload's own marks
out=gpurun_out/r3; mkdir -p $out
R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace_setup -o k -- python3 $R/bench.py --workload glass_setup --steps 5 --warmup 1 --no-cpu-baseline > $R/$out/trace_setup.out 2>&1 || { tail -5 $R/$out/trace_setup.out; exit 1; }
cd $R
python3 - <<'PY'
import csv, json
rows = list(csv.DictReader(open("gpurun_out/r3/trace_setup/k_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels in all: %.2f ms over 7 frames (one for the timing pass) = %.2f ms per frame" % (tot / 1e6, tot / 7e6))
for r in rows[:22]:
    print("   %-70s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
j = json.loads(open("gpurun_out/r3/trace_setup.out").read().strip().splitlines()[-1])
print("ms per step", j["ms_per_step"])
PY
rm -rf $out/trace_setup
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
PY
