#!/bin/bash
# Runs bench workloads with the product library and with variant libraries, prints ms per frame.
# usage: tools/exp_lib_compare.sh "<variant names>" "<workloads>"
for v in base $1; do
  for w in $2; do
    if [ $v = base ]; then unset RAYTRACER_AMD_LIB; else export RAYTRACER_AMD_LIB=build_variants/lib_$v.so; fi
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/cmp_${v}_$w.json 2> gpurun_out/cmp_${v}_$w.err || { tail -3 gpurun_out/cmp_${v}_$w.err; exit 1; }
    python - $v $w gpurun_out/cmp_${v}_$w.json <<'PY'
import json, sys
v, w, p = sys.argv[1:4]
j = json.loads(open(p).read().strip().splitlines()[-1])
k = j.get("roofline", {}).get("kernel_ms_per_frame", {})
print("%-8s %-14s %9.3f ms/step %9.1f Mrays/s  %s" % (v, w, j["ms_per_step"], j["value"], {a.replace("k_", ""): round(b, 3) for a, b in k.items()}))
PY
  done
done
