#!/bin/bash
# usage: tools/exp_variants.sh <workload> <variant names...>   (build_variants/lib_<name>.so; "base" = product library)
w=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset RAYTRACER_AMD_LIB; else export RAYTRACER_AMD_LIB=$PWD/build_variants/lib_$v.so; fi
  timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err || { echo "$v failed"; tail -3 gpurun_out/var_$v.err; exit 1; }
  python - $v gpurun_out/var_$v.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-8s %9.1f Mrays/s %9.3f ms" % (sys.argv[1], j["value"], j["ms_per_step"]))
PY
done
