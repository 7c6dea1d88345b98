for L in 4 2 1; do
  export RT_LEAF_MAX=$L
  echo "leaf max $L"
  timeout -k 10 200 python bench.py --workload teapot3_8k16 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])" || exit 1
  timeout -k 10 200 python bench.py --workload balls --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])" || exit 1
done
