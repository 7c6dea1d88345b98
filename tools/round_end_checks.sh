#!/bin/bash
# what the driver runs at the end of a round: the GPU suite, the smoke call, the default bench
out=gpurun_out/final; mkdir -p $out
timeout -k 10 900 python -u -m pytest tests/ -x -q -m gpu --timeout 300 --timeout-method thread > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open("gpurun_out/final/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["unit"], j["ms_per_step"], "ms", "frac", j["roofline"]["frac"], "pinned", j["frame_matches_pinned"], "cpu", j["cpu_baseline"]["value"])
PY
