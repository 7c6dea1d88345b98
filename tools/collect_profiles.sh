#!/bin/bash
# Collects the per-round evidence: bench JSON lines, rocprofv3 kernel stats, PMC summary.
# usage: tools/collect_profiles.sh <round-tag>   (run on the GPU box; copies land in gpurun_out/<tag>/)
tag=$1
out=gpurun_out/$tag
mkdir -p $out
R=$PWD
export TMPDIR=/tmp
for w in teapot3_8k16 teapot3 balls refract glass glass_setup gears balls2; do
  timeout -k 10 400 python bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err || { echo "bench $w failed"; tail -3 $out/bench_$w.err; exit 1; }
done
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace_8k16 -o k -- python3 $R/bench.py --workload teapot3_8k16 --steps 3 --warmup 1 --no-cpu-baseline > $R/$out/trace_8k16.out 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace_teapot3 -o k -- python3 $R/bench.py --workload teapot3 --steps 50 --warmup 5 --no-cpu-baseline > $R/$out/trace_teapot3.out 2>&1 || exit 1
cd $R
cp $out/trace_8k16/k_kernel_stats.csv $out/kernel_stats_teapot3_8k16.csv
cp $out/trace_teapot3/k_kernel_stats.csv $out/kernel_stats_teapot3.csv
tools/pmc_passes.sh $tag --workload teapot3_8k16 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_log.txt 2>&1 || { tail -5 $out/pmc_log.txt; exit 1; }
cp gpurun_out/pmc_${tag}_summary.txt $out/pmc_summary_teapot3_8k16.txt
cp gpurun_out/pmc_${tag}.json $out/pmc_teapot3_8k16.json
tools/pmc_quick.sh ${tag}q --workload teapot3_8k16 --steps 1 --warmup 0 --no-cpu-baseline > $out/units_teapot3_8k16.txt 2>&1 || { tail -5 $out/units_teapot3_8k16.txt; exit 1; }
rm -rf $out/trace_8k16 $out/trace_teapot3
ls $out
