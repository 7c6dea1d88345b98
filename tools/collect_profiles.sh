#!/bin/bash
# Collects the per-round evidence: bench JSON lines, rocprofv3 kernel stats, counter passes, traversal statistics.
# usage: tools/collect_profiles.sh <round-tag> [part]   (run on the GPU box; copies land in gpurun_out/<tag>/, from where
# the files to be judged are copied into profiles/<tag>/).  Needs build_variants/lib_count.so (tools/build_variant.sh count
# -DRT_COUNT) and build_variants/valu_rates, both built from the same sources as the product library.
# part: "counters" (walk statistics and counter passes of every workload), "bench" (bench lines and kernel traces; reads
# the counters from profiles/<tag>/), or nothing for both.
tag=$1
part=${2:-all}
out=gpurun_out/$tag
mkdir -p $out profiles/$tag
R=$PWD
export TMPDIR=/tmp
SMALL="teapot3 balls refract glass gears"
if [ $part = all ] || [ $part = counters ]; then
  # traversal statistics of this build and the issue rates they are priced with: read by bench.py, so first
  for w in teapot3_8k16 $SMALL; do
    RAYTRACER_AMD_LIB=build_variants/lib_count.so timeout -k 10 200 python tools/walk_stats.py $w --json=$out/walk_$w.json > $out/walk_$w.txt 2>&1 || { echo "walk_stats $w failed"; tail -3 $out/walk_$w.txt; exit 1; }
    echo "walk $w done"
  done
  timeout -k 10 120 build_variants/valu_rates > $out/valu_rates.jsonl || exit 1
  # counters of the default workload (separate passes), then the same for the small frames (more steps: their frames are
  # single launches of a fraction of a millisecond)
  tools/pmc_passes.sh $tag --workload teapot3_8k16 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_log.txt 2>&1 || { tail -5 $out/pmc_log.txt; exit 1; }
  cp gpurun_out/pmc_${tag}_summary.txt $out/pmc_summary_teapot3_8k16.txt
  cp gpurun_out/pmc_${tag}.json $out/pmc_teapot3_8k16.json
  echo "counters teapot3_8k16 done"
  for w in $SMALL; do
    tools/pmc_passes.sh ${tag}_$w --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $out/pmc_log_$w.txt 2>&1 || { tail -5 $out/pmc_log_$w.txt; exit 1; }
    cp gpurun_out/pmc_${tag}_${w}_summary.txt $out/pmc_summary_$w.txt
    cp gpurun_out/pmc_${tag}_$w.json $out/pmc_$w.json
    echo "counters $w done"
  done
  tools/pmc_quick.sh ${tag}q --workload teapot3_8k16 --steps 1 --warmup 0 --no-cpu-baseline > $out/units_teapot3_8k16.txt 2>&1 || { tail -5 $out/units_teapot3_8k16.txt; exit 1; }
  cp $out/pmc_*.json $out/walk_*.json profiles/$tag/
fi
if [ $part = all ] || [ $part = bench ]; then
  # bench lines (they read the counters through profiles/)
  for w in teapot3_8k16 teapot3 balls refract glass glass_setup gears balls2 tetra3; do
    timeout -k 10 400 python bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err || { echo "bench $w failed"; tail -3 $out/bench_$w.err; exit 1; }
    echo "bench $w done"
  done
  cd /tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace_8k16 -o k -- python3 $R/bench.py --workload teapot3_8k16 --steps 3 --warmup 1 --no-cpu-baseline > $R/$out/trace_8k16.out 2>&1 || exit 1
  for w in $SMALL; do
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace_$w -o k -- python3 $R/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > $R/$out/trace_$w.out 2>&1 || exit 1
  done
  cd $R
  cp $out/trace_8k16/k_kernel_stats.csv $out/kernel_stats_teapot3_8k16.csv
  rm -rf $out/trace_8k16
  for w in $SMALL; do cp $out/trace_$w/k_kernel_stats.csv $out/kernel_stats_$w.csv; rm -rf $out/trace_$w; done
fi
ls $out
