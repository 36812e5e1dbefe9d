#!/bin/bash
# Produces the rocprofv3 --kernel-trace --stats summaries that profiles/ keeps (run through gpurun).
# usage: tools/profile.sh <round-tag>
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/profiles
for sc in cube dodge; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$sc -o $sc --output-format csv -- python3 $R/bench.py --scene $sc --steps 20 --warmup 3 --no-cpu-baseline --no-tree-scenes --no-work-counters > $R/gpurun_out/profiles/${tag}_bench_${sc}_under_rocprof.json 2> $R/gpurun_out/prof_$sc.err
  cp $R/gpurun_out/prof_$sc/${sc}_kernel_stats.csv $R/gpurun_out/profiles/${tag}_${sc}_kernel_stats.csv
done
cd $R
for sc in cube dodge; do
  python bench.py --scene $sc --steps 20 --warmup 3 > gpurun_out/profiles/${tag}_bench_${sc}.json 2>/dev/null
  tail -c 600 gpurun_out/profiles/${tag}_bench_${sc}.json; echo
done
head -5 gpurun_out/profiles/${tag}_dodge_kernel_stats.csv
