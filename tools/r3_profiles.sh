#!/bin/bash
# round-3 profile set (run through gpurun): rocprofv3 kernel stats + PMC passes (VALU, traffic) for cube / dodge / wavy + the default bench line
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
for sc in cube dodge wavy; do
  extra=""; steps=20
  if [ $sc = wavy ]; then extra="--width 3840 --height 2160 --grid 16 --depth 8"; steps=3; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$sc -o $sc --output-format csv -- python3 $R/bench.py --scene $sc --steps $steps --warmup 3 --no-cpu-baseline --no-tree-scenes --no-work-counters $extra > $R/gpurun_out/profiles/${tag}_bench_${sc}_under_rocprof.json 2> $R/gpurun_out/prof_$sc.err || { tail -3 $R/gpurun_out/prof_$sc.err; exit 1; }
  cp $R/gpurun_out/prof_$sc/${sc}_kernel_stats.csv $R/gpurun_out/profiles/${tag}_${sc}_kernel_stats.csv
  echo "kernel stats $sc done"
done
cd $R
for sc in cube dodge; do bash tools/valu.sh $sc > gpurun_out/valu_$sc.txt 2>&1 || exit 1; bash tools/traffic.sh $sc traffic_$sc > gpurun_out/traffic_$sc.txt 2>&1 || exit 1; echo "pmc $sc done"; done
bash tools/valu.sh wavy --width 3840 --height 2160 --grid 16 --depth 8 > gpurun_out/valu_wavy.txt 2>&1 || exit 1
bash tools/traffic.sh wavy traffic_wavy --width 3840 --height 2160 --grid 16 --depth 8 > gpurun_out/traffic_wavy.txt 2>&1 || exit 1
echo "pmc wavy done"
