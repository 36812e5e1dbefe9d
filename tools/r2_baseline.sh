#!/bin/bash
# round-2 baseline: bench (cube, dodge) + RT_PROFILE step counters for dodge/cube
mkdir -p gpurun_out/r2base
for sc in cube dodge; do
  timeout -k 10 200 python bench.py --scene $sc --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2base/bench_$sc.json 2> gpurun_out/r2base/bench_$sc.err || exit 1
done
timeout -k 10 120 python tools/prof.py dodgeColorTest.obj 1920 1080 8 4 > gpurun_out/r2base/prof_dodge.log 2>&1 || exit 1
timeout -k 10 120 python tools/prof.py cube.obj 1920 1080 8 4 > gpurun_out/r2base/prof_cube.log 2>&1 || exit 1
tail -n 12 gpurun_out/r2base/prof_dodge.log
