#!/bin/bash
# bench lib variants: RT_LIB=<so> for each given suffix
mkdir -p gpurun_out/var
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/raytracer-in-cpp_amd/lib/librt_mi355x$v.so
  for sc in dodge wavy; do
    extra=""; steps=50
    if [ $sc = wavy ]; then extra="--width 3840 --height 2160 --grid 16 --depth 8"; steps=5; fi
    RT_LIB=$lib timeout -k 10 300 python bench.py --scene $sc --steps $steps --warmup 3 --no-cpu-baseline --no-tree-scenes --no-work-counters $extra > gpurun_out/var/bench_${sc}$v.json 2> gpurun_out/var/bench_${sc}$v.err || { tail -3 gpurun_out/var/bench_${sc}$v.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/var/bench_${sc}$v.json')); print('$sc$v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['roofline']['ms_per_frame']['instrumented_frame'])"
  done
done
