#!/bin/bash
# bench dodge+wavy under env variants: each arg is "NAME=VALUE"
mkdir -p gpurun_out/var
for v in "$@"; do
  for sc in ${SCENES:-dodge wavy}; do
    extra=""; steps=50
    if [ $sc = wavy ]; then extra="--width 3840 --height 2160 --grid 16 --depth 8"; steps=5; fi
    env $v timeout -k 10 300 python bench.py --scene $sc --steps $steps --warmup 3 --no-cpu-baseline --no-tree-scenes --no-work-counters $extra > gpurun_out/var/bench_${sc}_$v.json 2> gpurun_out/var/bench_${sc}_$v.err || { tail -3 gpurun_out/var/bench_${sc}_$v.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/var/bench_${sc}_$v.json')); print('$sc $v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['roofline']['ms_per_frame']['instrumented_frame'])"
  done
done
