#!/bin/bash
# cube bench under env variants: each arg is "NAME=VALUE"
mkdir -p gpurun_out/var
for v in "$@"; do
  env $v timeout -k 10 300 python bench.py --scene cube --steps 100 --warmup 5 --no-cpu-baseline --no-tree-scenes --no-work-counters > gpurun_out/var/cube_$v.json 2> gpurun_out/var/cube_$v.err || { tail -3 gpurun_out/var/cube_$v.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/var/cube_$v.json')); print('cube $v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['roofline']['ms_per_frame']['instrumented_frame'])"
done
