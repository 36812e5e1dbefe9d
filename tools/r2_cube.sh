#!/bin/bash
# cube bench for lib variants: tools/r2_cube.sh "" _w5 ...
mkdir -p gpurun_out/var
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/raytracer-in-cpp_amd/lib/librt_mi355x$v.so
  RT_LIB=$lib timeout -k 10 300 python bench.py --scene cube --steps 100 --warmup 5 --no-cpu-baseline --no-tree-scenes --no-work-counters > gpurun_out/var/cube$v.json 2> gpurun_out/var/cube$v.err || { tail -3 gpurun_out/var/cube$v.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/var/cube$v.json')); print('cube$v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['roofline']['ms_per_frame']['instrumented_frame'])"
done
