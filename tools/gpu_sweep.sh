#!/bin/bash
# parity subset, then dodge bench under a few leaf-task settings (used through gpurun)
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cube_256 or dodge_matches or counters or shards or material or culling or deep_tree or hipgraph" > gpurun_out/pytest_quick.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -3 gpurun_out/pytest_quick.log
if [ $rc -ne 0 ]; then echo "parity failed: not running the bench"; exit 1; fi
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --scene dodge --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$label.json 2> gpurun_out/bench_$label.err || tail -5 gpurun_out/bench_$label.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$label.json')); print('$label', d['ms_per_step'], 'ms', d['roofline']['ms_per_frame']['instrumented_frame'])"
}
for b in 0 750 1500 3000 6000; do
  run sb$b RT_SHADOW_BUDGET=$b
done
for b in 0 500 2000 4000; do
  run tb$b RT_TRACE_BUDGET=$b
done
run fused RT_STAGED_TRACE=0
