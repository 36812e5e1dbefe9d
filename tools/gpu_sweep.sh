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
for b in 500 1000 1500 2000 3000; do
  run b$b RT_STAGED_TRACE=1 RT_TRACE_BUDGET=$b RT_SHADOW_BUDGET=$b
done
run b1500_t750 RT_STAGED_TRACE=1 RT_TRACE_BUDGET=1500 RT_SHADOW_BUDGET=1500 RT_TASK_TARGET=750
run b3000_t1000 RT_STAGED_TRACE=1 RT_TRACE_BUDGET=3000 RT_SHADOW_BUDGET=3000 RT_TASK_TARGET=1000
run b1000_t2000 RT_STAGED_TRACE=1 RT_TRACE_BUDGET=1000 RT_SHADOW_BUDGET=1000 RT_TASK_TARGET=2000
run fused_sh1500 RT_STAGED_TRACE=0 RT_SHADOW_BUDGET=1500
