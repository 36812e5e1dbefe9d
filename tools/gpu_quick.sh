#!/bin/bash
# quick GPU iteration loop: a few parity tests + bench on both scenes (used through gpurun)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cube_256 or dodge_matches or counters or shards or material or culling or deep_tree or hipgraph" > gpurun_out/pytest_quick.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -3 gpurun_out/pytest_quick.log
if [ $rc -ne 0 ]; then echo "parity failed: not running the bench"; exit 1; fi
for sc in cube dodge; do
  python bench.py --scene $sc --steps 20 --warmup 3 > gpurun_out/bench_$sc.json 2> gpurun_out/bench_$sc.err || tail -5 gpurun_out/bench_$sc.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$sc.json')); print('$sc', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['roofline']['ms_per_frame']['instrumented_frame'], 'shadow GB/s', d['roofline']['achieved'], 'cpu', d.get('cpu_baseline',{}).get('value'))"
done
