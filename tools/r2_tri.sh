#!/bin/bash
# triangle-level shaft culling: parity subset, work counters and variants
mkdir -p gpurun_out/tri
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/tri/pytest.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -5 gpurun_out/tri/pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
for sc in dodge wavy; do
  extra=""; steps=50
  if [ $sc = wavy ]; then extra="--width 3840 --height 2160 --grid 16 --depth 8"; steps=5; fi
  timeout -k 10 300 python bench.py --scene $sc --steps $steps --warmup 3 --no-cpu-baseline --no-tree-scenes $extra > gpurun_out/tri/bench_$sc.json 2> gpurun_out/tri/bench_$sc.err || { tail -5 gpurun_out/tri/bench_$sc.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/tri/bench_$sc.json')); r=d['roofline']; print('$sc', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', r['ms_per_frame']['instrumented_frame'], r['frac'], r['work']['shadow'])"
done
bash tools/r2_variants.sh "$@"
