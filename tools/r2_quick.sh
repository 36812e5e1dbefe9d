#!/bin/bash
# quick GPU loop: parity tests + bench on cube/dodge (+ optional wavy cfg4)
mkdir -p gpurun_out/q
timeout -k 10 900 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/q/pytest.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -5 gpurun_out/q/pytest.log
if [ $rc -ne 0 ]; then echo "parity failed: not running the bench"; exit 1; fi
for sc in ${SCENES:-cube dodge}; do
  extra=""; steps=50
  if [ $sc = wavy ]; then extra="--width 3840 --height 2160 --grid 16 --depth 8"; steps=5; fi
  timeout -k 10 300 python bench.py --scene $sc --steps $steps --warmup 3 --no-cpu-baseline $extra > gpurun_out/q/bench_$sc.json 2> gpurun_out/q/bench_$sc.err || { tail -5 gpurun_out/q/bench_$sc.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/q/bench_$sc.json')); print('$sc', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['roofline']['ms_per_frame']['instrumented_frame'])"
done
