#!/bin/bash
# cfg4 bench under the env given on the command line: tools/gpu_wavy.sh label [VAR=val ...]
label=$1; shift
env "$@" timeout -k 10 400 python bench.py --scene wavy --width 3840 --height 2160 --grid 16 --depth 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bw.json 2>gpurun_out/bw.err || tail -3 gpurun_out/bw.err
python -c "
import json; d=json.load(open('gpurun_out/bw.json')); print('$label', d['ms_per_step'], d['roofline']['ms_per_frame']['instrumented_frame'])"
