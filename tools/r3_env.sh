#!/bin/bash
# dodge / wavy trace-stage sweeps over environment switches: tools/r3_env.sh <scene> "<VAR=val ...>" ...   (one bench run per quoted setting)
sc=$1; shift
R=$GRAFT_REPO_ROOT
extra=""; steps=40
if [ $sc = wavy ]; then extra="--width 3840 --height 2160 --grid 16 --depth 8"; steps=4; fi
for setting in "$@"; do
  ( for kv in $setting; do export $kv; done
    python3 $R/bench.py --scene $sc --steps $steps --warmup 5 --no-cpu-baseline --no-tree-scenes --no-work-counters $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; f=r.get('ms_per_frame',{})
print('%-60s ms/frame %.4f  instrumented %s' % ('$setting', d['ms_per_step'], f.get('instrumented_frame')))
" )
done
