#!/bin/bash
# cube headline, per-group times: tools/r3_cube.sh [VAR=val ...]
R=$GRAFT_REPO_ROOT
for kv in "$@"; do export $kv; done
python3 $R/bench.py --scene cube --steps 100 --warmup 10 --no-cpu-baseline --no-tree-scenes --no-work-counters 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; f=r.get('ms_per_frame',{})
print('cube ms/frame %.4f  graph yaw %.4f same-cam %.4f  %s' % (d['ms_per_step'], d.get('graph_replay_ms_per_frame_yaw_path') or -1, d.get('graph_replay_ms_per_frame_same_camera') or -1, f))
"
