#!/bin/bash
# per-dispatch kernel durations of the last frame of a short bench run (used through gpurun): tools/trace_frame.sh <scene>
sc=${1:-cube}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_$sc -o t --output-format csv -- python3 $R/bench.py --scene $sc --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/trace_$sc.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/trace_$sc/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "rtamd" in r["Kernel_Name"]]
# find the last k_resolve and print the dispatches of the frame that ends there
ends = [i for i, r in enumerate(rows) if "k_resolve" in r["Kernel_Name"]]
i1 = ends[-1]; i0 = ends[-2] + 1
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%7.1f us  %s" % ((e - s) / 1e3, (s - t0) / 1e3, r["Kernel_Name"].split("(")[0][:60]))
PY
