#!/bin/bash
# per-kernel durations of one scene's bench under rocprofv3: tools/kstats.sh <scene> [bench args]
sc=${1:-dodge}; shift 1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/kstats
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ks_$sc -o $sc --output-format csv -- python3 $R/bench.py --scene $sc --steps 20 --warmup 3 --no-cpu-baseline --no-tree-scenes --no-work-counters "$@" > $R/gpurun_out/kstats/$sc.json 2> $R/gpurun_out/kstats/$sc.err || { tail -3 $R/gpurun_out/kstats/$sc.err; exit 1; }
cp /tmp/ks_$sc/${sc}_kernel_stats.csv $R/gpurun_out/kstats/${sc}_kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/kstats/${sc}_kernel_stats.csv")):
    n = r["Name"].split("(")[0].replace("void ", "")
    if "rtamd" in n and float(r["TotalDurationNs"]) > 2e5: print("  %-45s calls %5s avg %9.1f us min %8.1f max %9.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
