#!/bin/bash
# A/B per-kernel durations: the product build vs librt_mi355x_ab.so (make ab AB_FLAGS=...), same box, same call.  tools/r3_ab.sh <tag> <scene> [bench args]
tag=${1:-ab}; sc=${2:-dodge}; shift 2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
for v in new ab; do
  if [ $v = ab ]; then export RT_LIB=$R/raytracer-in-cpp_amd/lib/librt_mi355x_ab.so; else unset RT_LIB; fi
  rocprofv3 --kernel-trace --stats -d /tmp/ks_${sc}_$v -o $sc --output-format csv -- python3 $R/bench.py --scene $sc --steps 20 --warmup 3 --no-cpu-baseline --no-tree-scenes --no-work-counters "$@" > $R/gpurun_out/$tag/${sc}_$v.json 2> $R/gpurun_out/$tag/${sc}_$v.err || { tail -3 $R/gpurun_out/$tag/${sc}_$v.err; exit 1; }
  cp /tmp/ks_${sc}_$v/${sc}_kernel_stats.csv $R/gpurun_out/$tag/${sc}_${v}_kernel_stats.csv
  echo "== $sc $v"
  python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/$tag/${sc}_${v}_kernel_stats.csv")):
    n = r["Name"].split("(")[0].replace("void ", "").replace("rtamd::", "")
    if "k_" in n and float(r["TotalDurationNs"]) > 1e5: print("  %-40s calls %5s avg %9.1f us min %8.1f max %9.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
