#!/bin/bash
# HBM traffic of the bench kernels from the TCC PMC counters, separate passes (MI355X_MICROARCH.md §HBM / rocprofv3 PMC slots):
#   pass 1: --pmc FETCH_SIZE    pass 2: --pmc WRITE_SIZE      (units: KiB; FETCH_SIZE under-reports wide 16-B/lane reads by 2x on gfx950)
# usage: tools/traffic.sh <scene> <tag> [extra bench args]
sc=${1:-cube}; tag=${2:-traffic}; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$tag
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/$tag/f -o f --output-format csv -- python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --no-tree-scenes --no-work-counters "$@" > $R/gpurun_out/$tag/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/$tag/w -o w --output-format csv -- python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --no-tree-scenes --no-work-counters "$@" > $R/gpurun_out/$tag/w.log 2>&1
python3 - <<PY
import csv, collections, glob, json
out = {}
for part, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    for f in glob.glob("$R/gpurun_out/$tag/%s/*counter_collection.csv" % part):
        per = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != name: continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            per[k].append(float(row["Counter_Value"]))
        for k, v in per.items():
            if "rtamd" not in k: continue
            v.sort()
            out.setdefault(k, {})[name + "_KiB_max_launch"] = v[-1]
            out[k][name + "_KiB_sum"] = sum(v)
            out[k][name + "_launches"] = len(v)
print(json.dumps(out, indent=1))
json.dump(out, open("$R/gpurun_out/$tag/traffic_$sc.json", "w"), indent=1)
PY
