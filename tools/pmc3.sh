#!/bin/bash
# latency / fetch counters: tools/pmc3.sh <scene>
sc=${1:-dodge}; shift 1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc3_$sc
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS -d $R/gpurun_out/pmc3_$sc/a -o a --output-format csv -- python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --no-tree-scenes --no-work-counters "$@" > $R/gpurun_out/pmc3_$sc/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc3_$sc/b -o b --output-format csv -- python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --no-tree-scenes --no-work-counters "$@" > $R/gpurun_out/pmc3_$sc/b.log 2>&1
python3 - <<PY
import csv, collections, glob
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc3_$sc/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rtamd" in k: per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in per.items():
    o = {c: max(v) for c, v in d.items()}
    def r(a,b): return (o.get(a,0)/o[b]) if o.get(b) else 0
    print(k)
    print("   ", {c: "%.3g" % x for c, x in o.items()})
    print("    avg latency (LEVEL/INSTS): ifetch %.0f vmem %.0f smem %.0f lds %.0f" % (r("SQ_IFETCH_LEVEL","SQ_IFETCH"), r("SQ_INST_LEVEL_VMEM","SQ_INSTS_VMEM_RD"), r("SQ_INST_LEVEL_SMEM","SQ_INSTS_SMEM"), r("SQ_INST_LEVEL_LDS","SQ_INSTS_LDS")))
PY
