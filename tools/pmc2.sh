#!/bin/bash
# SQ stall/instruction-mix counters of the bench kernels, two PMC passes (own runs, --pmc only): tools/pmc2.sh <scene> [bench args]
sc=${1:-dodge}; shift 1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc2_$sc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS -d $R/gpurun_out/pmc2_$sc/a -o a --output-format csv -- python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --no-tree-scenes --no-work-counters "$@" > $R/gpurun_out/pmc2_$sc/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc2_$sc/b -o b --output-format csv -- python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --no-tree-scenes --no-work-counters "$@" > $R/gpurun_out/pmc2_$sc/b.log 2>&1
python3 - <<PY
import csv, collections, glob, json
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc2_$sc/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rtamd" in k: per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, d in per.items():
    out[k] = {c: max(v) for c, v in d.items()}
    print(k, {c: "%.3g" % x for c, x in out[k].items()})
json.dump(out, open("$R/gpurun_out/pmc2_$sc/pmc2_$sc.json", "w"), indent=1)
PY
