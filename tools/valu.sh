#!/bin/bash
# VALU wave-instructions per launch of the bench kernels (SQ_INSTS_VALU, its own PMC pass): tools/valu.sh <scene> [bench args]
sc=${1:-cube}; shift 1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/valu_$sc
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/valu_$sc/a -o a --output-format csv -- python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --no-tree-scenes --no-work-counters "$@" > $R/gpurun_out/valu_$sc/a.log 2>&1
# second pass: the time share of VALU execution (FP64 runs at half rate: it shows here, not in the instruction count)
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -d $R/gpurun_out/valu_$sc/b -o b --output-format csv -- python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --no-tree-scenes --no-work-counters "$@" > $R/gpurun_out/valu_$sc/b.log 2>&1
python3 - <<PY
import csv, collections, glob, json
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/valu_$sc/a/*counter_collection.csv") + glob.glob("$R/gpurun_out/valu_$sc/b/*counter_collection.csv"):
    second = "/b/" in f
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rtamd" in k: per[k][("B_" if second else "") + row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, d in per.items():
    i = max(range(len(d["SQ_INSTS_VALU"])), key=lambda j: d["SQ_INSTS_VALU"][j])     # the heaviest (level-0) launch
    out[k] = {"valu_wave_instructions": d["SQ_INSTS_VALU"][i], "salu_wave_instructions": d["SQ_INSTS_SALU"][i], "waves": d["SQ_WAVES"][i],
              "gpu_cycles": d["GRBM_GUI_ACTIVE"][i] / 8.0, "launches": len(d["SQ_INSTS_VALU"])}
    if d.get("B_SQ_INSTS_VALU"):
        j = max(range(len(d["B_SQ_INSTS_VALU"])), key=lambda q: d["B_SQ_INSTS_VALU"][q])
        out[k].update({"active_inst_valu": d["B_SQ_ACTIVE_INST_VALU"][j], "busy_cycles": d["B_SQ_BUSY_CYCLES"][j], "wave_cycles": d["B_SQ_WAVE_CYCLES"][j]})
    print(k, out[k])
json.dump(out, open("$R/gpurun_out/valu_$sc/valu_$sc.json", "w"), indent=1)
PY
