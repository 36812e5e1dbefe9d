#!/bin/bash
# PMC passes (separate from kernel-trace, per the pool's rules): usage tools/pmc.sh <scene> <tag>
sc=${1:-dodge}; tag=${2:-pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/$tag/a -o a --output-format csv -- python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$tag/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD -d $R/gpurun_out/$tag/b -o b --output-format csv -- python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$tag/b.log 2>&1
python3 - <<PY
import csv, collections, glob
for part in ("a","b"):
    for f in glob.glob("$R/gpurun_out/$tag/%s/*counter_collection.csv" % part):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:40]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        for k, d in agg.items():
            if "rtamd" in k:
                print(part, k, {c: round(v) for c, v in d.items()})
                if "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
                    cyc = d["GRBM_GUI_ACTIVE"] / 8.0           # rocprofv3 sums the 8 XCDs
                    # one wave64 VALU instruction occupies its SIMD for one quad-cycle; 256 CUs x 4 SIMDs
                    print("   VALU utilisation %.3f  SALU/VALU %.2f  mean resident waves/SIMD %.2f" % (
                        d["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc), d["SQ_INSTS_SALU"] / max(d["SQ_INSTS_VALU"], 1.0),
                        d["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * cyc)))
PY
