#!/bin/bash
# SQ / LDS counters of one kernel-only run: bash tools/prof_sq.sh <tag> [kbench args]
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/sq_${TAG}_a -- python3 $R/tools/kbench.py "$@" > $R/gpurun_out/sq_${TAG}_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/sq_${TAG}_b -- python3 $R/tools/kbench.py "$@" > $R/gpurun_out/sq_${TAG}_b.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
for leg in "ab":
    fs=glob.glob("$R/gpurun_out/sq_${TAG}_%s/*/*_counter_collection.csv"%leg)
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "frad::" in r["Kernel_Name"]: agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in agg.items():
        print(k, {c: round(sum(v)/len(v)) for c,v in d.items()}, "n=",len(next(iter(d.values()))))
PY
