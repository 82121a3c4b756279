#!/bin/bash
# PMC passes (SQ issue / wait / LDS counters, then FETCH_SIZE and WRITE_SIZE each in a pass of its own) over ANY python
# workload of this repository, aggregated per FrAD kernel:
#   gpurun --timeout 600 -- 'bash tools/pmc_any.sh p1 tools/p1_probe.py'
# -> gpurun_out/pmc_<tag>.json {kernel: {counter: mean per launch, ..., hbm_bytes_per_launch_corrected}}
# Only --pmc + --kernel-trace are combined (the pool refuses --pmc with the runtime / sys traces).
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SCRIPT=$R/$1; shift
run() {  # leg name, counters...
    local leg=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_${leg} -- python3 $SCRIPT $ARGS > $R/gpurun_out/pmc_${TAG}_${leg}.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_${leg}.log; exit 1; }
}
ARGS="$*"
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
run f FETCH_SIZE
run w WRITE_SIZE
python3 - <<PY
import csv, glob, collections, json
out = {}
for leg in "abfw":
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob("$R/gpurun_out/pmc_${TAG}_%s/*/*_counter_collection.csv" % leg):
        for r in csv.DictReader(open(f)):
            if "frad::" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[r["Kernel_Name"]] = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Grid_Size", "Workgroup_Size") if k in r}
    for k, d in agg.items():
        e = out.setdefault(k.replace("void frad::", "").split("(")[0], {"launch": meta[k]})
        for c, v in d.items():
            e[c] = round(sum(v) / len(v), 1); e["launches_" + leg] = len(v)
for k, e in out.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:     # KiB per dispatch; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md)
        e["hbm_bytes_per_launch_corrected"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
    if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
        e["issue_frac_per_wave"] = round(e.get("SQ_ACTIVE_INST_ANY", 0) / e["SQ_WAVE_CYCLES"], 3)
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_frac"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"], 3)
json.dump(out, open("$R/gpurun_out/pmc_${TAG}.json", "w"), indent=1)
for k, e in out.items():
    print(k[:60], {c: e[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "issue_frac_per_wave", "lds_conflict_frac", "hbm_bytes_per_launch_corrected") if c in e})
PY
