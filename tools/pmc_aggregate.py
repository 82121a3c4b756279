#!/usr/bin/env python3
"""Aggregate the PMC passes of tools/pmc_any.sh: gpurun_out/pmc_<tag>_{a,b,f,w}/ -> gpurun_out/pmc_<tag>.json, per FrAD kernel the mean of
every counter per launch, HBM bytes with the gfx950 FETCH_SIZE x2 correction (MI355X_MICROARCH.md), issue and LDS-conflict ratios."""
import collections, csv, glob, json, os, sys

def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void frad::", "").replace("frad::", "")
    return n.split("(")[0]

def main(root, tag):
    out = {}
    for leg in "abfw":
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        files = glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}_{leg}", "*", "*_counter_collection.csv"))
        for f in sorted(files, key=os.path.getmtime)[-1:]:      # gpurun_out/ accumulates runs: the newest one only
            for r in csv.DictReader(open(f)):
                if "frad::" in r["Kernel_Name"]:
                    k = short(r["Kernel_Name"])
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta[k] = {x: r[x] for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Grid_Size", "Workgroup_Size") if x in r}
        for k, d in agg.items():
            e = out.setdefault(k, {"launch": meta[k]})
            for c, v in d.items():
                e[c] = round(sum(v) / len(v), 1); e["launches_" + leg] = len(v)
    for k, e in out.items():
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:     # KiB per dispatch; FETCH_SIZE x2 on gfx950
            e["hbm_bytes_per_launch_corrected"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
        if e.get("SQ_WAVE_CYCLES"):
            e["issue_frac_per_wave"] = round(e.get("SQ_ACTIVE_INST_ANY", 0) / e["SQ_WAVE_CYCLES"], 3)
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"], 3)
    json.dump(out, open(os.path.join(root, "gpurun_out", f"pmc_{tag}.json"), "w"), indent=1)
    for k, e in out.items():
        print(k[:60], {c: e[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "issue_frac_per_wave", "lds_conflict_frac", "hbm_bytes_per_launch_corrected") if c in e})

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
