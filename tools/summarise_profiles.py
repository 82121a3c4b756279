#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>_* (rocprofv3 CSVs) into the small, committed files under profiles/:
   <tag>_kernel_stats.csv   the --stats table, FrAD kernels only (+ everything else summed)
   <tag>_counters.json      per-kernel means of the PMC passes, HBM traffic with the gfx950 FETCH_SIZE x2
                            correction of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B requests at 64 B)
Usage: python tools/summarise_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    g = glob.glob(os.path.join(src, pattern))
    return max(g, key=os.path.getmtime) if g else None      # gpurun_out/ accumulates runs: take the newest


stats = one(f"prof_{tag}_trace/*/*_kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if "frad::" in r["Name"]]
    other = [r for r in rows if "frad::" not in r["Name"]]
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in keep:
            w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
        w.writerow(["(all other kernels: torch workload synthesis, overflow flag, memsets)", sum(int(r["Calls"]) for r in other),
                    sum(int(r["TotalDurationNs"]) for r in other), "", round(sum(float(r["Percentage"]) for r in other), 2), "", "", ""])

counters = {}
for leg in ("fetch", "write", "sq"):
    f = one(f"prof_{tag}_{leg}/*/*_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(f)):
        if "frad::" not in r["Kernel_Name"]:
            continue
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[r["Kernel_Name"]] = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
    for k, d in agg.items():
        e = counters.setdefault(k, {"launch": meta[k]})
        for c, v in d.items():
            e[c] = sum(v) / len(v)
for k, e in counters.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        # units: KiB per dispatch; FETCH_SIZE under-counts wide coalesced reads by exactly 2x on gfx950
        e["hbm_bytes_per_launch_corrected"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
        e["hbm_bytes_per_launch_raw"] = int((e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
# which build these counters belong to: bench.py only quotes `roofline.traffic` from a file whose kernel sources match
import hashlib
import subprocess
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(root, "frad_python_amd", "csrc", "*.h*")) + glob.glob(os.path.join(root, "frad_python_amd", "csrc", "*.inc"))):
    h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
try:
    git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
except Exception:
    git = None
counters["_meta"] = {"tag": tag, "csrc_sha": h.hexdigest()[:16], "git": git}
json.dump(counters, open(os.path.join(dst, f"{tag}_counters.json"), "w"), indent=1)
print("wrote", [f for f in os.listdir(dst) if f.startswith(tag)])
