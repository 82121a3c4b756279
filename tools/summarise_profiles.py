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

# timed-steps-only durations of the two headline kernels from the kernel trace (bench.py: 5 cold-start steps, then the
# pre-warm and warm-up steps, then the K timed ones, then the cold-rotation and yardstick sections)
trace = one(f"prof_{tag}_trace/*/*_kernel_trace.csv")
timed = {}
try:
    line = json.loads(open(os.path.join(src, f"prof_{tag}_bench_line.json")).read())
    before = 5 + line["config"]["prewarm_steps"] + line["warmup"]
    steps = line["steps"]
except Exception:
    line, before, steps = None, None, None
if trace and before is not None:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "k_p0_fwd_wave" in r["Kernel_Name"] or "k_p0_inv_wave" in r["Kernel_Name"]:
            per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, v in per.items():
        v.sort()
        d = [x[1] for x in v[before:before + steps]]
        if d:
            timed[k] = {"launches": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d), "skipped_before": before}
if stats and timed:
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "a", newline="") as f:
        w = csv.writer(f)
        w.writerow([])
        w.writerow(["# the same kernels over the K timed steps only (launches sorted by start time; cold-start, pre-warm and warm-up launches skipped)"])
        w.writerow(["Name", "Launches", "AverageNs", "MinNs", "MaxNs", "LaunchesSkippedBefore"])
        for k, t in timed.items():
            w.writerow([k, t["launches"], round(t["avg_ns"], 1), t["min_ns"], t["max_ns"], t["skipped_before"]])

counters = {}
for part in ("bench", "other"):
    f = os.path.join(src, f"pmc_{tag}_{part}.json")
    if os.path.exists(f):
        for k, e in json.load(open(f)).items():
            e["workload"] = "bench.py --steps 3 --warmup 1" if part == "bench" else "tools/pmc_workload.py"
            counters[k] = e
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
