#!/usr/bin/env python3
"""Condense a tools_r2.sh output directory (gpurun_out/<tag>/) into profiles/<name>/:
kernel_stats.csv (rocprofv3 --kernel-trace --stats), pmc_per_launch.json (averages per launch, separate --pmc passes)
and the HBM-traffic figure bench.py reports as roofline.traffic (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; the
factor 2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md §HBM for wide streaming reads).

    python tools_prof_summary.py gpurun_out/r2i agg profiles/r02_agg
    python tools_prof_summary.py gpurun_out/r2i emb profiles/r02_emb
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, which, dst = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, f"trace_{which}", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    # keep our kernels only (torch's fill / random kernels have kilobyte-long names)
    rows = list(csv.reader(open(stats[0])))
    with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(rows[0])
        for r in rows[1:]:
            if "at::" in r[0] or "rocclr" in r[0]:
                continue
            wr.writerow([r[0].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]] + r[1:])


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


out = {}
for d in sorted(glob.glob(os.path.join(src, f"pmc_{which}_*", "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        if "at::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]:
            continue
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {}).update({c: round(sum(x) / len(x), 1) for c, x in v.items()})
        out[k]["launches_sampled"] = max(out[k].get("launches_sampled", 0), max(len(x) for x in v.values()))
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch"] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
        v["mfma_util"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    if v.get("SQ_LDS_IDX_ACTIVE"):
        v["lds_conflict_frac"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 4)
json.dump(out, open(os.path.join(dst, "pmc_per_launch.json"), "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(k, {c: v[c] for c in ("hbm_bytes_per_launch", "mfma_util", "lds_conflict_frac") if c in v})
