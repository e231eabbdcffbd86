#!/usr/bin/env python3
"""Condense a tools/r2.sh output directory (gpurun_out/<tag>/) into profiles/<name>/:
kernel_stats.csv (rocprofv3 --kernel-trace --stats), pmc_per_launch.json (averages per launch, separate --pmc passes)
and the HBM-traffic figure bench.py reports as roofline.traffic (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; the
factor 2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md §HBM for wide streaming reads).

    python tools/prof_summary.py gpurun_out/r2i agg profiles/r02_agg
    python tools/prof_summary.py gpurun_out/r2i emb profiles/r02_emb
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

# the files bench.py reads for roofline.traffic
prof = os.path.dirname(os.path.abspath(dst))
rel = os.path.relpath(os.path.join(dst, "pmc_per_launch.json"), os.path.dirname(prof))
how = "(2*FETCH_SIZE + WRITE_SIZE) KiB, separate rocprofv3 --pmc passes (tools/r2.sh pmc_%s)" % which
if which == "agg":
    for fname, pref, alg in (("pmc_k_query_attend.json", "k_query_attend_split", 1337271040),
                             ("pmc_k_attend_f2.json", "k_attend_f2", 1337271040),
                             ("pmc_k_attend_f3.json", "k_attend_f3", 1337271040),
                             ("pmc_k_attend_bf16_res.json", "k_attend_bf16_res", 676774912),
                             ("pmc_k_query_attend_bf16.json", "k_query_attend_bf16", 676774912)):
        ks = [k for k in out if k.startswith(pref) and "hbm_bytes_per_launch" in out[k]]
        if ks:
            k = max(ks, key=lambda n: out[n].get("launches_sampled", 0))
            json.dump({"kernel": k, "hbm_bytes_per_launch": out[k]["hbm_bytes_per_launch"],
                       "mfma_util": out[k].get("mfma_util"), "source": rel + ": " + how,
                       "algorithmic_bytes_per_launch": alg}, open(os.path.join(prof, fname), "w"), indent=1)
if which == "emb" and stats:
    calls = {short(r[0]): int(r[1]) for r in rows[1:]}
    fwd = max([c for k, c in calls.items() if k.startswith("k_stem")] or [1])
    tot, n, names = 0, 0, []
    for k, v in out.items():
        if (k.startswith("k_conv_wino") or k.startswith("k_conv_s6") or k.startswith("k_conv<")) and "hbm_bytes_per_launch" in v:
            per_fwd = calls.get(k, 0) / fwd
            tot += v["hbm_bytes_per_launch"] * per_fwd
            n += per_fwd
            names.append("%s x %g" % (k, per_fwd))
    json.dump({"kernel": "conv kernels of one forward (bs = 256): " + "; ".join(names), "hbm_bytes_per_forward": int(tot),
               "launches_per_forward": n, "source": rel + ": sum over the conv launches of one forward of " + how},
              open(os.path.join(prof, "pmc_k_conv.json"), "w"), indent=1)
