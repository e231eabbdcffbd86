#!/usr/bin/env python3
"""profiles/rNN_{agg,emb}/ (rocprofv3 --kernel-trace --stats + separate --pmc passes of `bench.py --streams 1`, condensed by
tools/prof_summary.py) -> profiles/rNN_kernel_table.json: per leg, one row per kernel with launches per pass, average ms,
counter HBM bytes (2*FETCH_SIZE + WRITE_SIZE), algorithmic bytes / FLOPs where the kernel has a closed form, MFMA pipe
utilisation (SQ_VALU_MFMA_BUSY_CYCLES) and the fraction of the roofline that bounds it.  bench.py attaches the rows to its
JSON line (`kernels`), so the rooflines can be recomputed without opening profiles/.
    python tools/kernel_table.py r03
"""
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
PEAK_BF16, PEAK_HBM = 2500e12, 8000e9
N, K, Q, NB = 10000, 512, 128, 64


def load(which):
    d = os.path.join(P, f"{tag}_{which}")
    rows = list(csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))))
    pmc = json.load(open(os.path.join(d, "pmc_per_launch.json")))
    return rows, pmc


def row(r, pmc, per_pass, **extra):
    k = r["Name"]
    v = pmc.get(k, {})
    ms = float(r["AverageNs"]) * 1e-6
    out = {"kernel": k, "launches_per_pass": per_pass, "avg_ms": round(ms, 4), "ms_per_pass": round(ms * per_pass, 4),
           "hbm_bytes_counter": v.get("hbm_bytes_per_launch"), "mfma_util": v.get("mfma_util"),
           "lds_conflict_frac": v.get("lds_conflict_frac")}
    out.update(extra)
    if out.get("alg_bytes"):
        out["frac_of_hbm_peak"] = round(out["alg_bytes"] / (ms * 1e-3) / PEAK_HBM, 4)
        if out["hbm_bytes_counter"]:
            out["counter_over_alg_bytes"] = round(out["hbm_bytes_counter"] / out["alg_bytes"], 3)
    if out.get("executed_mfma_flops"):
        out["frac_of_bf16_mfma_peak"] = round(out["executed_mfma_flops"] / (ms * 1e-3) / PEAK_BF16, 4)
    return out


table = {"source": f"profiles/{tag}_agg, profiles/{tag}_emb: rocprofv3 --kernel-trace --stats and separate --pmc passes of "
                   f"`bench.py --streams 1` (tools/r3.sh or tools/r4.sh prof_* / pmc_*), condensed by tools/prof_summary.py + tools/kernel_table.py",
         "peaks": {"bf16_mfma_tflops": 2500, "hbm_gbs": 8000}}
rows, pmc = load("agg")
feat32, feat16 = NB * N * K * 4, NB * N * K * 2
mlp_flops = NB * (2 * N * K * Q + 2 * N * Q * Q)
passes32 = max(int(r["Calls"]) for r in rows if r["Name"].startswith("k_query_attend_split") or r["Name"].startswith("k_attend_f2") or r["Name"].startswith("k_attend_f3"))
passes16 = max(int(r["Calls"]) for r in rows if r["Name"].startswith("k_attend_bf16_res"))
agg32, agg16 = [], []
for r in rows:
    k, calls = r["Name"], int(r["Calls"])
    if k.startswith("k_attend_f2") or k.startswith("k_attend_f3"):
        agg32.append(row(r, pmc, 1, alg_bytes=feat32, executed_mfma_flops=3 * mlp_flops, alg_flops=mlp_flops,
                         note="round 5: tiles resident in LDS (features read once), query MLP as 3 fp16 plane products per fp32 MAC"
                              + ("; 32-row tiles, query weights resident in registers, one partial per (workgroup, bag)" if k.startswith("k_attend_f3") else "; 64-row tiles, weights from L2 per tile")))
    elif k.startswith("k_query_attend_split"):
        agg32.append(row(r, pmc, 1, alg_bytes=feat32, executed_mfma_flops=6 * mlp_flops, alg_flops=mlp_flops,
                         note="query MLP as 6 bf16 plane products per fp32 MAC; features read twice (MLP, value sum)"))
    elif k.startswith("k_attend_bf16_res"):
        agg16.append(row(r, pmc, 1, alg_bytes=feat16, executed_mfma_flops=mlp_flops, alg_flops=mlp_flops,
                         note="tile resident in LDS, weights in AGPRs: one read of every feature byte"))
    elif k.startswith("k_logits_stream<1, float>") or (k.startswith("k_logits_stream") and "float" in k):
        agg32.append(row(r, pmc, 1, alg_bytes=feat32))
    elif k.startswith("k_logits_stream"):
        agg16.append(row(r, pmc, 1, alg_bytes=feat16))
    elif k.startswith("k_qmax") and "float" in k:
        agg32.append(row(r, pmc, 1))
    elif k.startswith("k_qmax"):
        agg16.append(row(r, pmc, 1))
    elif k.startswith("k_finish") or k.startswith("k_pred"):
        note = "launched by both legs; time is the mix of both" if calls == passes32 + passes16 else None
        agg32.append(row(r, pmc, 1, note=note))
        agg16.append(row(r, pmc, 1, note=note))
table["aggregator"] = agg32
table["aggregator_bf16"] = agg16
try:
    rows, pmc = load("emb")
    fwd = max(int(r["Calls"]) for r in rows if r["Name"].startswith("k_stem"))
    emb = []
    for r in rows:
        k, calls = r["Name"], int(r["Calls"])
        if k.startswith("k_pack") or calls < fwd // 2:
            continue
        emb.append(row(r, pmc, round(calls / fwd, 2)))
    table["embedder"] = emb
    table["embedder_forward_ms"] = round(sum(e["ms_per_pass"] for e in emb), 3)
except FileNotFoundError:
    pass
for which, key in (("train", "train_step"), ("single", "single_bag")):
    # one launch of every kernel per fused train step / per single-bag forward (tools/r4.sh prof_fused / prof_single)
    try:
        rows, pmc = load(which)
    except FileNotFoundError:
        continue
    top = max(int(r["Calls"]) for r in rows)
    sec = [row(r, pmc, round(int(r["Calls"]) / top, 2)) for r in rows if int(r["Calls"]) >= top // 2]
    table[key] = sec
    table[key + "_gpu_ms"] = round(sum(e["ms_per_pass"] for e in sec), 4)
out = os.path.join(P, f"{tag}_kernel_table.json")
json.dump(table, open(out, "w"), indent=1)
print("wrote", out)
for leg in ("aggregator", "aggregator_bf16", "embedder", "train_step", "single_bag"):
    for e in table.get(leg, []):
        print(leg, e["kernel"][:50], e["ms_per_pass"], e.get("frac_of_bf16_mfma_peak"), e.get("frac_of_hbm_peak"), e.get("counter_over_alg_bytes"))
