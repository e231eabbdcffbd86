#!/usr/bin/env python3
"""Timeline of the JPEG -> features pipeline from a rocprofv3 --kernel-trace CSV (of `bench.py --workload slide_jpeg_half` or
tools/jpeg_chunk_sweep.py): every decode launch sequence (k_jpeg_*) with its start, per-kernel durations and the gap to the
previous one, and how busy the embedder kernels kept the device in between.
    python tools/jpeg_timeline.py <kernel_trace.csv> [last_n_decodes]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ev = []
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
dec = [e for e in ev if "jpeg" in e[2]]
emb = [e for e in ev if "jpeg" not in e[2]]
# group the decode kernels into launch sequences: a sequence starts with k_jpeg_unstuff
seqs = []
for e in dec:
    if "unstuff" in e[2] or not seqs:
        seqs.append([])
    seqs[-1].append(e)
seqs = seqs[-last_n:]
t0 = seqs[0][0][0]


def busy(lo, hi):
    """fraction of [lo, hi] covered by at least one embedder kernel"""
    iv = sorted((max(lo, a), min(hi, b)) for a, b, _ in emb if b > lo and a < hi)
    cov, end = 0, lo
    for a, b in iv:
        if b > end:
            cov += b - max(a, end)
            end = b
    return cov / max(1, hi - lo)


prev_end = None
for s in seqs:
    a, b = s[0][0], max(x[1] for x in s)
    parts = "  ".join(f"{x[2].split('<')[0].replace('k_jpeg_', '')} {(x[1] - x[0]) / 1e3:.0f}" for x in s)
    gap = "" if prev_end is None else f"  gap to previous decode {(a - prev_end) / 1e3:8.0f} us (embedder busy {busy(prev_end, a) * 100:3.0f} %)"
    print(f"decode at {(a - t0) / 1e3:9.0f} us, {(b - a) / 1e3:7.0f} us [{parts}], embedder busy under it {busy(a, b) * 100:3.0f} %{gap}")
    prev_end = b
lo, hi = seqs[0][0][0], max(x[1] for x in ev)
print(f"window {(hi - lo) / 1e3:.0f} us: embedder busy {busy(lo, hi) * 100:.0f} %; decode kernels {sum(max(x[1] for x in s) - s[0][0] for s in seqs) / 1e3:.0f} us")
