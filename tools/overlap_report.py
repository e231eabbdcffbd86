#!/usr/bin/env python3
"""Timeline report of a rocprofv3 --kernel-trace CSV (tools/overlap_run.py): per kernel the time it ran beside an attend
kernel of ANOTHER stream, the gaps between consecutive attend kernels, and a text timeline of the last passes.
    python tools/overlap_report.py <kernel_trace.csv> [window_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
ev = []
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
t_last = ev[-1][1]
att_all = [e for e in ev if "attend" in e[2]]
t_end = att_all[-max(4, len(att_all) // 4)][1]          # a window in the middle of the timed part, not the drain at its end
ev = [e for e in ev if t_end - win * 1000 * 6 <= e[0] and e[1] <= t_end + win * 1000]
att = [e for e in ev if "attend" in e[2]]
print("attend launches", len(att))
gaps = [(att[i + 1][0] - att[i][1]) / 1000 for i in range(len(att) - 1)]
durs = [(e[1] - e[0]) / 1000 for e in att]
if gaps:
    gs = sorted(gaps)
    print(f"attend duration us: median {sorted(durs)[len(durs) // 2]:.1f}  min {min(durs):.1f} max {max(durs):.1f}")
    print(f"gap between consecutive attends us: median {gs[len(gs) // 2]:.1f}  min {gs[0]:.1f}  max {gs[-1]:.1f}   (negative = overlap)")
    span = (att[-1][1] - att[0][0]) / 1000 / (len(att) - 1) if len(att) > 1 else 0
    print(f"attend start-to-start period us: {((att[-1][0] - att[0][0]) / 1000 / (len(att) - 1)):.1f}")
names = sorted({e[2] for e in ev})
for n in names:
    ks = [e for e in ev if e[2] == n]
    tot = sum(e[1] - e[0] for e in ks) / 1000
    under = 0
    for e in ks:
        for a in att:
            if a is e or a[3] == e[3] and a[4] == e[4]:
                continue
            lo, hi = max(a[0], e[0]), min(a[1], e[1])
            if hi > lo:
                under += hi - lo
    print(f"{n:28s} n={len(ks):4d} avg {tot / len(ks):8.1f} us   beside another stream's attend: {under / 1000 / max(tot, 1e-9) * 100:5.1f} %")
print("--- timeline (us from window start), last window")
t0 = t_end - int(win * 1000)
for e in ev:
    if e[1] >= t0 and e[0] <= t_end:
        print(f"{(e[0] - t0) / 1000:9.1f} .. {(e[1] - t0) / 1000:9.1f}  q{e[3]:>3s} s{e[4]:>3s}  {e[2]}")
