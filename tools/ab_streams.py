#!/usr/bin/env python3
"""A/B of two library builds at 1 and 3 streams, interleaved rounds: python tools/ab_streams.py <workload> <libA> <libB>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl, libs = sys.argv[1], sys.argv[2:]
res = {}
for rnd in range(2):
    for lib in libs:
        for st in (1, 3):
            e = dict(os.environ, DSMIL_NATIVE_LIB=lib)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--no-cpu-baseline", "--no-single-bag",
                                  "--streams", str(st), "--steps", "5", "--warmup", "2", "--min-seconds", "0.5"], env=e, capture_output=True, text=True, timeout=600)
            j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            res.setdefault((lib, st), []).append((j["value"], j["ms_per_pass"]))
for k, v in res.items():
    print(k, v, flush=True)
