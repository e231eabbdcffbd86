#!/usr/bin/env python3
"""A/B of kernel variants on the GPU box: runs bench.py legs in subprocesses under different environment settings
(experiment builds read their knobs once per process) and prints one line per variant.
Usage: python tools/variants.py <workload> NAME:K=V,K=V ...   (NAME: alone = product build, no knobs)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload = sys.argv[1]
rounds = int(os.environ.get("VARIANT_ROUNDS", "2"))
specs = []
for a in sys.argv[2:]:
    name, _, kv = a.partition(":")
    env = dict(x.split("=", 1) for x in kv.split(",") if x)
    specs.append((name, env))
res = {n: [] for n, _ in specs}
for r in range(rounds):   # interleaved rounds: box drift hits every variant alike
    for name, env in specs:
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--no-cpu-baseline",
                              "--no-single-bag", "--streams", "1", "--steps", "5", "--warmup", "2", "--min-seconds", "0.4"],
                             env=e, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            res[name].append(("FAIL", out.stderr[-600:]))
            continue
        j = json.loads(line[-1])
        rf = j.get("roofline", {})
        res[name].append((j["value"], rf.get("kernel_ms") or rf.get("kernel_ms_total"), j.get("ms_per_pass")))
for name, _ in specs:
    print(name, res[name], flush=True)
