#!/usr/bin/env python3
"""Experiment (round 3): does the 256 MiB Infinity Cache serve the aggregator's SECOND pass over the features?

The forward reads every feature row twice (k_logits_stream, then k_query_attend*).  With all 64 bags in one call the
first pass has streamed 0.66 / 1.31 GB before the second begins, so the second pass comes from HBM again.  Here the
same batch is run as G-bag groups (logits + attend of a group back to back); a group of G x 10 000 x 512 bf16 rows is
G x 10.2 MB (fp32: 20.5 MB).  Prints ms per 64 bags and the attend kernel's own time per group size.
    python tools/exp_groups.py [bf16|f32] [--reps 20]
"""
import _path  # noqa: F401
import ctypes
import sys
import time

import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops, _native
from dsmil_wsi_amd.synthetic import load_weights  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
dev = torch.device("cuda:0")
N, K, nb = 10000, 512, 64
wnp = load_weights("tcga" if which == "bf16" else "c16")
w = {k: torch.from_numpy(v).to(dev) for k, v in wnp.items()}
g = torch.Generator(device=dev).manual_seed(1234)
feats = torch.randn((nb * N, K), generator=g, device=dev, dtype=torch.float32)
if which == "bf16":
    feats = feats.to(torch.bfloat16)
L = _native.lib()


def run(G):
    for b0 in range(0, nb, G):
        ops.agg_forward(feats[b0 * N:(b0 + G) * N], [N] * G, w)


for G in (64, 32, 16, 8, 4, 2):
    for _ in range(3):
        run(G)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run(G)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    L.dsmil_profile_enable(1)
    for _ in range(4):
        run(G)
    torch.cuda.synchronize()
    tot, n = ctypes.c_double(0), ctypes.c_int64(0)
    L.dsmil_profile_collect(0, ctypes.byref(tot), ctypes.byref(n))
    L.dsmil_profile_enable(0)
    print(f"{which} group={G:2d} bags ({G * N * K * feats.element_size() / 1e6:7.1f} MB): {ms:7.3f} ms per 64 bags "
          f"= {nb / ms:7.1f} k bags/s; attend kernel {tot.value / 4:7.3f} ms per 64 bags ({n.value // 4} launches)", flush=True)
