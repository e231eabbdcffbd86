#!/usr/bin/env python3
"""A ragged batch in the 128-row regime (one 60 000-row bag among 300 bags of 64 rows + three of 9 000) against a uniform batch
of the same row count, 50 passes each — for rocprofv3 --kernel-trace --stats (per-kernel cost of the ragged launch sequence).
    python tools/ragged_probe.py [f32|bf16]"""
import _path  # noqa: F401
import sys
import time
import numpy as np
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops
from dsmil_wsi_amd.synthetic import load_weights

bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
w = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in load_weights("tcga" if bf16 else "c16").items()}
lengths = [64] * 150 + [60000] + [64] * 150 + [9000] * 3
x = torch.randn(sum(lengths), 512, device="cuda")
if bf16:
    x = x.to(torch.bfloat16)
nb_u = 16
n_u = sum(lengths) // nb_u
for name, xx, ll in (("ragged", x, lengths), ("uniform", x[:nb_u * n_u], [n_u] * nb_u)):
    for _ in range(3):
        ops.agg_forward(xx, ll, w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ops.agg_forward(xx, ll, w)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per pass, {sum(ll)} rows in {len(ll)} bags", flush=True)
