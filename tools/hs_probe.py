#!/usr/bin/env python3
"""Determinism / parity probe of the hidden-split attend kernel: repeated forwards of one bag, with and without a row map."""
import _path  # noqa: F401
import sys
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops
from dsmil_wsi_amd.synthetic import load_weights
w = {k: torch.from_numpy(v).cuda() for k, v in load_weights("c16").items()}
for N in (10000, 35000, 1000):
    g = torch.Generator(device="cuda").manual_seed(N)
    x = torch.randn(2 * N, 512, device="cuda", generator=g)
    rows = torch.randperm(2 * N, device="cuda", generator=g)[:N]
    xg = x.index_select(0, rows)
    ref = ops.agg_forward(xg, [N], w)
    torch.cuda.synchronize()
    for name, fn in (("gathered", lambda: ops.agg_forward(xg, [N], w)), ("row_map", lambda: ops.agg_forward(x, [N], w, row_map=rows))):
        bad = {}
        for it in range(30):
            out = fn()
            for k, (u, v) in enumerate(zip(out, ref)):
                if not torch.equal(u, v):
                    d = (u.float() - v.float()).abs()
                    bad.setdefault(k, []).append((it, int((d > 0).sum()), float(d.max())))
        print(N, name, "mismatches by output (classes, pred, A, B, idx):", {k: (len(v), v[:3]) for k, v in bad.items()})
