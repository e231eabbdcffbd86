#!/usr/bin/env python3
"""Experiment (round 6): the persistent batch kernel on FEWER than 256 workgroups while other streams' k_logits_stream
launches take the CUs it leaves free — does the pass (two feature reads) shorten when the second read of batch i+1 runs
beside the MFMA kernel of batch i?      python tools/exp_grid.py agg|bf16"""
import _path  # noqa: F401
import sys
import time
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops, _native
from dsmil_wsi_amd.synthetic import load_weights

which = sys.argv[1] if len(sys.argv) > 1 else "agg"
L = _native.lib()
dev = torch.device("cuda:0")
wnp = load_weights("c16" if which == "agg" else "tcga")
w = {k: torch.from_numpy(v).to(dev) for k, v in wnp.items()}
nb, N, K = 64, 10000, 512
g = torch.Generator(device=dev).manual_seed(1234)
batches = []
for i in range(3):
    f = torch.randn((nb * N, K), generator=g, device=dev)
    batches.append(f.to(torch.bfloat16) if which == "bf16" else f)
lengths = [N] * nb
offsets = ops.offsets_tensor(lengths, dev)
rounds = 243
for grid in (256, 248, 240, 224, 208, 192, 176, 160, 128):
    L.dsmil_agg_persistent_grid(grid)
    for S in (1, 2, 3, 4):
        pool = ops.StreamPool(S)
        for r in (24, rounds):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(r):
                fb = batches[i % 3]
                pool.run(lambda: ops.agg_forward(fb, lengths, w, offsets=offsets))
            pool.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{which} grid {grid} streams {S}: {nb * r / dt:.0f} bags/s, {dt / r * 1e3:.4f} ms per pass", flush=True)
L.dsmil_agg_persistent_grid(256)
