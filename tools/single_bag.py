#!/usr/bin/env python3
"""One MILNet.forward-sized call per iteration (a lone 10 000 x 512 bag): latency and, under rocprofv3, per-kernel times."""
import _path  # noqa: F401
import sys
import time
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops
from dsmil_wsi_amd.synthetic import load_weights
tag = sys.argv[1] if len(sys.argv) > 1 else "c16"
dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
w = {k: torch.from_numpy(v).cuda() for k, v in load_weights(tag).items()}
x = torch.randn(10000, 512, device="cuda").to(dt)
for _ in range(10):
    ops.agg_forward(x, [10000], w)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    ops.agg_forward(x, [10000], w)
torch.cuda.synchronize()
print(f"{tag} {dt}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per forward")
