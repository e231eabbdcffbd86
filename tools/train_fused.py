#!/usr/bin/env python3
"""Fused training steps only (training.FusedTrainStep = one dsmil_agg_train_step call per bag), for rocprofv3 kernel stats.
    python tools/train_fused.py [classes] [steps]"""
import _path  # noqa: F401
import sys
import time
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import training
from dsmil_wsi_amd.synthetic import build_net
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
net = build_net("c16" if C == 1 else "tcga", "cuda").train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.5, 0.9), weight_decay=1e-3)
fused = training.FusedTrainStep.create(net, torch.nn.BCEWithLogitsLoss(), opt)
bags = [torch.randn(10000, 512, device="cuda") for _ in range(8)]
y = torch.zeros(1, C, device="cuda")
y[0, 0] = 1
for i in range(20):
    fused(bags[i % 8], y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss = fused(bags[i % 8], y)
torch.cuda.synchronize()
print(f"C={C}: {(time.perf_counter() - t0) / steps * 1e6:.1f} us per fused step (no per-step sync), loss {loss.item():.4f}")
