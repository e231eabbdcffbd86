#!/usr/bin/env python3
"""Forward time of a trunk (--depth 18|34|50|101, InstanceNorm, bs B) through the native embedder: A/B of launch heuristics
that are chosen per layer (DSMIL_S6_TILE, DSMIL_WINO_KERNEL in experiment builds).
    DSMIL_NATIVE_LIB=libdsmil_hip_expt.so DSMIL_S6_TILE=44 python tools/trunk_time.py 50 64"""
import sys
import time

import _path  # noqa: F401
import torch
import torch.nn as nn
import dsmil
from dsmil_wsi_amd import resnet as R

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
net = getattr(R, f"resnet{depth}")(norm_layer=nn.InstanceNorm2d)
net.fc = nn.Identity()
feat = 512 if depth in (18, 34) else 2048
ic = dsmil.IClassifier(net, feat, output_class=2).eval().cuda()
for p in ic.parameters():
    p.requires_grad = False
x = torch.rand(B, 3, 224, 224, device="cuda")
with torch.no_grad():
    for _ in range(3):
        ic(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        ic(x)
    torch.cuda.synchronize()
print("depth %d bs %d: %.3f ms per forward, %.0f patches/s" % (depth, B, (time.perf_counter() - t0) / n * 1e3, B * n / (time.perf_counter() - t0)))
