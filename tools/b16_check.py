#!/usr/bin/env python3
"""Round 6: the opt-in bf16-activation trunk (precision "bf16") against the fp32-class trunk on the same patches, and its
forward time.   python tools/b16_check.py [batch]"""
import _path  # noqa: F401
import sys
import time
import torch
import torch.nn as nn
import dsmil
from dsmil_wsi_amd.resnet import resnet18

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
res = resnet18(norm_layer=nn.InstanceNorm2d)
res.fc = nn.Identity()
ic = dsmil.IClassifier(res, 512, output_class=2).eval().to(dev)
for p in ic.parameters():
    p.requires_grad = False
x = torch.rand(B, 3, 224, 224, device=dev)
out = {}
for prec in ("fp32", "half", "bf16"):
    ic.embed_precision = prec
    with torch.no_grad():
        f, c = ic(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            f, c = ic(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    out[prec] = f.float().cpu()
    print(f"{prec}: {dt * 1e3:.3f} ms per forward of {B} = {B / dt:.0f} patches/s; feats finite {bool(torch.isfinite(f).all())}  |f| max {float(f.abs().max()):.3f} mean {float(f.abs().mean()):.4f}")
for prec in ("half", "bf16"):
    d = (out[prec] - out["fp32"]).abs()
    print(f"{prec} vs fp32: max abs {float(d.max()):.3e}  mean abs {float(d.mean()):.3e}  rel-to-max {float(d.max() / out['fp32'].abs().max()):.3e}")
