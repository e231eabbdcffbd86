#!/usr/bin/env python3
"""Round 6: soak of the 16-bit-activation trunk (csrc/resnet_b16.h; `--precision half` = fp16 activations, `--precision bf16`)
over random patch shapes, batches and weights against the fp32-class trunk of the same module (which tests/test_resnet_gpu.py
holds to the oracle at odd sizes).  Every forward runs twice (bit-identical) and a second time after a forward of ANOTHER shape
(stale borders / scratch reuse).  Bars: those of tests/test_resnet_gpu.py (ResNet-18: half 5e-3 max, bf16 5e-2 max / 8e-3 mean; ResNet-34, twice the layers:
bf16 1e-1 / 2e-2, half 1.5e-2) for maps of 128 pixels and more per side; smaller maps are printed only (InstanceNorm over the
2 x 2 .. 3 x 3 pixel maps of layer 4 amplifies any rounding).   python tools/b16_soak.py [rounds] [seed]"""
import _path  # noqa: F401
import sys
import numpy as np
import torch
import torch.nn as nn
import dsmil
from dsmil_wsi_amd.resnet import resnet18, resnet34

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
BARS = {"half": (5e-3, 1e-3), "bf16": (5e-2, 8e-3)}             # ResNet-18 (16 conv + norm layers behind the stem)
BARS34 = {"half": (1.5e-2, 3e-3), "bf16": (1e-1, 2e-2)}          # ResNet-34 (32): the rounding of twice the layers
worst = {"half": [0.0, 0.0], "bf16": [0.0, 0.0]}
worst34 = {"half": [0.0, 0.0], "bf16": [0.0, 0.0]}
bad = 0
prev = None
for r in range(rounds):
    torch.manual_seed(seed * 1000 + r)
    deep = r % 8 == 7
    res = (resnet34 if deep else resnet18)(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    ic = dsmil.IClassifier(res, 512, output_class=int(rng.integers(1, 4))).eval().to(dev)
    for p in ic.parameters():
        p.requires_grad = False
    B = int(rng.integers(1, 49))
    small = r % 5 == 4
    lo, hi = (64, 128) if small else (128, 321)
    H, W = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
    x = torch.rand(B, 3, H, W, device=dev) * float(rng.choice([1.0, 1.0, 255.0]))
    out = {}
    with torch.no_grad():
        for prec in ("fp32", "half", "bf16"):
            ic.embed_precision = prec
            f1, c1 = ic(x)
            f2, c2 = ic(x)
            if prev is not None:   # another shape in between: the trunk's scratch and zero borders are re-laid
                ic(prev)
            f3, c3 = ic(x)
            torch.cuda.synchronize()
            if not (torch.equal(f1, f2) and torch.equal(f1, f3) and torch.equal(c1, c3)):
                print(f"round {r} {prec}: NOT bit-identical across repeats")
                bad += 1
            out[prec] = (f1.float().cpu(), c1.float().cpu())
    line = f"round {r:3d} {'r34' if deep else 'r18'} B {B:2d} {H:3d}x{W:3d} scale {float(x.max()):6.1f}:"
    ref = out["fp32"][0]
    for prec in ("half", "bf16"):
        f = out[prec][0]
        d = (f - ref).abs()
        mx, mean = float(d.max()), float(d.mean())
        fin = bool(torch.isfinite(f).all())
        line += f"  {prec} max {mx:.2e} mean {mean:.2e}{'' if fin else ' NON-FINITE'}"
        if not fin:
            bad += 1
        if not small:
            wr, bars = (worst34, BARS34) if deep else (worst, BARS)
            wr[prec][0] = max(wr[prec][0], mx)
            wr[prec][1] = max(wr[prec][1], mean)
            if mx > bars[prec][0] or mean > bars[prec][1]:
                line += " OVER-BAR"
                bad += 1
    print(line + ("  (small map: printed only)" if small else ""), flush=True)
    prev = x[: min(B, 3), :, : max(64, H - 17), : max(64, W - 9)].contiguous()
print(f"ResNet-34, maps >= 128 px: half max {worst34['half'][0]:.3e} mean {worst34['half'][1]:.3e}; bf16 max {worst34['bf16'][0]:.3e} mean {worst34['bf16'][1]:.3e}")
print(f"ResNet-18, maps >= 128 px: half max {worst['half'][0]:.3e} mean {worst['half'][1]:.3e}; bf16 max {worst['bf16'][0]:.3e} mean {worst['bf16'][1]:.3e}")
print("soak ok" if bad == 0 else f"soak FAILED: {bad} findings")
sys.exit(0 if bad == 0 else 1)
