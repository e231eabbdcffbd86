#!/usr/bin/env python3
"""Stage times of the multi-scale end-to-end path (bench.py e2e leg) on one GPU."""
import _path  # noqa: F401  (repo root on sys.path)
import time
import numpy as np
import torch
import torch.nn as nn
import dsmil
from dsmil_wsi_amd import pipeline as pl
import sys
from dsmil_wsi_amd.synthetic import build_net  # noqa: E402
from dsmil_wsi_amd.resnet import resnet18

dev = torch.device("cuda:0")
def mk(seed):
    torch.manual_seed(seed)
    r = resnet18(norm_layer=nn.InstanceNorm2d); r.fc = nn.Identity()
    ic = dsmil.IClassifier(r, 512, output_class=2).eval().to(dev)
    for p in ic.parameters(): p.requires_grad = False
    return ic
e_lo, e_hi = mk(11), mk(12)
if len(sys.argv) > 1:                      # python tools/e2e_stages.py half
    e_lo.embed_precision = e_hi.embed_precision = sys.argv[1]
net = build_net("tree", dev)
gy, gx = 24, 26
wsi = torch.randint(0, 256, (gy * 896, gx * 896, 3), device=dev, dtype=torch.uint8)
colors = [np.array([255, 40, 0]), np.array([0, 90, 255])]
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    ta = T()
    out = pl.multiscale_attention_map(wsi, e_lo, e_hi, net, [0.5, 0.5], colors)
    print("whole multiscale_attention_map: %.1f ms" % (1e3 * (T() - ta)))
    tb = T()
    pl.box_downsample_u8(wsi, 4)
    print("box_downsample_u8 alone: %.1f ms" % (1e3 * (T() - tb)))
    t0 = T()
    low, high, parent, pos = pl.pyramid_tiles(wsi, 224, 4)
    t1 = T()
    f_low, _ = pl.embed_tiles(e_lo, low, 256)
    t2 = T()
    f_high, _ = pl.embed_tiles(e_hi, high, 256)
    t3 = T()
    tree = torch.cat([f_high, f_low.index_select(0, parent)], dim=-1)
    with torch.no_grad():
        classes, pred, A, B = net(tree)
    t4 = T()
    prob = np.atleast_1d(torch.sigmoid(pred).squeeze().cpu().numpy())
    An, pn = A.cpu().numpy(), pos.cpu().numpy()
    t5 = T()
    cmap = pl.attention_colormap(An, pn, prob, [0.5, 0.5], colors, None, "slide", lambda *_: None, upsample_device=dev)
    t6 = T()
    print("ms: tiling %.1f  embed_low %.1f  embed_high %.1f  concat+aggregate %.1f  d2h %.1f  colormap %.1f  total %.1f" %
          tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0)), cmap.shape)
