#!/usr/bin/env python3
"""Experiment: independent passes dealt round-robin to S HIP streams (ops.StreamPool, one workspace per stream).
    python tools/streams.py agg|bf16|emb"""
import _path  # noqa: F401  (repo root on sys.path)
import sys
import time
import torch
import torch.nn as nn
import dsmil
from dsmil_wsi_amd import ops
from dsmil_wsi_amd.synthetic import load_weights  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "agg"
dev = torch.device("cuda:0")
if which in ("agg", "bf16"):
    wnp = load_weights("c16" if which == "agg" else "tcga")
    w = {k: torch.from_numpy(v).to(dev) for k, v in wnp.items()}
    nb, N, K = 64, 10000, 512
    g = torch.Generator(device=dev).manual_seed(1234)
    feats = torch.randn((nb * N, K), generator=g, device=dev)
    if which == "bf16":
        feats = feats.to(torch.bfloat16)
    lengths = [N] * nb
    offsets = ops.offsets_tensor(lengths, dev)
    call = lambda: ops.agg_forward(feats, lengths, w, offsets=offsets)
    units, rounds = nb, 240
else:
    from dsmil_wsi_amd.resnet import resnet18
    torch.manual_seed(0)
    res = resnet18(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    ic = dsmil.IClassifier(res, 512, output_class=2).eval().to(dev)
    for p in ic.parameters():
        p.requires_grad = False
    x = torch.rand(256, 3, 224, 224, device=dev)
    def call():
        with torch.no_grad():
            return ic(x)
    units, rounds = 256, 40
for S in (1, 2, 3, 4):
    pool = ops.StreamPool(S)
    for r in (max(4, rounds // 10), rounds):   # warm-up, then timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(r):
            out = pool.run(call)
        pool.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{which} streams {S}: {units * r / dt:.0f} units/s, {dt / r * 1e3:.4f} ms per pass", flush=True)
