#!/usr/bin/env python3
"""Six training steps on the same bags through training.FusedTrainStep (one dsmil_agg_train_step call per step) and through
the generic path (MILNet.bag_loss under autograd, loss.backward(), torch.optim.Adam.step()), from the same start: per step, the
largest difference of every parameter and of every first moment, relative to the tensor's largest magnitude — all zeros when the
two paths are bit-identical.
    python tools/fused_vs_generic.py [tcga|c16|...] [rows]"""
import _path  # noqa: F401
import sys
import numpy as np  # noqa: F401
import torch
import dsmil  # noqa
from dsmil_wsi_amd import training as T
from dsmil_wsi_amd.synthetic import make_bag, VARIANT, build_net
tag = sys.argv[1] if len(sys.argv) > 1 else "tcga"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
K, C, nonlinear, _ = VARIANT[tag]
nets = [build_net(tag, "cuda").train() for _ in range(2)]
hp = dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=1e-3)
opts = [torch.optim.Adam(n.parameters(), **hp) for n in nets]
crit = torch.nn.BCEWithLogitsLoss()
fused = T.FusedTrainStep.create(nets[1], crit, opts[1])
for step in range(6):
    x = torch.from_numpy(make_bag(900 + step, N, K)).cuda()
    y = torch.zeros(1, C).cuda()
    y[0, step % C] = float(step % 2) if C == 1 else 1.0
    opts[0].zero_grad()
    l0, _, _ = T.bag_loss(nets[0], crit, x, y, None)
    l0.backward()
    opts[0].step()
    l1 = fused(x, y, None)
    fused.sync()
    out = [f"step {step} loss {l0.item():.7f} {l1.item():.7f}"]
    for (n0, p0), (n1, p1) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        s0, s1 = opts[0].state[p0], opts[1].state[p1]
        m0, m1 = s0["exp_avg"], s1["exp_avg"]
        d = (m0 - m1).abs()
        out.append(f"{n0.split('.')[-2]}.{n0.split('.')[-1][0]} dm={float(d.max()/m0.abs().max()):.1e} dp={float((p0-p1).abs().max()/p0.abs().max()):.1e}")
        if n0.endswith("q.0.weight"):
            bad = d > 2e-4 * m0.abs().max()
            rows = bad.sum(1).nonzero().flatten().tolist()
            out.append(f"badrows={rows[:8]} n={int(bad.sum())}")
    print(" | ".join(out))
