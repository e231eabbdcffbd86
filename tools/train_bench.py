#!/usr/bin/env python3
"""Time one DSMIL training step (train_tcga.py:60-74: forward, loss, backward, Adam step) per bag on
the GPU: native backward (dsmil_agg_backward) vs the dense-product backward, plus the parts.
Usage: python tools/train_bench.py [--rows 10000] [--feats 512] [--classes 2] [--steps 50]"""
import _path  # noqa: F401  (repo root on sys.path)
import argparse
import json
import time

import torch

import dsmil as mil


def timed(fn, steps, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--feats", type=int, default=512)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda"
    net = mil.MILNet(mil.FCLayer(a.feats, a.classes), mil.BClassifier(a.feats, a.classes)).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.5, 0.9), weight_decay=1e-3)
    crit = torch.nn.BCEWithLogitsLoss()
    x = torch.randn(a.rows, a.feats, device=dev)
    xd = x.clone().requires_grad_(True)  # forces the dense-product backward
    y = torch.zeros(1, a.classes, device=dev)
    y[0, 0] = 1

    def step(inp):
        opt.zero_grad(set_to_none=True)
        ins, bag, _, _ = net(inp)
        mx, _ = torch.max(ins, 0)
        loss = 0.5 * crit(bag.view(1, -1), y) + 0.5 * crit(mx.view(1, -1), y)
        loss.backward()
        opt.step()

    def fwd_only():
        with torch.no_grad():
            net(x)

    from dsmil_wsi_amd import ops
    w = {"fc_w": net.i_classifier.fc[0].weight.detach(), "fc_b": net.i_classifier.fc[0].bias.detach()}
    w.update({k: v.detach() for k, v in net.b_classifier._weights().items()})
    classes, pred, A, B, idx = ops.agg_forward(x, [a.rows], w)
    gp = torch.ones(a.classes, device=dev)
    gc = torch.zeros(a.rows, a.classes, device=dev)

    def bwd_only():
        ops.agg_backward(x, w, A, B, idx, gp, g_classes=gc)

    out = {
        "rows": a.rows, "feats": a.feats, "classes": a.classes,
        "train_step_native_ms": timed(lambda: step(x), a.steps),
        "train_step_dense_ms": timed(lambda: step(xd), a.steps),
        "forward_only_ms": timed(fwd_only, a.steps),
        "native_backward_only_ms": timed(bwd_only, a.steps),
    }
    out["bags_per_s_native"] = 1e3 / out["train_step_native_ms"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
