"""CPU-only error study of the candidate MFMA operand forms of the patch embedder ("next" item 3).

Question: which cut of the fp32 operands (how many low-precision planes, which plane products) keeps the ResNet-18-IN
features inside the 1e-4 parity bar of BASELINE.md §4 after 20 convs + InstanceNorms — measured, not estimated, with zero
GPU minutes.  Every conv of the trunk is emulated as

    conv_form(x, w) = sum over the form's plane pairs (i, j) of conv_fp64(x_i, w_j)

with x = sum_i x_i, w = sum_j w_j the plane cuts; everything else (InstanceNorm, ReLU, max-pool, residual add, avg-pool)
runs in fp64, and the conv result is rounded to fp32 where the GPU stores it.  The fp32 accumulation order of the MFMA is
NOT emulated (it is the same for every form); what is compared is the error each form ADDS through its dropped plane
products and inexact cuts.  Reference: the same trunk entirely in fp64 (oracle/resnet_numpy.py's arithmetic through
torch's fp64 conv).  Winograd convs cut their operands in the transform domain on the GPU; this study cuts in the spatial
domain (same relative size of the dropped terms).

Forms
  bf16x3_9   three bf16 planes by truncation (exact cut), all nine products: every fp32 product exact
  bf16x3_6   the six largest (product library, rounds 2-4): drops x1w2 + x2w1 + x2w2 (<= 2^-23 |xw|)
  bf16x3_3   x0w0 + x0w1 + x1w0: drops terms of 2^-16 |xw|
  bf16x1     plain bf16 x bf16 (RNE), the `--dtype bf16` embedder leg
  f16x2_3    TWO fp16 planes by round-to-nearest (x = h0 + h1 + e, |e| <= 2^-24 |x| while h1 stays normal), products
             h0h0 + h0h1 + h1h0: drops h1h1 (<= 2^-24 |xw|) — fp32-class accuracy from THREE MFMAs instead of six.
             fp16 has 5 exponent bits: operands are pre-scaled by a power of two per tensor (exact) so that max |v| sits at
             2^SCALE_TOP; values whose second plane would fall below the fp16 subnormal quantum lose it (`ftz`: also below
             the smallest NORMAL, should the matrix pipe flush fp16 denormals)
  f16x2_4    the same with h1h1
  f16x1      plain fp16 x fp16 (RNE), scaled

    python tools/form_error_study.py [--batch 8] [--size 224] [--seed 11]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dsmil  # noqa: F401,E402
from dsmil_wsi_amd.synthetic import make_patches, make_resnet18_weights  # noqa: E402

SCALE_TOP = 12   # pre-scaled operands have max |v| in [2^11, 2^12): products of two tops stay far below fp32 overflow


def bf16_trunc_planes(v, n):
    """n bf16 planes by truncation: v = p0 + p1 + ... exactly for n = 3 (8 significand bits each)."""
    out, r = [], v.to(torch.float32).clone()
    for _ in range(n):
        bits = r.view(torch.int32) & ~0xFFFF
        p = bits.view(torch.float32)
        out.append(p.double())
        r = (r - p)
    return out


def bf16_rne(v):
    return [v.to(torch.float32).to(torch.bfloat16).double()]


def _pow2_scale(v):
    m = float(v.abs().max())
    if m == 0:
        return 1.0
    return 2.0 ** (SCALE_TOP - 1 - int(np.floor(np.log2(m))))


def f16_planes(v, n, ftz):
    """n fp16 planes by round-to-nearest of the pre-scaled value; returns (planes in the UNscaled domain, as fp64)."""
    s = _pow2_scale(v)
    r = (v.double() * s)
    out = []
    for _ in range(n):
        h = r.to(torch.float16)
        if ftz:
            h = torch.where(h.abs() < 2.0 ** -14, torch.zeros_like(h), h)
        hd = h.double()
        out.append(hd / s)
        r = r - hd
    return out


FORMS = {
    "bf16x3_9": (lambda v: bf16_trunc_planes(v, 3), [(i, j) for i in range(3) for j in range(3)]),
    "bf16x3_6": (lambda v: bf16_trunc_planes(v, 3), [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
    "bf16x3_3": (lambda v: bf16_trunc_planes(v, 2), [(0, 0), (0, 1), (1, 0)]),
    "bf16x3_4": (lambda v: bf16_trunc_planes(v, 2), [(0, 0), (0, 1), (1, 0), (1, 1)]),
    "bf16x1": (bf16_rne, [(0, 0)]),
    "f16x2_3": (lambda v: f16_planes(v, 2, False), [(0, 0), (0, 1), (1, 0)]),
    "f16x2_3_ftz": (lambda v: f16_planes(v, 2, True), [(0, 0), (0, 1), (1, 0)]),
    "f16x2_4": (lambda v: f16_planes(v, 2, False), [(0, 0), (0, 1), (1, 0), (1, 1)]),
    "f16x1": (lambda v: f16_planes(v, 1, False), [(0, 0)]),
}


def conv_form(x, w, stride, pad, form):
    if form is None:
        return F.conv2d(x, w, stride=stride, padding=pad)
    cut, pairs = FORMS[form]
    xs, ws = cut(x), cut(w)
    y = None
    for i, j in pairs:
        t = F.conv2d(xs[i], ws[j], stride=stride, padding=pad)
        y = t if y is None else y + t
    return y.float().double()   # the GPU stores the conv result as fp32


def inorm(x):
    return F.instance_norm(x, eps=1e-5)


def trunk(x, w, form):
    c = lambda x_, name, s, p: conv_form(x_, w[name + ".weight"], s, p, form)
    y = F.max_pool2d(F.relu(inorm(c(x, "conv1", 2, 3))), 3, 2, 1)
    for li in range(1, 5):
        for b in range(2):
            down = li > 1 and b == 0
            pre = f"layer{li}.{b}"
            s = 2 if down else 1
            out = F.relu(inorm(c(y, pre + ".conv1", s, 1)))
            out = inorm(c(out, pre + ".conv2", 1, 1))
            idn = inorm(c(y, pre + ".downsample.0", s, 0)) if down else y
            y = F.relu(out + idn)
    return y.mean(dim=(2, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--patch-seed", type=int, default=5)
    ap.add_argument("--forms", default=",".join(FORMS))
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    w = {k: torch.from_numpy(np.asarray(v)).double() for k, v in make_resnet18_weights(a.seed).items()}
    x = torch.from_numpy(make_patches(a.patch_seed, a.batch, a.size, a.size)).double()
    t0 = time.time()
    ref = trunk(x, w, None)
    print(f"fp64 reference: {time.time() - t0:.1f} s, features in [{float(ref.min()):.3f}, {float(ref.max()):.3f}], "
          f"mean {float(ref.mean()):.3f}", flush=True)
    rows = {}
    for form in a.forms.split(","):
        t0 = time.time()
        f = trunk(x, w, form)
        d = (f - ref).abs().flatten()
        rows[form] = {"products": len(FORMS[form][1]), "max_abs": float(d.max()), "p999_abs": float(torch.quantile(d, 0.999)),
                      "mean_abs": float(d.mean()), "margin_vs_1e-4": 1e-4 / max(float(d.max()), 1e-300)}
        print(f"{form:12s} products {rows[form]['products']}  max {rows[form]['max_abs']:.3e}  p99.9 {rows[form]['p999_abs']:.3e}  "
              f"mean {rows[form]['mean_abs']:.3e}  margin x{rows[form]['margin_vs_1e-4']:.1f}  ({time.time() - t0:.0f} s)", flush=True)
    if a.out:
        json.dump({"batch": a.batch, "size": a.size, "weight_seed": a.seed, "patch_seed": a.patch_seed, "scale_top": SCALE_TOP,
                   "forms": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
