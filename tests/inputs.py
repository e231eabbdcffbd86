"""Seeded synthetic inputs shared by the golden generator, the tests, smoke() and bench.py.

numpy's PCG64 ``standard_normal``/``random`` streams are what the committed golden vectors were
generated from; every golden case stores the sha256 of its input so that a numpy that produced
a different stream would be detected instead of silently mis-comparing.
"""
import numpy as np


def make_bag(seed, N, K, scale=1.0):
    """A bag of N instance feature rows, fp32 N(0, scale^2), row-major [N,K]."""
    rng = np.random.default_rng(int(seed))
    return (rng.standard_normal((N, K), dtype=np.float32) * np.float32(scale)).astype(np.float32)


def make_label(seed, C):
    """A 0/1 bag label vector of length C (train_tcga.py:26-33 builds one-hot / binary labels)."""
    rng = np.random.default_rng(int(seed) + 7919)
    y = np.zeros(C, np.float32)
    if C == 1:
        y[0] = float(rng.integers(0, 2))
    else:
        y[int(rng.integers(0, C))] = 1.0
    return y


def make_patches(seed, B, H=224, W=224):
    """A batch of synthetic RGB patches in [0,1), NCHW fp32 — the range VF.to_tensor yields
    (compute_feats.py:35-39, no mean/std normalisation)."""
    rng = np.random.default_rng(int(seed))
    return rng.random((B, 3, H, W), dtype=np.float32)


# torchvision ResNet-18 conv tensors in registration order: (name, cout, cin, k)
RESNET18_CONVS = [("conv1", 64, 3, 7)] + [
    (f"layer{li}.{b}.{c}", co, (ci if (b == 0 and c != "conv2") else co), (1 if c == "downsample.0" else 3))
    for li, ci, co in ((1, 64, 64), (2, 64, 128), (3, 128, 256), (4, 256, 512))
    for b in (0, 1)
    for c in (("conv1", "conv2") + (("downsample.0",) if (b == 0 and li > 1) else ()))]


def make_resnet18_weights(seed=11):
    """Seeded kaiming-normal(fan_out, relu) conv weights — torchvision's ResNet init — as an ordered dict
    name -> [Cout,Cin,k,k] (seed 11).  Same stream as oracle/resnet_oracle.make_weights."""
    import collections
    import torch
    g = torch.Generator().manual_seed(seed)
    w = collections.OrderedDict()
    for name, cout, cin, k in RESNET18_CONVS:
        std = (2.0 / (cout * k * k)) ** 0.5
        w[name + ".weight"] = torch.randn((cout, cin, k, k), generator=g, dtype=torch.float32) * std
    return w
