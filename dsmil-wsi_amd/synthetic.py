"""Seeded synthetic inputs and the example weight sets — shared by bench.py, the entry-point demos, smoke() and the tests.

Nothing here touches ``oracle/`` or ``tests/``: this is product-side data plumbing.

* ``make_bag`` / ``make_label`` / ``make_patches``: numpy PCG64 streams.  The committed golden vectors
  (tests/golden/agg_golden.npz) were generated from exactly these, and every golden case stores the sha256 of its input, so
  a numpy that produced a different stream is detected instead of silently mis-comparing.
* ``make_resnet18_weights``: torchvision's ResNet init (kaiming-normal, fan_out) from a seeded torch generator — the
  reference ships no embedder checkpoint (download.py:56-57), SURVEY.md §8(d) config 4 fixes seed 11.
* ``load_weights`` / ``build_net``: the aggregator weight sets under ``data/``: the reference's two shipped files
  (example_aggregator_weights/{c16,tcga}_aggregator.pth, testing_c16.py:121 / testing_tcga.py:129) re-packed as ``.npz`` by
  tests/golden/make_golden.py (the ``.pth`` files live in /root/reference, which does not exist on the GPU box), plus seeded
  variants for the other widths / flags of dsmil.py:28-44 (MUSK K = 166, tree K = 1024, nonlinear=False, passing_v=True).
"""
import collections
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

VARIANT = {  # tag -> (K, C, nonlinear, passing_v)
    "c16": (512, 1, True, False), "tcga": (512, 2, True, False), "musk": (166, 1, True, False),
    "tree": (1024, 2, True, False), "linq": (64, 3, False, False), "passv": (64, 2, True, True),
}


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
    import torch
    g = torch.Generator().manual_seed(seed)
    w = collections.OrderedDict()
    for name, cout, cin, k in RESNET18_CONVS:
        std = (2.0 / (cout * k * k)) ** 0.5
        w[name + ".weight"] = torch.randn((cout, cin, k, k), generator=g, dtype=torch.float32) * std
    return w


def load_weights(tag):
    """Aggregator parameters of weight set `tag` as a dict of fp32 numpy arrays (fc_w, fc_b, q0_w, ..., fcc_b)."""
    z = np.load(os.path.join(DATA, f"weights_{tag}.npz"))
    return {k: z[k] for k in z.files}


def state_dict_from_npz(p, nonlinear=True, passing_v=False):
    """Map the flat parameter names back to the reference state_dict keys."""
    import torch
    sd = {"i_classifier.fc.0.weight": p["fc_w"], "i_classifier.fc.0.bias": p["fc_b"],
          "b_classifier.fcc.weight": p["fcc_w"], "b_classifier.fcc.bias": p["fcc_b"]}
    if nonlinear:
        sd.update({"b_classifier.q.0.weight": p["q0_w"], "b_classifier.q.0.bias": p["q0_b"],
                   "b_classifier.q.2.weight": p["q2_w"], "b_classifier.q.2.bias": p["q2_b"]})
    else:
        sd.update({"b_classifier.q.weight": p["q0_w"], "b_classifier.q.bias": p["q0_b"]})
    if passing_v:
        sd.update({"b_classifier.v.1.weight": p["v_w"], "b_classifier.v.1.bias": p["v_b"]})
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def build_net(tag, device="cpu"):
    """MILNet(FCLayer, BClassifier) of weight set `tag`, loaded strictly, in eval mode on `device`."""
    from . import modules as M
    K, C, nonlinear, passing_v = VARIANT[tag]
    net = M.MILNet(M.FCLayer(in_size=K, out_size=C),
                   M.BClassifier(input_size=K, output_class=C, dropout_v=0.0, nonlinear=nonlinear, passing_v=passing_v))
    net.load_state_dict(state_dict_from_npz(load_weights(tag), nonlinear, passing_v), strict=True)
    return net.eval().to(device)
