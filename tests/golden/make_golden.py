#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

It imports /root/reference/dsmil.py unmodified, loads the two shipped aggregator weight files
(example_aggregator_weights/{c16,tcga}_aggregator.pth, used by testing_c16.py:121 and
testing_tcga.py:129), feeds seeded synthetic bags and stores
  * the weights re-packed as plain .npz (so that tests and bench can run without the reference),
  * the reference outputs (classes, pred, A, B, critical index),
  * reference autograd gradients of the train_tcga.py:67-71 objective for small bags.
Inputs are not stored: they are regenerated from ``dsmil-wsi_amd/synthetic.py`` and guarded by a sha256.
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(os.path.dirname(HERE)), "dsmil-wsi_amd", "data")   # the re-packed weight sets ship with the package
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
import dsmil as ref  # noqa: E402  (the reference module)
import importlib.util  # noqa: E402
# the seeded input generators live in the package (dsmil-wsi_amd/synthetic.py); loaded by path because `import dsmil` is the
# REFERENCE module in this script
_spec = importlib.util.spec_from_file_location("_synthetic", os.path.join(os.path.dirname(DATA), "synthetic.py"))
_syn = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_syn)
make_bag, make_label = _syn.make_bag, _syn.make_label

torch.set_num_threads(1)
torch.manual_seed(0)

KEYMAP = {
    "i_classifier.fc.0.weight": "fc_w", "i_classifier.fc.0.bias": "fc_b",
    "b_classifier.q.0.weight": "q0_w", "b_classifier.q.0.bias": "q0_b",
    "b_classifier.q.2.weight": "q2_w", "b_classifier.q.2.bias": "q2_b",
    "b_classifier.q.weight": "q0_w", "b_classifier.q.bias": "q0_b",
    "b_classifier.v.1.weight": "v_w", "b_classifier.v.1.bias": "v_b",
    "b_classifier.fcc.weight": "fcc_w", "b_classifier.fcc.bias": "fcc_b",
}


def sd_to_np(sd):
    return {KEYMAP[k]: v.detach().cpu().numpy().astype(np.float32) for k, v in sd.items()}


def build(K, C, nonlinear=True, passing_v=False):
    return ref.MILNet(ref.FCLayer(in_size=K, out_size=C),
                      ref.BClassifier(input_size=K, output_class=C, dropout_v=0.0,
                                      nonlinear=nonlinear, passing_v=passing_v)).eval()


def ortho_init(net, seed):
    """train_tcga.py:229-239 style init (orthogonal weights, zero bias -> here small random
    bias so the bias path is exercised)."""
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, (torch.nn.Linear, torch.nn.Conv1d)):
            torch.nn.init.orthogonal_(m.weight, generator=g)
            with torch.no_grad():
                m.bias.copy_(0.05 * torch.randn(m.bias.shape, generator=g))


def run_fwd(net, x):
    with torch.no_grad():
        xt = torch.from_numpy(x)
        classes, pred, A, B = net(xt)
        _, m_idx = torch.sort(classes, 0, descending=True)
    return dict(classes=classes.numpy(), pred=pred.numpy(), A=A.numpy(), B=B.numpy(),
                idx=m_idx[0].numpy().astype(np.int64))


def run_grad(net, x, label):
    """train_tcga.py:67-72 on one bag."""
    net.zero_grad()
    crit = torch.nn.BCEWithLogitsLoss()
    xt = torch.from_numpy(x)
    yt = torch.from_numpy(label)
    ins, bag, _, _ = net(xt)
    mx, _ = torch.max(ins, 0)
    loss = 0.5 * crit(bag.view(1, -1), yt.view(1, -1)) + 0.5 * crit(mx.view(1, -1), yt.view(1, -1))
    loss.backward()
    out = {"loss": np.float32(loss.item())}
    for k, prm in net.named_parameters():
        out["g_" + KEYMAP[k]] = prm.grad.numpy().copy()
    return out


def sha(x):
    return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()


def main():
    cases = {}
    # ---- shipped weights -------------------------------------------------------------
    for tag, C in (("c16", 1), ("tcga", 2)):
        sd = torch.load(f"/root/reference/example_aggregator_weights/{tag}_aggregator.pth",
                        map_location="cpu")
        net = build(512, C)
        net.load_state_dict(sd, strict=True)
        np.savez(os.path.join(DATA, f"weights_{tag}.npz"), **sd_to_np(sd))
        for N in (1, 2, 37, 128, 500, 2000, 10000):   # 10000 x 512: the exact shape of BASELINE configs[1] / [2]
            seed = 1000 + N
            x = make_bag(seed, N, 512)
            name = f"{tag}_N{N}"
            out = run_fwd(net, x)
            out["x_sha"] = np.array(sha(x))
            out["seed"] = np.int64(seed)
            cases[name] = out
        for N in (5, 200):
            seed = 2000 + N
            x = make_bag(seed, N, 512)
            label = make_label(seed, C)
            out = run_grad(net, x, label)
            out["x_sha"] = np.array(sha(x))
            out["seed"] = np.int64(seed)
            out["label"] = label
            cases[f"{tag}_grad_N{N}"] = out
    # ---- seeded-init variants: MUSK1 width (train_mil.py:129), tree width (README:204),
    #      nonlinear=False and passing_v=True (dsmil.py:33-39) --------------------------
    variants = [
        ("musk", 166, 1, True, False, (3, 40)),
        ("tree", 1024, 2, True, False, (300,)),
        ("linq", 64, 3, False, False, (50,)),
        ("passv", 64, 2, True, True, (50,)),
    ]
    for tag, K, C, nonlinear, passing_v, Ns in variants:
        net = build(K, C, nonlinear, passing_v)
        ortho_init(net, seed={"musk": 11, "tree": 12, "linq": 13, "passv": 14}[tag])
        np.savez(os.path.join(DATA, f"weights_{tag}.npz"), **sd_to_np(net.state_dict()))
        for N in Ns:
            seed = 3000 + N + K
            x = make_bag(seed, N, K)
            out = run_fwd(net, x)
            out["x_sha"] = np.array(sha(x))
            out["seed"] = np.int64(seed)
            cases[f"{tag}_N{N}"] = out
        if nonlinear and not passing_v:
            N = Ns[-1] if Ns[-1] <= 64 else 33
            seed = 4000 + N + K
            x = make_bag(seed, N, K)
            label = make_label(seed, C)
            out = run_grad(net, x, label)
            out["x_sha"] = np.array(sha(x))
            out["seed"] = np.int64(seed)
            out["label"] = label
            cases[f"{tag}_grad_N{N}"] = out
    flat = {}
    for name, d in cases.items():
        for k, v in d.items():
            flat[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "agg_golden.npz"), **flat)
    print("wrote", len(cases), "cases:", ", ".join(sorted(cases)))


if __name__ == "__main__":
    main()
