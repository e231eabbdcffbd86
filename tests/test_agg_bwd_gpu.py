"""dsmil_agg_backward (csrc/agg_bwd.hip) through the C-ABI against an fp64 autograd restatement of
dsmil.py:46-62 with DENSE random upstream gradients on every forward output — a stricter check
than the training objective alone (tests/test_agg_gpu.py::test_gradients_vs_reference_autograd
pins the same path to the reference's own autograd through tests/golden).  Needs a real MI355X.

Tolerance: each gradient within 2e-4 of its own max-abs + 2e-5 absolute (the absolute term covers
gradients that are exactly zero in exact arithmetic, e.g. the query stream of a one-instance bag
whose softmax is constant; fp32 MFMA accumulation over up to 70k
instances vs fp64), stated per assert."""
import numpy as np
import pytest
import torch

from conftest import load_weights
from inputs import make_bag
from util import VARIANT

pytestmark = pytest.mark.gpu

KEYS = ("fc_w", "fc_b", "q0_w", "q0_b", "q2_w", "q2_b", "fcc_w", "fcc_b")


def _params(tag, dev, dtype):
    p = load_weights(tag)
    return {k: torch.from_numpy(np.ascontiguousarray(p[k])).to(dev, dtype) for k in KEYS if k in p}


def _autograd_f64(x, vals, p, idx, nonlinear, g):
    """Plain fp64 restatement (CPU) of FCLayer + BClassifier given the critical indices."""
    x = x.double()
    P = {k: v.double().requires_grad_(True) for k, v in p.items()}
    V = vals.double().requires_grad_(True) if vals is not None else x
    c = x @ P["fc_w"].T + P["fc_b"]
    h = x @ P["q0_w"].T + P["q0_b"]
    Q = torch.tanh(torch.relu(h) @ P["q2_w"].T + P["q2_b"]) if nonlinear else h
    s = Q @ Q[idx].T / np.sqrt(128.0)
    A = torch.softmax(s, 0)
    B = A.T @ V
    pred = torch.einsum("ock,ck->o", P["fcc_w"], B) + P["fcc_b"]
    obj = (pred * g["pred"].double()).sum()
    for name, t in (("classes", c), ("A", A), ("B", B)):
        if g.get(name) is not None:
            obj = obj + (t * g[name].double()).sum()
    obj.backward()
    out = {k: v.grad for k, v in P.items() if v.grad is not None}
    if vals is not None:
        out["vals"] = V.grad
    return out


CASES = [  # tag, N, which upstream grads are dense
    ("tcga", 1, "pcAB"), ("tcga", 31, "pcAB"), ("tcga", 33, "pc"), ("tcga", 700, "pcAB"), ("c16", 5000, "pcAB"),
    ("musk", 40, "pcAB"), ("musk", 333, "p"), ("tree", 300, "pcAB"), ("linq", 50, "pcAB"), ("linq", 1000, "pA"),
    ("passv", 50, "pcAB"), ("tcga", 70000, "pcAB"),
]


@pytest.mark.parametrize("tag,N,which", CASES)
def test_backward_dense_upstream(tag, N, which):
    from dsmil_wsi_amd import ops
    K, C, nonlinear, passing_v = VARIANT[tag]
    dev = "cuda"
    rng = np.random.default_rng(77 + N + K)
    x = torch.from_numpy(make_bag(900 + N, N, K))
    p = _params(tag, "cpu", torch.float32)
    vals = None
    if passing_v:  # the value rows are whatever the caller's v layer produced; any matrix serves
        vals = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32))
    pg = {k: v.to(dev) for k, v in p.items()}
    xg = x.to(dev)
    vg = vals.to(dev) if vals is not None else None
    classes, pred, A, B, idx = ops.agg_forward(xg, [N], pg, vals=vg, nonlinear=nonlinear)
    g = {"pred": torch.from_numpy(rng.standard_normal(C).astype(np.float32))}
    if "c" in which:
        g["classes"] = torch.from_numpy(rng.standard_normal((N, C)).astype(np.float32))
    if "A" in which:
        g["A"] = torch.from_numpy(rng.standard_normal((N, C)).astype(np.float32))
    if "B" in which:
        g["B"] = torch.from_numpy(rng.standard_normal((C, K)).astype(np.float32))
    gg = {k: v.to(dev) for k, v in g.items()}
    got = ops.agg_backward(xg, pg, A, B, idx, gg["pred"], g_classes=gg.get("classes"), g_A=gg.get("A"),
                           g_B=gg.get("B"), vals=vg, nonlinear=nonlinear, want_g_vals=passing_v)
    torch.cuda.synchronize()
    ref = _autograd_f64(x, vals, p, idx[0].cpu(), nonlinear, g)
    for k, r in ref.items():
        if k in ("fc_w", "fc_b") and "c" not in which:
            assert k not in got
            continue
        r = r.numpy()
        scale = max(float(np.abs(r).max()), 1e-12)
        err = float(np.abs(got[k].cpu().numpy().astype(np.float64) - r).max())
        assert err <= 2e-4 * scale + 2e-5, f"{tag} N={N} {k}: max err {err:.3e} vs scale {scale:.3e}"


def test_backward_is_deterministic():
    """Fixed-order two-stage reductions: two runs give bit-identical gradients."""
    from dsmil_wsi_amd import ops
    K, C, nonlinear, _ = VARIANT["tcga"]
    N = 3000
    xg = torch.from_numpy(make_bag(5, N, K)).cuda()
    pg = _params("tcga", "cuda", torch.float32)
    classes, pred, A, B, idx = ops.agg_forward(xg, [N], pg, nonlinear=nonlinear)
    gp = torch.ones(C, device="cuda")
    gc = torch.full((N, C), 0.01, device="cuda")
    a = ops.agg_backward(xg, pg, A, B, idx, gp, g_classes=gc, nonlinear=nonlinear)
    a = {k: v.clone() for k, v in a.items()}
    b = ops.agg_backward(xg, pg, A, B, idx, gp, g_classes=gc, nonlinear=nonlinear)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_backward_rejects_bad_workspace():
    from dsmil_wsi_amd import _native
    L = _native.lib()
    assert L.dsmil_agg_backward_workspace_bytes(0, 512, 512, 2) == 0
    assert L.dsmil_agg_backward_workspace_bytes(1000, 512, 512, 2) > 1000 * 128 * 4 * 4


def test_training_step_matches_dense_backward():
    """A MILNet training step (train_tcga.py:60-74) through the native backward equals the same step
    with the dense-product backward (the path taken when the input rows need gradients)."""
    from util import build_net
    from dsmil_wsi_amd.modules import _AggFunction
    crit = torch.nn.BCEWithLogitsLoss()
    y = torch.tensor([[1.0, 0.0]], device="cuda")
    grads = []
    for dense in (False, True):
        net = build_net("tcga", "cuda").train()
        x = torch.from_numpy(make_bag(123, 777, 512)).cuda().requires_grad_(dense)
        ins, bag, _, _ = net(x)
        mx, _ = torch.max(ins, 0)
        loss = 0.5 * crit(bag.view(1, -1), y) + 0.5 * crit(mx.view(1, -1), y)
        loss.backward()
        grads.append({k: p.grad.clone() for k, p in net.named_parameters()})
    for k in grads[0]:
        ref = grads[1][k]
        scale = float(ref.abs().max()) + 1e-12
        assert float((grads[0][k] - ref).abs().max()) <= 2e-4 * scale, k
