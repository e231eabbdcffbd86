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


# ---- fused training objective + dropout_patches as a row map --------------------------------
@pytest.mark.parametrize("tag,N", [("c16", 5), ("c16", 200), ("tcga", 5), ("tcga", 200), ("musk", 40), ("tree", 33)])
def test_fused_bag_loss_vs_reference_autograd(golden, tag, N):
    """MILNet.bag_loss (dsmil_agg_forward_ex + dsmil_agg_loss_head + dsmil_agg_backward_ex with the SPARSE max-stream
    gradient) against the reference's own loss and autograd gradients for train_tcga.py:67-72 (tests/golden)."""
    from util import build_net
    name = f"{tag}_grad_N{N}"
    net = build_net(tag, "cuda").train()
    x = torch.from_numpy(make_bag(int(golden[f"{name}/seed"]), N, VARIANT[tag][0])).cuda()
    y = torch.from_numpy(golden[f"{name}/label"]).cuda()
    loss, bag, mx = net.bag_loss(x, y)
    loss.backward()
    assert abs(loss.item() - float(golden[f"{name}/loss"])) < 1e-5
    keymap = {"i_classifier.fc.0.weight": "fc_w", "i_classifier.fc.0.bias": "fc_b",
              "b_classifier.q.0.weight": "q0_w", "b_classifier.q.0.bias": "q0_b",
              "b_classifier.q.2.weight": "q2_w", "b_classifier.q.2.bias": "q2_b",
              "b_classifier.fcc.weight": "fcc_w", "b_classifier.fcc.bias": "fcc_b"}
    for k, prm in net.named_parameters():
        ref = golden[f"{name}/g_{keymap[k]}"]
        scale = max(1e-6, float(np.abs(ref).max()))
        np.testing.assert_allclose(prm.grad.cpu().numpy(), ref, atol=1e-4 * scale + 1e-7, rtol=1e-3, err_msg=k)
    # the two extra outputs are the forward's bag logits and the max over instances
    with torch.no_grad():
        ins, bag2, _, _ = net(x)
    assert torch.equal(bag, bag2) and torch.equal(mx, ins.max(0).values)


@pytest.mark.parametrize("tag,N,keep", [("tcga", 3000, 0.7), ("c16", 70000, 0.5), ("musk", 333, 0.9)])
def test_row_map_equals_gathered_rows(tag, N, keep):
    """dropout_patches (train_tcga.py:78-83: `feats[randperm(N)[:int(N*p)]]`) as an index list: forward outputs, loss
    and every parameter gradient equal those of the gathered copy (same kernels, same logical row order)."""
    from util import build_net
    from dsmil_wsi_amd import ops
    K, C = VARIANT[tag][0], VARIANT[tag][1]
    x = torch.from_numpy(make_bag(31 + N, N, K)).cuda()
    g = torch.Generator(device="cuda").manual_seed(N)
    rows = torch.randperm(N, device="cuda", generator=g)[: int(N * keep)]
    p = _params(tag, "cuda", torch.float32)
    a = ops.agg_forward(x, [rows.numel()], p, row_map=rows)
    b = ops.agg_forward(x.index_select(0, rows), [rows.numel()], p)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    y = torch.zeros(C, device="cuda")
    y[0] = 1.0
    grads = []
    for mapped in (True, False):
        net = build_net(tag, "cuda").train()
        if mapped:
            loss, _, _ = net.bag_loss(x, y, rows)
        else:
            loss, _, _ = net.bag_loss(x.index_select(0, rows), y)
        loss.backward()
        grads.append((loss.item(), {k: q.grad.clone() for k, q in net.named_parameters()}))
    assert grads[0][0] == grads[1][0]
    for k in grads[0][1]:
        assert torch.equal(grads[0][1][k], grads[1][1][k]), k


def test_loss_head_matches_torch_bce():
    """dsmil_agg_loss_head against torch's BCEWithLogitsLoss composition (train_tcga.py:68-71), incl. its gradients."""
    from dsmil_wsi_amd import ops
    g = torch.Generator().manual_seed(3)
    for C in (1, 2, 5):
        N = 50
        classes = (torch.randn(N, C, generator=g) * 3).cuda()
        pred = (torch.randn(1, C, generator=g) * 3).cuda().requires_grad_(True)
        y = (torch.rand(C, generator=g) > 0.5).float().cuda()
        idx = classes.argmax(0)
        mxv = classes[idx, torch.arange(C, device="cuda")].detach().requires_grad_(True)
        crit = torch.nn.BCEWithLogitsLoss()
        ref = 0.5 * crit(pred.view(1, -1), y.view(1, -1)) + 0.5 * crit(mxv.view(1, -1), y.view(1, -1))
        ref.backward()
        loss, mx, g_pred, g_max = ops.agg_loss_head(classes, pred.detach(), idx, y)
        assert abs(loss.item() - ref.item()) < 1e-6
        assert torch.equal(mx, mxv.detach())
        np.testing.assert_allclose(g_pred.cpu().numpy(), pred.grad.view(-1).cpu().numpy(), atol=1e-7, rtol=1e-5)
        np.testing.assert_allclose(g_max.cpu().numpy(), mxv.grad.view(-1).cpu().numpy(), atol=1e-7, rtol=1e-5)


def test_train_loop_uses_fused_objective_and_learns():
    """training.train over resident bags with dropout_patch > 0: the fused path (row maps, one progress sync per step)
    reduces the loss on a separable toy set."""
    import argparse
    from dsmil_wsi_amd import training as T
    import dsmil as mil
    rng = np.random.default_rng(0)
    direction = rng.standard_normal(64).astype(np.float32)
    bags = []
    for b in range(16):
        lab = b % 2
        X = rng.standard_normal((150 + 11 * b, 64)).astype(np.float32)
        if lab:
            X[:6] += 2.5 * direction
        bags.append(torch.from_numpy(np.concatenate([X, np.full((X.shape[0], 1), lab, np.float32)], 1)).cuda())
    args = argparse.Namespace(feats_size=64, num_classes=1, dropout_patch=0.3, dropout_node=0.0, non_linearity=1,
                              lr=2e-3, weight_decay=1e-4, num_epochs=8, average=False)
    torch.manual_seed(0)
    np.random.seed(0)   # training.train orders the bags with sklearn.utils.shuffle = numpy's global generator (train_tcga.py:57)
    net, crit, opt, sched = T.init_model(args, mil, torch.device("cuda"))
    losses = [T.train(args, bags, net, crit, opt, log=False) for _ in range(8)]
    assert losses[-1] < 0.8 * losses[0], losses
    tl, score, aucs, th = T.test(args, bags, net, crit, log=False)
    assert aucs[0] > 0.9, aucs   # seeded run (torch + numpy generators above): deterministic; 16 bags, one swapped pair is 1/64 of AUC


def test_adam_kernel_matches_torch_adam():
    """dsmil_adam_step against torch.optim.Adam (the optimiser of train_tcga.py:241) over several steps with weight decay:
    same update to within a few fp32 roundings (the scalar factors are formed in double on both sides; torch runs the
    update as a chain of foreach kernels, this is one fused expression per element)."""
    from dsmil_wsi_amd import ops
    g = torch.Generator().manual_seed(5)
    shapes = [(2, 512), (2,), (128, 512), (128,), (128, 128), (128,), (2, 2, 512), (2,)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    ours = [p.detach().clone() for p in ref]
    m = [torch.zeros_like(p) for p in ours]
    v = [torch.zeros_like(p) for p in ours]
    hp = dict(lr=2e-4, betas=(0.5, 0.9), eps=1e-8, weight_decay=1e-3)
    opt = torch.optim.Adam(ref, **hp)
    for step in range(1, 8):
        grads = [(torch.randn(s, generator=g) * (10.0 ** (step % 3 - 1))).cuda() for s in shapes]
        for p, gr in zip(ref, grads):
            p.grad = gr.clone()
        opt.step()
        ops.adam_step(ours, grads, m, v, step, hp["lr"], hp["betas"], hp["eps"], hp["weight_decay"])
        for a, b in zip(ours, ref):
            np.testing.assert_allclose(a.cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-5, atol=2e-6)
    for a, p in zip(m, ref):
        np.testing.assert_allclose(a.cpu().numpy(), opt.state[p]["exp_avg"].cpu().numpy(), rtol=1e-5, atol=1e-7)
    for a, p in zip(v, ref):
        np.testing.assert_allclose(a.cpu().numpy(), opt.state[p]["exp_avg_sq"].cpu().numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("tag,N,drop", [("tcga", 3000, 0.0), ("c16", 10000, 0.0), ("tcga", 777, 0.4), ("musk", 150, 0.0),
                                        ("linq", 400, 0.25), ("tree", 500, 0.0)])
def test_fused_train_step_follows_the_generic_path(tag, N, drop):
    """training.FusedTrainStep (ONE native call per train_tcga.py:60-75 step: forward + loss + backward + Adam) against the
    generic path — MILNet.bag_loss under autograd, loss.backward(), torch.optim.Adam.step() — from the same start, on the
    same bags and row maps: per-step losses to 1e-5, parameters after 6 steps to 1e-4 of their scale (bit-identical, moments
    included, without dropout), and the optimiser
    state (step count, moments) left consistent for a following generic step."""
    from dsmil_wsi_amd import training as T
    from util import build_net
    K, C, nonlinear, _ = VARIANT[tag]
    nets = [build_net(tag, "cuda").train() for _ in range(2)]
    hp = dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=1e-3)   # a larger lr than the default makes 6 steps move the weights
    opts = [torch.optim.Adam(n.parameters(), **hp) for n in nets]
    crit = torch.nn.BCEWithLogitsLoss()
    fused = T.FusedTrainStep.create(nets[1], crit, opts[1])
    assert fused is not None
    gen = torch.Generator().manual_seed(11)
    for step in range(6):
        x = torch.from_numpy(make_bag(900 + step, N, K)).cuda()
        y = torch.zeros(1, C).cuda()
        y[0, step % C] = float(step % 2) if C == 1 else 1.0
        keep = int(N * (1 - drop))
        rows = torch.randperm(N, generator=gen)[:keep].cuda() if keep < N else None
        opts[0].zero_grad()
        l0, _, _ = T.bag_loss(nets[0], crit, x, y, rows)
        l0.backward()
        opts[0].step()
        l1 = fused(x, y, rows)
        assert abs(l0.item() - l1.item()) < 1e-5 * max(1.0, abs(l0.item())), (step, l0.item(), l1.item())
    fused.sync()
    for (n0, p0), (n1, p1) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        a, b = p0.detach().cpu().numpy(), p1.detach().cpu().numpy()
        np.testing.assert_allclose(b, a, atol=1e-4 * max(1e-3, float(np.abs(a).max())), rtol=0, err_msg=n0)
        s0, s1 = opts[0].state[p0], opts[1].state[p1]
        assert float(s0["step"]) == float(s1["step"]) == 6.0
        np.testing.assert_allclose(s1["exp_avg"].cpu().numpy(), s0["exp_avg"].cpu().numpy(),
                                   atol=2e-4 * max(1e-6, float(s0["exp_avg"].abs().max())), rtol=0, err_msg=n0)
        if drop == 0.0:
            # stronger, on this image's torch: the same kernels form the gradients on both paths and k_bwd_reduce's Adam is
            # torch.optim.Adam's arithmetic operation for operation (explicit fmaf, agg_bwd.hip adam_elem), so parameters and
            # both moments come out BIT-identical (tools/fused_vs_generic.py prints the per-step differences)
            assert np.array_equal(a, b), n0
            assert torch.equal(s0["exp_avg"], s1["exp_avg"]) and torch.equal(s0["exp_avg_sq"], s1["exp_avg_sq"]), n0
    # the inference path sees the updated weights (the packed-weight caches are keyed on the version counters)
    x = torch.from_numpy(make_bag(1, 300, K)).cuda()
    with torch.no_grad():
        o0, o1 = nets[0].eval()(x), nets[1].eval()(x)
    np.testing.assert_allclose(o1[1].cpu().numpy(), o0[1].cpu().numpy(), atol=1e-4)
    # and a generic step after the fused ones continues from the same optimiser state
    nets[1].train()
    opts[1].zero_grad()
    l, _, _ = T.bag_loss(nets[1], crit, x, torch.ones(1, C).cuda())
    l.backward()
    opts[1].step()
    assert float(opts[1].state[next(nets[1].parameters())]["step"]) == 7.0
