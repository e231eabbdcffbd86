"""BASELINE config 2 — "TCGA-lung precomputed feats, 2-class DSMIL aggregator bf16": bf16 storage
of features and weights, f32 accumulation.  The reference is fp32 only, so the oracle is the
reference arithmetic (numpy oracle, fp64) fed the SAME bf16-rounded features and parameters.

Stated tolerances: instance logits and the critical index involve only exact bf16 products
accumulated in f32 -> 1e-4 abs / exact.  The query MLP additionally rounds the hidden layer to
bf16 for the second bf16 MFMA (2^-9 relative), which perturbs scores by ~1e-3: attention within
3e-2 relative (+1e-6 abs), bag embedding / bag logits within 2e-2 relative (+2e-3 abs)."""
import numpy as np
import pytest
import torch

import agg_oracle as orc
from conftest import load_weights
from inputs import make_bag
from util import build_net

pytestmark = pytest.mark.gpu


def _round_bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float32).numpy()


def _check(out, ref, N):
    classes, pred, A, B = [o.float().cpu().numpy() for o in out[:4]]
    np.testing.assert_allclose(classes, ref[0], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(A, ref[2], atol=1e-6, rtol=3e-2)
    np.testing.assert_allclose(B.reshape(ref[3].shape), ref[3], atol=2e-3, rtol=2e-2)
    np.testing.assert_allclose(pred, ref[1], atol=2e-3, rtol=2e-2)
    np.testing.assert_allclose(A.sum(axis=0), 1.0, atol=1e-4)


@pytest.mark.parametrize("tag", ["tcga", "c16"])
@pytest.mark.parametrize("N", [1, 37, 500, 2000, 10000])
def test_bf16_storage_path_vs_oracle_on_rounded_values(tag, N):
    import dsmil_wsi_amd.ops as ops
    p = load_weights(tag)
    pr = {k: _round_bf16(v) for k, v in p.items()}
    x = make_bag(600 + N, N, 512)
    xr = _round_bf16(x)
    ref = orc.milnet_forward(xr, pr, dtype="f64")
    w = {k: torch.from_numpy(v).cuda() for k, v in p.items()}          # fp32 master weights: rounded inside
    out = ops.agg_forward(torch.from_numpy(x).cuda().to(torch.bfloat16), [N], w)
    _check(out, ref, N)
    assert np.array_equal(out[4].cpu().numpy()[0], ref[4])


def test_bf16_module_and_varlen_batch():
    """module.bfloat16() + bf16 bags through MILNet.forward / forward_bags; many-bag launch
    (4-wave tiles) equals the per-bag results."""
    net = build_net("tcga", "cuda").to(torch.bfloat16)
    p = {k: _round_bf16(v) for k, v in load_weights("tcga").items()}
    lengths = [3000 + 41 * i for i in range(24)]
    bags = [torch.from_numpy(make_bag(40 + i, n, 512)).cuda().to(torch.bfloat16) for i, n in enumerate(lengths)]
    with torch.no_grad():
        outs = net.forward_bags(bags)
        single = net(bags[5])
    assert single[0].dtype == torch.bfloat16 and single[1].shape == (1, 2)
    for i in (0, 5, 23):
        ref = orc.milnet_forward(bags[i].float().cpu().numpy(), p, dtype="f64")
        _check(outs[i], ref, lengths[i])
    np.testing.assert_allclose(single[1].float().cpu().numpy(), outs[5][1].float().cpu().numpy(), atol=2e-2, rtol=2e-2)


def test_bf16_unsupported_width_raises():
    import dsmil_wsi_amd.ops as ops
    p = load_weights("musk")                                               # K = 166: not a multiple of 8
    w = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    x = torch.from_numpy(make_bag(1, 40, 166)).cuda().to(torch.bfloat16)
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.agg_forward(x, [40], w)


@pytest.mark.parametrize("tag,N,nonlinear", [("tcga", 70000, True), ("c16", 100000, True), ("tree", 70000, True), ("linq", 70000, False)])
def test_bf16_dma_kernel_at_the_wide_launch(tag, N, nonlinear):
    """>= 512 tiles of 128 rows: k_query_attend_bf16_dma (LDS-DMA staged, 64-k steps; K = 64 is a single step, K = 1024
    sixteen, the linear query has no second GEMM).  Same oracle and tolerances as above; ten runs bit-identical (the
    completion counting of the DMA pipeline is done by hand)."""
    import dsmil_wsi_amd.ops as ops
    from util import VARIANT
    K = VARIANT[tag][0]
    p = load_weights(tag)
    pr = {k: _round_bf16(v) for k, v in p.items()}
    x = make_bag(800 + N + K, N, K)
    xr = _round_bf16(x)
    ref = orc.milnet_forward(xr, pr, dtype="f64", nonlinear=nonlinear)
    w = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    xb = torch.from_numpy(x).cuda().to(torch.bfloat16)
    out = ops.agg_forward(xb, [N], w, nonlinear=nonlinear)
    _check(out, ref, N)
    assert np.array_equal(out[4].cpu().numpy()[0], ref[4])
    first = [t.clone() for t in out]
    for _ in range(9):
        again = ops.agg_forward(xb, [N], w, nonlinear=nonlinear)
        for a, b in zip(again, first):
            assert torch.equal(a, b)
