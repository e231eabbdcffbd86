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
    np.testing.assert_allclose(A.sum(axis=0, dtype=np.float64), 1.0, atol=1e-4)


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


def _oracle_bags(bags, p, nonlinear=True):
    pr = {k: _round_bf16(v) for k, v in p.items()}
    return [orc.milnet_forward(_round_bf16(b), pr, dtype="f64", nonlinear=nonlinear) for b in bags]


@pytest.mark.parametrize("K,C,nonlinear,lengths", [
    (512, 2, True, [1, 2, 127, 128, 129, 255, 256, 257, 1000, 31, 4097] * 6 + [70000]),   # ragged, one long bag
    (512, 1, True, [10000] * 8),                                                          # configs[2]'s shape, C = 1
    (256, 2, True, [3000 + 37 * i for i in range(30)]),                                   # 4 feature chunks
    (256, 1, False, [2500 + 11 * i for i in range(40)]),                                  # linear query
    (512, 2, False, [9000] * 9),                                                          # linear query, two classes
    (64, 2, False, [2500 + 11 * i for i in range(40)]),                                   # ring kernel: 1 chunk, linear query
    (448, 1, False, [9000] * 9),                                                          # ring kernel: 7 chunks
])
def test_bf16_resident_tile_kernel(K, C, nonlinear, lengths):
    """k_attend_bf16_res (agg_res.h): the persistent kernel with the 128-row tile resident in LDS and the query weights
    resident in registers — taken for >= 512 tiles of 128 rows when K is 512 or 256 and C <= 2 (K = 64 / 448 keep the ring
    kernel).  Ragged bags (tiles past a bag's end are skipped by every wave alike, partial last tiles, 1-row bags), linear
    query; oracle and tolerances as above; five runs bit-identical (hand-counted DMA completion, value sum on the matrix
    pipe through transposed LDS reads)."""
    import dsmil_wsi_amd.ops as ops
    rng = np.random.default_rng(K + C)
    p = {"fc_w": rng.standard_normal((C, K), dtype=np.float32) * 0.05, "fc_b": rng.standard_normal(C, dtype=np.float32) * 0.1,
         "q0_w": rng.standard_normal((128, K), dtype=np.float32) * np.float32(1.0 / np.sqrt(K)),
         "q0_b": rng.standard_normal(128, dtype=np.float32) * 0.1,
         "q2_w": rng.standard_normal((128, 128), dtype=np.float32) * np.float32(1.0 / np.sqrt(128)),
         "q2_b": rng.standard_normal(128, dtype=np.float32) * 0.1,
         "fcc_w": rng.standard_normal((C, C, K), dtype=np.float32) * 0.05, "fcc_b": rng.standard_normal(C, dtype=np.float32) * 0.1}
    bags = [make_bag(9000 + i, n, K) for i, n in enumerate(lengths)]
    w = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    x = torch.from_numpy(np.concatenate(bags)).cuda().to(torch.bfloat16)
    assert sum(n // 128 + 1 for n in lengths) >= 512
    out = ops.agg_forward(x, lengths, w, nonlinear=nonlinear)
    torch.cuda.synchronize()
    classes, pred, A, B, idx = [o.float().cpu().numpy() if o.dtype != torch.int64 else o.cpu().numpy() for o in out]
    off = np.concatenate([[0], np.cumsum(lengths)])
    check = sorted(set([0, 1, 2, 5, len(lengths) - 1, len(lengths) // 2]))
    refs = _oracle_bags([bags[i] for i in check], p, nonlinear)
    for i, ref in zip(check, refs):
        sl = slice(off[i], off[i + 1])
        np.testing.assert_allclose(classes[sl], ref[0], atol=1e-4, rtol=1e-5)
        np.testing.assert_allclose(A[sl], ref[2], atol=1e-6, rtol=3e-2)
        np.testing.assert_allclose(B[i].reshape(ref[3].shape), ref[3], atol=2e-3, rtol=2e-2)
        np.testing.assert_allclose(pred[i:i + 1], ref[1], atol=2e-3, rtol=2e-2)
        assert np.array_equal(idx[i], ref[4])
    for i in range(len(lengths)):
        np.testing.assert_allclose(A[off[i]:off[i + 1]].sum(axis=0, dtype=np.float64), 1.0, atol=1e-4)
    for _ in range(4):
        again = ops.agg_forward(x, lengths, w, nonlinear=nonlinear)
        for a_, b_ in zip(again, out):
            assert torch.equal(a_, b_)
    # round 6: the co-resident logits / q_max / combine kernels (dsmil_agg_logits_form 2 = always; they exist for K = 512,
    # C <= 2; the default takes them when calls arrive on several streams) against the kernels every other shape takes
    # (form 0): same arithmetic in the same order, bit for bit
    from dsmil_wsi_amd import _native
    L = _native.lib()
    prev = L.dsmil_agg_logits_form(2)
    try:
        forced = ops.agg_forward(x, lengths, w, nonlinear=nonlinear)
        L.dsmil_agg_logits_form(0)
        plain = ops.agg_forward(x, lengths, w, nonlinear=nonlinear)
        torch.cuda.synchronize()
    finally:
        L.dsmil_agg_logits_form(prev)
    assert prev == 1
    for a_, b_, c_ in zip(forced, plain, out):
        assert torch.equal(a_, c_) and torch.equal(b_, c_)


def test_coresident_passes_on_three_streams_equal_the_serial_result():
    """Round 6: with several streams in flight the logits / q_max / combine kernels of one batch run on the SAME compute
    units as the persistent attend kernel of another (registers and LDS are budgeted for it, agg_res.h).  Sharing a CU
    must not change a bit: three different batches dealt to a three-stream pool, 12 rounds, against their one-stream
    results on the plain kernels."""
    import dsmil_wsi_amd.ops as ops
    from dsmil_wsi_amd import _native
    L = _native.lib()
    w = {k: torch.from_numpy(v).cuda() for k, v in load_weights("tcga").items()}
    lengths = [2400] * 32
    xs = [torch.from_numpy(make_bag(7100 + i, sum(lengths), 512)).cuda().to(torch.bfloat16) for i in range(3)]
    prev = L.dsmil_agg_logits_form(0)
    try:
        want = [[t.clone() for t in ops.agg_forward(x, lengths, w)] for x in xs]
        torch.cuda.synchronize()
    finally:
        L.dsmil_agg_logits_form(prev)
    pool = ops.StreamPool(3)
    got = []
    for r in range(12):
        x = xs[r % 3]
        got.append(pool.run(lambda x=x: [t.clone() for t in ops.agg_forward(x, lengths, w)]))
    pool.join()
    torch.cuda.synchronize()
    for r, g in enumerate(got):
        for a_, b_ in zip(g, want[r % 3]):
            assert torch.equal(a_, b_)
