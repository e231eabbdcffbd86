"""The numpy oracle (oracle/agg_oracle.py) against vectors produced by the reference itself
(tests/golden/make_golden.py ran /root/reference/dsmil.py).  CPU only."""
import hashlib

import numpy as np
import pytest

import agg_oracle as orc
from conftest import load_weights
from inputs import make_bag

FWD_CASES = [(t, n) for t in ("c16", "tcga") for n in (1, 2, 37, 128, 500, 2000, 10000)] + \
            [("musk", 3), ("musk", 40), ("tree", 300), ("linq", 50), ("passv", 50)]
KDIM = {"c16": 512, "tcga": 512, "musk": 166, "tree": 1024, "linq": 64, "passv": 64}
GRAD_CASES = [("c16", 5), ("c16", 200), ("tcga", 5), ("tcga", 200), ("musk", 40), ("tree", 33)]


def _input(golden, name, K, N):
    x = make_bag(int(golden[f"{name}/seed"]), N, K)
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(golden[f"{name}/x_sha"]), \
        "seeded input stream differs from the one the golden vectors were generated with"
    return x


@pytest.mark.parametrize("tag,N", FWD_CASES)
def test_forward_matches_reference(golden, tag, N):
    name = f"{tag}_N{N}"
    p = load_weights(tag)
    x = _input(golden, name, KDIM[tag], N)
    classes, pred, A, B, idx = orc.milnet_forward(
        x, p, nonlinear=(tag != "linq"), passing_v=(tag == "passv"))
    ref_cls = golden[f"{name}/classes"]
    np.testing.assert_allclose(classes, ref_cls, atol=2e-5, rtol=1e-5)
    # index: the reference's own values at our index must be its column maxima (tie-safe form)
    C = ref_cls.shape[1]
    assert np.array_equal(ref_cls[idx, np.arange(C)], ref_cls.max(axis=0))
    assert np.array_equal(idx, golden[f"{name}/idx"])
    np.testing.assert_allclose(pred, golden[f"{name}/pred"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(A, golden[f"{name}/A"], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(B, golden[f"{name}/B"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(A.sum(axis=0, dtype=np.float64), 1.0, atol=1e-5)   # (fp64 sum: a float32 column sum of N near-equal weights has a systematic rounding bias)


@pytest.mark.parametrize("tag,N", GRAD_CASES)
def test_gradients_match_reference_autograd(golden, tag, N):
    name = f"{tag}_grad_N{N}"
    p = load_weights(tag)
    x = _input(golden, name, KDIM[tag], N)
    loss, g = orc.train_loss_and_grads(x, golden[f"{name}/label"], p, dtype="f64")
    assert abs(loss - float(golden[f"{name}/loss"])) < 2e-6
    for k in ("fc_w", "fc_b", "q0_w", "q0_b", "q2_w", "q2_b", "fcc_w", "fcc_b"):
        ref = golden[f"{name}/g_{k}"]
        scale = max(1e-6, float(np.abs(ref).max()))
        np.testing.assert_allclose(g[k], ref, atol=2e-5 * scale + 1e-8, rtol=2e-4, err_msg=k)


def test_fp64_and_fp32_oracle_agree():
    p = load_weights("tcga")
    x = make_bag(77, 300, 512)
    a = orc.milnet_forward(x, p, dtype="f32")
    b = orc.milnet_forward(x, p, dtype="f64")
    for u, v in zip(a[:4], b[:4]):
        np.testing.assert_allclose(u, v, atol=2e-5, rtol=1e-4)
    assert np.array_equal(a[4], b[4])
