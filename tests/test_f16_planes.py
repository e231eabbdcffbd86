"""The claims behind the fp16 two-plane forms of round 5 (csrc/agg_f2.h, csrc/agg_f3.h, the embedder's PlaneProducts<3>), restated
in numpy (no GPU needed):

* x' = x * 2^e with x' in [2^13, 2^14) for the largest element; h0 = rne16(x'), h1 = rne16(x' - h0): x' = h0 + h1 (1 + d),
  |d| <= 2^-11 of h1, i.e. the cut keeps ~22 significand bits while h1 is a normal fp16 number;
* every plane product is exact in fp32 (11 x 11 significand bits), and h0 w0 + h0 w1 + h1 w0 differs from x w by ~2^-22 |x w|;
* a POWER-OF-TWO change of the scale does not change the significands of the planes — so scaling the hidden layer by an upper
  bound of its row maximum (k_attend_f3: max_j ||W1[j]||_1 * 2^14 / row scale + max |b|) instead of the maximum itself gives the
  same products bit for bit as long as no second plane leaves the normal range, and costs < 2^-33 of the row maximum where one
  does;
* the bound is a bound."""
import numpy as np


def f2_scale(m):
    """csrc/agg_f2.h f2_scale: the power of two s with m * s in [2^13, 2^14); 1 for 0 / subnormal / non-finite m."""
    m = np.float32(m)
    e = int((m.view(np.uint32) >> 23) & 0xFF)
    se = 267 - e
    if e == 0 or e == 255:
        se = 127
    se = min(max(se, 2), 252)
    return np.uint32(se << 23).view(np.float32)


def cut2(xs):
    """two fp16 planes by round-to-nearest of already scaled fp32 values"""
    xs = np.asarray(xs, np.float32)
    h0 = xs.astype(np.float16)
    h1 = (xs - h0.astype(np.float32)).astype(np.float16)      # (the subtraction is exact in fp32)
    return h0, h1


def test_two_plane_cut_keeps_22_bits_while_the_second_plane_is_normal():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * 10.0 ** rng.uniform(-3, 3, 200000)).astype(np.float32)
    s = f2_scale(np.abs(x).max())
    xs = (x * s).astype(np.float32)
    assert np.array_equal(xs.astype(np.float64), x.astype(np.float64) * float(s))      # a power of two: exact
    assert 2.0 ** 13 <= np.abs(xs).max() < 2.0 ** 14
    h0, h1 = cut2(xs)
    err = np.abs(h0.astype(np.float64) + h1.astype(np.float64) - xs.astype(np.float64))
    normal = np.abs(h1.astype(np.float64)) >= 2.0 ** -14
    big = normal & (xs != 0)
    assert np.all(err[big] <= np.abs(xs[big].astype(np.float64)) * 2.0 ** -22)
    assert np.all(err <= np.maximum(np.abs(xs.astype(np.float64)) * 2.0 ** -22, 2.0 ** -25))   # below: half a subnormal quantum


def test_three_products_are_exact_in_fp32_and_close_to_the_product():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(100000).astype(np.float32)
    w = (rng.standard_normal(100000) * 0.05).astype(np.float32)
    sx, sw = f2_scale(np.abs(x).max()), f2_scale(np.abs(w).max())
    x0, x1 = cut2(x * sx)
    w0, w1 = cut2(w * sw)
    total = np.zeros(x.shape, np.float64)
    for a, b in ((x0, w1), (x1, w0), (x0, w0)):
        p32 = (a.astype(np.float32) * b.astype(np.float32)).astype(np.float32)           # what the fp16 MFMA forms (fp32)
        assert np.array_equal(p32.astype(np.float64), a.astype(np.float64) * b.astype(np.float64))
        total += p32.astype(np.float64)
    exact = x.astype(np.float64) * w.astype(np.float64) * float(sx) * float(sw)
    nz = np.abs(exact) > 2.0 ** -20 * np.abs(exact).max()
    rel = np.abs(total[nz] - exact[nz]) / np.abs(exact[nz])
    assert rel.max() < 2.0 ** -20 and np.median(rel) < 2.0 ** -22


def test_a_power_of_two_scale_change_leaves_the_planes_unchanged():
    rng = np.random.default_rng(2)
    h = np.maximum(rng.standard_normal((64, 128)).astype(np.float32) * 3.0, 0.0)          # a ReLU'd hidden layer
    for r in range(h.shape[0]):
        true = f2_scale(h[r].max())
        loose = np.float32(float(true) * 2.0 ** -6)                                        # a bound 64x above the maximum
        a0, a1 = cut2(h[r] * true)
        b0, b1 = cut2(h[r] * loose)
        # the first planes agree everywhere, the second wherever it stays a normal fp16 number under the loose scale
        assert np.array_equal(a0.astype(np.float64), b0.astype(np.float64) * 64.0)
        keep = np.abs(b1.astype(np.float64)) >= 2.0 ** -14
        assert np.array_equal(a1.astype(np.float64)[keep], b1.astype(np.float64)[keep] * 64.0)
        lost = np.abs(a1.astype(np.float64) - b1.astype(np.float64) * 64.0)[~keep]
        if lost.size:   # <= half a subnormal quantum of the loose scale, in units of the true scale
            assert lost.max() <= 2.0 ** -25 * 64.0
            assert lost.max() / (float(h[r].max()) * float(true)) < 2.0 ** -32


def test_the_hidden_bound_is_a_bound():
    rng = np.random.default_rng(3)
    K = 512
    W1 = (rng.standard_normal((128, K)) * 0.05).astype(np.float32)
    b1 = (rng.standard_normal(128) * 0.05).astype(np.float32)
    n1 = np.abs(W1.astype(np.float64)).sum(1).max()
    for scale in (1e-3, 1.0, 300.0):
        x = (rng.standard_normal((256, K)) * scale).astype(np.float32)
        x[7] = 0.0
        hid = np.maximum(x.astype(np.float64) @ W1.T.astype(np.float64) + b1, 0.0)
        for r in range(x.shape[0]):
            m = np.abs(x[r]).max()
            sinv = 1.0 / float(f2_scale(m))                     # max |x| < 2^14 / scale
            bound = n1 * 16384.0 * sinv + np.abs(b1).max()
            assert hid[r].max() <= bound
            s = float(f2_scale(np.float32(bound)))
            assert hid[r].max() * s < 2.0 ** 14                 # far from fp16's 65504
