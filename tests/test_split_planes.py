"""The claim behind the bf16-MFMA forms (csrc/agg_split.h, k_conv_wino_s3): cutting an fp32 value into
three bf16 planes BY TRUNCATION is exact (h + m + l == x, every plane has <= 8 significand bits; for
|x| >= 2^-100, i.e. while no residual is subnormal), and
every plane product is exact in fp32 — so the nine plane products sum to the exact fp32 product.
A numpy restatement of `split3` / `cut4` (no GPU needed)."""
import numpy as np


def cut3(x):
    x = np.asarray(x, np.float32)
    mask = np.uint32(0xFFFF0000)
    h = (x.view(np.uint32) & mask).view(np.float32)
    r1 = (x - h).astype(np.float32)
    m = (r1.view(np.uint32) & mask).view(np.float32)
    r2 = (r1 - m).astype(np.float32)
    low = (r2.view(np.uint32) & mask).view(np.float32)
    return h, m, low, r2


def _values():
    rng = np.random.default_rng(0)
    v = [rng.standard_normal(200000).astype(np.float32),
         (rng.standard_normal(50000) * 1e-20).astype(np.float32),
         (rng.standard_normal(50000) * 1e20).astype(np.float32),
         np.array([0.0, -0.0, 1.0, -1.0, 3.4028235e38, 2.0 ** -100, 255.99998, 1 + 2 ** -23], np.float32),
         rng.integers(0, 2 ** 32, 100000, dtype=np.uint64).astype(np.uint32).view(np.float32)]
    x = np.concatenate(v)
    # the cut is exact as long as no RESIDUAL is subnormal: |x| >= 2^-100 (or 0) keeps all three planes
    # normal; below that the low plane of the (irrelevant, < 1e-30) value is truncated
    return x[np.isfinite(x) & ((np.abs(x) >= 2.0 ** -100) | (x == 0))]


def test_three_plane_cut_is_exact():
    x = _values()
    h, m, low, r2 = cut3(x)
    # residual after two cuts already fits one bf16 plane: truncating it loses nothing
    assert np.array_equal(low.view(np.uint32), r2.view(np.uint32))
    s = h.astype(np.float64) + m.astype(np.float64) + low.astype(np.float64)
    assert np.array_equal(s, x.astype(np.float64))
    for p in (h, m, low):   # each plane is a bf16 value: low 16 bits of the fp32 pattern are zero
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))


def test_nine_plane_products_sum_to_the_exact_product():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(100000).astype(np.float32)
    b = (rng.standard_normal(100000) * 0.05).astype(np.float32)
    pa, pb = cut3(a)[:3], cut3(b)[:3]
    total = np.zeros(a.shape, np.float64)
    for u in pa:
        for w in pb:
            prod32 = (u * w).astype(np.float32)                       # what a bf16 MFMA forms (fp32)
            assert np.array_equal(prod32.astype(np.float64), u.astype(np.float64) * w.astype(np.float64))
            total += prod32.astype(np.float64)
    assert np.array_equal(total, a.astype(np.float64) * b.astype(np.float64))
    # the six-product form leaves out (m,l), (l,m), (l,l): with truncating cuts |m| < 2^-7 |x| and
    # |l| < 2^-14 |x|, so the omission is below 2^-20 of the product (typically ~2^-23)
    six = total - (pa[1].astype(np.float64) * pb[2] + pa[2].astype(np.float64) * pb[1] + pa[2].astype(np.float64) * pb[2])
    exact = a.astype(np.float64) * b.astype(np.float64)
    nz = exact != 0
    assert np.max(np.abs(six[nz] - exact[nz]) / np.abs(exact[nz])) < 2.0 ** -20
