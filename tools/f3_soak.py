"""Soak of k_attend_f3 (dsmil_agg_batch_form 2) against k_query_attend_split (form 0) and itself: random ragged batches in the
128-row regime (bag lengths from 1 row to 20 000, random scales, K in {128, 256, 384, 512}, C in {1, 2}), each run twice.

    python tools/f3_soak.py [rounds]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for d in ("", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import dsmil  # noqa: F401,E402
from dsmil_wsi_amd import ops, _native  # noqa: E402

L = _native.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2026)
worst = {"A": 0.0, "B": 0.0, "pred": 0.0}
for it in range(rounds):
    K = int(rng.choice([128, 256, 384, 512]))
    C = int(rng.choice([1, 2]))
    w = {"fc_w": rng.normal(0, 0.05, (C, K)), "fc_b": rng.normal(0, 0.05, (C,)), "q0_w": rng.normal(0, 0.06, (128, K)),
         "q0_b": rng.normal(0, 0.05, (128,)), "q2_w": rng.normal(0, 0.08, (128, 128)), "q2_b": rng.normal(0, 0.05, (128,)),
         "fcc_w": rng.normal(0, 0.05, (C, C, K)), "fcc_b": rng.normal(0, 0.05, (C,))}
    p = {k: torch.from_numpy(v.astype(np.float32)).cuda() for k, v in w.items()}
    nb = int(rng.integers(8, 40))
    lengths = [int(x) for x in rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 500, 3000, 9000, 12000, 20000], nb)]
    while sum(lengths) // 128 + nb < 512:
        lengths.append(int(rng.choice([9000, 12000, 20000])))
    assert L.dsmil_agg_tile_rows(len(lengths), sum(lengths)) == 128
    x = torch.randn(sum(lengths), K, device="cuda")
    off = np.concatenate([[0], np.cumsum(lengths)])
    for b in range(len(lengths)):   # per-bag scales over six decades
        x[int(off[b]):int(off[b + 1])] *= float(10.0 ** rng.uniform(-3, 3))
    prev = L.dsmil_agg_batch_form(2)
    try:
        g1 = [t.clone() for t in ops.agg_forward(x, lengths, p)]
        g2 = [t.clone() for t in ops.agg_forward(x, lengths, p)]
        L.dsmil_agg_batch_form(0)
        ref = [t.clone() for t in ops.agg_forward(x, lengths, p)]
    finally:
        L.dsmil_agg_batch_form(prev)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b), f"round {it}: two runs differ"
    assert torch.equal(g1[4], ref[4]), f"round {it}: critical instances differ"
    for b in range(len(lengths)):
        sl = slice(int(off[b]), int(off[b + 1]))
        a_ref = ref[2][sl]
        worst["A"] = max(worst["A"], float((g1[2][sl] - a_ref).abs().max() / a_ref.abs().max()))
        sc = max(1e-30, float(ref[3][b].abs().max()))
        worst["B"] = max(worst["B"], float((g1[3][b] - ref[3][b]).abs().max() / sc))
        worst["pred"] = max(worst["pred"], float((g1[1][b] - ref[1][b]).abs().max() / max(1.0, float(ref[1][b].abs().max()))))
    assert all(torch.isfinite(t).all() for t in g1[:4]), f"round {it}: non-finite output"
    print(f"round {it}: K {K} C {C} bags {len(lengths)} rows {sum(lengths)}  worst so far {worst}", flush=True)
assert worst["A"] < 2e-4 and worst["B"] < 2e-5 and worst["pred"] < 2e-5, worst
print("soak ok", worst)
