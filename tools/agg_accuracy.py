"""Error of the three batch kernels (dsmil_agg_batch_form 0 / 1 / 2) against the fp64 oracle on one ragged batch in the 128-row
regime, bags on different scales: max |A - A_ref| / max A_ref per bag, max |B - B_ref| / max |B_ref|, max |pred - pred_ref|.

    python tools/agg_accuracy.py [c16|tcga]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for d in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import dsmil  # noqa: F401,E402
from dsmil_wsi_amd import ops, _native  # noqa: E402
from dsmil_wsi_amd.synthetic import load_weights  # noqa: E402
from inputs import make_bag  # noqa: E402
import agg_oracle as orc  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "c16"
L = _native.lib()
w = load_weights(tag)
p = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in w.items()}
K = w["q0_w"].shape[1]
lengths = [9000, 7000, 12000, 3000, 8000, 10000, 9500, 8500]
scales = [1.0, 1e-3, 300.0, 1.0, 1.0, 1.0, 1.0, 1.0]
bags = []
for i, (n, sc) in enumerate(zip(lengths, scales)):
    x = make_bag(5000 + i, n, K) * np.float32(sc)
    if i == 4:
        x *= (10.0 ** np.random.default_rng(5).uniform(-2, 2, size=(n, 1))).astype(np.float32)
    bags.append(x)
x = torch.from_numpy(np.concatenate(bags)).cuda()
refs = [orc.milnet_forward(b, w, dtype="f64") for b in bags]
off = np.concatenate([[0], np.cumsum(lengths)])
for form in (0, 1, 2):
    prev = L.dsmil_agg_batch_form(form)
    got = [t.cpu().numpy() for t in ops.agg_forward(x, lengths, p)]
    L.dsmil_agg_batch_form(prev)
    ea = eb = ep = 0.0
    for b, r in enumerate(refs):
        sl = slice(int(off[b]), int(off[b + 1]))
        ea = max(ea, float(np.abs(got[2][sl] - r[2]).max() / np.abs(r[2]).max()))
        eb = max(eb, float(np.abs(got[3][b:b + 1] - r[3]).max() / max(1e-30, np.abs(r[3]).max())))
        ep = max(ep, float(np.abs(got[1][b:b + 1] - r[1]).max() / max(1.0, np.abs(r[1]).max())))
    print(f"form {form}: A {ea:.2e}  B {eb:.2e}  pred {ep:.2e}")
