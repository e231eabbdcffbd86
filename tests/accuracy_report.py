#!/usr/bin/env python3
"""Accuracy of the aggregator forward against the fp64 oracle for the MFMA form selected by
DSMIL_MLP (f32 = v_mfma_f32_32x32x2_f32; s9 / s6 = bf16 MFMA over exact three-plane cuts, 9 or 6
plane products).  Prints one JSON line per (weights, N).  Checker tooling (uses oracle/): lives in tests/."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), HERE]
import agg_oracle as orc  # noqa: E402
from inputs import make_bag  # noqa: E402

import dsmil  # noqa: E402,F401
from dsmil_wsi_amd import ops  # noqa: E402

mode = os.environ.get("DSMIL_MLP", "default")
for tag, N in (("c16", 10000), ("tcga", 10000), ("tcga", 100000), ("tree", 3000), ("musk", 500)):
    p = dict(np.load(os.path.join(os.path.dirname(HERE), "dsmil-wsi_amd", "data", f"weights_{tag}.npz")))
    K = p["q0_w"].shape[1]
    x = make_bag(4242 + N, N, K)
    ref = orc.milnet_forward(x, p, dtype="f64")
    pg = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in p.items()}
    classes, pred, A, B, idx = ops.agg_forward(torch.from_numpy(x).cuda(), [N], pg)
    torch.cuda.synchronize()
    rc, rp, rA, rB = ref[0], ref[1], ref[2], ref[3]
    out = {"mode": mode, "weights": tag, "N": N,
           "pred_abs": float(np.abs(pred.cpu().numpy().astype(np.float64) - rp).max()),
           "pred_scale": float(np.abs(rp).max()),
           "A_rel_to_max": float(np.abs(A.cpu().numpy().astype(np.float64) - rA).max() / rA.max()),
           "B_abs": float(np.abs(B.cpu().numpy().astype(np.float64) - rB).max()),
           "B_scale": float(np.abs(rB).max())}
    print(json.dumps(out))
