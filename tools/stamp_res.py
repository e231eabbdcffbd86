#!/usr/bin/env python3
"""Phase timeline of k_attend_bf16_res from in-kernel s_memtime stamps (trace build:
python dsmil-wsi_amd/build.py --variant trace -DDSMIL_EXPERIMENTS -DDSMIL_TRACE).
Run: DSMIL_NATIVE_LIB=libdsmil_hip_trace.so DSMIL_EXPT=64 python tools/stamp_res.py"""
import _path  # noqa: F401
import os
import numpy as np
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops
from dsmil_wsi_amd.synthetic import load_weights  # noqa: E402

assert int(os.environ.get("DSMIL_EXPT", "0")) & 64
nb, N, K = 64, 10000, 512
w = {k: torch.from_numpy(v).cuda() for k, v in load_weights("tcga").items()}
torch.manual_seed(0)
x = torch.randn(nb * N, K, device="cuda").to(torch.bfloat16)
for _ in range(3):
    classes, pred, A, B, idx = ops.agg_forward(x, [N] * nb, w)
torch.cuda.synchronize()
a = A.view(torch.int64).cpu().numpy().reshape(nb, N)       # C = 2: one int64 per row
ntile = N // 64
T = np.array([a[b, t * 64:t * 64 + 8] for b in range(nb) for t in range(ntile)])   # [tiles, 8] stamps of wave 0
names = ["start", "S (tile landed)", "GEMM1 done", "E0 (H exchange + GEMM2)", "E1 (tanh+partial scores)", "E3 (softmax)", "value sum done", "past T"]
d = np.diff(T, axis=1)
ok = (np.abs(d) < 1e6).all(axis=1)
print("tiles", len(T), "sane", int(ok.sum()), "tile total median", int(np.median(T[ok, -1] - T[ok, 0])))
for i in range(d.shape[1]):
    print(f"  {names[i]:>26s} -> {names[i + 1]:<26s} median {int(np.median(d[ok, i])):6d}  p10 {int(np.percentile(d[ok, i], 10)):6d}  p90 {int(np.percentile(d[ok, i], 90)):6d}")
