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
from conftest import load_weights

assert int(os.environ.get("DSMIL_EXPT", "0")) & 64
nb, N, K = 64, 10000, 512
w = {k: torch.from_numpy(v).cuda() for k, v in load_weights("tcga").items()}
torch.manual_seed(0)
x = torch.randn(nb * N, K, device="cuda").to(torch.bfloat16)
for _ in range(3):
    classes, pred, A, B, idx = ops.agg_forward(x, [N] * nb, w)
torch.cuda.synchronize()
a = A.view(torch.int64).cpu().numpy().reshape(nb, N)       # C = 2: one int64 per row
ntile = N // 128
T = np.array([a[b, t * 128:t * 128 + 64] for b in range(nb) for t in range(ntile)])   # [tiles, 64]: 0..31 compute, 32..63 loader
comp, load = T[:, :27], T[:, 32:32 + 22]
names_c = ["start"] + [f"{'at' if i % 2 == 0 else 'past'} B{i // 2}" for i in range(16)] + ["past B8", "past B9", "at E0", "past E0", "at E1", "past E2", "vs0 done", "past E3", "vs1 done", "past E4"]
dc = np.diff(comp, axis=1)
ok = (np.abs(dc) < 1e6).all(axis=1)
print("tiles", len(T), "sane", int(ok.sum()))
print("compute wave 0: median cycles per phase (tile total median %d)" % np.median((comp[ok, -1] - comp[ok, 0])))
for i in range(dc.shape[1]):
    print(f"  {names_c[i]:>10s} -> {names_c[i + 1]:<10s} median {int(np.median(dc[ok, i])):6d}  p10 {int(np.percentile(dc[ok, i], 10)):6d}  p90 {int(np.percentile(dc[ok, i], 90)):6d}")
dl = np.diff(load, axis=1)
okl = (np.abs(dl) < 1e6).all(axis=1)
names_l = ["start"] + [f"{'landed' if i % 2 == 0 else 'past B'}{i // 2}" for i in range(16)] + ["past E0", "past E2", "past E3", "issued 0,1", "past E4"]
print("feature-stream wave: median cycles per phase")
for i in range(dl.shape[1]):
    print(f"  {names_l[i]:>10s} -> {names_l[i + 1]:<10s} median {int(np.median(dl[okl, i])):6d}  p10 {int(np.percentile(dl[okl, i], 10)):6d}  p90 {int(np.percentile(dl[okl, i], 90)):6d}")
