#!/usr/bin/env python3
"""Phase timeline of k_query_attend_split from in-kernel s_memtime stamps.
Needs a trace build of the library:  python dsmil-wsi_amd/build.py --variant trace -DDSMIL_EXPERIMENTS -DDSMIL_TRACE
Run:  DSMIL_NATIVE_LIB=libdsmil_hip_trace.so DSMIL_EXPT=68 python tools/stamp.py   (64 = stamps, 4 = stop after the MLP so the tail does not overwrite them; +8: the XE variant)"""
import _path  # noqa: F401  (repo root on sys.path)
import os
import numpy as np
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops

assert int(os.environ.get("DSMIL_EXPT", "0")) & 64
nb, N, K = 64, 10000, 512
torch.manual_seed(0)
x = torch.randn(nb * N, K, device="cuda")
w = {"fc_w": torch.randn(1, K, device="cuda") * 0.05, "fc_b": torch.zeros(1, device="cuda"),
     "q0_w": torch.randn(128, K, device="cuda") * 0.05, "q0_b": torch.zeros(128, device="cuda"),
     "q2_w": torch.randn(128, 128, device="cuda") * 0.1, "q2_b": torch.zeros(128, device="cuda"),
     "fcc_w": torch.randn(1, 1, K, device="cuda") * 0.05, "fcc_b": torch.zeros(1, device="cuda")}
for _ in range(3):
    classes, pred, A, B, idx = ops.agg_forward(x, [N] * nb, w)
torch.cuda.synchronize()
a = A.view(torch.int32).cpu().numpy().reshape(nb, N).astype(np.int64) & 0xFFFFFFFF
ntile = (N + 127) // 128
T = []
for b in range(nb):
    for t in range(ntile - 1):
        r = a[b, t * 128:t * 128 + 124].reshape(62, 2)
        T.append(r[:, 0] | (r[:, 1] << 32))
T = np.array(T)  # [tiles, 62]
n = 2 + 32 + 8
T = T[:, :n]
d = np.diff(T, axis=1)
print("tiles", len(T), "total cycles per tile (MLP): mean", (T[:, -1] - T[:, 0]).mean(), "median", np.median(T[:, -1] - T[:, 0]))
print("prologue mean", d[:, 0].mean(), "median", np.median(d[:, 0]))
print("GEMM1 step mean", d[:, 1:33].mean(), "median", np.median(d[:, 1:33]), "p90", np.percentile(d[:, 1:33], 90))
print("  even steps", d[:, 1:33:2].mean(), "odd steps", d[:, 2:33:2].mean())
print("GEMM2 step mean", d[:, 33:41].mean(), "first", d[:, 33].mean(), "rest", d[:, 34:41].mean())
print("per-step means:", np.round(d.mean(axis=0)).astype(int).tolist())
good = (np.abs(d) < 1e6).all(axis=1)
dg = d[good]
print("tiles with sane stamps:", int(good.sum()))
print("per-step medians:", np.round(np.median(dg, axis=0)).astype(int).tolist())
print("per-step p10:", np.round(np.percentile(dg, 10, axis=0)).astype(int).tolist())
print("per-step p90:", np.round(np.percentile(dg, 90, axis=0)).astype(int).tolist())
tot = dg.sum(axis=1)
print("tile total (sane): median", np.median(tot), "p10", np.percentile(tot, 10), "p90", np.percentile(tot, 90))
import time
ok = (T[:, -1] - T[:, 0] > 0) & (T[:, -1] - T[:, 0] < 1e6)
print("valid tiles", ok.sum())
Tv = T[ok]
print("span of all tiles (cycles): p0.5 start -> p99.5 end:", np.percentile(Tv[:, -1], 99.5) - np.percentile(Tv[:, 0], 0.5))
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(10):
    ops.agg_forward(x, [N] * nb, w)
ev1.record()
torch.cuda.synchronize()
print("ms per forward (stamped run, all kernels):", ev0.elapsed_time(ev1) / 10)
for i in (0, 1000, 3000):
    print("tile", i, (T[i] - T[i, 0]).tolist())
