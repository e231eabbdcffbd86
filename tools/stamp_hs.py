#!/usr/bin/env python3
"""Phase timeline of k_attend_hs from in-kernel s_memtime stamps (trace build, DSMIL_EXPT=64):
  python dsmil-wsi_amd/build.py --variant trace -DDSMIL_EXPERIMENTS -DDSMIL_TRACE
  DSMIL_NATIVE_LIB=libdsmil_hip_trace.so DSMIL_EXPT=64 [DSMIL_NW=1] python tools/stamp_hs.py [bags]"""
import _path  # noqa: F401
import sys
import numpy as np
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops
from dsmil_wsi_amd.synthetic import load_weights
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N, K = 10000, 512
w = {k: torch.from_numpy(v).cuda() for k, v in load_weights("c16").items()}
x = torch.randn(nb * N, K, device="cuda")
for _ in range(3):
    classes, pred, A, B, idx = ops.agg_forward(x, [N] * nb, w)
torch.cuda.synchronize()
a = A.view(torch.int32).cpu().numpy().reshape(nb, N).astype(np.int64) & 0xFFFFFFFF
ntile = N // 64
T = []
for b in range(nb):
    for t in range(ntile):
        r = a[b, t * 64:t * 64 + 64].reshape(32, 2)
        T.append(r[:, 0] | (r[:, 1] << 32))
T = np.array(T).astype(np.float64)
names = {0: "compute entry", 1: "frags of step 0 ready", 8: "s=0", 9: "s=8", 10: "s=16", 11: "s=24", 2: "GEMM-1 main loop done", 12: "last GEMM-1 step starts", 5: "hidden layer exchanged", 6: "2 GEMM-2 steps done", 7: "8 GEMM-2 steps done",
         3: "GEMM 2 done", 4: "tail done", 16: "cutter entry", 17: "cutter prologue issued", 18: "chunk 0 landed", 20: "chunk 1 landed",
         21: "chunk 8 landed", 19: "cutter done"}
t0 = T[:, 0:1]
rel = T - t0
order = [16, 17, 18, 20, 21, 19, 0, 1, 8, 9, 10, 11, 2, 12, 5, 6, 7, 3, 4]
print("tiles:", len(T), " (cycles relative to the compute wave's entry; 100 MHz-class counter ticks if s_memtime is the constant clock)")
for i in order:
    v = rel[:, i]
    print(f"  {names[i]:28s} median {np.median(v):9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f}")
print("kernel span (max end - min start):", T[:, 4].max() - T[:, 0].min())
