"""Phase timeline of k_attend_f2 from s_memtime stamps (experiment build, ABL 32; DSMIL_EXPT=64 skips k_finish so that the stamps
stay in A).  Wave 0 (compute) and wave 4 (cutter) of every full tile.

    DSMIL_NATIVE_LIB=libdsmil_hip_expt.so DSMIL_F2_ABL=32 DSMIL_EXPT=64 python tools/f2_stamps.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dsmil  # noqa: F401,E402
from dsmil_wsi_amd import ops  # noqa: E402
from dsmil_wsi_amd.synthetic import load_weights  # noqa: E402

p = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in load_weights("c16").items()}
n_bags, rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 10000)
x = torch.randn(n_bags * rows, 512, device="cuda")
for _ in range(3):
    out = ops.agg_forward(x, [rows] * n_bags, p)
torch.cuda.synchronize()
A = out[2].cpu().numpy().reshape(-1)
tiles = []
for b in range(n_bags):
    for t in range(rows // 64):
        r0 = b * rows + t * 64
        st = A[r0:r0 + 64].view(np.uint64)
        tiles.append((b, t, st[:16].astype(np.int64), st[16:32].astype(np.int64)))
names_c = ["start", "chunk0 barrier", "step 8", "step 16", "step 24", "GEMM1 end", "B1", "B2", "half0 done", "B4", "GEMM2 end/tail", "tail end"]
names_x = ["start", "chunk0", "chunk3", "chunk7", "chunk11", "chunk15", "barriers B1-4 done", "tail end"]
dc = np.array([np.diff(t[2][:12]) for t in tiles if t[2][0] > 0 and t[2][11] > t[2][0]])
dx = np.array([np.diff(t[3][:8]) for t in tiles if t[3][0] > 0 and t[3][7] > t[3][0]])
tot = np.array([t[2][11] - t[2][0] for t in tiles if t[2][0] > 0 and t[2][11] > t[2][0]])
print(f"{len(dc)} tiles; ticks are s_memtime (100 MHz constant clock: 10 ns per tick)")
print("compute wave 0: median ticks per phase")
for n, v in zip(names_c[1:], np.median(dc, axis=0)):
    print(f"  -> {n:18s} {v:8.0f}")
print(f"  tile total median {np.median(tot):.0f}, mean {tot.mean():.0f}")
print("cutter wave 4: median ticks per phase")
for n, v in zip(names_x[1:], np.median(dx, axis=0)):
    print(f"  -> {n:18s} {v:8.0f}")

