"""Phase timeline of k_attend_f3 from s_memtime stamps (experiment build, DSMIL_F3_DBG=1; DSMIL_EXPT=64 skips k_finish so that the
stamps stay in A).  Wave 0 of every full 32-row tile.

    DSMIL_NATIVE_LIB=libdsmil_hip_expt.so DSMIL_F3_DBG=1 DSMIL_EXPT=64 python tools/f3_stamps.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dsmil  # noqa: F401,E402
from dsmil_wsi_amd import ops, _native  # noqa: E402
from dsmil_wsi_amd.synthetic import load_weights  # noqa: E402

_native.lib().dsmil_agg_batch_form(2)
p = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in load_weights("c16").items()}
n_bags, rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 10000)
x = torch.randn(n_bags * rows, 512, device="cuda")
for _ in range(3):
    out = ops.agg_forward(x, [rows] * n_bags, p)
torch.cuda.synchronize()
A = out[2].cpu().numpy().reshape(-1)
names = ["tile start", "S barrier", "GEMM 1 (+ cut of the next tile)", "bias/ReLU/row max + B1", "hidden cut + B2", "GEMM 2", "tanh + dots",
         "T1", "stats", "value sum"]
rows_ = []
for b in range(n_bags):
    for t in range(rows // 32):
        r0 = b * rows + t * 32
        st = A[r0:r0 + 32].view(np.uint64)[:14].astype(np.int64)
        if st[0] > 0 and st[9] > st[0]:
            rows_.append(st)
st = np.array(rows_)
d = np.diff(st[:, :10], axis=1)
print(f"{len(st)} tiles")
for n, v in zip(names[1:], np.median(d, axis=0)):
    print(f"  -> {n:34s} {v:8.0f}")
print("  GEMM 1 quarters (steps 0-7, 8-15, 16-23, 24-31):", np.median(np.diff(st[:, [1, 10, 11, 12, 2]], axis=1), axis=0))
print("  value sum end -> flush check done:", np.median(st[:, 13] - st[:, 9]))
print(f"  stamped part of the tile: median {np.median(st[:, 9] - st[:, 0]):.0f}")
# tile-to-tile period: consecutive tiles of one workgroup are consecutive tiles of a bag
per = []
for b in range(n_bags):
    base = b * (rows // 32)
for i in range(1, len(st)):
    dd = st[i][0] - st[i - 1][0]
    if 0 < dd < 100000:
        per.append(dd)
print(f"  tile period (start to start, same workgroup): median {np.median(per):.0f}")
