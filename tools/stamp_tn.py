#!/usr/bin/env python3
"""Step timeline of k_tn_split from in-kernel s_memtime stamps (trace build only):
    python dsmil-wsi_amd/build.py --variant trace -DDSMIL_EXPERIMENTS -DDSMIL_TRACE
    DSMIL_NATIVE_LIB=libdsmil_hip_trace.so python tools/stamp_tn.py [classes]
Stamps of thread 0 of every workgroup: 0 entry, 1 first loads issued, 2.. one per 32-row step, 15 epilogue stores issued
(s_memtime ticks: the 100 MHz constant clock)."""
import _path  # noqa: F401
import ctypes
import sys
import numpy as np
import torch
import train_fused  # noqa: F401  (runs a few fused training steps on 10 000 x 512 bags)
from dsmil_wsi_amd import _native
lib = _native.lib()
W = 1024 * 16
buf = (ctypes.c_ulonglong * W)()
lib.dsmil_debug_tn_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.dsmil_debug_tn_trace(buf, W) == 0
T = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.float64)
T = T[T[:, 0] > 0]
print("workgroups with stamps:", len(T))
t0 = T[:, 0].min()
for i in range(16):
    v = T[:, i]
    ok = v > 0
    if ok.sum() == 0:
        continue
    d = (v[ok] - T[ok, 0])
    print(f"stamp {i:2d}: n {ok.sum():4d}  since own entry: median {np.median(d):7.0f} p10 {np.percentile(d, 10):7.0f} p90 {np.percentile(d, 90):7.0f}   | since first entry: median {np.median(v[ok] - t0):7.0f}")
print("entry spread (last - first workgroup entry):", T[:, 0].max() - t0, " kernel span:", T[:, 15].max() - t0)
