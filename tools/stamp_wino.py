#!/usr/bin/env python3
"""Step timeline of k_conv_wino_pp from in-kernel s_memtime stamps (trace build only):
    python dsmil-wsi_amd/build.py --variant trace -DDSMIL_EXPERIMENTS -DDSMIL_TRACE
    DSMIL_NATIVE_LIB=libdsmil_hip_trace.so DSMIL_WINO_TRACE=<k> python tools/stamp_wino.py
k = which Winograd launch of the process records (1 = first conv of the first forward; one forward of ResNet-18 has 13).
k_conv_wino_s3 (unit kernel; wave 0 = role 0, last wave = role 1; a step = one chunk): 0 top, 1 after the MFMAs, 2 after
raw_write + raw_load, 3 after barrier 1, 4 after the transform, 5 after barrier 2.
k_conv_wino_pp: multiply wave (role 0) slots: 0 step top, 1 after position 3, 2 after position 7, 3 after the barrier, 4 after the epilogue.
Staging wave (role 1) slots: 0 top, 1 after the transform, 2 after raw_write, 3 after raw_load + statistics, 4 after the barrier.
k_conv_wino_w1 (DSMIL_WINO_KERNEL=w1; waves 0 and 3): slot P = start of pair-block P of the chunk (8 blocks of 12 MFMAs)."""
import _path  # noqa: F401  (repo root on sys.path)
import ctypes
import os
import numpy as np
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import _native, ops
from dsmil_wsi_amd.resnet import resnet18
import torch.nn as nn

torch.manual_seed(0)
res = resnet18(norm_layer=nn.InstanceNorm2d)
res.fc = nn.Identity()
ic = dsmil.IClassifier(res, 512, output_class=1).eval().cuda()
for p in ic.parameters():
    p.requires_grad = False
x = torch.rand(256, 3, 224, 224, device="cuda")
with torch.no_grad():
    ic(x)
torch.cuda.synchronize()
lib = _native.lib()
W = 4 * 2 * 256 * 8
buf = (ctypes.c_ulonglong * W)()
f = lib.dsmil_debug_wino_trace
f.restype = ctypes.c_int
rc = f(buf, W)
assert rc == 0, rc
T = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(4, 2, 256, 8)
if (T[:, :, 255, 3] > 0).any():   # k_conv_wino_w1: slot 255 = kernel start, loop start, epilogue start, end
    for wg in range(4):
        for role in range(2):
            t = T[wg, role, 255]
            if t[3] > 0:
                print(f"wg {wg} wave {'0' if role == 0 else 'last'}: prologue {t[1] - t[0]}, chunk loop {t[2] - t[1]}, epilogue {t[3] - t[2]} ticks")
    T[:, :, 255, :] = 0
for wg in range(2):
    for role, name in ((0, "multiply"), (1, "staging")):
        t = T[wg, role]
        n = int((t[:, 0] > 0).sum())
        if n < 3:
            print("wg", wg, name, "no stamps"); continue
        t = t[:n]
        step = np.diff(t[:, 0])
        print(f"wg {wg} {name}: steps {n}, step period median {np.median(step):.0f} mean {step.mean():.0f} ticks")
        ns = 8 if (t[:, 7] > 0).any() else 6 if (t[:, 5] > 0).any() else 5
        segs = np.diff(t[:, :ns], axis=1)
        print("   segment medians (slot k -> k+1):", np.median(segs, axis=0).astype(int).tolist())
        print("   first 24 step periods:", step[:24].tolist())
