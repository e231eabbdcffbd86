"""Ablation timing of k_attend_f2 (experiment build): one process per DSMIL_F2_ABL variant (compile-time ablations: clean code generation), the 64 x 10 000 x 512 batch, attend-kernel
time from the library's own HIP events (dsmil_profile_*).  Timing only — ablated runs compute garbage.

    DSMIL_NATIVE_LIB=libdsmil_hip_expt.so python tools/f2_ablate.py            (driver: spawns the variants)
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

MASKS = [("full", 0), ("variant: nt feature loads", 64), ("variant: weights 7 ahead", 128), ("variant: both", 192), ("variant: refill per two cuts", 256), ("no_weight_loads", 1), ("no_feature_loads", 2), ("no_mfma", 4), ("no_value_sum", 8),
         ("no_cut", 16), ("no_w_no_x", 3), ("no_w_x_mfma", 7), ("nothing", 31)]


def one():
    import ctypes
    import torch
    import dsmil  # noqa: F401
    from dsmil_wsi_amd import ops, _native
    from dsmil_wsi_amd.synthetic import load_weights
    import numpy as np
    L = _native.lib()
    if os.environ.get("DSMIL_BATCH_FORM"):               # 1 = k_attend_f2, 2 = k_attend_f3
        L.dsmil_agg_batch_form(int(os.environ["DSMIL_BATCH_FORM"]))
    p = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in load_weights(os.environ.get("F2_TAG", "c16")).items()}
    n_bags, rows = 64, 10000
    x = torch.randn(n_bags * rows, 512, device="cuda")
    for _ in range(3):
        ops.agg_forward(x, [rows] * n_bags, p)
    torch.cuda.synchronize()
    L.dsmil_profile_enable(1)
    for _ in range(20):
        ops.agg_forward(x, [rows] * n_bags, p)
    torch.cuda.synchronize()
    ms, n = ctypes.c_double(), ctypes.c_int64()
    L.dsmil_profile_collect(0, ctypes.byref(ms), ctypes.byref(n))
    print(f"RESULT {ms.value / max(1, n.value):.4f} ms per attend launch ({n.value} launches)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stagger":   # start skew sweep: DSMIL_EXPT bits 20.. = 1 + units of 64 cycles per eighth
        for units in (0, 8, 16, 24, 32, 48, 64):
            e = dict(os.environ, DSMIL_NATIVE_LIB="libdsmil_hip_expt.so", DSMIL_EXPT=str((units + 1) << 20))
            r = subprocess.run([sys.executable, __file__, "one"], env=e, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            print(f"stagger {units:3d} x 64 cycles per eighth  {line[0] if line else r.stderr[-300:]}", flush=True)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "masks":     # an explicit list: python tools/f2_ablate.py masks 0,512
        MASKS = [(f"mask {m}", int(m)) for m in sys.argv[2].split(",")]
    for name, mask in MASKS:
        e = dict(os.environ, DSMIL_NATIVE_LIB="libdsmil_hip_expt.so", DSMIL_F2_ABL=str(mask))
        r = subprocess.run([sys.executable, __file__, "one"], env=e, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print(f"{name:20s} {mask:#8x}  {line[0] if line else r.stderr[-300:]}", flush=True)
