#!/usr/bin/env python3
"""Experiment (round 6): is the fp32 batch kernel (k_attend_f3) bound by the chip's power / clock budget rather than by a pipe?
Runs the 64 x 10 000 x 512 aggregator pass for ~2 s on persistent grids of 64 .. 256 workgroups (dsmil_agg_persistent_grid) while a
thread samples the GPU's sysfs power and shader-clock sensors; prints the attend kernel's own time (HIP events of the library's
profiling channel), the per-CU tile rate, and the median power / clock.     python tools/exp_power.py [agg|bf16|emb]"""
import _path  # noqa: F401
import glob
import statistics
import sys
import threading
import time
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops, _native
from dsmil_wsi_amd.synthetic import load_weights

which = sys.argv[1] if len(sys.argv) > 1 else "agg"
L = _native.lib()
dev = torch.device("cuda:0")


def sensors():
    out = {}
    for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        try:
            out["power_w"] = int(open(p).read()) / 1e6
        except Exception:  # noqa: BLE001
            pass
    for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"):
        try:
            out["sclk_mhz"] = int(open(p).read()) / 1e6
        except Exception:  # noqa: BLE001
            pass
    return out


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.rows = []

    def run(self):
        while not self.stop:
            s = sensors()
            if s:
                self.rows.append(s)
            time.sleep(0.02)

    def med(self, k):
        v = [r[k] for r in self.rows if k in r]
        return statistics.median(v) if v else float("nan")


print("sensors at idle:", sensors(), flush=True)
if which in ("agg", "bf16"):
    w = {k: torch.from_numpy(v).to(dev) for k, v in load_weights("c16" if which == "agg" else "tcga").items()}
    nb, N, K = 64, 10000, 512
    g = torch.Generator(device=dev).manual_seed(1234)
    feats = torch.randn((nb * N, K), generator=g, device=dev)
    if which == "bf16":
        feats = feats.to(torch.bfloat16)
    lengths = [N] * nb
    offsets = ops.offsets_tensor(lengths, dev)
    call = lambda: ops.agg_forward(feats, lengths, w, offsets=offsets)  # noqa: E731
    grids = (256, 224, 192, 160, 128, 96, 64)
else:
    import torch.nn as nn
    from dsmil_wsi_amd.resnet import resnet18
    torch.manual_seed(0)
    res = resnet18(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    ic = dsmil.IClassifier(res, 512, output_class=2).eval().to(dev)
    x = torch.rand(256, 3, 224, 224, device=dev)

    def call():
        with torch.no_grad():
            return ic(x)
    grids = (256,)
for grid in grids:
    L.dsmil_agg_persistent_grid(grid)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    smp = Sampler()
    smp.start()
    t0 = time.perf_counter()
    n = 0
    ev = []
    while time.perf_counter() - t0 < 2.0:
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            call()
            b.record()
            ev.append((a, b))
            n += 1
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    smp.stop = True
    smp.join()
    ms = statistics.median(a.elapsed_time(b) for a, b in ev)
    kern = ""
    if which in ("agg", "bf16"):   # the attend kernel's own time: the library's HIP-event channel 0
        import ctypes
        L.dsmil_profile_enable(1)
        for _ in range(100):
            call()
        torch.cuda.synchronize()
        tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        L.dsmil_profile_collect(0, ctypes.byref(tot), ctypes.byref(cnt))
        L.dsmil_profile_enable(0)
        k_ms = tot.value / max(1, cnt.value)
        kern = f"attend kernel {k_ms:.4f} ms = {k_ms * grid / 256:.4f} ms x 256 / grid (per-CU rate {0.4 / (k_ms * grid / 256):.2f} of a 0.4 ms full-chip launch); "
    print(f"{which} grid {grid}: pass {dt / n * 1e3:.4f} ms wall, {ms:.4f} ms by events; {kern}power {smp.med('power_w'):.0f} W, sclk {smp.med('sclk_mhz'):.0f} MHz "
          f"({len(smp.rows)} samples)", flush=True)
L.dsmil_agg_persistent_grid(256)
