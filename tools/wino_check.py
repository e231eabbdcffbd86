#!/usr/bin/env python3
"""Same-box check of a Winograd kernel variant of the experiment library against the product kernel: the features of a few
seeded batches are computed in two subprocesses (the knobs are read once per process) and compared bit for bit, then the
embedder leg of bench.py is timed both ways.
Usage (GPU box): DSMIL_NATIVE_LIB=libdsmil_hip_expt.so python tools/wino_check.py [K=V ...]
base = DSMIL_WINO_KERNEL=unit (k_conv_wino_s3 everywhere) unless the first argument is base:K=V,...; variant = the library's
default (k_conv_wino_w1 on the 128-cout layers, max-pool fused into the stem) plus the given knobs."""
import os
import subprocess
import sys

import _path  # noqa: F401
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(5, 96, 96), (3, 224, 224), (33, 64, 160), (2, 225, 231), (4, 250, 250), (64, 224, 224)]


def run(out):
    import torch
    import torch.nn as nn
    import dsmil
    from dsmil_wsi_amd.resnet import resnet18
    from dsmil_wsi_amd.synthetic import make_patches, make_resnet18_weights
    res = resnet18(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    res.load_state_dict(make_resnet18_weights(11), strict=True)
    ic = dsmil.IClassifier(res, 512, output_class=2).eval().cuda()
    outs = []
    with torch.no_grad():
        for i, (b, h, w) in enumerate(SHAPES):
            f, _ = ic(torch.from_numpy(make_patches(100 + i, b, h, w)).cuda())
            outs.append(f.float().cpu().numpy().ravel())
    torch.cuda.synchronize()
    np.save(out, np.concatenate(outs))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "against":   # against <ref.npy>: the first, third-from-last ... = the 4 shapes of an older reference file
        SHAPES[:] = [(5, 96, 96), (3, 224, 224), (33, 64, 160), (64, 224, 224)]
        o = os.path.join(ROOT, "gpurun_out", "wino_check_now.npy")
        os.makedirs(os.path.dirname(o), exist_ok=True)
        run(o)
        a, b = np.load(sys.argv[2]), np.load(o)
        print("against", sys.argv[2], "bit-identical", bool(np.array_equal(a, b)), "max abs diff %.3e" % float(np.abs(a - b).max()))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "run":
        run(sys.argv[2])
        sys.exit(0)
    base = {"DSMIL_WINO_KERNEL": "unit"}
    args = sys.argv[1:]
    if args and args[0].startswith("base:"):   # another baseline: base:K=V,K=V (e.g. base:DSMIL_STEM_FUSE=0)
        base = dict(x.split("=", 1) for x in args[0][5:].split(",") if x)
        args = args[1:]
    knobs = dict(a.split("=", 1) for a in args)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    outs = []
    for tag, env in (("base", base), ("var", knobs)):
        e = dict(os.environ)
        e.update(env)
        o = os.path.join(ROOT, "gpurun_out", f"wino_check_{tag}.npy")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "run", o], env=e, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print(tag, "FAILED", r.stderr[-2000:])
            sys.exit(1)
        outs.append(np.load(o))
    a, b = outs
    print("features:", a.shape, "finite", bool(np.isfinite(b).all()), "bit-identical", bool(np.array_equal(a, b)),
          "max abs diff %.3e" % float(np.abs(a - b).max()), "max |base| %.3f" % float(np.abs(a).max()), flush=True)
    spec = ",".join(f"{k}={v}" for k, v in knobs.items())
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "variants.py"), "embedder", "base:" + ",".join(f"{k}={v}" for k, v in base.items()), "var:" + spec])
