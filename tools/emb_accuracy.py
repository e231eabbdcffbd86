"""Feature error of the embedder's MFMA forms on the GPU against the fp64 restatement (oracle/resnet_oracle.py): one JSON
line per form.  Checker tooling (uses oracle/; never imported by the product).

    python tools/emb_accuracy.py                 spawns one process per form (experiment build: DSMIL_WINO / DSMIL_CONV)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]

FORMS = [("h3 (product: fp16 x 2 planes, 3 products)", {}),
         ("s6 (bf16 x 3 planes, 6 products)", {"DSMIL_WINO": "s6", "DSMIL_CONV": "s6"}),
         ("s9 / s6 direct (bf16 x 3 planes, 9 products in the Winograd convs)", {"DSMIL_WINO": "s9", "DSMIL_CONV": "s6"}),
         ("f32 MFMA", {"DSMIL_WINO": "f32", "DSMIL_CONV": "f32"})]


def one(name):
    import numpy as np
    import torch
    import torch.nn as nn
    import dsmil
    import resnet_oracle as ro
    from dsmil_wsi_amd.resnet import resnet18
    from dsmil_wsi_amd.synthetic import make_patches
    res = resnet18(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    w = ro.make_weights(seed=11)
    res.load_state_dict(w, strict=True)
    for p in res.parameters():
        p.requires_grad = False
    ic = dsmil.IClassifier(res, 512, output_class=2).eval()
    xp = torch.from_numpy(make_patches(5, 16, 224, 224))
    with torch.no_grad():
        rf, _ = ro.iclassifier_forward(xp.double(), {k: v.double() for k, v in w.items()}, ic.fc.weight.double(), ic.fc.bias.double())
        f, _ = ic.cuda()(xp.cuda())
    d = (f.cpu().double() - rf).abs().flatten()
    print("RESULT " + json.dumps({"form": name, "patches": 16, "max_abs": float(d.max()), "p999_abs": float(torch.quantile(d, 0.999)),
                                  "mean_abs": float(d.mean()), "feature_scale": float(rf.abs().max()), "bar": 1e-4}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "one":
        one(sys.argv[2])
        sys.exit(0)
    for name, env in FORMS:
        e = dict(os.environ, **env)
        if env:
            e["DSMIL_NATIVE_LIB"] = "libdsmil_hip_expt.so"
        r = subprocess.run([sys.executable, __file__, "one", name], env=e, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        print(line[0][7:] if line else json.dumps({"form": name, "error": r.stderr[-400:]}), flush=True)
