"""Helper for tests/test_forms_gpu.py: runs in a SUBPROCESS (the MFMA-form knobs DSMIL_MLP / DSMIL_WINO are
read once per process) and checks one small aggregator bag and one small embedder batch against the
oracles.  Prints 'FORM-OK <mlp form> <pred>' on success."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), HERE]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import agg_oracle as orc  # noqa: E402
import resnet_oracle as ro  # noqa: E402
import dsmil  # noqa: E402
import dsmil_wsi_amd._native as nat  # noqa: E402
from conftest import load_weights  # noqa: E402
from dsmil_wsi_amd.resnet import resnet18  # noqa: E402
from inputs import make_bag, make_patches  # noqa: E402
from util import build_net  # noqa: E402

for N in (700, 70000):   # 1-wave and 4-wave tile geometry
    net = build_net("tcga", "cuda")
    x = make_bag(31 + N, N, 512)
    with torch.no_grad():
        classes, pred, A, B = net(torch.from_numpy(x).cuda())
    ref = orc.milnet_forward(x, load_weights("tcga"), dtype="f64")
    np.testing.assert_allclose(classes.cpu().numpy(), ref[0], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(pred.cpu().numpy(), ref[1], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(A.cpu().numpy(), ref[2], atol=1e-6, rtol=1e-3)
    np.testing.assert_allclose(B.cpu().numpy(), ref[3], atol=1e-4, rtol=1e-5)

res = resnet18(norm_layer=nn.InstanceNorm2d)
res.fc = nn.Identity()
w = ro.make_weights(seed=19)
res.load_state_dict(w, strict=True)
for p in res.parameters():
    p.requires_grad = False
ic = dsmil.IClassifier(res, 512, output_class=2).eval()
xp = torch.from_numpy(make_patches(3, 3, 224, 224))
with torch.no_grad():
    rf, rc = ro.iclassifier_forward(xp.double(), {k: v.double() for k, v in w.items()},
                                    ic.fc.weight.double(), ic.fc.bias.double())
    f, c = ic.cuda()(xp.cuda())
np.testing.assert_allclose(f.cpu().numpy(), rf.numpy(), atol=1e-4, rtol=1e-4)
np.testing.assert_allclose(c.cpu().numpy(), rc.numpy(), atol=1e-4, rtol=1e-4)
print("FORM-OK", nat.lib().dsmil_agg_mlp_form(), pred.cpu().numpy().ravel().tolist())
