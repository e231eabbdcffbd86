"""CPU oracle for the patch-embedder hot path: ResNet-18 with InstanceNorm2d and fc = Identity,
wrapped as dsmil.IClassifier (reference call sites: compute_feats.py:146-170,211; dsmil.py:14-25).

TEST INFRASTRUCTURE ONLY — never imported by the product package.

PARITY UNPINNED.  The arithmetic of this path lives in torchvision.models.resnet (version
unpinned by the reference: env.yml lists neither torch nor torchvision), and torchvision is not
installed in the build image, nor is its source on disk.  The reference ships no embedder weights
(test/weights/embedder.pth is a download, download.py:56-57), no expected outputs and no tests.
This file is therefore a restatement of the PUBLIC torchvision ResNet-18 / BasicBlock definition
as constrained by the reference's call sites:
  * norm_layer = nn.InstanceNorm2d (affine=False, track_running_stats=False, eps=1e-5)
    compute_feats.py:147, simclr/models/resnet_simclr.py:10
  * all convolutions bias-free; layers [2,2,2,2]; stride on the block's first 3x3 conv;
    1x1 stride-2 ``downsample`` conv + norm in the first block of layers 2-4
  * child order conv1,bn1,relu,maxpool,layer1-4,avgpool (simclr/models/resnet_simclr.py:16)
  * fc = Identity (compute_feats.py:170); IClassifier adds Linear(512,C) (dsmil.py:19,24)
It uses torch's own CPU kernels (F.conv2d, F.instance_norm, F.max_pool2d) in fp32 (or fp64 for
a tighter truth).  What IS pinned: the 20-tensor / shape table of SURVEY.md §2.2 and the state-dict
key order that compute_feats.py:226-231 relies on (tests/test_resnet_host.py).
"""
import torch
import torch.nn.functional as F

# (name, Cout, Cin, k, stride, pad) in torchvision state_dict order — the 20 conv tensors
CONV_TABLE = [("conv1", 64, 3, 7, 2, 3)]
for _li, (_cin, _cout) in enumerate([(64, 64), (64, 128), (128, 256), (256, 512)], start=1):
    _s = 1 if _li == 1 else 2
    CONV_TABLE.append((f"layer{_li}.0.conv1", _cout, _cin, 3, _s, 1))
    CONV_TABLE.append((f"layer{_li}.0.conv2", _cout, _cout, 3, 1, 1))
    if _li > 1:
        CONV_TABLE.append((f"layer{_li}.0.downsample.0", _cout, _cin, 1, _s, 0))
    CONV_TABLE.append((f"layer{_li}.1.conv1", _cout, _cout, 3, 1, 1))
    CONV_TABLE.append((f"layer{_li}.1.conv2", _cout, _cout, 3, 1, 1))
assert len(CONV_TABLE) == 20


def make_weights(seed=11, dtype=torch.float32):
    """Seeded kaiming-normal(fan_out, relu) conv weights — torchvision's ResNet init — as an
    ordered dict name -> [Cout,Cin,k,k] (seed 11)."""
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, cout, cin, k, _s, _p in CONV_TABLE:
        std = (2.0 / (cout * k * k)) ** 0.5
        w[name + ".weight"] = (torch.randn((cout, cin, k, k), generator=g, dtype=torch.float32) * std).to(dtype)
    return w


def _in(x):
    return F.instance_norm(x, eps=1e-5)


def _block(x, w, prefix, stride, down):
    out = F.relu(_in(F.conv2d(x, w[prefix + ".conv1.weight"], stride=stride, padding=1)))
    out = _in(F.conv2d(out, w[prefix + ".conv2.weight"], stride=1, padding=1))
    idn = _in(F.conv2d(x, w[prefix + ".downsample.0.weight"], stride=stride)) if down else x
    return F.relu(out + idn)


def resnet18_in_features(x, w, return_intermediates=False):
    """x [B,3,H,W] in [0,1] -> feats [B,512].  torchvision ResNet._forward_impl order."""
    inter = {}
    y = F.conv2d(x, w["conv1.weight"], stride=2, padding=3)
    inter["conv1_raw"] = y
    y = F.relu(_in(y))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    inter["pool"] = y
    for li in (1, 2, 3, 4):
        y = _block(y, w, f"layer{li}.0", 1 if li == 1 else 2, li > 1)
        inter[f"layer{li}.0"] = y
        y = _block(y, w, f"layer{li}.1", 1, False)
        inter[f"layer{li}.1"] = y
    feats = torch.flatten(F.adaptive_avg_pool2d(y, 1), 1)
    return (feats, inter) if return_intermediates else feats


def iclassifier_forward(x, w, fc_w, fc_b):
    """dsmil.py:21-25 — (feats.view(B,-1), Linear(feats))."""
    feats = resnet18_in_features(x, w)
    return feats, feats @ fc_w.t() + fc_b
