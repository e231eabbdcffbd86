"""Second, independent CPU restatement of the patch-embedder arithmetic in plain numpy (fp64): convolution
as an explicit sum over kernel taps of strided input windows times the tap's [Cout,Cin] matrix, InstanceNorm
from its definition (biased variance over H x W per image and channel, eps inside the square root), max-pool
and average-pool by explicit window walks.  No torch operator is involved, so agreement with
oracle/resnet_oracle.py (which leans on torch's F.conv2d / F.instance_norm / F.max_pool2d CPU kernels) is not
agreement by construction: tests/test_resnet_host.py compares the two on seeded inputs.

TEST INFRASTRUCTURE ONLY — never imported by the product package.

PARITY UNPINNED (like resnet_oracle.py): the reference ships neither embedder weights nor vectors, and its
backbone is torchvision's resnet18 (absent from the image).  What this file follows:
  * wiring: public torchvision ResNet / BasicBlock as constrained by compute_feats.py:146-170 and
    simclr/models/resnet_simclr.py:10,16 (norm_layer = InstanceNorm2d, fc = Identity, child order);
  * nn.InstanceNorm2d defaults: affine=False, track_running_stats=False, eps=1e-5 (biased variance);
  * nn.MaxPool2d(3, 2, 1): padding acts as -inf;  AdaptiveAvgPool2d(1): plain mean over H x W;
  * dsmil.IClassifier (dsmil.py:21-25): feats.view(B,-1), Linear(512, C);
  * the other backbones of compute_feats.py:158-167: resnet34 (BasicBlock, blocks [3,4,6,3]), resnet50 / resnet101
    (Bottleneck, blocks [3,4,6,3] / [3,4,23,3], expansion 4, 2048-d): public torchvision v1.5 Bottleneck — conv1 1x1,
    conv2 3x3 carrying the stride, conv3 1x1, a 1x1 (strided) downsample + norm in the FIRST block of every layer
    (layer 1 too: 64 -> 256 channels);
  * eval-mode nn.BatchNorm2d (compute_feats.py:149-154, --norm_layer batch): (x - running_mean) / sqrt(running_var + eps)
    * weight + bias, eps = 1e-5.
"""
import numpy as np


def conv2d(x, w, stride, pad):
    """x [B,Cin,H,W], w [Cout,Cin,k,k] -> [B,Cout,Ho,Wo];  y[b,o,i,j] = sum_{c,u,v} w[o,c,u,v] x[b,c,i*s+u-p,j*s+v-p]."""
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xp = np.zeros((B, Cin, H + 2 * pad, W + 2 * pad), dtype=x.dtype)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    y = np.zeros((B, Cout, Ho, Wo), dtype=x.dtype)
    for u in range(k):
        for v in range(k):
            win = xp[:, :, u:u + (Ho - 1) * stride + 1:stride, v:v + (Wo - 1) * stride + 1:stride]   # [B,Cin,Ho,Wo]
            y += np.einsum("oc,bcij->boij", w[:, :, u, v], win, optimize=True)
    return y


def instance_norm(x, eps=1e-5):
    mu = x.mean(axis=(2, 3), keepdims=True)
    var = ((x - mu) ** 2).mean(axis=(2, 3), keepdims=True)   # biased
    return (x - mu) / np.sqrt(var + eps)


def max_pool_3x3_s2_p1(x):
    B, C, H, W = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    xp = np.full((B, C, H + 2, W + 2), -np.inf, dtype=x.dtype)
    xp[:, :, 1:1 + H, 1:1 + W] = x
    y = np.full((B, C, Ho, Wo), -np.inf, dtype=x.dtype)
    for u in range(3):
        for v in range(3):
            y = np.maximum(y, xp[:, :, u:u + (Ho - 1) * 2 + 1:2, v:v + (Wo - 1) * 2 + 1:2])
    return y


def relu(x):
    return np.maximum(x, 0)


def basic_block(x, w, prefix, stride, down):
    out = relu(instance_norm(conv2d(x, w[prefix + ".conv1.weight"], stride, 1)))
    out = instance_norm(conv2d(out, w[prefix + ".conv2.weight"], 1, 1))
    idn = instance_norm(conv2d(x, w[prefix + ".downsample.0.weight"], stride, 0)) if down else x
    return relu(out + idn)


def resnet_in_features(x, w, blocks=(2, 2, 2, 2)):
    """x [B,3,H,W] in [0,1], w: name -> ndarray (torchvision names) -> feats [B,512] (fp64)."""
    x = np.asarray(x, np.float64)
    w = {k: np.asarray(v, np.float64) for k, v in w.items()}
    y = max_pool_3x3_s2_p1(relu(instance_norm(conv2d(x, w["conv1.weight"], 2, 3))))
    for li, n in enumerate(blocks, start=1):
        for b in range(n):
            first_down = li > 1 and b == 0
            y = basic_block(y, w, f"layer{li}.{b}", 2 if first_down else 1, first_down)
    return y.mean(axis=(2, 3))


def _norm(x, w, name, kind, eps=1e-5):
    """kind 'instance': per-image statistics; 'batch': the frozen running statistics + affine stored under `name`."""
    if kind == "instance":
        return instance_norm(x, eps)
    sh = (1, -1, 1, 1)
    return ((x - w[name + ".running_mean"].reshape(sh)) / np.sqrt(w[name + ".running_var"].reshape(sh) + eps)
            * w[name + ".weight"].reshape(sh) + w[name + ".bias"].reshape(sh))


def basic_block_n(x, w, prefix, stride, down, kind):
    out = relu(_norm(conv2d(x, w[prefix + ".conv1.weight"], stride, 1), w, prefix + ".bn1", kind))
    out = _norm(conv2d(out, w[prefix + ".conv2.weight"], 1, 1), w, prefix + ".bn2", kind)
    idn = _norm(conv2d(x, w[prefix + ".downsample.0.weight"], stride, 0), w, prefix + ".downsample.1", kind) if down else x
    return relu(out + idn)


def bottleneck_block(x, w, prefix, stride, down, kind):
    out = relu(_norm(conv2d(x, w[prefix + ".conv1.weight"], 1, 0), w, prefix + ".bn1", kind))
    out = relu(_norm(conv2d(out, w[prefix + ".conv2.weight"], stride, 1), w, prefix + ".bn2", kind))   # v1.5: stride on the 3x3
    out = _norm(conv2d(out, w[prefix + ".conv3.weight"], 1, 0), w, prefix + ".bn3", kind)
    idn = _norm(conv2d(x, w[prefix + ".downsample.0.weight"], stride, 0), w, prefix + ".downsample.1", kind) if down else x
    return relu(out + idn)


ARCH = {18: ("basic", (2, 2, 2, 2)), 34: ("basic", (3, 4, 6, 3)), 50: ("bottleneck", (3, 4, 6, 3)), 101: ("bottleneck", (3, 4, 23, 3))}


def resnet_features(x, w, depth=18, kind="instance"):
    """Any of the reference's backbones (compute_feats.py:146-167) with InstanceNorm ('instance') or frozen BatchNorm
    ('batch'): x [B,3,H,W] in [0,1], w: torchvision state-dict names -> ndarray;  -> [B, 512 | 2048] (fp64)."""
    x = np.asarray(x, np.float64)
    w = {k: np.asarray(v, np.float64) for k, v in w.items() if not k.endswith("num_batches_tracked")}
    block, blocks = ARCH[depth]
    y = max_pool_3x3_s2_p1(relu(_norm(conv2d(x, w["conv1.weight"], 2, 3), w, "bn1", kind)))
    for li, n in enumerate(blocks, start=1):
        for b in range(n):
            stride = 2 if (li > 1 and b == 0) else 1
            if block == "basic":
                y = basic_block_n(y, w, f"layer{li}.{b}", stride, li > 1 and b == 0, kind)
            else:
                y = bottleneck_block(y, w, f"layer{li}.{b}", stride, b == 0, kind)
    return y.mean(axis=(2, 3))


def iclassifier_forward(x, w, fc_w, fc_b):
    feats = resnet_in_features(x, w)
    return feats, feats @ np.asarray(fc_w, np.float64).T + np.asarray(fc_b, np.float64)
