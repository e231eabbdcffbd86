"""ResNet-18/34 constructors with torchvision's attribute names and state_dict order.

The reference builds its embedder with ``torchvision.models.resnet18(norm_layer=nn.InstanceNorm2d)``
and replaces ``fc`` by ``nn.Identity()`` (compute_feats.py:146-170); its checkpoint loaders zip
tensors onto the model BY POSITION (compute_feats.py:226-231), so the registration order
conv1, bn1, layer1..4 (block: conv1, bn1, conv2, bn2, downsample.0, downsample.1), fc is part
of the contract.  torchvision is not installed in this image, so the entry points import this
module instead; with InstanceNorm and a CUDA(HIP) input the forward runs in libdsmil_hip.so
(dsmil_resnet18in_forward; eval-mode BatchNorm: dsmil_resnet18bn_forward), otherwise (CPU tensors,
training-mode BatchNorm, ResNet-34, autograd) it runs the plain torch ops of the same graph.
"""
import torch
import torch.nn as nn

from . import ops


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        idn = x if self.downsample is None else self.downsample(x)
        return self.relu(out + idn)


class Bottleneck(nn.Module):
    """torchvision's Bottleneck (ResNet v1.5: the stride sits on the 3x3 conv2)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, stride=1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        idn = x if self.downsample is None else self.downsample(x)
        return self.relu(out + idn)


class ResNet(nn.Module):
    def __init__(self, layers, num_classes=1000, norm_layer=None, block=BasicBlock):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self._norm_layer = norm_layer
        self._block = block
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        block, down = self._block, None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                 self._norm_layer(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, down, self._norm_layer)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, norm_layer=self._norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    # ---- torch-op graph (CPU tensors, BatchNorm, training) --------------------------------
    def _features_torch(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return torch.flatten(self.avgpool(x), 1)

    # ---- native path ----------------------------------------------------------------------
    def _native_trunk(self, x):
        """(convs, bn_norms or None) when this forward can run in libdsmil_hip.so, else None."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3):
            return None
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return None  # embedder training (SimCLR) is outside the hot path
        convs = resnet18_in_convs(self)
        if convs is not None:
            return convs, None
        return resnet18_bn_parts(self)   # eval-mode BatchNorm (`--norm_layer batch`)

    def forward_with_head(self, x, fc_w, fc_b):
        """(feats, Linear(feats)) in one native call sequence — what IClassifier.forward needs."""
        trunk = self._native_trunk(x)
        if trunk is not None and isinstance(self.fc, nn.Identity):
            return ops.resnet18in_forward(x, trunk[0], fc_w, fc_b, bn_norms=trunk[1])
        feats = self.forward(x)
        feats = feats.view(feats.shape[0], -1)
        return feats, torch.nn.functional.linear(feats, fc_w, fc_b)

    def forward(self, x):
        trunk = self._native_trunk(x)
        if trunk is not None:
            feats, _ = ops.resnet18in_forward(x, trunk[0], bn_norms=trunk[1])
        else:
            feats = self._features_torch(x)
        return self.fc(feats)


def _is_plain_instance_norm(m):
    return isinstance(m, nn.InstanceNorm2d) and not m.affine and not m.track_running_stats and abs(m.eps - 1e-5) < 1e-12


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _conv_is(m, k, stride, pad):
    """A plain bias-free nn.Conv2d with exactly the geometry the native kernels implement."""
    return (isinstance(m, nn.Conv2d) and m.bias is None and m.groups == 1 and _pair(m.kernel_size) == (k, k)
            and _pair(m.stride) == (stride, stride) and _pair(m.padding) == (pad, pad) and _pair(m.dilation) == (1, 1)
            and getattr(m, "padding_mode", "zeros") == "zeros")


def _resnet18_parts(model):
    """(convs, norms) of a structurally STOCK BasicBlock ResNet-18 / ResNet-34 module (ours or torchvision's),
    else None.  Besides the tensor shapes, every conv's kernel / stride / padding / dilation / groups / bias,
    the 3x3-s2-p1 max-pool, the ReLUs' presence and the global average pool are checked: a modified trunk
    (maxpool = Identity, a changed stride — common in SimCLR variants) must take the torch graph, not be
    silently computed as the stock architecture."""
    try:
        if not _conv_is(model.conv1, 7, 2, 3):
            return None
        mp = model.maxpool
        if not (isinstance(mp, nn.MaxPool2d) and _pair(mp.kernel_size) == (3, 3) and _pair(mp.stride) == (2, 2)
                and _pair(mp.padding) == (1, 1) and _pair(mp.dilation) == (1, 1) and not mp.ceil_mode):
            return None
        ap = model.avgpool
        if not (isinstance(ap, nn.AdaptiveAvgPool2d) and _pair(ap.output_size) == (1, 1)):
            return None
        if not isinstance(getattr(model, "relu", None), nn.ReLU):
            return None
        convs = [model.conv1.weight]
        norms = [model.bn1]
        for li in (1, 2, 3, 4):
            layer = getattr(model, f"layer{li}")
            for bi, blk in enumerate(layer):
                if not hasattr(blk, "conv2") or not isinstance(getattr(blk, "relu", None), nn.ReLU):
                    return None
                stride = 2 if (li > 1 and bi == 0) else 1
                if hasattr(blk, "conv3"):   # Bottleneck: 1x1, 3x3 (stride), 1x1; downsample in the first block of EVERY layer
                    if not (_conv_is(blk.conv1, 1, 1, 0) and _conv_is(blk.conv2, 3, stride, 1) and _conv_is(blk.conv3, 1, 1, 0)):
                        return None
                    convs += [blk.conv1.weight, blk.conv2.weight, blk.conv3.weight]
                    norms += [blk.bn1, blk.bn2, blk.bn3]
                    need_down = bi == 0
                else:
                    if not (_conv_is(blk.conv1, 3, stride, 1) and _conv_is(blk.conv2, 3, 1, 1)):
                        return None
                    convs += [blk.conv1.weight, blk.conv2.weight]
                    norms += [blk.bn1, blk.bn2]
                    need_down = stride != 1
                if blk.downsample is not None:
                    if len(blk.downsample) != 2 or not _conv_is(blk.downsample[0], 1, stride, 0):
                        return None
                    convs.append(blk.downsample[0].weight)
                    norms.append(blk.downsample[1])
                elif need_down:
                    return None
    except (AttributeError, TypeError):
        return None
    if ops.resnet_depth_of(convs) is None:
        return None
    return convs, norms


def resnet18_in_convs(model):
    """If ``model`` is structurally a ResNet-18 / ResNet-34 with plain InstanceNorm2d everywhere (ours or
    torchvision's), return its conv weights in state_dict order (20 / 36 tensors); else None."""
    parts = _resnet18_parts(model)
    if parts is None or not all(_is_plain_instance_norm(n) for n in parts[1]):
        return None
    return parts[0]


def resnet18_bn_parts(model):
    """If ``model`` is structurally a ResNet-18 whose 20 norms are BatchNorm2d with running statistics
    in EVAL mode (the reference's `--norm_layer batch` extractor under i_classifier.eval(),
    compute_feats.py:149-154), return (convs, norms); else None."""
    parts = _resnet18_parts(model)
    if parts is None:
        return None
    for n in parts[1]:
        if not isinstance(n, nn.BatchNorm2d) or n.training or not n.track_running_stats or n.running_mean is None:
            return None
    return parts


def resnet18(pretrained=False, weights=None, norm_layer=None, **kwargs):
    """torchvision.models.resnet18 signature subset used by the reference
    (compute_feats.py:157 ``pretrained=``, testing_c16.py:113 ``weights=None``)."""
    if pretrained or weights is not None:
        raise ValueError("pretrained ImageNet weights need torchvision and a download; not available offline")
    return ResNet([2, 2, 2, 2], norm_layer=norm_layer, **kwargs)


def resnet34(pretrained=False, weights=None, norm_layer=None, **kwargs):
    if pretrained or weights is not None:
        raise ValueError("pretrained ImageNet weights need torchvision and a download; not available offline")
    return ResNet([3, 4, 6, 3], norm_layer=norm_layer, **kwargs)


def resnet50(pretrained=False, weights=None, norm_layer=None, **kwargs):
    """compute_feats.py:161-163 (`--backbone resnet50`, 2048-d features)."""
    if pretrained or weights is not None:
        raise ValueError("pretrained ImageNet weights need torchvision and a download; not available offline")
    return ResNet([3, 4, 6, 3], norm_layer=norm_layer, block=Bottleneck, **kwargs)


def resnet101(pretrained=False, weights=None, norm_layer=None, **kwargs):
    """compute_feats.py:164-167 (`--backbone resnet101`, 2048-d features)."""
    if pretrained or weights is not None:
        raise ValueError("pretrained ImageNet weights need torchvision and a download; not available offline")
    return ResNet([3, 4, 23, 3], norm_layer=norm_layer, block=Bottleneck, **kwargs)
