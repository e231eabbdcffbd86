"""Host-side checks of the embedder boundary (CPU): our ResNet constructor has torchvision's
state_dict names/order (the positional-zip loader of compute_feats.py:226-231 depends on it), and
its torch-op graph equals the oracle restatement."""
import collections

import numpy as np
import pytest
import torch
import torch.nn as nn

import dsmil
import resnet_oracle as ro
from dsmil_wsi_amd.resnet import resnet18, resnet18_in_convs
from inputs import make_patches


def test_state_dict_is_the_20_conv_tensors_in_torchvision_order():
    net = resnet18(pretrained=False, norm_layer=nn.InstanceNorm2d)
    net.fc = nn.Identity()                                     # compute_feats.py:170
    keys = list(net.state_dict().keys())
    assert keys == [n + ".weight" for n, *_ in ro.CONV_TABLE]
    for (name, cout, cin, k, _s, _p), (kk, v) in zip(ro.CONV_TABLE, net.state_dict().items()):
        assert tuple(v.shape) == (cout, cin, k, k), kk
    assert sum(v.numel() for v in net.state_dict().values()) == 11166912


def test_positional_zip_load_like_compute_feats():
    """compute_feats.py:219-233: pop 4 projection-head tensors of the SimCLR checkpoint, zip the
    rest BY POSITION onto IClassifier's keys, load with strict=False."""
    res = resnet18(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    ic = dsmil.IClassifier(res, 512, output_class=2)
    w = ro.make_weights(seed=3)
    simclr = collections.OrderedDict(("features." + str(i), v) for i, v in enumerate(w.values()))
    for n in ("l1.weight", "l1.bias", "l2.weight", "l2.bias"):           # resnet_simclr.py:19-20
        simclr[n] = torch.zeros(1)
    for _ in range(4):
        simclr.popitem()
    new_sd = collections.OrderedDict()
    for (k, v), (k0, _v0) in zip(simclr.items(), ic.state_dict().items()):
        new_sd[k0] = v
    missing = ic.load_state_dict(new_sd, strict=False)
    assert set(missing.missing_keys) == {"fc.weight", "fc.bias"}
    for (name, *_), got in zip(ro.CONV_TABLE, resnet18_in_convs(res)):
        assert torch.equal(got, w[name + ".weight"])


def test_module_graph_equals_oracle_on_cpu():
    res = resnet18(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    w = ro.make_weights(seed=5)
    res.load_state_dict(w, strict=True)
    ic = dsmil.IClassifier(res, 512, output_class=2).eval()
    x = torch.from_numpy(make_patches(1, 2, 64, 64))
    with torch.no_grad():
        feats, c = ic(x)
        ref_f, ref_c = ro.iclassifier_forward(x, w, ic.fc.weight, ic.fc.bias)
    np.testing.assert_allclose(feats.numpy(), ref_f.numpy(), atol=1e-5)
    np.testing.assert_allclose(c.numpy(), ref_c.numpy(), atol=1e-5)
    assert feats.shape == (2, 512) and c.shape == (2, 2)


def test_uint8_nhwc_ingest_equals_to_tensor_on_cpu():
    """Decoded images (uint8 NHWC) give exactly what VF.to_tensor + the fp32 path give
    (compute_feats.py:35-39): HWC->CHW, float32, IEEE division by 255."""
    res = resnet18(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    res.load_state_dict(ro.make_weights(seed=6), strict=True)
    ic = dsmil.IClassifier(res, 512, output_class=2).eval()
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (2, 64, 64, 3), generator=g, dtype=torch.uint8)
    x = img.permute(0, 3, 1, 2).to(torch.float32).div(255)
    with torch.no_grad():
        f8, c8 = ic(img)
        f, c = ic(x)
    assert torch.equal(f8, f) and torch.equal(c8, c)
    with pytest.raises(ValueError):
        ic(img.permute(0, 3, 1, 2).contiguous())


def test_structural_detection_rejects_other_norms():
    assert resnet18_in_convs(resnet18(norm_layer=nn.BatchNorm2d)) is None
    assert resnet18_in_convs(resnet18(norm_layer=nn.InstanceNorm2d)) is not None
    assert resnet18_in_convs(nn.Linear(3, 3)) is None


def test_frozen_batchnorm_detection_and_fold():
    """Eval-mode BatchNorm trunks are recognised (train-mode ones are not) and the (x - m) * r fold that
    dsmil_resnet18bn_forward consumes reproduces nn.BatchNorm2d.eval()."""
    from dsmil_wsi_amd import ops
    from dsmil_wsi_amd.modules import resnet_convs_of
    from dsmil_wsi_amd.resnet import resnet18_bn_parts
    res = resnet18(norm_layer=nn.BatchNorm2d)
    res.fc = nn.Identity()
    assert resnet18_bn_parts(res) is None and resnet_convs_of(res) is None     # training mode
    res.eval()
    convs, norms = resnet_convs_of(res)
    assert len(convs) == 20 and len(norms) == 20 and resnet18_in_convs(res) is None
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for n in norms:
            n.running_mean.normal_(0, 0.3, generator=g)
            n.running_var.uniform_(0.2, 2.0, generator=g)
            n.weight.uniform_(-1.5, 1.5, generator=g)
            n.weight[n.weight.abs() < 0.05] = 0.5
            n.bias.normal_(0, 0.3, generator=g)
    m, r = ops._folded_bn(norms, torch.device("cpu"))
    assert m.numel() == r.numel() == sum(n.num_features for n in norms) == 4800
    o = 0
    for n in norms:
        C = n.num_features
        x = torch.randn(2, C, 3, 3, generator=g)
        with torch.no_grad():
            ref = n(x)
        got = (x - m[o:o + C].view(1, C, 1, 1)) * r[o:o + C].view(1, C, 1, 1)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)
        o += C
    with torch.no_grad():
        norms[3].weight[7] = 0.0
    with pytest.raises(NotImplementedError):
        ops._folded_bn(norms, torch.device("cpu"))


def test_resnet34_is_recognised_with_36_convs():
    from dsmil_wsi_amd import ops
    from dsmil_wsi_amd.modules import resnet_convs_of
    from dsmil_wsi_amd.resnet import resnet34
    res = resnet34(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    convs, norms = resnet_convs_of(res)
    assert norms is None and len(convs) == 36 and ops.resnet_depth_of(convs) == 34
    assert [tuple(w.shape) for w in convs] == ops.resnet_conv_shapes(34)
    assert [k for k in res.state_dict() if "conv" in k or "downsample.0" in k] == \
        [k for k, v in res.state_dict().items() if v.dim() == 4]
    import dsmil_wsi_amd._native as nat
    L = nat.lib()
    assert L.dsmil_resnet_num_convs(18) == 20 and L.dsmil_resnet_num_convs(34) == 36 and L.dsmil_resnet_num_convs(77) == 0
    assert L.dsmil_resnet_norm_channels(18) == 4800
    assert L.dsmil_resnet_packed_bytes(18) == L.dsmil_resnet18_packed_bytes() > 0
    assert L.dsmil_resnet_packed_bytes(34) > L.dsmil_resnet_packed_bytes(18)


@pytest.mark.parametrize("B,H,W", [(2, 96, 96), (1, 125, 131), (2, 64, 70), (1, 250, 250)])
def test_two_independent_restatements_agree(B, H, W):
    """oracle/resnet_oracle.py leans on torch's CPU operators (F.conv2d, F.instance_norm, F.max_pool2d);
    oracle/resnet_numpy.py restates the same network from the definitions in plain numpy (tap-by-tap
    convolution, explicit statistics, window walks).  The embedder oracle is parity-UNPINNED by the reference
    (no vectors ship), so this cross-check is what keeps it from being self-consistent by construction:
    odd sizes included (250 -> 125 -> 63 -> 32 -> 16 -> 8)."""
    import resnet_numpy as rn
    w = ro.make_weights(seed=23)
    x = make_patches(40 + B, B, H, W)
    g = torch.Generator().manual_seed(5)
    fc_w, fc_b = torch.randn((2, 512), generator=g) * 0.1, torch.randn((2,), generator=g) * 0.1
    f_np, c_np = rn.iclassifier_forward(x, {k: v.numpy() for k, v in w.items()}, fc_w.numpy(), fc_b.numpy())
    with torch.no_grad():
        f_t, c_t = ro.iclassifier_forward(torch.from_numpy(x).double(), {k: v.double() for k, v in w.items()},
                                          fc_w.double(), fc_b.double())
        f32, _ = ro.iclassifier_forward(torch.from_numpy(x), w, fc_w, fc_b)
    np.testing.assert_allclose(f_np, f_t.numpy(), atol=1e-11, rtol=1e-10)
    np.testing.assert_allclose(c_np, c_t.numpy(), atol=1e-11, rtol=1e-10)
    # and the fp32 evaluation of the torch restatement sits well inside the 1e-4 parity bar
    np.testing.assert_allclose(f32.numpy(), f_np, atol=2e-5, rtol=1e-4)


def test_numpy_restatement_building_blocks_against_hand_computed_values():
    """Known-answer checks of the numpy pieces themselves (values worked out by hand)."""
    import resnet_numpy as rn
    x = np.arange(16, dtype=np.float64).reshape(1, 1, 4, 4)
    # 3x3 all-ones kernel, stride 1, pad 1: corner = 0+1+4+5, centre (1,1) = sum of the 3x3 block at the origin
    y = rn.conv2d(x, np.ones((1, 1, 3, 3)), 1, 1)
    assert y[0, 0, 0, 0] == 10 and y[0, 0, 1, 1] == 45 and y[0, 0, 3, 3] == 10 + 11 + 14 + 15
    # stride 2, pad 0, 1x1 kernel = subsampling
    assert np.array_equal(rn.conv2d(x, np.full((1, 1, 1, 1), 2.0), 2, 0)[0, 0], 2 * x[0, 0, ::2, ::2])
    # max-pool 3x3 s2 p1 of 0..15: windows centred on (0,0),(0,2),(2,0),(2,2)
    assert np.array_equal(rn.max_pool_3x3_s2_p1(x)[0, 0], np.array([[5., 7.], [13., 15.]]))
    # instance norm: zero mean, biased unit variance (up to eps)
    z = rn.instance_norm(x)
    assert abs(z.mean()) < 1e-12 and abs((z ** 2).mean() - 21.25 / (21.25 + 1e-5)) < 1e-12


@pytest.mark.parametrize("depth,nconv,params", [(50, 53, 23454912), (101, 104, 42394816)])
def test_bottleneck_trunks_are_recognised(depth, nconv, params):
    """`--backbone resnet50|resnet101` (compute_feats.py:161-167): torchvision's Bottleneck wiring (stride on the 3x3
    conv, downsample in the first block of EVERY layer), state_dict order conv1, conv2, conv3, downsample.0; the native
    library agrees on conv count, feature width and per-norm channel total; a modified block falls back to torch."""
    from dsmil_wsi_amd import ops
    from dsmil_wsi_amd.modules import resnet_convs_of
    from dsmil_wsi_amd import resnet as R
    import dsmil_wsi_amd._native as nat
    res = getattr(R, f"resnet{depth}")(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    assert sum(p.numel() for p in res.parameters()) == params          # torchvision's count without the fc layer
    convs, norms = resnet_convs_of(res)
    assert norms is None and len(convs) == nconv and ops.resnet_depth_of(convs) == depth
    assert [tuple(w.shape) for w in convs] == ops.resnet_conv_shapes(depth)
    assert [tuple(v.shape) for v in res.state_dict().values() if v.dim() == 4] == ops.resnet_conv_shapes(depth)
    L = nat.lib()
    assert L.dsmil_resnet_num_convs(depth) == nconv and L.dsmil_resnet_feature_dim(depth) == 2048
    assert L.dsmil_resnet_norm_channels(depth) == sum(s[0] for s in ops.resnet_conv_shapes(depth))
    assert L.dsmil_resnet_workspace_bytes(depth, 4, 224, 224) > L.dsmil_resnet_workspace_bytes(18, 4, 224, 224)
    x = torch.from_numpy(make_patches(3, 1, 64, 64))
    with torch.no_grad():
        assert res(x).shape == (1, 2048)
    res.layer3[0].conv2.stride = (1, 1)                                  # not the stock architecture any more
    assert resnet_convs_of(res) is None


@pytest.mark.parametrize("depth,norm", [(34, "instance"), (34, "batch"), (50, "instance"), (50, "batch"), (101, "instance")])
def test_other_backbones_module_graph_equals_numpy_oracle(depth, norm):
    """The product's resnet34 / resnet50 / resnet101 constructors (dsmil-wsi_amd/resnet.py: BasicBlock / Bottleneck wiring,
    torchvision key names) evaluated in fp64 on the CPU against oracle/resnet_numpy.py — an independent plain-numpy
    restatement of the public torchvision definition (Bottleneck v1.5: stride on conv2, a downsample in the first block of
    every layer).  This is what lets the GPU tests use either as truth."""
    import resnet_numpy as rnp
    from dsmil_wsi_amd import resnet as R
    g = torch.Generator().manual_seed(900 + depth)
    res = getattr(R, f"resnet{depth}")(norm_layer=nn.InstanceNorm2d if norm == "instance" else nn.BatchNorm2d)
    res.fc = nn.Identity()
    with torch.no_grad():
        for m in res.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (m.weight.shape[0] * m.weight.shape[2] ** 2)) ** 0.5)
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_((torch.rand(m.weight.shape, generator=g) * 0.5 + 0.6) *
                               torch.where(torch.rand(m.weight.shape, generator=g) < 0.1, -1.0, 1.0))
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    res.eval()
    x = torch.from_numpy(make_patches(5 + depth, 2, 64, 96))
    ref = rnp.resnet_features(x.numpy(), {k: v.numpy() for k, v in res.state_dict().items()}, depth, norm)
    with torch.no_grad():
        got = res.double()(x.double()).numpy()
    assert got.shape == ref.shape == (2, 512 if depth == 34 else 2048)
    np.testing.assert_allclose(got, ref, atol=1e-10, rtol=1e-10)
