"""Parity of the HIP patch embedder (ResNet-18 + InstanceNorm, through the C-ABI) with the CPU
oracle restatement (oracle/resnet_oracle.py; PARITY UNPINNED by the reference, see its header).
Tolerance: 1e-4 abs on the 512-d features (= the quantum of the reference's '%.4f' CSV,
compute_feats.py:82) and on the instance logits."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import dsmil
import resnet_numpy as rnp
import resnet_oracle as ro
from dsmil_wsi_amd.resnet import resnet18
from inputs import make_patches

pytestmark = pytest.mark.gpu


def _build(seed, C=2):
    res = resnet18(pretrained=False, norm_layer=nn.InstanceNorm2d)
    for p in res.parameters():
        p.requires_grad = False                                # compute_feats.py:168-169
    res.fc = nn.Identity()
    w = ro.make_weights(seed=seed)
    res.load_state_dict(w, strict=True)
    ic = dsmil.IClassifier(res, 512, output_class=C)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        ic.fc.weight.copy_(torch.randn(ic.fc.weight.shape, generator=g) * 0.1)
        ic.fc.bias.copy_(torch.randn(ic.fc.bias.shape, generator=g) * 0.1)
    return ic.eval(), w


def _ref(x, w, ic, dtype=torch.float64):
    w64 = {k: v.to(dtype) for k, v in w.items()}
    f, c = ro.iclassifier_forward(x.cpu().to(dtype), w64, ic.fc.weight.detach().cpu().to(dtype),
                                  ic.fc.bias.detach().cpu().to(dtype))
    return f.numpy(), c.numpy()


@pytest.mark.parametrize("B,H,W", [(1, 224, 224), (3, 224, 224), (5, 224, 224), (2, 256, 256), (4, 96, 96), (2, 160, 224)])
def test_embedder_vs_oracle(B, H, W):
    ic, w = _build(seed=11)
    x = torch.from_numpy(make_patches(7 + B, B, H, W))
    ref_f, ref_c = _ref(x, w, ic)
    icg = ic.cuda()
    with torch.no_grad():
        feats, c = icg(x.cuda())
    assert feats.shape == (B, 512) and c.shape == (B, 2)
    np.testing.assert_allclose(feats.cpu().numpy(), ref_f, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(c.cpu().numpy(), ref_c, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("B,H,W,u8", [(5, 224, 224, False), (33, 224, 224, True), (2, 225, 231, False)])
def test_opt_in_half_precision_path_within_its_stated_tolerance(B, H, W, u8):
    """IClassifier.embed_precision = "half" (compute_feats.py --precision half): on this ResNet-18 InstanceNorm trunk the
    fp16-ACTIVATION trunk of round 6 (dsmil_resnet_forward_ex, precision = 3: fp16 activations behind the stem, one fp16 MFMA
    product per MAC, f32 accumulation and norm statistics; rounds 5: precision = 1, fp32 activations and one fp16 plane per
    conv operand — still what other trunks take).  NOT the 1e-4 parity path: the bar stated in include/dsmil_hip.h is 5e-3 abs
    on features of O(1) (measured 2.6e-3); the default path of the same module stays at 1e-4, and the two differ (the switch
    does something)."""
    ic, w = _build(seed=11)
    x = torch.from_numpy(make_patches(40 + B, B, H, W))
    ref_f, ref_c = _ref(x, w, ic)
    icg = ic.cuda()
    xin = (x * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().cuda() if u8 else x.cuda()
    if u8:
        ref_f, ref_c = _ref(xin.cpu().permute(0, 3, 1, 2).to(torch.float32).div(255), w, ic)
    with torch.no_grad():
        f32, c32 = icg(xin)
        icg.embed_precision = "half"
        fh, ch = icg(xin)
        icg.embed_precision = "fp32"
    np.testing.assert_allclose(f32.cpu().numpy(), ref_f, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(fh.cpu().numpy(), ref_f, atol=5e-3, rtol=0)
    np.testing.assert_allclose(ch.cpu().numpy(), ref_c, atol=5e-3, rtol=0)
    assert float((fh - f32).abs().max()) > 1e-5


@pytest.mark.parametrize("B,H,W,u8", [(5, 224, 224, False), (33, 224, 224, True), (256, 224, 224, False), (2, 225, 231, False),
                                      (3, 96, 128, False), (2, 128, 160, False)])
def test_opt_in_bf16_activation_trunk_within_its_stated_tolerance(B, H, W, u8):
    """IClassifier.embed_precision = "bf16" (dsmil_resnet_forward_ex, precision = 2; compute_feats.py --precision bf16; round 6):
    behind the stem every activation is STORED in bf16 (NHWC with a one-pixel zero border) and every conv is one bf16 MFMA
    product per MAC with f32 accumulation, f32 InstanceNorm statistics (csrc/resnet_b16.h).  NOT the 1e-4 parity path: the bar
    stated in include/dsmil_hip.h is bf16 rounding through 16 conv + norm layers — max 5e-2 abs, mean 8e-3 abs on features of
    O(1) against the fp64 oracle (measured: 2e-2 / 3e-3).  Batch sizes whose 256- / 512-position tiles straddle images, odd and
    small patch sizes (225 x 231 -> 57 x 58 -> ... -> 8 x 8; 96 x 128 -> 24 x 32 -> 3 x 4: below that an InstanceNorm over a
    handful of bf16 pixels amplifies the rounding — 64 x 64 patches, 2 x 2 maps in layer 4, reach 0.12), uint8 input.  Two runs are bit-identical
    (no atomics: the statistics are fixed-order partial sums)."""
    ic, w = _build(seed=11)
    x = torch.from_numpy(make_patches(40 + B, B, H, W))
    icg = ic.cuda()
    xin = (x * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().cuda() if u8 else x.cuda()
    xr = xin.cpu().permute(0, 3, 1, 2).to(torch.float32).div(255) if u8 else x
    sub = list(range(B)) if B <= 33 else [0, 1, 7, 100, 255]          # (the oracle runs in fp64 on the CPU)
    ref_f, ref_c = _ref(xr[sub], w, ic)
    with torch.no_grad():
        icg.embed_precision = "bf16"
        try:
            fb, cb = icg(xin)
            fb2, _ = icg(xin)
        finally:
            icg.embed_precision = "fp32"
    assert fb.shape == (B, 512) and torch.isfinite(fb).all() and torch.equal(fb, fb2)
    err = np.abs(fb.cpu().numpy()[sub] - ref_f)
    assert err.max() < 5e-2 and err.mean() < 8e-3, (err.max(), err.mean())
    assert np.abs(cb.cpu().numpy()[sub] - ref_c).max() < 5e-2
    assert err.max() > 1e-4          # (the switch does something)


def test_bf16_activation_trunk_resnet34_and_unsupported_trunks():
    """precision "bf16" on `--backbone resnet34` (blocks [3,4,6,3]) against oracle/resnet_numpy.py; a frozen-BatchNorm trunk
    and a Bottleneck trunk are refused (ValueError), not silently run at another precision."""
    from dsmil_wsi_amd.resnet import resnet34, resnet50
    import dsmil_wsi_amd.ops as ops
    from dsmil_wsi_amd.modules import resnet_convs_of
    g = torch.Generator().manual_seed(41)
    res = resnet34(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    with torch.no_grad():
        for m in res.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (m.weight.shape[0] * m.weight.shape[2] ** 2)) ** 0.5)
    ic = dsmil.IClassifier(res, 512, output_class=2).eval()
    for p in ic.parameters():
        p.requires_grad = False
    x = torch.from_numpy(make_patches(78, 3, 224, 224))
    rf = rnp.resnet_features(x.numpy(), {k: v.numpy() for k, v in res.state_dict().items()}, 34, "instance")
    icg = ic.cuda()
    icg.embed_precision = "bf16"
    with torch.no_grad():
        f, _ = icg(x.cuda())
    err = np.abs(f.cpu().numpy() - rf)
    # 32 conv + norm layers instead of 16; tools/b16_soak.py over random weights and shapes: max 3.1e-2 .. 5.7e-2, mean 7.2e-3 .. 1.23e-2
    assert err.max() < 1e-1 and err.mean() < 2e-2, (err.max(), err.mean())
    # refused: frozen BatchNorm, Bottleneck
    icb = _build_bn(seed=5).cuda()
    icb.embed_precision = "bf16"
    with pytest.raises(ValueError), torch.no_grad():
        icb(x.cuda())
    # "half" on the same frozen-BatchNorm trunk: the fp16-activation trunk does not apply — the one-plane form runs instead
    with torch.no_grad():
        icb.embed_precision = "fp32"
        f32, _ = icb(x.cuda())
        icb.embed_precision = "half"
        fh, _ = icb(x.cuda())
    assert torch.isfinite(fh).all() and float((fh - f32).abs().max()) < 5e-3 * max(1.0, float(f32.abs().max()))   # (BatchNorm features are not O(1))
    r50 = resnet50(norm_layer=nn.InstanceNorm2d)
    r50.fc = nn.Identity()
    convs50 = [t.cuda() for t in resnet_convs_of(r50)[0]]
    with pytest.raises(ValueError), torch.no_grad():
        ops.resnet18in_forward(x.cuda(), convs50, precision="bf16")


@pytest.mark.parametrize("B,H,W", [(256, 224, 224), (7, 224, 224), (33, 224, 224), (255, 224, 224), (257, 224, 224),
                                   (3, 250, 250), (2, 225, 231), (9, 231, 225)])
def test_embedder_at_the_benchmarked_batch_and_odd_sizes(B, H, W):
    """BASELINE configs[3] times bs = 256 at 224x224: at that size the Winograd host search picks different
    unit shapes (several images per unit) and the direct convs' 128-pixel tiles straddle images differently
    than at B <= 6 — parity is checked THERE, plus the batch sizes around it and odd patch sizes
    (250 -> 125 -> 63 -> 32 -> 16 -> 8; 225x231 -> 113x116 -> 57x58 -> 29x29 -> 15x15 -> 8x8).
    Oracle: the torch restatement in fp64 (cross-checked against the numpy restatement on the CPU,
    tests/test_resnet_host.py).  Same 1e-4 bar (compute_feats.py:82)."""
    ic, w = _build(seed=11)
    x = torch.from_numpy(make_patches(700 + B, B, H, W))
    ref_f, ref_c = _ref(x, w, ic)
    icg = ic.cuda()
    with torch.no_grad():
        feats, c = icg(x.cuda())
    assert feats.shape == (B, 512) and c.shape == (B, 2)
    err = np.abs(feats.cpu().numpy() - ref_f).max()
    np.testing.assert_allclose(feats.cpu().numpy(), ref_f, atol=1e-4, rtol=1e-4, err_msg=f"max abs err {err:.3e}")
    np.testing.assert_allclose(c.cpu().numpy(), ref_c, atol=1e-4, rtol=1e-4)


def test_embedder_rows_at_bs256_equal_the_rows_of_small_batches():
    """Row i of a 256-patch batch is the row a 3-patch batch containing patch i produces (per-image
    statistics): ties the benchmarked batch to the small-batch parity cases bit-for-bit-close."""
    ic, w = _build(seed=11)
    x = torch.from_numpy(make_patches(956, 256, 224, 224)).cuda()
    icg = ic.cuda()
    with torch.no_grad():
        full, cfull = icg(x)
        for lo in (0, 101, 253):
            part, cpart = icg(x[lo:lo + 3])
            np.testing.assert_allclose(part.cpu().numpy(), full[lo:lo + 3].cpu().numpy(), atol=2e-6, rtol=1e-5)
            np.testing.assert_allclose(cpart.cpu().numpy(), cfull[lo:lo + 3].cpu().numpy(), atol=2e-6, rtol=1e-5)


def test_images_are_independent_of_batch_composition():
    """InstanceNorm uses per-image statistics: sharding a slide's patches over ranks must not
    change any row.  Tiles of 128 flattened pixels straddle images in layers 2-4."""
    ic, w = _build(seed=12)
    x = torch.from_numpy(make_patches(99, 6, 224, 224)).cuda()
    icg = ic.cuda()
    with torch.no_grad():
        full, _ = icg(x)
        a, _ = icg(x[:2])
        b, _ = icg(x[2:])
    np.testing.assert_allclose(torch.cat([a, b]).cpu().numpy(), full.cpu().numpy(), atol=2e-6, rtol=1e-5)


def test_constant_image_does_not_produce_nan():
    """A flat (background) tile has zero variance in conv1's border-free region; eps keeps IN finite."""
    ic, w = _build(seed=13)
    x = torch.full((2, 3, 224, 224), 0.8)
    ref_f, _ = _ref(x, w, ic)
    with torch.no_grad():
        feats, _ = ic.cuda()(x.cuda())
    assert torch.isfinite(feats).all()
    np.testing.assert_allclose(feats.cpu().numpy(), ref_f, atol=2e-3, rtol=2e-3)


def test_weight_update_invalidates_packed_cache():
    ic, w = _build(seed=14)
    icg = ic.cuda()
    x = torch.from_numpy(make_patches(5, 2, 224, 224)).cuda()
    with torch.no_grad():
        f1, _ = icg(x)
        w2 = ro.make_weights(seed=15)
        icg.feature_extractor.load_state_dict({k: v.cuda() for k, v in w2.items()}, strict=True)
        f2, _ = icg(x)
    ref2, _ = _ref(x.cpu(), w2, ic)
    assert not np.allclose(f1.cpu().numpy(), f2.cpu().numpy(), atol=1e-3)
    np.testing.assert_allclose(f2.cpu().numpy(), ref2, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("B,H,W", [(3, 224, 224), (2, 96, 160)])
def test_uint8_nhwc_ingest_is_bit_identical_to_fp32_entry(B, H, W):
    """dsmil_resnet18in_forward_u8 (decoded uint8 NHWC images, ToTensor fused into the stem) against
    dsmil_resnet18in_forward on VF.to_tensor's output: same bits, and both within tolerance of the oracle."""
    ic, w = _build(seed=13)
    g = torch.Generator().manual_seed(21 + B)
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    x = img.permute(0, 3, 1, 2).to(torch.float32).div(255).contiguous()
    ref_f, ref_c = _ref(x, w, ic)
    icg = ic.cuda()
    with torch.no_grad():
        f8, c8 = icg(img.cuda())
        f, c = icg(x.cuda())
    assert torch.equal(f8, f) and torch.equal(c8, c)
    np.testing.assert_allclose(f8.cpu().numpy(), ref_f, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(c8.cpu().numpy(), ref_c, atol=1e-4, rtol=1e-4)


def _build_bn(seed, C=2):
    """`--norm_layer batch` extractor (compute_feats.py:149-154) with non-trivial frozen statistics and
    affine parameters, including negative weights (max-pool then needs the window MIN)."""
    res = resnet18(pretrained=False, norm_layer=nn.BatchNorm2d)
    res.fc = nn.Identity()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in res.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 1.5 + 0.25)
                w = torch.rand(m.weight.shape, generator=g) * 0.8 + 0.6
                sign = torch.where(torch.rand(m.weight.shape, generator=g) < 0.15, -1.0, 1.0)
                m.weight.copy_(w * sign)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
    ic = dsmil.IClassifier(res, 512, output_class=C)
    with torch.no_grad():
        ic.fc.weight.copy_(torch.randn(ic.fc.weight.shape, generator=g) * 0.1)
        ic.fc.bias.copy_(torch.randn(ic.fc.bias.shape, generator=g) * 0.1)
    return ic.eval()


@pytest.mark.parametrize("B,H,W,u8", [(3, 224, 224, False), (2, 96, 160, False), (2, 224, 224, True),
                                      (256, 224, 224, False), (33, 224, 224, True), (3, 250, 250, False),
                                      (2, 225, 231, True)])
def test_frozen_batchnorm_trunk_vs_torch_fp64(B, H, W, u8):
    """dsmil_resnet18bn_forward (eval-mode BatchNorm folded into the InstanceNorm kernels' (x-m)*r step).  Truth for the
    small batches (B <= 5): oracle/resnet_numpy.py (kind="batch": plain numpy fp64, no torch operator); for the large ones
    (minutes of numpy) the same torch module evaluated on the CPU in fp64.  Tolerance 1e-4 abs + 1e-4 rel."""
    import copy
    ic = _build_bn(seed=17)
    g = torch.Generator().manual_seed(5 + B)
    if u8:
        img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
        x = img.permute(0, 3, 1, 2).to(torch.float32).div(255).contiguous()
    else:
        x = torch.from_numpy(make_patches(9 + B, B, H, W))
        img = None
    if B <= 5:
        sd = {k: v.numpy() for k, v in ic.feature_extractor.state_dict().items()}
        rf = torch.from_numpy(rnp.resnet_features(x.numpy(), sd, 18, "batch"))
        rc = rf @ ic.fc.weight.detach().double().T + ic.fc.bias.detach().double()
    else:
        ref = copy.deepcopy(ic).double()
        with torch.no_grad():
            rf, rc = ref(x.double())
    icg = ic.cuda()
    with torch.no_grad():
        f, c = icg(img.cuda() if u8 else x.cuda())
    np.testing.assert_allclose(f.cpu().numpy(), rf.numpy(), atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(c.cpu().numpy(), rc.numpy(), atol=1e-4, rtol=1e-4)
    # training-mode BatchNorm is NOT the frozen-statistics trunk: it must take the torch graph
    from dsmil_wsi_amd.modules import resnet_convs_of
    icg.feature_extractor.train()
    assert resnet_convs_of(icg.feature_extractor) is None


@pytest.mark.parametrize("norm", ["instance", "batch"])
def test_resnet34_trunk_vs_numpy_oracle(norm):
    """`--backbone resnet34` (compute_feats.py:158-160): the same kernels over blocks [3,4,6,3]
    (dsmil_resnet_forward, depth 34) against oracle/resnet_numpy.py in fp64."""
    import copy
    from dsmil_wsi_amd.resnet import resnet34
    from dsmil_wsi_amd.modules import resnet_convs_of
    g = torch.Generator().manual_seed(41)
    res = resnet34(norm_layer=nn.InstanceNorm2d if norm == "instance" else nn.BatchNorm2d)
    res.fc = nn.Identity()
    with torch.no_grad():
        for m in res.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (m.weight.shape[0] * m.weight.shape[2] ** 2)) ** 0.5)
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.6)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    ic = dsmil.IClassifier(res, 512, output_class=2).eval()
    for p in ic.parameters():
        p.requires_grad = False
    trunk = resnet_convs_of(ic.feature_extractor)
    assert trunk is not None and len(trunk[0]) == 36
    x = torch.from_numpy(make_patches(77, 2, 224, 224))
    # truth: oracle/resnet_numpy.py (plain numpy fp64, no torch operator) on the module's state dict
    rf = torch.from_numpy(rnp.resnet_features(x.numpy(), {k: v.numpy() for k, v in res.state_dict().items()}, 34, norm))
    rc = rf @ ic.fc.weight.double().T + ic.fc.bias.double()
    icg = ic.cuda()
    with torch.no_grad():
        f, c = icg(x.cuda())
    assert f.shape == (2, 512)
    np.testing.assert_allclose(f.cpu().numpy(), rf.numpy(), atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(c.cpu().numpy(), rc.numpy(), atol=1e-4, rtol=1e-4)


def test_resnet101_trunk_vs_numpy_oracle_bar_is_1p5x_reference_fp32_error_not_1e_4():
    """Depth 101 separately, because its bar is NOT the 1e-4 of every other trunk: 104 fp32 conv + norm layers on random
    weights put the reference's own fp32 evaluation ~3e-4 from fp64, so the bar is 1.5x that error, measured here; the
    assertion message reports the native error against fp64 and against 1e-4."""
    _bottleneck_case(101, "instance", 2, 224, 224)


@pytest.mark.parametrize("depth,norm,B,H,W", [(50, "instance", 2, 224, 224), (50, "batch", 3, 224, 224), (50, "instance", 33, 224, 224)])
def test_bottleneck_trunks_vs_numpy_oracle(depth, norm, B, H, W):
    _bottleneck_case(depth, norm, B, H, W, strict=True)          # the 1e-4 bar, no relaxation


def test_resnet50_small_maps_bar_is_1p5x_reference_fp32_error_not_1e_4():
    """96 x 160 inputs leave layer 4 with 3 x 5 maps: InstanceNorm over 15 values amplifies rounding, and the reference's
    own fp32 evaluation is ~1.1e-4 from fp64 here — this one case takes the relaxed bar (1.5x that error, measured)."""
    _bottleneck_case(50, "instance", 5, 96, 160)


def _bottleneck_case(depth, norm, B, H, W, strict=False):
    """`--backbone resnet50 | resnet101` (compute_feats.py:161-167: Bottleneck blocks, 2048-d features): the native
    trunk (dsmil_resnet_forward, depth 50 / 101: 1x1 and strided 3x3 convs on the direct MFMA kernel, stride-1 3x3 convs
    on the Winograd kernel, fused InstanceNorm / folded frozen BatchNorm) against the same torch module evaluated on the
    CPU in fp64.  Tolerance 1e-4 abs + 1e-4 rel on features and instance logits — except where the reference's OWN
    fp32 evaluation (the same torch module in fp32 on the CPU) is further than that from fp64: 104 fp32 conv + norm
    layers drift by ~3e-4 on random weights (depth 101), and no fp32 implementation can be closer to fp64 than fp32
    arithmetic allows; there the bar is 1.5x the reference's own fp32 error, measured in the test."""
    import copy
    from dsmil_wsi_amd.resnet import resnet50, resnet101
    from dsmil_wsi_amd.modules import resnet_convs_of
    g = torch.Generator().manual_seed(500 + depth)
    ctor = resnet50 if depth == 50 else resnet101
    res = ctor(norm_layer=nn.InstanceNorm2d if norm == "instance" else nn.BatchNorm2d)
    res.fc = nn.Identity()
    with torch.no_grad():
        for m in res.modules():
            if isinstance(m, nn.Conv2d):   # kaiming(fan_out), torchvision's init, seeded
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (m.weight.shape[0] * m.weight.shape[2] ** 2)) ** 0.5)
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_((torch.rand(m.weight.shape, generator=g) * 0.5 + 0.6) *
                               torch.where(torch.rand(m.weight.shape, generator=g) < 0.1, -1.0, 1.0))
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    ic = dsmil.IClassifier(res, 2048, output_class=2).eval()
    with torch.no_grad():
        ic.fc.weight.copy_(torch.randn(ic.fc.weight.shape, generator=g) * 0.05)
        ic.fc.bias.copy_(torch.randn(ic.fc.bias.shape, generator=g) * 0.1)
    for p in ic.parameters():
        p.requires_grad = False
    trunk = resnet_convs_of(ic.feature_extractor)
    assert trunk is not None and len(trunk[0]) == (53 if depth == 50 else 104)
    x = torch.from_numpy(make_patches(80 + B, B, H, W))
    with torch.no_grad():
        if B <= 5:   # truth: oracle/resnet_numpy.py (plain numpy fp64) on the module's state dict
            rf = torch.from_numpy(rnp.resnet_features(x.numpy(), {k: v.numpy() for k, v in res.state_dict().items()}, depth, norm))
            rc = rf @ ic.fc.weight.double().T + ic.fc.bias.double()
        else:        # 33 x 224 x 224 through ResNet-50 is minutes of numpy: the torch module in fp64, which
            rf, rc = copy.deepcopy(ic).double()(x.double())   # tests/test_resnet_host.py ties to the numpy oracle
        f32_ref, _ = ic(x)                                     # the reference's arithmetic: torch fp32 on the CPU
    ref_err = float((f32_ref.double() - rf).abs().max())
    icg = ic.cuda()
    with torch.no_grad():
        f, c = icg(x.cuda())
    assert f.shape == (B, 2048) and c.shape == (B, 2)
    err = float((f.cpu().double() - rf).abs().max())
    tol = max(1e-4, 1.5 * ref_err)
    np.testing.assert_allclose(f.cpu().numpy(), rf.numpy(), atol=tol, rtol=1e-4,
                               err_msg=f"max abs err vs fp64 {err:.3e} ({err / 1e-4:.2f} x the 1e-4 bar); reference fp32 vs fp64 {ref_err:.3e}")
    if strict:
        assert err <= 1e-4 * (1 + float(rf.abs().max())), f"max abs err vs fp64 {err:.3e} exceeds the 1e-4 bar (reference fp32 {ref_err:.3e})"
    np.testing.assert_allclose(c.cpu().numpy(), rc.numpy(), atol=tol, rtol=1e-4)
