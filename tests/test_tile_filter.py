"""Background filters of the reference's tilers: deepzoom_tiler.py:56-61 (PIL FIND_EDGES band sums)
and test_crop_single.py:17-24 (mean ubyte HSV saturation).  The CPU restatement is pinned to the REAL dependency
where it is installed (Pillow: ImageFilter.FIND_EDGES + ImageStat, run here and on the GPU box); skimage is absent, so
the saturation half follows skimage's published rgb2hsv / img_as_ubyte definitions (float64, rint) with hand-worked
known answers.  The HIP kernel (dsmil_tile_stats) must reproduce the exact integer sums."""
import numpy as np
import pytest
import torch

import dsmil  # noqa: F401
from dsmil_wsi_amd import pipeline as pl


def _tiles(seed, B, H, W):
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    if B > 1:
        t[1] = 230                                                  # flat background
    if B > 2:
        t[2] = (np.arange(W)[None, :, None] * 255 // max(1, W - 1)).astype(np.uint8)   # smooth horizontal ramp
    if B > 3:
        t[3, :, :, 1] = 0                                           # saturated (no green)
    return t


def _pil_edge_sums(tile):
    from PIL import Image, ImageFilter, ImageStat
    return ImageStat.Stat(Image.fromarray(tile).filter(ImageFilter.FIND_EDGES)).sum


@pytest.mark.parametrize("B,H,W", [(5, 224, 224), (4, 32, 40), (3, 7, 5), (2, 3, 3), (2, 225, 231)])
def test_edge_sums_equal_pillow(B, H, W):
    t = _tiles(11 + H, B, H, W)
    st = pl.tile_stats_reference(torch.from_numpy(t))
    for i in range(B):
        assert [int(v) for v in st[i, :3]] == [int(v) for v in _pil_edge_sums(t[i])]
    # the tiler's decision (deepzoom_tiler.py:59-61) on 224-px tiles: noise is tissue, flat and smooth tiles are background
    if (H, W) == (224, 224):
        keep = pl.background_keep_mask(torch.from_numpy(t), edge_threshold=15)
        assert keep.tolist() == [True, False, False, True, True]


def test_saturation_known_answers():
    """rgb2hsv saturation = (max - min) / max (0 for grey), img_as_ubyte = rint(255 s): worked by hand."""
    px = np.array([[[255, 0, 0], [10, 10, 10], [0, 0, 0], [200, 100, 50], [2, 1, 1], [255, 254, 255]]], np.uint8)[None]
    st = pl.tile_stats_reference(torch.from_numpy(px))
    # 255 | 0 | 0 | rint(255*150/200 = 191.25) = 191 | rint(127.5) = 128 (half to even) | rint(255/255 = 1.0) = 1
    assert int(st[0, 3]) == 255 + 0 + 0 + 191 + 128 + 1
    t = _tiles(5, 4, 64, 64)
    keep = pl.background_keep_mask(torch.from_numpy(t), edge_threshold=None, sat_threshold=30)   # test_crop_single.py:36
    assert keep.tolist() == [True, False, False, True]


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(64, 224, 224), (5, 32, 40), (3, 7, 5), (2, 225, 231), (2, 1000, 1024), (3, 2, 9)])
def test_hip_tile_stats_are_exact(B, H, W):
    from dsmil_wsi_amd import ops
    t = _tiles(100 + W, B, H, W)
    got = ops.tile_stats(torch.from_numpy(t).cuda()).cpu().numpy()
    ref = pl.tile_stats_reference(torch.from_numpy(t))
    assert np.array_equal(got, ref)
    for i in range(min(B, 4)):
        assert [int(v) for v in got[i, :3]] == [int(v) for v in _pil_edge_sums(t[i])]
    a = pl.background_keep_mask(torch.from_numpy(t).cuda(), 15, 30).cpu()
    b = pl.background_keep_mask(torch.from_numpy(t), 15, 30)
    assert torch.equal(a, b)
