#!/usr/bin/env python3
"""Randomised soak of the device JPEG decoder against Pillow: sizes 1..300, qualities 1..100, every subsampling, optimised tables,
restart intervals, grey, photographic / noisy / flat content.     python tools/jpeg_soak.py [cases] [seed]"""
import _path  # noqa: F401
import io
import sys
import numpy as np
from PIL import Image
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = dev = pil = 0
batch, meta = {}, {}
for c in range(cases):
    h, w = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    if c % 5 == 0:
        h = w = 224
    kind = int(rng.integers(0, 4))
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:
        a = rng.integers(0, 256, (h, w, 3))
    elif kind == 1:
        a = np.stack([(xx * 3 + yy) % 256, (yy * 5) % 256, (xx ^ yy) % 256], -1)
    elif kind == 2:
        base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3)).repeat(8, 0).repeat(8, 1)[:h, :w]
        a = np.clip(base + rng.normal(0, float(rng.uniform(0, 30)), (h, w, 3)), 0, 255)
    else:
        a = np.full((h, w, 3), int(rng.integers(0, 256)))
    a = a.astype(np.uint8)
    kw = dict(quality=int(rng.integers(1, 101)))
    grey = rng.random() < 0.1
    if not grey:
        kw["subsampling"] = int(rng.integers(0, 3))
    if rng.random() < 0.3:
        kw["optimize"] = True
    if rng.random() < 0.3 and h >= 8 and w >= 8:
        kw["restart_marker_blocks"] = int(rng.integers(1, 9))
    b = io.BytesIO()
    try:
        Image.fromarray(a[:, :, 0] if grey else a).save(b, "JPEG", **kw)
    except OSError:      # (Pillow's encoder refuses a few combinations, e.g. restart markers on tiny images)
        continue
    batch.setdefault((h, w), []).append(b.getvalue())
    meta.setdefault((h, w), []).append(kw)
for (h, w), blobs in batch.items():
    st = {}
    got = ops.jpeg_decode(blobs, "cuda", size=(h, w), stats=st).cpu().numpy()
    dev += st.get("device", 0)
    pil += st.get("pillow", 0)
    for i, bl in enumerate(blobs):
        ref = np.array(Image.open(io.BytesIO(bl)).convert("RGB"))
        if not np.array_equal(got[i], ref):
            bad += 1
            print("MISMATCH", h, w, meta[(h, w)][i], int(np.abs(got[i].astype(int) - ref.astype(int)).max()), flush=True)
print(f"{cases} files in {len(batch)} size groups: {dev} decoded on the device, {pil} by Pillow, {bad} mismatches")
sys.exit(1 if bad else 0)
