"""Golden vectors of the JPEG decode path: small baseline-JPEG files and what Pillow — the reference's decoder,
`Image.open` at /root/reference/compute_feats.py:28 — makes of them.  Run in the build container (Pillow 12.2.0 over
libjpeg-turbo 3.1.4.1):   python tests/golden/make_jpeg_golden.py
Writes tests/golden/jpeg_golden.npz: for every case the file's bytes (uint8) and the decoded RGB array (uint8 [H, W, 3])."""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def image(rng, h, w, kind):
    if kind == 0:
        a = rng.integers(0, 256, (h, w, 3))
    elif kind == 1:
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([(xx * 3 + yy) % 256, (yy * 5) % 256, (xx ^ yy) % 256], -1)
    else:
        base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3)).repeat(8, 0).repeat(8, 1)[:h, :w]
        a = np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255)
    return a.astype(np.uint8)


CASES = [  # name, h, w, kind, save kwargs
    ("tile_q70_420", 64, 64, 2, dict(quality=70)),                       # what deepzoom_tiler.py:64,250 writes (smaller)
    ("noise_q95_444", 24, 40, 0, dict(quality=95, subsampling=0)),
    ("ramp_q30_422", 33, 31, 1, dict(quality=30, subsampling=1)),
    ("odd_q70_420", 17, 23, 2, dict(quality=70, subsampling=2)),
    ("one_pixel", 1, 1, 0, dict(quality=70)),
    ("opt_q85_420", 48, 48, 2, dict(quality=85, optimize=True)),
    ("rst_q70_420", 48, 64, 2, dict(quality=70, restart_marker_blocks=3)),
    ("rst_rows_q85_422", 40, 56, 2, dict(quality=85, subsampling=1, restart_marker_rows=1)),
    ("grey_q70", 40, 40, 2, dict(quality=70)),
]


def main():
    rng = np.random.default_rng(20260930)
    out = {}
    for name, h, w, kind, kw in CASES:
        a = image(rng, h, w, kind)
        if name.startswith("grey"):
            a = a[:, :, 0]
        b = io.BytesIO()
        Image.fromarray(a).save(b, "JPEG", **kw)
        blob = b.getvalue()
        out[name + "/file"] = np.frombuffer(blob, np.uint8)
        out[name + "/rgb"] = np.array(Image.open(io.BytesIO(blob)).convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "jpeg_golden.npz"), **out)
    print("wrote", len(CASES), "cases,", os.path.getsize(os.path.join(HERE, "jpeg_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
