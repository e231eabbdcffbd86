"""Batched baseline-JPEG decode on the device (dsmil_jpeg_decode) against Pillow's own decode of the same bytes —
the reference's `Image.open` (compute_feats.py:28,107) — byte for byte: 4:2:0 / 4:2:2 / 4:4:4 / grey, qualities 30-95, optimised
Huffman tables, restart intervals, sizes that are not multiples of the MCU, the tiler's 224 x 224 tiles at quality 70; files outside
the decoder's scope (progressive) take the Pillow path inside the same call."""
import io

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _img(rng, h, w, kind):
    if kind == 0:
        a = rng.integers(0, 256, (h, w, 3))
    elif kind == 1:
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([(xx * 3 + yy) % 256, (yy * 5) % 256, (xx ^ yy) % 256], -1)
    else:
        base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3)).repeat(8, 0).repeat(8, 1)[:h, :w]
        a = np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255)
    return a.astype(np.uint8)


def _jpeg(a, **kw):
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", **kw)
    return b.getvalue()


def _pil(blob):
    return np.array(Image.open(io.BytesIO(blob)).convert("RGB"))


@pytest.mark.parametrize("h,w", [(224, 224), (16, 16), (17, 23), (33, 31), (8, 8), (1, 1), (3, 5), (250, 250), (96, 160)])
def test_device_decode_equals_pillow(h, w):
    from dsmil_wsi_amd import ops
    rng = np.random.default_rng(h * 1000 + w)
    blobs = []
    for kind in (0, 1, 2):
        for q in (30, 70, 95):
            for ss in (0, 1, 2):
                blobs.append(_jpeg(_img(rng, h, w, kind), quality=q, subsampling=ss))
    blobs.append(_jpeg(_img(rng, h, w, 2), quality=70, optimize=True))
    blobs.append(_jpeg(_img(rng, h, w, 2)[:, :, 0], quality=70))                       # grey
    if h >= 16 and w >= 16:
        blobs.append(_jpeg(_img(rng, h, w, 2), quality=70, restart_marker_blocks=3))
        blobs.append(_jpeg(_img(rng, h, w, 2), quality=85, subsampling=1, restart_marker_rows=1))
    blobs.append(_jpeg(_img(rng, h, w, 2), quality=70, progressive=True))              # outside the scope: the Pillow path
    stats = {}
    got = ops.jpeg_decode(blobs, "cuda", size=(h, w), stats=stats).cpu().numpy()
    assert stats["pillow"] == 1 and stats["device"] == len(blobs) - 1
    for i, b in enumerate(blobs):
        ref = _pil(b)
        assert np.array_equal(got[i], ref), (i, int(np.abs(got[i].astype(int) - ref.astype(int)).max()))


def test_a_slide_worth_of_tiles_and_the_oracle():
    """2 048 tiles of 224 x 224 at the tiler's quality 70 (deepzoom_tiler.py:250): device decode == Pillow for every tile; the
    numpy oracle (oracle/jpeg_oracle.py) agrees on a sample (it is pinned to Pillow by tests/test_jpeg_host.py)."""
    import jpeg_oracle as jo
    from dsmil_wsi_amd import ops
    rng = np.random.default_rng(5)
    blobs = [_jpeg(_img(rng, 224, 224, 2), quality=70) for _ in range(64)]
    blobs = [blobs[i % 64] if i % 7 else _jpeg(_img(rng, 224, 224, i % 3), quality=70) for i in range(2048)]
    got = ops.jpeg_decode(blobs, "cuda").cpu().numpy()
    for i in range(0, 2048, 1):
        if i % 7 == 0 or i < 64:
            assert np.array_equal(got[i], _pil(blobs[i])), i
    assert np.array_equal(got[7], jo.decode(blobs[7]))


def test_corrupt_stream_falls_back_to_pillow():
    """A truncated entropy-coded segment: the device decoder flags the image (DSMIL_E_INVALID) or decodes the zero-fed tail as
    libjpeg does; either way the batch's result is what Pillow gives for every file Pillow can read."""
    from dsmil_wsi_amd import ops
    rng = np.random.default_rng(9)
    good = _jpeg(_img(rng, 64, 64, 2), quality=70)
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x5A                       # a flipped byte in the middle of the scan
    blobs = [good, bytes(bad), good]
    from PIL import ImageFile
    ImageFile.LOAD_TRUNCATED_IMAGES = True
    try:
        ref1 = _pil(bytes(bad))
    except Exception:  # noqa: BLE001  Pillow refuses the file: nothing to compare
        ref1 = None
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = False
    stats = {}
    try:
        got = ops.jpeg_decode(blobs, "cuda", stats=stats).cpu().numpy()
    except Exception:  # noqa: BLE001  (Pillow raised on the corrupt file inside the fallback: the reference's loader would too)
        return
    assert np.array_equal(got[0], _pil(good)) and np.array_equal(got[2], _pil(good))
    assert got[1].shape == (64, 64, 3) and (ref1 is None or ref1.shape == (64, 64, 3))


def test_compute_feats_with_device_decode_gives_the_same_csv(tmp_path, monkeypatch):
    """compute_feats.py --gpu_decode (new, default off): the tiles' JPEG files are decoded by dsmil_jpeg_decode instead of by
    Pillow in DataLoader workers (compute_feats.py:28,55) — the decoded bytes are identical, so every feature row is
    BIT-identical to the default path's (a progressive JPEG among the tiles takes Pillow inside the same call)."""
    import glob
    import os
    import zlib
    import pandas as pd
    from test_entry_points import _jpeg as write_tile, _simclr_checkpoint
    from dsmil_wsi_amd import pipeline as pl
    monkeypatch.chdir(tmp_path)
    import compute_feats as cf
    _simclr_checkpoint("simclr/runs/r0/checkpoints/model.pth", 31)
    for slide in ("s1", "s2"):
        for i in range(9):
            write_tile(f"WSI/toy/single/0_x/{slide}/{i}_{i + 1}.jpeg", zlib.crc32(f'{slide}/{i}'.encode()) % 10000, size=224)
    # one progressive tile: outside the device decoder's scope
    with Image.open("WSI/toy/single/0_x/s1/0_1.jpeg") as im:
        im.convert("RGB").save("WSI/toy/single/0_x/s1/0_1.jpeg", "JPEG", quality=70, progressive=True)
    try:
        cf.main(["--dataset", "toy", "--weights", "r0", "--batch_size", "4", "--num_workers", "0", "--save_npy"])
        ref = {f: np.load(f) for f in sorted(glob.glob("datasets/toy/0_x/*.npy"))}
        for f in ref:
            os.remove(f)
        seen = {}
        real = pl._embed_jpeg_chunks      # (round 6: the device-decode path of embed_files deals its batches to the stream pool)

        def spy(*a, **k):
            seen["calls"] = seen.get("calls", 0) + 1
            return real(*a, **k)
        monkeypatch.setattr(pl, "_embed_jpeg_chunks", spy)
        cf.main(["--dataset", "toy", "--weights", "r0", "--batch_size", "4", "--num_workers", "2", "--save_npy", "--gpu_decode"])
        assert seen.get("calls", 0) == 2 and pl.GPU_DECODE[0] is False      # one call per bag; the flag is reset behind main()
        got = {f: np.load(f) for f in sorted(glob.glob("datasets/toy/0_x/*.npy"))}
    finally:
        pl.GPU_DECODE[0] = False
    assert list(ref) == list(got) and len(ref) == 2
    for f in ref:
        assert ref[f].shape == (9, 512) and np.array_equal(ref[f], got[f]), f
    assert len(pd.read_csv("datasets/toy/0_x/s1.csv")) == 9


def _with_fill_bytes(blob):
    """A restart-interval JPEG with an 0xFF fill byte in front of every RSTn marker of its scan (T.81 B.1.1.2 allows any number)."""
    import jpeg_oracle as jo
    beg = jo.parse(blob)["ecs"][0]
    out = bytearray(blob[:beg])
    i = beg
    while i < len(blob):
        if blob[i] == 0xFF and i + 1 < len(blob) and 0xD0 <= blob[i + 1] <= 0xD7:
            out += b"\xff"
        out.append(blob[i])
        i += 1
    return bytes(out)


def test_fill_bytes_truncation_and_rgb_component_ids():
    """Streams a tiler does not write but the standard allows: fill bytes in front of restart markers decode like Pillow; a file
    without its EOI is NOT decoded on the device (Pillow judges it, as in the reference's loader)."""
    import jpeg_oracle as jo
    from dsmil_wsi_amd import ops
    rng = np.random.default_rng(21)
    a = _img(rng, 64, 80, 2)
    rst = _jpeg(a, quality=70, restart_marker_blocks=2)
    filled = _with_fill_bytes(rst)
    assert len(filled) > len(rst)
    ref = _pil(filled)
    assert np.array_equal(ref, _pil(rst)) and np.array_equal(jo.decode(filled), ref)
    stats = {}
    got = ops.jpeg_decode([filled, rst], "cuda", stats=stats).cpu().numpy()
    assert stats["pillow"] == 0 and np.array_equal(got[0], ref) and np.array_equal(got[1], ref)
    # truncated: the parser hands it to Pillow
    cut = rst[:-2]
    _, _, recs = ops.jpeg_parse([rst, cut])
    assert recs["status"][0] == 0 and recs["status"][1] == -1


def test_chunked_decode_walks_both_staging_buffers(tmp_path):
    """embed_files(gpu_decode=True) and embed_jpeg_blobs over SEVEN chunks of 8 files (the two staging buffers are each reused
    three times, the decode of chunk i+1 runs on the side stream under the embedding of chunk i): feature rows bit-identical to the
    Pillow loader's, in order."""
    import os
    from test_resnet_gpu import _build
    from dsmil_wsi_amd import pipeline as pl
    rng = np.random.default_rng(31)
    files, blobs = [], []
    for i in range(53):
        b = _jpeg(_img(rng, 96, 96, i % 3), quality=70)
        f = os.path.join(tmp_path, f"{i}_{i + 1}.jpeg")
        with open(f, "wb") as fh:
            fh.write(b)
        files.append(f)
        blobs.append(b)
    ic, _ = _build(seed=11)
    ic = ic.cuda()
    ref_f, ref_c = pl.embed_files(ic, files, batch_size=4, num_workers=0, gpu_decode=False)
    old = pl.DECODE_BATCH[0]
    pl.DECODE_BATCH[0] = 8
    try:
        got_f, got_c = pl.embed_files(ic, files, batch_size=4, num_workers=2, gpu_decode=True)
    finally:
        pl.DECODE_BATCH[0] = old
    assert torch.equal(ref_f, got_f) and torch.equal(ref_c, got_c)
    st = {}
    f2, c2 = pl.embed_jpeg_blobs(ic, blobs, batch_size=4, decode_batch=8, streams=3, stats=st)
    torch.cuda.synchronize()
    assert st == {"device": 53, "pillow": 0}
    assert torch.equal(ref_f, f2) and torch.equal(ref_c, c2)


def test_decodes_in_flight_with_host_decoded_files_and_a_chunk_without_a_device_launch():
    """pipeline.embed_jpeg_blobs keeps several decodes in flight (ops.jpeg_decode_begin / _end, DECODE_AHEAD on DECODE_STREAMS
    side streams, the staging buffers reused round-robin).  Thirteen chunks of 4 files: progressive files (host-decoded inside
    jpeg_decode_end, copied in behind later chunks' launches) in several chunks, ONE chunk with no file for the device at all
    (no launch, no event of its own) — rows bit-identical to the embedding of Pillow's decode, in order, for 1-3 decodes ahead on
    1-2 streams; and the two halves used directly equal jpeg_decode."""
    from test_resnet_gpu import _build
    from dsmil_wsi_amd import ops, pipeline as pl
    rng = np.random.default_rng(47)
    blobs = []
    for i in range(51):
        prog = i in (5, 22, 23, 49) or 28 <= i < 32          # chunk 7 (files 28-31) is all-progressive
        blobs.append(_jpeg(_img(rng, 96, 96, i % 3), quality=70, progressive=prog))
    n_prog = 8
    ic, _ = _build(seed=13)
    ic = ic.cuda()
    tiles = torch.from_numpy(np.stack([_pil(b) for b in blobs])).cuda()
    ref_f, ref_c = pl.embed_tiles(ic, tiles, 4, streams=1)
    old = pl.DECODE_AHEAD[0], pl.DECODE_STREAMS[0]
    try:
        for ahead, nst in ((1, 1), (2, 2), (3, 2), (3, 1)):
            pl.DECODE_AHEAD[0], pl.DECODE_STREAMS[0] = ahead, nst
            for _ in range(2):
                st = {}
                f, c = pl.embed_jpeg_blobs(ic, blobs, batch_size=4, decode_batch=4, streams=3, stats=st)
                torch.cuda.synchronize()
                assert st == {"device": 51 - n_prog, "pillow": n_prog}, (ahead, nst, st)
                assert torch.equal(ref_f, f) and torch.equal(ref_c, c), (ahead, nst)
    finally:
        pl.DECODE_AHEAD[0], pl.DECODE_STREAMS[0] = old
    # the halves by hand, two decodes in flight on the current stream, a wait hook in front of the launch
    called = []
    p0 = ops.jpeg_decode_begin(blobs[:8], "cuda", before_launch=lambda: called.append(0))
    p1 = ops.jpeg_decode_begin(blobs[28:32], "cuda", before_launch=lambda: called.append(1))
    assert called == [0, 1] and p0.ev is not None and p1.ev is None
    o0, r0 = ops.jpeg_decode_end(p0)
    o1, r1 = ops.jpeg_decode_end(p1)
    assert (r0, r1) == (1, 4)
    assert torch.equal(o0, tiles[:8]) and torch.equal(o1, tiles[28:32])
