"""CPU side of the JPEG decode path: the numpy restatement (oracle/jpeg_oracle.py) against the committed golden
vectors (tests/golden/jpeg_golden.npz: files + Pillow's decode of them, made by tests/golden/make_jpeg_golden.py) and against
Pillow itself on freshly encoded images; the library's HOST marker parser (dsmil_jpeg_parse) against the oracle's."""
import io
import os

import numpy as np
import pytest
from PIL import Image

import dsmil  # noqa: F401
import jpeg_oracle as jo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_golden.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files})


@pytest.mark.parametrize("name", NAMES)
def test_oracle_equals_golden_pillow_decode(name):
    blob = GOLD[name + "/file"].tobytes()
    assert np.array_equal(jo.decode(blob), GOLD[name + "/rgb"])


@pytest.mark.parametrize("h,w", [(16, 16), (17, 23), (3, 5), (40, 56)])
def test_oracle_equals_pillow_on_fresh_encodes(h, w):
    rng = np.random.default_rng(h * 100 + w)
    for q in (30, 70, 95):
        for ss in (0, 1, 2):
            a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            b = io.BytesIO()
            Image.fromarray(a).save(b, "JPEG", quality=q, subsampling=ss)
            ref = np.array(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))
            assert np.array_equal(jo.decode(b.getvalue()), ref), (q, ss)


def test_progressive_is_out_of_scope_for_the_oracle_too():
    b = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(b, "JPEG", progressive=True)
    with pytest.raises(jo.Unsupported):
        jo.decode(b.getvalue())


def test_host_parser_agrees_with_the_oracle_and_deduplicates_tables():
    """dsmil_jpeg_parse (host code of the library, no device): records of a batch of golden files + a progressive one + a non-JPEG."""
    from dsmil_wsi_amd import ops
    blobs = [GOLD[n + "/file"].tobytes() for n in NAMES]
    b = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(b, "JPEG", progressive=True)
    blobs += [b.getvalue(), b"\x89PNG\r\n\x1a\n" + bytes(64)]
    data, plan, recs = ops.jpeg_parse(blobs)
    off = np.concatenate([[0], np.cumsum([len(x) for x in blobs])])
    for i, n in enumerate(NAMES):
        h = jo.parse(blobs[i])
        r = recs[i]
        assert r["status"] == 0
        assert (r["width"], r["height"], r["ncomp"]) == (h["width"], h["height"], len(h["comps"]))
        assert (r["hsamp"], r["vsamp"]) == (h["comps"][0][1], h["comps"][0][2]) or len(h["comps"]) == 1
        assert r["restart_interval"] == h["restart_interval"]
        assert r["ecs_begin"] == off[i] + h["ecs"][0] and r["ecs_end"] == off[i + 1]
    assert recs[len(NAMES)]["status"] == -2 and recs[len(NAMES) + 1]["status"] == -1
    n_qt, n_ht = plan[:16].view(np.int32)[1:3]
    # Pillow writes the standard tables scaled per quality: far fewer distinct tables than files x 4
    assert 2 <= n_qt <= 2 * len(NAMES) and 2 <= n_ht <= 8
    same = [i for i, n in enumerate(NAMES) if n in ("tile_q70_420", "odd_q70_420", "rst_q70_420")]
    assert len({tuple(recs[i]["qt"]) for i in same}) == 1 and len({tuple(recs[i]["dc"]) for i in same}) == 1
