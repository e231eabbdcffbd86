"""The reference's feature files (compute_feats.py:80-82: `pd.DataFrame(feats).to_csv(path, index=False, float_format='%.4f')`,
read back by train_tcga.py:27-32): dsmil_csv_format_f32 / pipeline.feats_csv_bytes / save_feats_csv write the SAME BYTES as pandas
— exact decimal rounding (ties to even), NaN, infinities, negative zero, subnormals, values of 1e9 and more — and the writer
thread of the compute_feats loops leaves complete files behind.  HOST code only: runs without a GPU."""
import ctypes
import io
import os

import numpy as np
import pandas as pd
import pytest

import dsmil  # noqa: F401
from dsmil_wsi_amd import _native, pipeline as pl


def _pandas_bytes(a, decimals=4):
    b = io.StringIO()
    pd.DataFrame(a).to_csv(b, index=False, float_format=f"%.{decimals}f")
    return b.getvalue().encode()


def _edge_matrix(rng, rows=300, cols=64):
    a = rng.standard_normal((rows, cols)).astype(np.float32)
    a[0, :12] = [0.0, -0.0, 1e-5, -1e-5, 0.00005, -0.00005, 0.03125, 0.09375, np.nan, np.inf, -np.inf, 3.4e38]
    a[1, :10] = [1e9, -1e9, 999999999.0, 123456.789, 1e-30, -1e-30, 0.99995, 2.5, 1e-45, -1e-45]
    a[2, :] = np.arange(cols, dtype=np.float32) / np.float32(32768.0)                       # exact binary fractions: ties
    a[3, :] = (rng.integers(-2 ** 20, 2 ** 20, cols) / np.float32(2 ** 15)).astype(np.float32)
    a[4, :] = (rng.integers(-2 ** 23, 2 ** 23, cols).astype(np.float64) / 2 ** 5).astype(np.float32)   # k / 32: '.03125' ties
    a[5, :] = np.float32(10.0) ** rng.integers(-12, 12, cols).astype(np.float32)
    return a


@pytest.mark.parametrize("decimals", [4, 0, 1, 6, 9])
def test_same_bytes_as_pandas(decimals):
    rng = np.random.default_rng(3 + decimals)
    a = _edge_matrix(rng)
    assert pl.feats_csv_bytes(a, decimals=decimals) == _pandas_bytes(a, decimals)
    assert pl.feats_csv_bytes(a, decimals=decimals, threads=1) == pl.feats_csv_bytes(a, decimals=decimals, threads=5)


def test_every_kind_of_float32_bit_pattern():
    """Random BIT PATTERNS: all exponents, subnormals, NaNs with payloads, both infinities."""
    rng = np.random.default_rng(17)
    a = rng.integers(0, 2 ** 32, (2000, 50), dtype=np.uint64).astype(np.uint32).view(np.float32)
    assert pl.feats_csv_bytes(a) == _pandas_bytes(a)


def test_a_bag_sized_file_round_trips_like_the_reference(tmp_path):
    rng = np.random.default_rng(5)
    feats = (rng.standard_normal((1200, 512)) * rng.choice([1e-3, 1.0, 30.0], (1200, 1))).astype(np.float32)
    path = os.path.join(tmp_path, "datasets", "toy", "s1.csv")
    pl.save_feats_csv(feats, path, npy=True)
    with open(path, "rb") as fh:
        assert fh.read() == _pandas_bytes(feats)
    back = pd.read_csv(path)                                # train_tcga.py:27
    assert list(back.columns) == [str(i) for i in range(512)] and back.shape == (1200, 512)
    assert np.abs(back.to_numpy() - feats).max() <= 5.0001e-5 + 1e-6 * np.abs(feats).max()
    assert np.array_equal(np.load(os.path.splitext(path)[0] + ".npy"), feats)
    # not a float32 matrix: pandas itself writes the file
    p2 = os.path.join(tmp_path, "d", "x.csv")
    pl.save_feats_csv(feats[:5].astype(np.float64), p2)
    with open(p2, "rb") as fh:
        assert fh.read() == _pandas_bytes(feats[:5].astype(np.float64))
    assert pl.feats_csv_bytes(np.zeros((0, 3), np.float32)) == b"0,1,2\n"


def test_c_abi_argument_checks_and_strided_rows():
    L = _native.lib()
    a = np.arange(12, dtype=np.float32).reshape(3, 4) / 8
    buf = np.empty(3 * 4 * 64, np.uint8)
    w = L.dsmil_csv_format_f32(a.ctypes.data, 3, 2, 4, 4, buf.ctypes.data, buf.size)      # the first two columns of every row
    assert buf[:w].tobytes() == b"0.0000,0.1250\n0.5000,0.6250\n1.0000,1.1250\n"
    assert L.dsmil_csv_format_f32(a.ctypes.data, 3, 4, 4, 4, buf.ctypes.data, 100) == _native.DSMIL_E_WORKSPACE
    for bad in ((None, 3, 4, 4, 4), (a.ctypes.data, 3, 0, 4, 4), (a.ctypes.data, 3, 4, 3, 4), (a.ctypes.data, 3, 4, 4, 10), (a.ctypes.data, -1, 4, 4, 4)):
        assert L.dsmil_csv_format_f32(*bad, buf.ctypes.data, buf.size) == _native.DSMIL_E_INVALID
    assert L.dsmil_csv_format_f32(a.ctypes.data, 0, 4, 4, 4, buf.ctypes.data, buf.size) == 0
    assert ctypes.sizeof(ctypes.c_int64) == 8


def test_writer_thread_finishes_the_files_and_reports_failures(tmp_path):
    rng = np.random.default_rng(9)
    bags = [rng.standard_normal((50 + 7 * i, 16)).astype(np.float32) for i in range(6)]
    with pl.FeatWriter(depth=2) as w:
        for i, b in enumerate(bags):
            w.save(b, os.path.join(tmp_path, "out", f"b{i}.csv"))
    for i, b in enumerate(bags):
        with open(os.path.join(tmp_path, "out", f"b{i}.csv"), "rb") as fh:
            assert fh.read() == _pandas_bytes(b)
    blocker = os.path.join(tmp_path, "file")
    open(blocker, "w").close()
    with pytest.raises(OSError):
        with pl.FeatWriter() as w:
            w.save(bags[0], os.path.join(blocker, "sub", "b.csv"))      # a directory cannot be made under a file


def test_reading_a_feature_file_gives_the_float32_values_of_the_pandas_path(tmp_path):
    """train_tcga.py:27-32 + :49 — pd.read_csv -> float64 -> torch.tensor(float32): pipeline.read_feats_csv (dsmil_csv_parse_f32)
    gives the same BITS, on any number of threads, and sklearn's shuffle permutes the array as it permutes the DataFrame."""
    import torch
    from sklearn.utils import shuffle
    rng = np.random.default_rng(23)
    feats = (rng.standard_normal((3000, 96)) * rng.choice([1e-3, 1.0, 300.0], (3000, 1))).astype(np.float32)
    feats[0, :8] = [0.0, -0.0, np.nan, np.inf, -np.inf, 1e15, 1e9, -1e-5]
    path = os.path.join(tmp_path, "bag.csv")
    pl.save_feats_csv(feats, path)
    ref = torch.tensor(pd.read_csv(path).to_numpy(), dtype=torch.float32).numpy()
    for th in (1, 3, 8):
        got = pl.read_feats_csv(path, threads=th)
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), th
    np.random.seed(4)
    a = torch.tensor(shuffle(pd.read_csv(path)).reset_index(drop=True).to_numpy(), dtype=torch.float32).numpy()
    np.random.seed(4)
    b = shuffle(pl.read_feats_csv(path))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_parser_on_fields_the_writer_does_not_produce(tmp_path):
    """Other writers' files: exponents, more digits than the fast path takes, '+', CRLF line ends, blank lines, empty fields —
    equal to pandas' values after the float32 cast; a non-numeric field or a ragged row hands the file to pandas (None)."""
    import torch
    text = ("a,b,c,d\r\n"
            "1.5e-3,+2.25,-0.000123456789012345678,12345678901234567890\r\n"
            "\r\n"
            ",inf,-inf,nan\r\n"
            "7,0.1,1E5,-3.\n"
            "\n"
            ".5,-.25,100,1e-40")
    path = os.path.join(tmp_path, "other.csv")
    with open(path, "w", newline="") as fh:
        fh.write(text)
    ref = torch.tensor(pd.read_csv(path).to_numpy(), dtype=torch.float32).numpy()
    got = pl.read_feats_csv(path)
    assert got.shape == (4, 4) == ref.shape
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_array_equal(np.nan_to_num(got, nan=7.0), np.nan_to_num(ref, nan=7.0))
    for bad in ("0,1\n1.0,abc\n", "0,1\n1.0\n", "0,1\n1.0,2.0,3.0\n"):
        with open(path, "w") as fh:
            fh.write(bad)
        assert pl.read_feats_csv(path) is None
    L = _native.lib()
    out = np.empty((2, 2), np.float32)
    t = np.frombuffer(b"1,2\n3,4\n5,6\n", np.uint8)
    assert L.dsmil_csv_parse_f32(t.ctypes.data, t.size, 2, out.ctypes.data, 2) == _native.DSMIL_E_INVALID      # more rows than room
    assert L.dsmil_csv_parse_f32(t.ctypes.data, 8, 2, out.ctypes.data, 2) == 2 and out.tolist() == [[1.0, 2.0], [3.0, 4.0]]
    assert L.dsmil_csv_parse_f32(None, 8, 2, out.ctypes.data, 2) == _native.DSMIL_E_INVALID


def test_generate_pt_files_reads_bags_like_the_reference(tmp_path, monkeypatch):
    """training.get_bag_feats (train_tcga.py:21-38): label, the shuffled float32 rows and the path — against the pandas path of
    the reference under the same numpy seed."""
    import argparse
    import torch
    from sklearn.utils import shuffle
    from dsmil_wsi_amd import training
    rng = np.random.default_rng(31)
    feats = rng.standard_normal((257, 32)).astype(np.float32)
    path = os.path.join(tmp_path, "s7.csv")
    pl.save_feats_csv(feats, path)
    row = pd.Series([path, 1])
    args = argparse.Namespace(dataset="toy", num_classes=2)
    np.random.seed(11)
    label, got, p = training.get_bag_feats(row, args)
    np.random.seed(11)
    ref = shuffle(pd.read_csv(path)).reset_index(drop=True).to_numpy()
    assert p == path and label.tolist() == [0.0, 1.0]
    assert np.array_equal(torch.tensor(np.array(got), dtype=torch.float32).numpy(), torch.tensor(ref, dtype=torch.float32).numpy())


def test_read_files_gives_the_files_bytes_and_raises_like_open(tmp_path):
    """pipeline.read_files (dsmil_read_files: the loader's file reads, compute_feats.py:21-56, one C call per group of files): the
    bytes `open(p, 'rb').read()` gives, an empty file, a missing file (FileNotFoundError as the Python loader would raise), and the
    raw C-ABI: sizes only / a buffer that is too small."""
    rng = np.random.default_rng(2)
    paths, ref = [], []
    for i in range(70):
        b = b"" if i == 5 else rng.integers(0, 256, int(rng.integers(1, 30000)), dtype=np.uint8).tobytes()
        p = os.path.join(tmp_path, f"t{i}.jpeg")
        with open(p, "wb") as fh:
            fh.write(b)
        paths.append(p)
        ref.append(b)
    got = pl.read_files(paths)
    assert len(got) == 70 and all(bytes(g) == r for g, r in zip(got, ref))
    assert pl.read_files([]) == []
    with pytest.raises(FileNotFoundError):
        pl.read_files(paths[:3] + [os.path.join(tmp_path, "missing.jpeg")])
    L = _native.lib()
    enc = [os.fsencode(p) for p in paths[:4]]
    blob = np.frombuffer(b"\0".join(enc) + b"\0", np.uint8)
    off = np.zeros(4, np.int64)
    np.cumsum([len(e) + 1 for e in enc[:-1]], out=off[1:])
    sizes = np.empty(4, np.int64)
    total = L.dsmil_read_files(blob.ctypes.data, off.ctypes.data, 4, None, 0, sizes.ctypes.data)
    assert total == sum(len(r) for r in ref[:4]) and sizes.tolist() == [len(r) for r in ref[:4]]
    small = np.empty(max(1, total - 1), np.uint8)
    assert L.dsmil_read_files(blob.ctypes.data, off.ctypes.data, 4, small.ctypes.data, total - 1, sizes.ctypes.data) == _native.DSMIL_E_WORKSPACE
    assert L.dsmil_read_files(None, off.ctypes.data, 4, None, 0, sizes.ctypes.data) == _native.DSMIL_E_INVALID
