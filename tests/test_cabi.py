"""The C-ABI shared library loads on a CPU-only machine and exports EVERY function that
include/dsmil_hip.h declares (no compute calls here — those need a GPU and live in the -m gpu tests);
the ctypes table of the binding lists exactly the same set; error paths that need no device work."""
import ctypes
import os
import re

import pytest

import dsmil  # noqa: F401  (registers the dsmil_wsi_amd package)
import dsmil_wsi_amd._native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dsmil_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dsmil_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for must in ("dsmil_agg_forward", "dsmil_agg_backward", "dsmil_agg_forward_bf16", "dsmil_fc_forward",
                 "dsmil_resnet18in_forward", "dsmil_resnet18in_forward_u8", "dsmil_resnet18bn_forward",
                 "dsmil_resnet_forward", "dsmil_agg_shard_argmax", "dsmil_agg_shard_attend", "dsmil_strerror"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(nat.LIB_PATH), "build with `python __graft_entry__.py`"
    lib = ctypes.CDLL(nat.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, f"declared in include/dsmil_hip.h but not exported: {missing}"


def test_binding_table_matches_the_header():
    assert sorted(nat.SIGNATURES) == _declared()
    L = nat.lib()   # resolves every symbol and sets its signature; raises NativeLibraryError otherwise
    assert L.dsmil_abi_version() >= 1
    assert L.dsmil_agg_mlp_form() in (0, 6, 9)


def test_status_codes_and_sizes_without_a_device():
    L = nat.lib()
    assert L.dsmil_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5):
        assert L.dsmil_strerror(code) not in (b"ok", b"")
    assert L.dsmil_agg_workspace_bytes(1, 10000, 512, 512, 2) > 0
    assert L.dsmil_agg_workspace_bytes(64, 640000, 512, 512, 1) > L.dsmil_agg_workspace_bytes(1, 10000, 512, 512, 1)
    assert L.dsmil_agg_backward_workspace_bytes(10000, 512, 512, 2) > 0
    assert L.dsmil_resnet18_workspace_bytes(256, 224, 224) > 0
    assert L.dsmil_resnet18_workspace_bytes(0, 224, 224) == 0
    assert L.dsmil_agg_tile_rows(64, 640000) == 128 and L.dsmil_agg_tile_rows(1, 10000) == 32
    # null pointers are rejected before any launch
    assert L.dsmil_agg_forward(None, None, None, 1, 10, 10, None, None, None, None, None, None, None, None, 0, None) == -1
    assert L.dsmil_fc_forward(None, 10, 512, 2, None, None, None, None) != 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(nat.NativeLibraryError):
        nat.lib()
