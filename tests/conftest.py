"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path.

``-m "not gpu"`` runs on the CPU-only build container (oracle vs golden vectors, host logic,
C-ABI symbol checks, gloo world_size-2 tests); ``-m gpu`` are the parity tests proper and call
the HIP kernels through the C-ABI on a real MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


import dsmil  # noqa: E402,F401  the root shim registers the `dsmil_wsi_amd` package (the directory name
#                          `dsmil-wsi_amd` is not importable by itself): tests may import it in any order / alone


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "agg_golden.npz"))


def load_weights(tag):
    from dsmil_wsi_amd.synthetic import load_weights as lw
    return lw(tag)
