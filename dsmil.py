"""Drop-in ``dsmil`` module (the reference scripts do ``import dsmil as mil``: compute_feats.py:1,
attention_map.py:1, train_tcga.py:225, train_mil.py:123).  Exposes FCLayer, IClassifier,
BClassifier and MILNet with the reference's signatures, backed by the gfx950 kernels in
``dsmil-wsi_amd/``."""
import importlib.util
import os
import sys

_PKG = "dsmil_wsi_amd"


def _load_package():
    if _PKG in sys.modules:
        return sys.modules[_PKG]
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dsmil-wsi_amd")
    spec = importlib.util.spec_from_file_location(_PKG, os.path.join(d, "__init__.py"),
                                                  submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_PKG] = mod
    spec.loader.exec_module(mod)
    return mod


_pkg = _load_package()
FCLayer = _pkg.FCLayer
IClassifier = _pkg.IClassifier
BClassifier = _pkg.BClassifier
MILNet = _pkg.MILNet

__all__ = ["FCLayer", "IClassifier", "BClassifier", "MILNet"]
