"""dsmil-wsi_amd — MI355X-native implementation of the DSMIL hot paths.

The directory name is fixed by the project layout and is not a valid Python identifier; the
repo-root ``dsmil.py`` shim registers this package as ``dsmil_wsi_amd`` and re-exports the model
API so that the reference scripts' ``import dsmil as mil`` keeps working unchanged.
"""
from .modules import FCLayer, IClassifier, BClassifier, MILNet  # noqa: F401

__all__ = ["FCLayer", "IClassifier", "BClassifier", "MILNet"]
