"""Helpers shared by the tests: MILNet built from the example weight sets (dsmil-wsi_amd/synthetic.py)."""
import dsmil  # noqa: F401  (registers the package)
from dsmil_wsi_amd.synthetic import VARIANT, build_net, load_weights, state_dict_from_npz  # noqa: F401
