"""Seeded synthetic inputs: re-exported from the package (dsmil-wsi_amd/synthetic.py), where bench.py and smoke() also
take them from — the golden generator (tests/golden/make_golden.py) and the tests import them under this name."""
import dsmil  # noqa: F401  (registers the package)
from dsmil_wsi_amd.synthetic import (RESNET18_CONVS, make_bag, make_label, make_patches,  # noqa: F401
                                     make_resnet18_weights)
