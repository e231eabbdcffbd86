"""Every selectable MFMA form stays parity-green: the default suite runs the default forms (aggregator
query MLP and Winograd convs on bf16 MFMA over exact three-plane cuts); this file re-runs a small
aggregator + embedder check in subprocesses with the alternatives selected.  The product library has ONE form; the
knobs (read once per process: DSMIL_MLP=f32 / s9, DSMIL_WINO=f32 / s9, DSMIL_CONV=f32) exist in the experiment build
only (libdsmil_hip_expt.so, built by __graft_entry__.build() next to the product library)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("env,form", [({"DSMIL_MLP": "f32", "DSMIL_WINO": "f32", "DSMIL_CONV": "f32"}, 0),
                                      ({"DSMIL_MLP": "s9", "DSMIL_WINO": "s9"}, 9), ({}, 6)])
def test_alternative_mfma_forms(env, form):
    e = dict(os.environ)
    for k in ("DSMIL_MLP", "DSMIL_WINO", "DSMIL_CONV"):
        e.pop(k, None)
    e.update(env)
    if env:   # alternative forms live in the experiment build
        lib = os.path.join(os.path.dirname(HERE), "dsmil-wsi_amd", "libdsmil_hip_expt.so")
        assert os.path.exists(lib), "python dsmil-wsi_amd/build.py --variant expt -DDSMIL_EXPERIMENTS (done by __graft_entry__.build())"
        e["DSMIL_NATIVE_LIB"] = "libdsmil_hip_expt.so"
    out = subprocess.run([sys.executable, os.path.join(HERE, "_form_check.py")], env=e, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("FORM-OK")]
    assert line and int(line[0].split()[1]) == form
