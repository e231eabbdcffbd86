"""Every selectable MFMA form stays parity-green: the default suite runs the default forms (aggregator query MLP of a lone bag on
bf16 MFMA over exact three-plane cuts, six products; batches and — since round 5 — every conv of the embedder on fp16 MFMA over
two-plane cuts, three products); this file re-runs a small
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
                                      ({"DSMIL_MLP": "s9", "DSMIL_WINO": "s9"}, 9),
                                      ({"DSMIL_WINO": "s6", "DSMIL_CONV": "s6", "DSMIL_FORM_CHECK_LIB": "expt"}, 6),   # the embedder's round 2-4 form
                                      ({}, 6)])
def test_alternative_mfma_forms(env, form):
    e = dict(os.environ)
    for k in ("DSMIL_MLP", "DSMIL_WINO", "DSMIL_CONV"):
        e.pop(k, None)
    e.update(env)
    e.pop("DSMIL_FORM_CHECK_LIB", None)
    if env:   # alternative forms live in the experiment build
        lib = os.path.join(os.path.dirname(HERE), "dsmil-wsi_amd", "libdsmil_hip_expt.so")
        assert os.path.exists(lib), "python dsmil-wsi_amd/build.py --variant expt -DDSMIL_EXPERIMENTS (done by __graft_entry__.build())"
        e["DSMIL_NATIVE_LIB"] = "libdsmil_hip_expt.so"
    out = subprocess.run([sys.executable, os.path.join(HERE, "_form_check.py")], env=e, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("FORM-OK")]
    assert line and int(line[0].split()[1]) == form


def test_one_wave_winograd_unit_is_bit_identical_to_the_two_wave_unit(tmp_path):
    """k_conv_wino_w1 (one wave per SIMD, tiled weights, layers with Cout % 128 == 0; the product default) computes exactly
    what k_conv_wino_s3 computes — same plane cuts, product order and inverse-transform order — so the features of the
    same inputs must agree BIT FOR BIT between the default and DSMIL_WINO_KERNEL=unit (experiment build), incl. odd sizes
    and batch sizes that leave partial units."""
    import numpy as np
    lib = os.path.join(os.path.dirname(HERE), "dsmil-wsi_amd", "libdsmil_hip_expt.so")
    assert os.path.exists(lib), "python dsmil-wsi_amd/build.py --variant expt -DDSMIL_EXPERIMENTS (done by __graft_entry__.build())"
    tool = os.path.join(os.path.dirname(HERE), "tools", "wino_check.py")
    outs = []
    for tag, env in (("unit", {"DSMIL_WINO_KERNEL": "unit"}), ("w1", {})):
        e = dict(os.environ)
        e.pop("DSMIL_WINO_KERNEL", None)
        e.update(env)
        e["DSMIL_NATIVE_LIB"] = "libdsmil_hip_expt.so"
        o = str(tmp_path / f"{tag}.npy")
        r = subprocess.run([sys.executable, tool, "run", o], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outs.append(np.load(o))
    assert np.isfinite(outs[1]).all()
    assert np.array_equal(outs[0], outs[1]), float(np.abs(outs[0] - outs[1]).max())


def test_fused_stem_maxpool_is_bit_identical_to_the_two_kernel_path(tmp_path):
    """The InstanceNorm trunks take the 3x3 stride-2 max-pool inside the stem kernel (pooled raw map + halo rows, then
    k_pool_fix_norm); DSMIL_STEM_FUSE=0 (experiment build) keeps the raw stem output + k_norm_relu_maxpool.  max is exact
    and the normalisation is the same expression, so the features must agree BIT FOR BIT, incl. sizes whose tiles are
    partly outside the image."""
    import numpy as np
    lib = os.path.join(os.path.dirname(HERE), "dsmil-wsi_amd", "libdsmil_hip_expt.so")
    assert os.path.exists(lib), "python dsmil-wsi_amd/build.py --variant expt -DDSMIL_EXPERIMENTS (done by __graft_entry__.build())"
    tool = os.path.join(os.path.dirname(HERE), "tools", "wino_check.py")
    outs = []
    for tag, env in (("two", {"DSMIL_STEM_FUSE": "0"}), ("fused", {})):
        e = dict(os.environ)
        e.pop("DSMIL_STEM_FUSE", None)
        e.update(env)
        e["DSMIL_NATIVE_LIB"] = "libdsmil_hip_expt.so"
        o = str(tmp_path / f"{tag}.npy")
        r = subprocess.run([sys.executable, tool, "run", o], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outs.append(np.load(o))
    assert np.isfinite(outs[1]).all()
    assert np.array_equal(outs[0], outs[1]), float(np.abs(outs[0] - outs[1]).max())
