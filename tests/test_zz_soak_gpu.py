"""Soaks and stress tests: LAST in the collection order (the driver runs `pytest -m gpu -x`: a flaky long test must never
stand between the runner and the parity tests of a SURVEY §8 row — tests/test_00_rows_gpu.py holds one canonical parity case per
row and runs first).

* the in-launch hand-off of the critical query (k_attend_hs) under load with a NaN-poisoned workspace, forward and train step,
  bit for bit against the separate-launch path;
* tests/soak_f3.py: random ragged, six-decade-scaled batches through k_attend_f3 against the fp64 oracle;
* the co-resident bf16 pass (k_logits_pipe / 4-wave k_qmax / lean k_finish beside another stream's k_attend_bf16_res) on two and
  three streams over ragged batches, 120 + 120 passes, bit for bit against the one-stream result of the plain kernels.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import agg_oracle as orc
from conftest import load_weights
from inputs import make_bag, make_label
from test_agg_gpu import _cmp

pytestmark = pytest.mark.gpu


class _Load:
    """A bandwidth-hungry kernel (k_fc over a 640 000 x 512 batch, 1.3 GB per launch) kept running on a second stream
    so that the chip is UNEVENLY loaded while the hand-off below is exercised (idle chips hide stale reads)."""

    def __init__(self):
        from dsmil_wsi_amd import ops
        self.ops = ops
        self.s = torch.cuda.Stream()
        self.x = torch.randn(640_000, 512, device="cuda")
        self.w = torch.randn(2, 512, device="cuda")
        self.b = torch.zeros(2, device="cuda")
        self.ev = []

    def kick(self):
        if len(self.ev) >= 6:             # bounded queue: at most six launches ahead
            self.ev.pop(0).synchronize()
        with torch.cuda.stream(self.s):
            self.ops.fc_forward(self.x, self.w, self.b)
            e = torch.cuda.Event()
            e.record(self.s)
        self.ev.append(e)


def _poison_workspace(ops):
    """Every word of the native workspace of the current stream (q_max, hand-off flags, tile partials) becomes
    0xFFFFFFFF = NaN / "flag set": a read of anything the CURRENT call has not written shows up as a NaN or, for a
    flag that was not cleared, as a tile that did not wait."""
    if ops._ws_last[0] is not None:
        ops._ws_last[0].fill_(0xFF)


def test_inline_query_handoff_stress_forward():
    """The in-launch hand-off of the critical query (k_attend_hs, csrc/agg_hs.h: flag poll -> agent acquire -> barrier ->
    plain loads) under the conditions that expose an invalid one (round 4 shipped one: stale q_max, wrong pred): 2 400
    lone-bag forwards alternating between bags whose critical queries differ, the workspace NaN-poisoned before every
    call, a 1.3 GB streaming kernel running beside them, every output word compared with the separate-launch path
    (dsmil_agg_inline_query(0): k_qmax between the logits pass and the attend kernel), bit for bit."""
    from dsmil_wsi_amd import ops, _native
    L = _native.lib()
    for tag, sizes in (("tcga", (10000, 2000, 700, 5000)), ("c16", (10000, 3000))):
        p = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in load_weights(tag).items()}
        bags = [torch.from_numpy(make_bag(321 + 7 * i, n, 512)).cuda() for i, n in enumerate(sizes)]
        prev = L.dsmil_agg_inline_query(0)
        try:
            refs = [[t.clone() for t in ops.agg_forward(b, [b.shape[0]], p)] for b in bags]
            torch.cuda.synchronize()
            assert L.dsmil_agg_inline_query(1) == 0
            load = _Load()
            bad = torch.zeros((), dtype=torch.int64, device="cuda")
            n_iter = 1600 if tag == "tcga" else 800
            for it in range(n_iter):
                if it % 3 == 0:
                    load.kick()
                _poison_workspace(ops)
                i = (it * 7 + it // 5) % len(bags)
                out = ops.agg_forward(bags[i], [bags[i].shape[0]], p)
                for a, b in zip(out, refs[i]):
                    bad += (a != b).sum()          # (NaN != x is True)
            torch.cuda.synchronize()
            assert int(bad) == 0, f"{tag}: {int(bad)} output words differ from the k_qmax path"
        finally:
            L.dsmil_agg_inline_query(prev if prev in (0, 1) else 1)
    # the reference of the comparison is itself right: fp64 oracle on one bag (the last weight set of the loop, c16)
    r = orc.milnet_forward(bags[0].cpu().numpy(), load_weights("c16"), dtype="f64")
    _cmp(refs[0][:4], r[0], r[1], r[2], r[3])


def test_inline_query_handoff_stress_train_step():
    """The same hand-off inside dsmil_agg_train_step (its forward is k_attend_hs): 600 fused Adam steps over alternating
    bags under load with a poisoned workspace follow the separate-launch path bit for bit — every step's loss and the
    final parameters and moments."""
    from dsmil_wsi_amd import ops, _native
    L = _native.lib()
    names = ("fc_w", "fc_b", "q0_w", "q0_b", "q2_w", "q2_b", "fcc_w", "fcc_b")
    w0 = load_weights("tcga")
    sizes = (3000, 10000, 1200)
    bags = [torch.from_numpy(make_bag(77 + 3 * i, n, 512)).cuda() for i, n in enumerate(sizes)]
    labels = [torch.from_numpy(make_label(5 + i, 2)).cuda().float() for i in range(len(sizes))]
    n_steps = 600

    def run(mode, stressed):
        prev = L.dsmil_agg_inline_query(mode)
        try:
            params = [torch.from_numpy(np.ascontiguousarray(w0[k])).cuda().float().contiguous() for k in names]
            m = [torch.zeros_like(t) for t in params]
            v = [torch.zeros_like(t) for t in params]
            losses = torch.zeros(n_steps, device="cuda")
            load = _Load() if stressed else None
            for it in range(n_steps):
                if stressed and it % 3 == 0:
                    load.kick()
                if stressed:
                    _poison_workspace(ops)
                i = (it * 5 + it // 7) % len(bags)
                ops.agg_train_step(bags[i], labels[i], params, m, v, it + 1, 1e-4, (0.5, 0.9), 1e-8, 5e-3,
                                   loss_out=losses[it:it + 1])
            torch.cuda.synchronize()
            return losses, params, m, v
        finally:
            L.dsmil_agg_inline_query(prev if prev in (0, 1) else 1)

    ref = run(0, False)
    got = run(1, True)
    assert torch.isfinite(ref[0]).all()
    assert torch.equal(ref[0], got[0]), f"losses differ at steps {torch.nonzero(ref[0] != got[0]).flatten()[:8].tolist()}"
    for group_r, group_g in zip(ref[1:], got[1:]):
        for a, b in zip(group_r, group_g):
            assert torch.equal(a, b)


def test_batch_form_f3_random_ragged_batches_soak():
    """tests/soak_f3.py, six seeded rounds: random ragged batches (bags of 1 .. 20 000 rows, per-bag scales over six decades, K
    in {128, 256, 384, 512}, C in {1, 2}) through k_attend_f3 — every bag against the fp64 oracle at the parity bar (B / pred
    errors in units of max(1, |B|max)), two runs bit-identical, two other persistent grids within the same bar."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "soak_f3.py"), "6"], capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_coresident_bf16_pass_soak_on_two_and_three_streams():
    """Round 6: the kernels of one bf16 batch share compute units with the persistent attend kernel of another (registers and
    LDS are budgeted for it: csrc/agg_res.h, dsmil_agg_logits_form).  Sharing a CU must never change a bit: three RAGGED
    batches (short bags, one long bag, partial last tiles) dealt to two, then three streams, 120 passes each, every output of
    every pass compared with the one-stream result of the plain kernels (form 0)."""
    import dsmil_wsi_amd.ops as ops
    from dsmil_wsi_amd import _native
    L = _native.lib()
    w = {k: torch.from_numpy(v).cuda() for k, v in load_weights("tcga").items()}
    rng = np.random.default_rng(606)
    batches = []
    for i in range(3):
        lengths = [int(n) for n in rng.integers(900, 4000, size=28)] + [30000 + 517 * i, 1, 129]
        x = torch.from_numpy(make_bag(8100 + i, sum(lengths), 512)).cuda().to(torch.bfloat16)
        batches.append((x, lengths))
    prev = L.dsmil_agg_logits_form(0)
    try:
        want = [[t.clone() for t in ops.agg_forward(x, lengths, w)] for x, lengths in batches]
        torch.cuda.synchronize()
    finally:
        L.dsmil_agg_logits_form(prev)
    for streams in (2, 3):
        pool = ops.StreamPool(streams)
        got = []
        for r in range(120):
            x, lengths = batches[r % 3]
            got.append(pool.run(lambda x=x, lengths=lengths: [t.clone() for t in ops.agg_forward(x, lengths, w)]))
        pool.join()
        torch.cuda.synchronize()
        for r, g in enumerate(got):
            for k, (a_, b_) in enumerate(zip(g, want[r % 3])):
                assert torch.equal(a_, b_), (streams, r, k)
