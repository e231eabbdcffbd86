"""CPU oracle for the DSMIL dual-stream aggregator hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the product
package (``dsmil-wsi_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

This is a numpy restatement (not a copy) of the arithmetic in the reference file
``dsmil.py`` (FCLayer :6-12, BClassifier :27-62, MILNet :64-74) and of the training
objective in ``train_tcga.py:67-71``.  It is pinned against the reference itself:
``tests/golden/make_golden.py`` imports ``/root/reference/dsmil.py`` in the build container,
runs it on seeded inputs with the two shipped weight files and stores the outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against those vectors.

Parameter dictionary layout (numpy arrays), names follow the reference state_dict:
    fc_w  [C,K]   fc_b  [C]            i_classifier.fc.0.{weight,bias}
    q0_w  [Q,K]   q0_b  [Q]            b_classifier.q.0.*   (or b_classifier.q.* if linear)
    q2_w  [Q,Q]   q2_b  [Q]            b_classifier.q.2.*   (absent when nonlinear=False)
    v_w   [K,K]   v_b   [K]            b_classifier.v.1.*   (only when passing_v=True)
    fcc_w [C,C,K] fcc_b [C]            b_classifier.fcc.*
"""
from __future__ import annotations

import numpy as np

Q_DIM = 128  # dsmil.py:31,33 hard-codes the query width


def _f(dtype):
    return np.float64 if dtype in ("f64", np.float64) else np.float32


def instance_logits(x, fc_w, fc_b):
    """dsmil.py:9-12 — FCLayer: c = x @ W^T + b, returned beside the untouched feats."""
    return x @ fc_w.T + fc_b


def query_mlp(x, p, nonlinear=True):
    """dsmil.py:30-33 — q = Linear(K,128)-ReLU-Linear(128,128)-Tanh, or a single Linear."""
    h = x @ p["q0_w"].T + p["q0_b"]
    if not nonlinear:
        return h
    h = np.maximum(h, 0)
    return np.tanh(h @ p["q2_w"].T + p["q2_b"])


def value_proj(x, p, passing_v=False):
    """dsmil.py:34-41 — v = Identity, or Dropout(eval: identity)-Linear(K,K)-ReLU."""
    if not passing_v:
        return x
    return np.maximum(x @ p["v_w"].T + p["v_b"], 0)


def critical_index(c):
    """dsmil.py:52 — row 0 of a descending sort of c along instances = per-class arg-max.

    The reference sort is unstable, so on exact ties its index is implementation defined;
    the build fixes lowest-index-wins (numpy argmax semantics) and parity is asserted on
    tie-free inputs (SURVEY.md §7 'Argmax tie semantics').
    """
    return np.argmax(c, axis=0).astype(np.int64)


def bclassifier_forward(feats, c, p, nonlinear=True, passing_v=False, dtype="f32"):
    """dsmil.py:46-62.  Returns (pred[1,C], A[N,C], B[1,C,Kv], idx[C])."""
    ft = _f(dtype)
    feats = np.asarray(feats, ft)
    c = np.asarray(c, ft)
    p = {k: np.asarray(v, ft) for k, v in p.items()}
    V = value_proj(feats, p, passing_v)                       # :48
    Qm = query_mlp(feats, p, nonlinear)                       # :49
    idx = critical_index(c)                                   # :52
    m_feats = feats[idx]                                      # :53
    q_max = query_mlp(m_feats, p, nonlinear)                  # :54
    s = (Qm @ q_max.T) / np.sqrt(ft(Qm.shape[1]))             # :55-56 (scale by sqrt(128))
    s = s - s.max(axis=0, keepdims=True)
    e = np.exp(s)
    A = e / e.sum(axis=0, keepdims=True)                      # :56 softmax over instances
    B = A.T @ V                                               # :57  [C,Kv]
    pred = np.einsum("ock,ck->o", p["fcc_w"], B) + p["fcc_b"]  # :44,:59-61 Conv1d(C,C,K)
    return pred[None, :], A, B[None], idx


def milnet_forward(x, p, nonlinear=True, passing_v=False, dtype="f32"):
    """dsmil.py:70-74.  Returns (classes[N,C], pred[1,C], A[N,C], B[1,C,K], idx[C])."""
    ft = _f(dtype)
    x = np.asarray(x, ft)
    classes = instance_logits(x, np.asarray(p["fc_w"], ft), np.asarray(p["fc_b"], ft))
    pred, A, B, idx = bclassifier_forward(x, classes, p, nonlinear, passing_v, dtype)
    return classes, pred, A, B, idx


# ----------------------------------------------------------------------------------------
# Training objective and analytic gradients (train_tcga.py:67-72); nonlinear q, v=Identity
# ----------------------------------------------------------------------------------------

def _bce_with_logits(z, y):
    """torch.nn.BCEWithLogitsLoss (mean reduction), train_tcga.py:240."""
    return np.mean(np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z))))


def _sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def train_loss(x, label, p, dtype="f64"):
    """train_tcga.py:67-71 — 0.5*BCE(bag_prediction) + 0.5*BCE(max over instances)."""
    classes, pred, _, _, _ = milnet_forward(x, p, dtype=dtype)
    mx = classes.max(axis=0)
    return 0.5 * _bce_with_logits(pred[0], label) + 0.5 * _bce_with_logits(mx, label)


def train_loss_and_grads(x, label, p, dtype="f64"):
    """Loss of train_tcga.py:67-71 and its gradient w.r.t. every parameter (what
    ``loss.backward()`` at train_tcga.py:72 produces for nonlinear=True, passing_v=False).

    Derivation follows dsmil.py:46-62 backwards.  The sort/index_select indices are not
    differentiable; gradient reaches ``fc`` through (i) the max stream and (ii) nothing else
    (``c`` only feeds the indices), and reaches ``q`` both through Q (all rows) and through
    q_max (the critical rows).
    """
    ft = _f(dtype)
    x = np.asarray(x, ft)
    label = np.asarray(label, ft)
    p = {k: np.asarray(v, ft) for k, v in p.items()}
    N, K = x.shape
    C = p["fc_w"].shape[0]
    # ---- forward with saved intermediates
    c = x @ p["fc_w"].T + p["fc_b"]
    idx = critical_index(c)
    pre1 = x @ p["q0_w"].T + p["q0_b"]
    h1 = np.maximum(pre1, 0)
    Qm = np.tanh(h1 @ p["q2_w"].T + p["q2_b"])
    mx_pre1 = pre1[idx]
    mx_h1 = h1[idx]
    q_max = Qm[idx]  # identical rows of the same MLP
    scale = 1.0 / np.sqrt(ft(Q_DIM))
    s = (Qm @ q_max.T) * scale
    e = np.exp(s - s.max(axis=0, keepdims=True))
    A = e / e.sum(axis=0, keepdims=True)
    B = A.T @ x
    pred = np.einsum("ock,ck->o", p["fcc_w"], B) + p["fcc_b"]
    mx = c[idx, np.arange(C)]
    loss = 0.5 * _bce_with_logits(pred, label) + 0.5 * _bce_with_logits(mx, label)
    # ---- backward
    g = {}
    d_pred = 0.5 * (_sigmoid(pred) - label) / C
    d_mx = 0.5 * (_sigmoid(mx) - label) / C
    # max stream -> fc (only the critical rows)
    g["fc_w"] = np.zeros_like(p["fc_w"])
    g["fc_b"] = np.zeros_like(p["fc_b"])
    for cc in range(C):
        g["fc_w"][cc] += d_mx[cc] * x[idx[cc]]
        g["fc_b"][cc] += d_mx[cc]
    # bag head
    g["fcc_b"] = d_pred.copy()
    g["fcc_w"] = d_pred[:, None, None] * B[None]
    dB = np.einsum("o,ock->ck", d_pred, p["fcc_w"])
    dA = x @ dB.T                                        # [N,C]
    ds = A * (dA - (A * dA).sum(axis=0, keepdims=True))  # softmax over instances
    ds *= scale
    dQ = ds @ q_max                                      # [N,Q]
    dqmax = ds.T @ Qm                                    # [C,Q]
    np.add.at(dQ, idx, dqmax)                            # q_max rows are rows of Q
    dz2 = dQ * (1.0 - Qm * Qm)
    g["q2_w"] = dz2.T @ h1
    g["q2_b"] = dz2.sum(axis=0)
    dh1 = (dz2 @ p["q2_w"]) * (pre1 > 0)
    g["q0_w"] = dh1.T @ x
    g["q0_b"] = dh1.sum(axis=0)
    del mx_pre1, mx_h1
    return loss, g
