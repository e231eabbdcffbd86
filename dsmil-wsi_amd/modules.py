"""Drop-in mirror of the reference model API (dsmil.py:6-74): FCLayer, IClassifier, BClassifier,
MILNet — same constructor signatures, forward tuples, attribute names and state_dict keys.

Parameters live in ordinary nn.Linear / nn.Conv1d sub-modules so that ``.apply(init)``
(train_tcga.py:229-239), ``state_dict()/load_state_dict()`` (train_tcga.py:186, testing_c16.py:122),
``copy.deepcopy`` and ``.cuda()/.cpu()`` behave exactly as with the reference.  When the input
is a CUDA(HIP) tensor the forward runs in libdsmil_hip.so (hand-written gfx950 kernels); a CPU
tensor takes the plain torch-CPU route (BASELINE config 0: MUSK1 plumbing via train_mil.py).
There is no silent GPU fallback: a missing native library raises.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

Q_DIM = 128


class FCLayer(nn.Module):
    """dsmil.py:6-12."""

    def __init__(self, in_size, out_size=1):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(in_size, out_size))

    def forward(self, feats):
        lin = self.fc[0]
        if feats.is_cuda:
            x = _FCFunction.apply(feats, lin.weight, lin.bias)
        else:
            x = lin(feats)
        return feats, x


def resnet_convs_of(fe):
    """(convs, bn_norms) when ``fe`` is a ResNet-18 trunk with fc = Identity that the native embedder
    covers — InstanceNorm (bn_norms None) or eval-mode BatchNorm — else None."""
    from .resnet import resnet18_bn_parts, resnet18_in_convs
    if not isinstance(getattr(fe, "fc", None), nn.Identity):
        return None
    convs = resnet18_in_convs(fe)
    if convs is not None:
        return convs, None
    return resnet18_bn_parts(fe)


class IClassifier(nn.Module):
    """dsmil.py:14-25: (feats.view(B,-1), Linear(feats)) around an arbitrary feature extractor."""

    def __init__(self, feature_extractor, feature_size, output_class):
        super().__init__()
        self.feature_extractor = feature_extractor
        self.fc = nn.Linear(feature_size, output_class)
        # not part of the reference API: "fp32" (the parity path), "half" or "bf16" — OPT-IN reduced precisions of the native
        # trunk (fp16 activations behind the stem / one fp16 plane per conv operand: ~2.6e-3 feature error; bf16 activations:
        # ~2e-2 — ops.resnet18in_forward)
        self.embed_precision = "fp32"

    def forward(self, x):
        fe = self.feature_extractor
        if x.dtype == torch.uint8 and (x.dim() != 4 or x.shape[3] != 3):
            # decoded images, uint8 NHWC [B,H,W,3] (new ingest path)
            raise ValueError(f"uint8 patches must be NHWC [B,H,W,3], got {tuple(x.shape)}")
        grad_on = torch.is_grad_enabled()
        trunk = None
        if x.is_cuda and x.dtype in (torch.float32, torch.uint8) and x.dim() == 4 and not (
                grad_on and (x.requires_grad or any(p.requires_grad for p in fe.parameters()))):
            # ResNet-18/34 + InstanceNorm / frozen BatchNorm with fc = Identity (ours or torchvision's,
            # compute_feats.py:157,170): the trunk runs in one native launch sequence
            trunk = resnet_convs_of(fe)
        if trunk is not None:
            head_trains = grad_on and (self.fc.weight.requires_grad or self.fc.bias.requires_grad)
            if not head_trains:
                # features and instance logits from the same launch sequence (uint8: ToTensor fused in the stem)
                return ops.resnet18in_forward(x, trunk[0], self.fc.weight, self.fc.bias, bn_norms=trunk[1],
                                              precision=self.embed_precision)
            # frozen trunk + trainable linear head (what compute_feats.py:168-173 / attention_map.py set up):
            # the reference differentiates through self.fc (dsmil.py:24), so the head goes through autograd
            feats, _ = ops.resnet18in_forward(x, trunk[0], bn_norms=trunk[1], precision=self.embed_precision)
            return feats, _FCFunction.apply(feats, self.fc.weight, self.fc.bias)
        if x.dtype == torch.uint8:
            # no native stem will take the bytes: apply VF.to_tensor here (compute_feats.py:35-39)
            x = x.permute(0, 3, 1, 2).to(torch.float32).div(255)
        feats = fe(x)
        feats = feats.view(feats.shape[0], -1)
        if feats.is_cuda:
            c = _FCFunction.apply(feats, self.fc.weight, self.fc.bias)
        else:
            c = self.fc(feats)
        return feats, c


class BClassifier(nn.Module):
    """dsmil.py:27-62."""

    def __init__(self, input_size, output_class, dropout_v=0.0, nonlinear=True, passing_v=False):
        super().__init__()
        if nonlinear:
            self.q = nn.Sequential(nn.Linear(input_size, Q_DIM), nn.ReLU(), nn.Linear(Q_DIM, Q_DIM), nn.Tanh())
        else:
            self.q = nn.Linear(input_size, Q_DIM)
        if passing_v:
            self.v = nn.Sequential(nn.Dropout(dropout_v), nn.Linear(input_size, input_size), nn.ReLU())
        else:
            self.v = nn.Identity()
        self.fcc = nn.Conv1d(output_class, output_class, kernel_size=input_size)

    # -- helpers --------------------------------------------------------------------------
    @property
    def nonlinear(self):
        return isinstance(self.q, nn.Sequential)

    @property
    def passing_v(self):
        return not isinstance(self.v, nn.Identity)

    def _weights(self):
        if self.nonlinear:
            q0, q2 = self.q[0], self.q[2]
            d = {"q0_w": q0.weight, "q0_b": q0.bias, "q2_w": q2.weight, "q2_b": q2.bias}
        else:
            d = {"q0_w": self.q.weight, "q0_b": self.q.bias, "q2_w": None, "q2_b": None}
        d["fcc_w"] = self.fcc.weight
        d["fcc_b"] = self.fcc.bias
        return d

    def _forward_cpu(self, feats, c):
        V = self.v(feats)
        Q = self.q(feats).view(feats.shape[0], -1)
        m_indices = torch.argmax(c, dim=0)  # lowest index on ties
        q_max = self.q(feats.index_select(0, m_indices))
        A = F.softmax(Q.mm(q_max.t()) / (Q.shape[1] ** 0.5), dim=0)
        B = A.t().mm(V).unsqueeze(0)
        C = self.fcc(B).view(1, -1)
        return C, A, B

    def forward(self, feats, c):
        if not feats.is_cuda:
            return self._forward_cpu(feats, c)
        vals = self.v(feats) if self.passing_v else None  # passing_v: unused by every script
        w = self._weights()
        pred, A, B = _AggFunction.apply(feats, c, vals, None, None, w["q0_w"], w["q0_b"], w["q2_w"],
                                        w["q2_b"], w["fcc_w"], w["fcc_b"], self.nonlinear)[1:4]
        return pred, A, B


class MILNet(nn.Module):
    """dsmil.py:64-74."""

    def __init__(self, i_classifier, b_classifier):
        super().__init__()
        self.i_classifier = i_classifier
        self.b_classifier = b_classifier

    def forward(self, x):
        ic, bc = self.i_classifier, self.b_classifier
        if (x.is_cuda and isinstance(ic, FCLayer) and isinstance(bc, BClassifier)
                and not bc.passing_v and x.dim() == 2):
            # one fused native call: instance logits + aggregator (dsmil.py:70-74)
            w = bc._weights()
            lin = ic.fc[0]
            classes, pred, A, B = _AggFunction.apply(x, None, None, lin.weight, lin.bias, w["q0_w"],
                                                     w["q0_b"], w["q2_w"], w["q2_b"], w["fcc_w"],
                                                     w["fcc_b"], bc.nonlinear)[0:4]
            return classes, pred, A, B
        feats, classes = ic(x)
        prediction_bag, A, B = bc(feats, classes)
        return classes, prediction_bag, A, B

    def graphed(self, n_rows):
        """A hipGraph-replayed forward for bags of exactly ``n_rows`` rows (inference; weights frozen): returns a
        callable feats -> (classes, pred [1,C], A, B [1,C,K]).  Single-bag latency is launch-bound otherwise."""
        ic, bc = self.i_classifier, self.b_classifier
        if not (isinstance(ic, FCLayer) and isinstance(bc, BClassifier) and not bc.passing_v):
            raise NotImplementedError("graphed forward: FCLayer + BClassifier with v = Identity")
        w = {k: (v.detach() if v is not None else None) for k, v in bc._weights().items()}
        lin = ic.fc[0]
        w["fc_w"], w["fc_b"] = lin.weight.detach(), lin.bias.detach()
        g = ops.GraphedAggForward(w, n_rows, lin.in_features, nonlinear=bc.nonlinear, device=lin.weight.device)

        def run(feats):
            classes, pred, A, B, _ = g(feats)
            return classes, pred, A, B
        run.graph = g
        return run

    def bag_loss(self, feats, label, row_map=None):
        """The training objective of one bag, train_tcga.py:64-71 / train_mil.py, as ONE native forward + loss head
        (and one native backward under autograd):
            bag_feats = feats[row_map]                      (dropout_patches :78-83, folded into the row loads)
            ins, bag, _, _ = milnet(bag_feats);  mx = max(ins, 0)
            loss = 0.5 BCEWithLogitsLoss(bag, y) + 0.5 BCEWithLogitsLoss(mx, y)
        Returns (loss [], bag_prediction [1,C], max_prediction [C]).  CUDA fp32 bags with FCLayer + BClassifier
        (v = Identity) take the fused path; everything else composes the same objective from torch ops."""
        ic, bc = self.i_classifier, self.b_classifier
        if (feats.is_cuda and feats.dtype == torch.float32 and feats.dim() == 2 and isinstance(ic, FCLayer)
                and isinstance(bc, BClassifier) and not bc.passing_v and not feats.requires_grad
                and ic.fc[0].out_features <= 64):   # dsmil_agg_loss_head: one wave of classes; more take the torch expression
            if row_map is not None and row_map.numel():
                # an out-of-range index would become an out-of-bounds device read in the row loads: checked once per
                # bag on the device, surfaced with the step's only host sync (the loss .item() of train_tcga.py:74)
                torch._assert_async((row_map.min() >= 0) & (row_map.max() < feats.shape[0]),
                                    "row_map index out of range")
            w = bc._weights()
            lin = ic.fc[0]
            loss, pred, mx = _BagLossFunction.apply(feats, label, row_map, lin.weight, lin.bias, w["q0_w"], w["q0_b"],
                                                    w["q2_w"], w["q2_b"], w["fcc_w"], w["fcc_b"], bc.nonlinear)
            return loss, pred, mx
        x = feats if row_map is None else feats.index_select(0, row_map)
        ins, bag, _, _ = self(x)
        mx, _ = torch.max(ins, 0)
        y = label.view(1, -1).to(bag.dtype)
        loss = 0.5 * F.binary_cross_entropy_with_logits(bag.view(1, -1), y) + \
            0.5 * F.binary_cross_entropy_with_logits(mx.view(1, -1), y)
        return loss, bag, mx

    @torch.no_grad()
    def forward_bags(self, bags):
        """Batched inference over many independent bags in ONE native call sequence ("varlen").

        ``bags`` is a list of [N_i, K] CUDA tensors, or a tuple (feats [sum N_i, K], lengths).
        Returns a list of (classes, pred, A, B) tuples shaped like ``forward``'s.  New capability
        (the reference loops one bag per iteration, train_tcga.py:92-99)."""
        ic, bc = self.i_classifier, self.b_classifier
        if not (isinstance(ic, FCLayer) and isinstance(bc, BClassifier) and not bc.passing_v):
            return [self.forward(b) for b in bags]
        if isinstance(bags, tuple):
            feats, lengths = bags
        else:
            lengths = [int(b.shape[0]) for b in bags]
            feats = torch.cat(list(bags), dim=0)
        w = bc._weights()
        lin = ic.fc[0]
        w["fc_w"], w["fc_b"] = lin.weight, lin.bias
        classes, pred, A, B, _ = ops.agg_forward(feats, lengths, {k: (v.detach() if v is not None else None)
                                                                  for k, v in w.items()},
                                                 nonlinear=bc.nonlinear)
        out, o = [], 0
        for i, n in enumerate(lengths):
            out.append((classes[o:o + n], pred[i:i + 1], A[o:o + n], B[i:i + 1]))
            o += n
        return out


# ---------------------------------------------------------------------------------------------
# autograd glue
# ---------------------------------------------------------------------------------------------
class _FCFunction(torch.autograd.Function):
    """c = x W^T + b in the native library; backward is three small dense products."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        if x.dtype != torch.float32:  # non-fp32 rows (bf16 storage): f32 accumulate, result in x.dtype
            return ops.fc_forward(x.detach().float(), w.detach().float(), b.detach().float()).to(x.dtype)
        return ops.fc_forward(x.detach(), w.detach(), b.detach())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g32, x32 = g.float(), x.float()   # bf16-storage rows: gradients accumulate in f32 like the forward
        gx = g32.mm(w.float()).to(x.dtype) if ctx.needs_input_grad[0] else None
        gw = g32.t().mm(x32).to(w.dtype) if ctx.needs_input_grad[1] else None
        gb = g32.sum(0).to(w.dtype) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


class _BagLossFunction(torch.autograd.Function):
    """Forward = dsmil_agg_forward_ex (row map) + dsmil_agg_loss_head; backward = dsmil_agg_backward_ex with the sparse
    max-stream gradient.  Replaces, per training step, the row gather, torch.max, two BCEWithLogitsLoss graphs and the
    dense [N,C] instance-logit gradient of train_tcga.py:64-72."""

    @staticmethod
    def forward(ctx, feats, label, row_map, fc_w, fc_b, q0_w, q0_b, q2_w, q2_b, fcc_w, fcc_b, nonlinear):
        det = lambda t: t.detach() if t is not None else None
        w = {"fc_w": det(fc_w), "fc_b": det(fc_b), "q0_w": det(q0_w), "q0_b": det(q0_b),
             "q2_w": det(q2_w), "q2_b": det(q2_b), "fcc_w": det(fcc_w), "fcc_b": det(fcc_b)}
        N = int(row_map.numel()) if row_map is not None else feats.shape[0]
        classes, pred, A, B, idx = ops.agg_forward(feats.detach(), [N], w, nonlinear=nonlinear, row_map=row_map)
        loss, max_pred, g_pred, g_max = ops.agg_loss_head(classes, pred, idx, label.detach())
        ctx.nonlinear = nonlinear
        ctx.save_for_backward(feats, row_map, fc_w, q0_w, q0_b, q2_w, q2_b, fcc_w, A, B, idx, g_pred, g_max)
        ctx.mark_non_differentiable(pred, max_pred)
        return loss, pred, max_pred

    @staticmethod
    def backward(ctx, g_loss, _g_pred, _g_max):
        feats, row_map, fc_w, q0_w, q0_b, q2_w, q2_b, fcc_w, A, B, idx, g_pred, g_max = ctx.saved_tensors
        w = {"fc_w": fc_w, "fc_b": None, "q0_w": q0_w, "q0_b": q0_b, "q2_w": q2_w, "q2_b": q2_b,
             "fcc_w": fcc_w, "fcc_b": None}
        # every parameter gradient is linear in (g_pred, g_max): the upstream scalar scales those two [C] vectors
        g = ops.agg_backward(feats, w, A, B, idx, g_pred * g_loss, g_max=g_max * g_loss, row_map=row_map,
                             nonlinear=ctx.nonlinear)
        return (None, None, None, g["fc_w"], g["fc_b"], g["q0_w"], g["q0_b"], g.get("q2_w"), g.get("q2_b"),
                g["fcc_w"], g["fcc_b"], None)


class _AggFunction(torch.autograd.Function):
    """Forward = dsmil_agg_forward (HIP).  Backward = analytic gradient of dsmil.py:46-62 (the
    arg-max indices are constants, as in the reference's autograd graph); it re-derives Q from
    the saved inputs (dsmil_agg_backward, csrc/agg_bwd.hip — SURVEY.md §8(f) row N1)."""

    @staticmethod
    def forward(ctx, feats, c_in, vals, fc_w, fc_b, q0_w, q0_b, q2_w, q2_b, fcc_w, fcc_b, nonlinear):
        det = lambda t: t.detach() if t is not None else None
        w = {"fc_w": det(fc_w), "fc_b": det(fc_b), "q0_w": det(q0_w), "q0_b": det(q0_b),
             "q2_w": det(q2_w), "q2_b": det(q2_b), "fcc_w": det(fcc_w), "fcc_b": det(fcc_b)}
        N = feats.shape[0]
        classes, pred, A, B, idx = ops.agg_forward(feats.detach(), [N], w, classes_in=det(c_in),
                                                   vals=det(vals), nonlinear=nonlinear)
        if feats.dtype == torch.bfloat16:
            # bf16-storage path (BASELINE config 2) is inference only; results keep the input dtype
            ctx.bf16 = True
            out = tuple(t.to(torch.bfloat16) for t in (classes, pred, A, B))
            ctx.mark_non_differentiable(*out, idx)
            return (*out, idx)
        ctx.bf16 = False
        ctx.nonlinear = nonlinear
        ctx.has_cin = c_in is not None
        ctx.has_vals = vals is not None
        ctx.save_for_backward(feats, vals, fc_w, q0_w, q0_b, q2_w, q2_b, fcc_w, A, B, idx)
        ctx.mark_non_differentiable(idx)
        return classes, pred, A, B, idx

    @staticmethod
    def backward(ctx, g_cls, g_pred, g_A, g_B, _g_idx):
        if ctx.bf16:
            raise NotImplementedError("the bf16-storage aggregator path is inference only")
        if not ctx.needs_input_grad[0]:
            return _AggFunction._backward_native(ctx, g_cls, g_pred, g_A, g_B)
        return _AggFunction._backward_dense(ctx, g_cls, g_pred, g_A, g_B)

    @staticmethod
    def _backward_native(ctx, g_cls, g_pred, g_A, g_B):
        """dsmil_agg_backward (HIP): every parameter gradient, and g_vals for a trainable v."""
        feats, vals, fc_w, q0_w, q0_b, q2_w, q2_b, fcc_w, A, B, idx = ctx.saved_tensors
        C = fcc_w.shape[0]
        if g_pred is None:
            g_pred = torch.zeros((1, C), device=feats.device)
        w = {"fc_w": fc_w, "fc_b": None, "q0_w": q0_w, "q0_b": q0_b, "q2_w": q2_w, "q2_b": q2_b,
             "fcc_w": fcc_w, "fcc_b": None}
        want_v = ctx.has_vals and ctx.needs_input_grad[2]
        g = ops.agg_backward(feats, w, A, B, idx, g_pred, g_classes=None if ctx.has_cin else g_cls,
                             g_A=g_A, g_B=g_B[0] if g_B is not None else None,
                             vals=vals if ctx.has_vals else None, nonlinear=ctx.nonlinear, want_g_vals=want_v)
        return (None, None, g.get("vals"), g.get("fc_w"), g.get("fc_b"), g["q0_w"], g["q0_b"],
                g.get("q2_w"), g.get("q2_b"), g["fcc_w"], g["fcc_b"], None)

    @staticmethod
    def _backward_dense(ctx, g_cls, g_pred, g_A, g_B):
        """Only when the gradient of the INPUT rows is requested (no reference script does: bags are
        data, train_tcga.py:60-66): the same analytic gradient composed from dense GPU products."""
        feats, vals, fc_w, q0_w, q0_b, q2_w, q2_b, fcc_w, A, B, idx = ctx.saved_tensors
        x = feats
        V = vals if ctx.has_vals else feats
        idx = idx[0]
        scale = Q_DIM ** -0.5
        C = fcc_w.shape[0]
        zeros = torch.zeros
        g_pred = g_pred if g_pred is not None else zeros((1, C), device=x.device)
        # ---- bag head
        g_fcc_b = g_pred[0]
        g_fcc_w = g_pred[0][:, None, None] * B[0][None]
        gB = torch.einsum("o,ock->ck", g_pred[0], fcc_w)
        if g_B is not None:
            gB = gB + g_B[0]
        gA = V.mm(gB.t())
        if g_A is not None:
            gA = gA + g_A
        gs = A * (gA - (A * gA).sum(0, keepdim=True)) * scale
        # ---- recompute the query stream
        pre1 = F.linear(x, q0_w, q0_b)
        if ctx.nonlinear:
            h1 = pre1.clamp_min(0)
            Q = torch.tanh(F.linear(h1, q2_w, q2_b))
        else:
            Q = pre1
        qmax = Q[idx]
        gQ = gs.mm(qmax)
        gQ.index_add_(0, idx, gs.t().mm(Q))
        if ctx.nonlinear:
            gz2 = gQ * (1 - Q * Q)
            g_q2_w = gz2.t().mm(h1)
            g_q2_b = gz2.sum(0)
            gh1 = gz2.mm(q2_w) * (pre1 > 0)
        else:
            g_q2_w = g_q2_b = None
            gh1 = gQ
        g_q0_w = gh1.t().mm(x)
        g_q0_b = gh1.sum(0)
        # ---- instance stream (FCLayer fused in) and inputs
        g_fc_w = g_fc_b = g_cin = None
        if ctx.has_cin:
            g_cin = None  # c only feeds the (non-differentiable) indices: dsmil.py:52
        elif g_cls is not None:
            g_fc_w = g_cls.t().mm(x)
            g_fc_b = g_cls.sum(0)
        g_x = g_vals = None
        if ctx.needs_input_grad[0]:
            g_x = gh1.mm(q0_w)
            if not ctx.has_cin and g_cls is not None:
                g_x = g_x + g_cls.mm(fc_w)
            if not ctx.has_vals:
                g_x = g_x + A.mm(gB)
        if ctx.has_vals and ctx.needs_input_grad[2]:
            g_vals = A.mm(gB)
        return (g_x, g_cin, g_vals, g_fc_w, g_fc_b, g_q0_w, g_q0_b, g_q2_w, g_q2_b,
                g_fcc_w, g_fcc_b, None)
