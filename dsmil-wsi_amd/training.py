"""Aggregator training / evaluation drivers with the semantics of the reference's train_tcga.py
(one bag per optimiser step, loss = 0.5*BCE(bag logits) + 0.5*BCE(max instance logits),
Adam(betas=(0.5,0.9)) + cosine annealing, ROC-optimal thresholds, three evaluation schemes) and
train_mil.py (classical MIL benchmarks, 10-fold CV).  The per-bag forward/backward goes through
dsmil.MILNet, i.e. the HIP aggregator when the bag lives on the GPU.

Differences that do not change results: bags are cached in HBM after their first load instead of
being re-read from disk every iteration (train_tcga.py:62; 288 GB of HBM holds whole feature
sets), and ``dropout_patches`` with rate 0 skips the full-bag row permutation
(train_tcga.py:65,78-83: the aggregator is permutation-invariant).
"""
import copy
import datetime
import glob
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn


# ---- data plumbing (train_tcga.py:19-51) ---------------------------------------------------
def get_bag_feats(csv_file_df, args):
    import pandas as pd
    from sklearn.utils import shuffle
    if args.dataset == "TCGA-lung-default":
        feats_csv_path = "datasets/tcga-dataset/tcga_lung_data_feats/" + csv_file_df.iloc[0].split("/")[1] + ".csv"
    else:
        feats_csv_path = csv_file_df.iloc[0]
    # (train_tcga.py:32; the file through dsmil_csv_parse_f32 — the float32 values torch.tensor(..., float32) makes of pandas'
    # float64 ones, ~10x faster than pd.read_csv — and sklearn's shuffle of the ROWS draws the same permutation for an array as
    # for the DataFrame; a file that is not a plain numeric table is left to pandas)
    from .pipeline import read_feats_csv
    arr = read_feats_csv(feats_csv_path)
    feats = shuffle(arr) if arr is not None else shuffle(pd.read_csv(feats_csv_path)).reset_index(drop=True).to_numpy()
    label = np.zeros(args.num_classes)
    if args.num_classes == 1:
        label[0] = csv_file_df.iloc[1]
    elif int(csv_file_df.iloc[1]) <= (len(label) - 1):
        label[int(csv_file_df.iloc[1])] = 1
    return label, feats, feats_csv_path


def generate_pt_files(args, df, temp_train_dir="temp_train"):
    """CSV -> one [N, feats_size + C] tensor per bag (features || repeated label)."""
    import shutil
    if os.path.exists(temp_train_dir):
        shutil.rmtree(temp_train_dir, ignore_errors=True)
    os.makedirs(temp_train_dir, exist_ok=True)
    print("Creating intermediate training files.")
    for i in range(len(df)):
        label, feats, path = get_bag_feats(df.iloc[i], args)
        bag_feats = torch.tensor(np.array(feats), dtype=torch.float32)
        bag_label = torch.tensor(np.array([label]), dtype=torch.float32).repeat(bag_feats.size(0), 1)
        torch.save(torch.cat((bag_feats, bag_label), dim=1),
                   os.path.join(temp_train_dir, os.path.splitext(path)[0].split(os.sep)[-1] + ".pt"))


class BagCache:
    """path -> (feats [N,K] contiguous, label [1,C]) resident on the training device.  The reference re-reads the
    stacked [N, K+C] tensor from disk every iteration and slices it (train_tcga.py:62-64: a strided view that the
    following gather densifies); here the split happens once per bag, when it is first loaded."""

    def __init__(self, device, feats_size=None):
        self.device = device
        self.feats_size = feats_size
        self.store = {}

    def _split(self, stacked, feats_size):
        return (stacked[:, :feats_size].contiguous().float(), stacked[0, feats_size:].unsqueeze(0).float())

    def get(self, item, feats_size=None):
        feats_size = feats_size or self.feats_size
        if torch.is_tensor(item):
            return self._split(item.to(self.device), feats_size)
        t = self.store.get(item)
        if t is None:
            t = self._split(torch.load(item, map_location=self.device), feats_size)
            self.store[item] = t
        return t


def dropout_rows(n, p, device):
    """train_tcga.py:78-83 as an index list: the int(n*p) randomly chosen rows of a bag (p = 1 - dropout rate) in the
    reference's random order, or None when every row is kept (the aggregator is permutation-invariant)."""
    keep = int(n * p)
    if keep >= n:
        return None
    return torch.randperm(n, device=device)[:keep]


def dropout_patches(feats, p):
    """train_tcga.py:78-83 — keep int(N*p) randomly chosen rows (p = 1 - dropout rate), as a gathered copy."""
    idx = dropout_rows(feats.size(0), p, feats.device)
    return feats if idx is None else feats.index_select(0, idx)


def _is_plain_bce(criterion):
    return (isinstance(criterion, nn.BCEWithLogitsLoss) and criterion.reduction == "mean"
            and criterion.weight is None and criterion.pos_weight is None)


def bag_loss(milnet, criterion, bag_feats, bag_label, row_map=None):
    """train_tcga.py:64-71.  ``row_map``: dropout_patches as an index list (rows of bag_feats that enter the bag).
    With the stock criterion and a MILNet(FCLayer, BClassifier) the whole objective is one native forward + loss
    head (MILNet.bag_loss); otherwise the same expression from torch ops."""
    if _is_plain_bce(criterion) and hasattr(milnet, "bag_loss"):
        return milnet.bag_loss(bag_feats, bag_label, row_map)
    if row_map is not None:
        bag_feats = bag_feats.index_select(0, row_map)
    ins_prediction, bag_prediction, _, _ = milnet(bag_feats)
    max_prediction, _ = torch.max(ins_prediction, 0)
    loss = 0.5 * criterion(bag_prediction.view(1, -1), bag_label.view(1, -1)) + \
        0.5 * criterion(max_prediction.view(1, -1), bag_label.view(1, -1))
    return loss, bag_prediction, max_prediction


class FusedTrainStep:
    """train_tcga.py:60-75 as ONE native call per bag (dsmil_agg_train_step): forward, the two-BCE objective, backward and
    the Adam update of all parameter tensors are enqueued by a single C call — the Python side of a step is that call plus
    the loss read-back of the progress line.  Numerically it is the step the generic path takes (same kernels, same Adam
    arithmetic as torch.optim.Adam's; tests/test_agg_bwd_gpu.py compares the two trajectories).

    Eligible (``FusedTrainStep.create`` returns None otherwise, and ``train`` keeps the generic autograd path):
    MILNet(FCLayer, BClassifier) with v = Identity on a GPU, every parameter trainable fp32, the stock
    BCEWithLogitsLoss, and a plain torch.optim.Adam (one parameter group holding exactly the model's parameters; amsgrad,
    maximize, capturable, differentiable off).  The optimiser's own state tensors (exp_avg, exp_avg_sq) are updated in
    place, ``step`` is written back by ``sync()`` — so optimizer.state_dict(), LR schedulers and a later generic step see a
    consistent optimiser.  ``.grad`` is not populated (as after zero_grad(set_to_none=True))."""

    def __init__(self, milnet, optimizer, params):
        self.net, self.opt, self.params = milnet, optimizer, params
        self.group = optimizer.param_groups[0]
        self.m, self.v, steps = [], [], []
        for p in params:
            if p is None:
                self.m.append(None); self.v.append(None)
                continue
            st = optimizer.state[p]
            if len(st) == 0:   # what torch.optim.Adam._init_group creates on a parameter's first step
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            self.m.append(st["exp_avg"]); self.v.append(st["exp_avg_sq"])
            steps.append(int(float(st["step"])))
        if len(set(steps)) != 1:
            raise RuntimeError("Adam state with different step counts per parameter")
        self.step = steps[0]
        self.nonlinear = milnet.b_classifier.nonlinear
        self._live = [p for p in params if p is not None] + [t for t in self.m + self.v if t is not None]

    @staticmethod
    def create(milnet, criterion, optimizer):
        from .modules import BClassifier, FCLayer
        try:
            ic, bc = milnet.i_classifier, milnet.b_classifier
            if not (isinstance(ic, FCLayer) and isinstance(bc, BClassifier)) or bc.passing_v or not _is_plain_bce(criterion):
                return None
            if type(optimizer) is not torch.optim.Adam or len(optimizer.param_groups) != 1:
                return None
            g = optimizer.param_groups[0]
            if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
                return None
            w = bc._weights()
            lin = ic.fc[0]
            params = [lin.weight, lin.bias, w["q0_w"], w["q0_b"], w["q2_w"], w["q2_b"], w["fcc_w"], w["fcc_b"]]
            live = [p for p in params if p is not None]
            if {id(p) for p in live} != {id(p) for p in g["params"]} or len(live) != len(list(milnet.parameters())):
                return None
            if not all(p.is_cuda and p.dtype == torch.float32 and p.requires_grad and p.is_contiguous() for p in live):
                return None
            if lin.out_features > 64:
                return None
            return FusedTrainStep(milnet, optimizer, params)
        except (AttributeError, KeyError):
            return None

    def accepts(self, bag_feats):
        return bag_feats.is_cuda and bag_feats.dtype == torch.float32 and bag_feats.dim() == 2 and bag_feats.is_contiguous()

    def __call__(self, bag_feats, bag_label, row_map=None):
        """One optimiser step on one bag; returns the loss (0-dim device tensor, detached)."""
        from . import ops
        g = self.group
        with torch.no_grad():
            loss = ops.agg_train_step(bag_feats, bag_label, [p.data if p is not None else None for p in self.params], self.m,
                                      self.v, self.step + 1, g["lr"], g["betas"], g["eps"], g["weight_decay"],
                                      nonlinear=self.nonlinear, row_map=row_map)
        self.step += 1   # only once the native step was enqueued: a call that raised applied no update (sync() stays consistent)
        # the kernels wrote the parameters through raw pointers: tell torch (version counters key the packed-weight caches
        # of the inference path, and autograd's saved-tensor checks)
        torch.autograd.graph.increment_version(self._live)
        return loss.reshape(())

    def resync(self):
        """Re-read the step count from the optimiser's state (after a generic optimizer.step() between fused steps)."""
        steps = {int(float(self.opt.state[p]["step"])) for p in self.params if p is not None}
        if len(steps) != 1:
            raise RuntimeError("Adam state with different step counts per parameter")
        self.step = steps.pop()

    def sync(self):
        """Write the step count back into the optimiser's state (torch keeps it as a CPU float tensor per parameter)."""
        for p in self.params:
            if p is not None:
                self.opt.state[p]["step"] = torch.tensor(float(self.step), dtype=torch.float32)


class LossReadback:
    """The per-step `loss.item()` of train_tcga.py:75 without the per-step stall: every step's loss is still copied to the host
    and reported, but through a pinned two-slot ring — `push(loss)` enqueues this step's device-to-host copy and hands back the
    PREVIOUS step's value (its copy finished while this step was being enqueued), `flush()` the last one.  The host stays at
    most one step ahead of the GPU, every value arrives, in order; only the moment of the read moves by one step."""

    def __init__(self, device):
        self.cuda = torch.device(device).type == "cuda"
        self.buf = torch.zeros(2, dtype=torch.float32, pin_memory=True) if self.cuda else None
        self.ev = [torch.cuda.Event(), torch.cuda.Event()] if self.cuda else None
        self.n = 0
        self.pending = None

    def push(self, loss):
        """loss: 0-dim tensor of this step.  Returns the previous step's loss as a float (None on the first call)."""
        if not self.cuda:
            prev, self.pending = self.pending, float(loss.detach())
            return prev
        k = self.n & 1
        self.buf[k:k + 1].copy_(loss.detach().reshape(1), non_blocking=True)
        self.ev[k].record()
        prev = None
        if self.n > 0:
            self.ev[k ^ 1].synchronize()
            prev = float(self.buf[k ^ 1])
        self.n += 1
        return prev

    def flush(self):
        if not self.cuda:
            prev, self.pending = self.pending, None
            return prev
        if self.n == 0:
            return None
        k = (self.n - 1) & 1
        self.ev[k].synchronize()
        return float(self.buf[k])


def train(args, train_df, milnet, criterion, optimizer, cache=None, log=True):
    """train_tcga.py:55-76: one optimiser step per bag, bags in random order."""
    from sklearn.utils import shuffle
    milnet.train()
    device = next(milnet.parameters()).device
    cache = cache or BagCache(device)
    total_loss = 0.0
    dirs = shuffle(list(train_df))
    losses = []
    # the whole step as one native call when the model / criterion / optimiser are the reference's (FusedTrainStep)
    fused = FusedTrainStep.create(milnet, criterion, optimizer) if getattr(args, "fused_step", True) else None
    readback = LossReadback(device)
    try:
        for i, item in enumerate(dirs):
            bag_feats, bag_label = cache.get(item, args.feats_size)
            rows = dropout_rows(bag_feats.size(0), 1 - args.dropout_patch, bag_feats.device)
            if fused is not None and fused.accepts(bag_feats):
                loss = fused(bag_feats, bag_label, rows)
            else:
                if fused is not None:
                    fused.sync()          # the optimiser's own step count = every update so far, fused ones included
                optimizer.zero_grad()
                loss, _, _ = bag_loss(milnet, criterion, bag_feats, bag_label, rows)
                loss.backward()
                optimizer.step()
                if fused is not None:
                    fused.resync()        # ... and the fused counter follows the generic update
            losses.append(loss.detach())
            if log:   # the progress line of train_tcga.py:74-75, every bag's loss, written one step late (LossReadback): no stall
                prev = readback.push(loss)
                if prev is not None:
                    sys.stdout.write("\r Training bag [%d/%d] bag loss: %.4f" % (i - 1, len(dirs), prev))
    finally:
        # also when a step raised: the loss of the last COMPLETED step is still reported and the optimiser's step count is
        # written back (a checkpoint saved by the caller's handler is then consistent)
        if log and dirs:
            last = readback.flush()
            if last is not None:
                sys.stdout.write("\r Training bag [%d/%d] bag loss: %.4f" % (len(losses) - 1, len(dirs), last))
        if fused is not None:
            fused.sync()
    if losses:
        total_loss = float(torch.stack(losses).sum().item())
    return total_loss / max(1, len(dirs))


def optimal_thresh(fpr, tpr, thresholds, p=0):
    loss = (fpr - tpr) - p * tpr / (fpr + tpr + 1)
    idx = np.argmin(loss, axis=0)
    return fpr[idx], tpr[idx], thresholds[idx]


def multi_label_roc(labels, predictions, num_classes, pos_label=1, log=True):
    from sklearn.metrics import roc_auc_score, roc_curve
    aucs, thresholds, thresholds_optimal = [], [], []
    if predictions.ndim == 1:
        predictions = predictions[:, None]
    if labels.ndim == 1:
        labels = labels[:, None]
    for c in range(num_classes):
        fpr, tpr, threshold = roc_curve(labels[:, c], predictions[:, c], pos_label=1)
        _, _, th = optimal_thresh(fpr, tpr, threshold)
        try:
            c_auc = roc_auc_score(labels[:, c], predictions[:, c])
        except ValueError as e:
            if "Only one class present" not in str(e):
                raise
            c_auc = 1
        if log:
            print("ROC AUC score:", c_auc)
        aucs.append(c_auc)
        thresholds.append(threshold)
        thresholds_optimal.append(th)
    return aucs, thresholds, thresholds_optimal


@torch.no_grad()
def test(args, test_df, milnet, criterion, thresholds=None, return_predictions=False, cache=None, log=True):
    """train_tcga.py:85-132."""
    milnet.eval()
    device = next(milnet.parameters()).device
    cache = cache or BagCache(device)
    # The reference reads the loss (twice), the label and the prediction on the host after every bag: four stalls per forward.
    # Here everything stays on the device until the loop ends (one transfer each), the progress line reads each loss one step
    # late (LossReadback): same numbers, same line for every bag.
    total_loss, losses, labels, preds = 0.0, [], [], []
    readback = LossReadback(device)
    for i, item in enumerate(test_df):
        bag_feats, bag_label = cache.get(item, args.feats_size)
        rows = dropout_rows(bag_feats.size(0), 1 - args.dropout_patch, bag_feats.device)   # train_tcga.py:96
        loss, bag_prediction, max_prediction = bag_loss(milnet, criterion, bag_feats, bag_label, rows)
        losses.append(loss.detach().reshape(()))
        if log:
            prev = readback.push(loss)
            if prev is not None:
                sys.stdout.write("\r Testing bag [%d/%d] bag loss: %.4f" % (i - 1, len(test_df), prev))
        labels.append(bag_label.reshape(-1))
        if args.average:
            preds.append((torch.sigmoid(max_prediction) + torch.sigmoid(bag_prediction)).reshape(-1))
        else:
            preds.append(torch.sigmoid(bag_prediction).reshape(-1))
    if log and len(losses):
        sys.stdout.write("\r Testing bag [%d/%d] bag loss: %.4f" % (len(losses) - 1, len(test_df), readback.flush()))
    if losses:
        total_loss = float(torch.stack(losses).double().sum().item())
    sq = (lambda a: a.squeeze(-1)) if args.num_classes == 1 else (lambda a: a)   # (the reference's per-bag .squeeze())
    test_labels = sq(torch.stack(labels).cpu().numpy().astype(int)) if labels else np.array([])
    test_predictions = sq(torch.stack(preds).cpu().numpy()) if preds else np.array([])
    auc_value, _, thresholds_optimal = multi_label_roc(test_labels, test_predictions, args.num_classes, log=log)
    if thresholds:
        thresholds_optimal = thresholds
    if args.num_classes == 1:
        test_predictions = (test_predictions >= thresholds_optimal[0]).astype(test_predictions.dtype)
        test_labels = np.squeeze(test_labels)
    else:
        for c in range(args.num_classes):
            test_predictions[:, c] = (test_predictions[:, c] >= thresholds_optimal[c]).astype(test_predictions.dtype)
    bag_score = sum(np.array_equal(test_labels[i], test_predictions[i]) for i in range(len(test_df)))
    avg_score = bag_score / max(1, len(test_df))
    if return_predictions:
        return total_loss / len(test_df), avg_score, auc_value, thresholds_optimal, test_predictions, test_labels
    return total_loss / max(1, len(test_df)), avg_score, auc_value, thresholds_optimal


# ---- model / optimiser construction (train_tcga.py:229-243) -------------------------------
def apply_sparse_init(m):
    if isinstance(m, (nn.Linear, nn.Conv2d, nn.Conv1d)):
        nn.init.orthogonal_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


def init_model(args, mil, device):
    i_classifier = mil.FCLayer(in_size=args.feats_size, out_size=args.num_classes)
    b_classifier = mil.BClassifier(input_size=args.feats_size, output_class=args.num_classes,
                                   dropout_v=args.dropout_node, nonlinear=args.non_linearity)
    milnet = mil.MILNet(i_classifier, b_classifier).to(device)
    milnet.apply(apply_sparse_init)
    criterion = nn.BCEWithLogitsLoss()
    optimizer = torch.optim.Adam(milnet.parameters(), lr=args.lr, betas=(0.5, 0.9), weight_decay=args.weight_decay)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, args.num_epochs, 0.000005)
    return milnet, criterion, optimizer, scheduler


def save_model(args, fold, run, save_path, model, thresholds_optimal):
    save_name = os.path.join(save_path, f"fold_{fold}_{run + 1}.pth")
    torch.save(model.state_dict(), save_name)
    print("Best model saved at: " + save_name)
    print("Best thresholds ===>>> " + "|".join("class-{}>>{}".format(*k) for k in enumerate(thresholds_optimal)))
    with open(os.path.join(save_path, f"fold_{fold}_{run + 1}.json"), "w") as f:
        json.dump([float(x) for x in thresholds_optimal], f)


def fit(args, mil, device, train_path, val_path, tag, run, save_path, cache, keep_best=False):
    """The epoch loop shared by the three schemes (train_tcga.py:272-287 and its two copies)."""
    milnet, criterion, optimizer, scheduler = init_model(args, mil, device)
    best_score, best_ac, best_auc, counter, best = 0, 0, [0.0] * args.num_classes, 0, None
    for epoch in range(1, args.num_epochs + 1):
        counter += 1
        train_loss = train(args, train_path, milnet, criterion, optimizer, cache)
        test_loss, avg_score, aucs, th = test(args, val_path, milnet, criterion, cache=cache)
        print("\r Epoch [%d/%d] train loss: %.4f test loss: %.4f, average score: %.4f, AUC: " %
              (epoch, args.num_epochs, train_loss, test_loss, avg_score) +
              "|".join("class-{}>>{}".format(*k) for k in enumerate(aucs)))
        scheduler.step()
        current = (sum(aucs) + avg_score) / 2
        if current > best_score:
            counter, best_score, best_ac, best_auc = 0, current, avg_score, aucs
            save_model(args, tag, run, save_path, milnet, th)
            if keep_best:
                best = (copy.deepcopy(milnet), th)
        if counter > args.stop_epochs:
            break
    return best_ac, best_auc, best, criterion


def run_eval_scheme(args, mil, device, bags_path=None):
    """train_tcga.py:252-429."""
    from sklearn.model_selection import KFold
    from sklearn.utils import shuffle
    bags_path = bags_path if bags_path is not None else glob.glob("temp_train/*.pt")
    save_path = os.path.join("weights", datetime.date.today().strftime("%Y%m%d"))
    os.makedirs(save_path, exist_ok=True)
    run = len(glob.glob(os.path.join(save_path, "*.pth")))
    cache = BagCache(device)
    fold_results = []
    if args.eval_scheme == "5-fold-cv":
        kf = KFold(n_splits=5, shuffle=True, random_state=42)
        for fold, (tr, te) in enumerate(kf.split(bags_path)):
            print(f"Starting CV fold {fold}.")
            ac, auc, _, _ = fit(args, mil, device, [bags_path[i] for i in tr], [bags_path[i] for i in te],
                                fold, run, save_path, cache)
            fold_results.append((ac, auc))
    elif args.eval_scheme == "5-time-train+valid+test":
        for it in range(5):
            print(f"Starting iteration {it + 1}.")
            bags_path = shuffle(bags_path)
            n = len(bags_path)
            train_end = int(n * (1 - args.split - 0.1))
            val_end = train_end + int(n * 0.1)
            ac, auc, best, criterion = fit(args, mil, device, bags_path[:train_end], bags_path[train_end:val_end],
                                           it, run, save_path, cache, keep_best=True)
            if best is not None:  # the reference calls test() with its arguments swapped here (:341)
                test(args, bags_path[val_end:], best[0], criterion, cache=cache)
            fold_results.append((ac, auc))
    elif args.eval_scheme == "5-fold-cv-standalone-test":
        from scipy.stats import mode
        from sklearn.metrics import accuracy_score, balanced_accuracy_score, hamming_loss
        bags_path = shuffle(bags_path)
        n_res = int(args.split * len(bags_path))
        reserved, bags_path = bags_path[:n_res], bags_path[n_res:]
        kf = KFold(n_splits=5, shuffle=True, random_state=42)
        fold_models = []
        for fold, (tr, te) in enumerate(kf.split(bags_path)):
            print(f"Starting CV fold {fold}.")
            ac, auc, best, criterion = fit(args, mil, device, [bags_path[i] for i in tr],
                                           [bags_path[i] for i in te], fold, run, save_path, cache, keep_best=True)
            fold_results.append((ac, auc))
            fold_models.append(best)
        fold_predictions = []
        for model, th in fold_models:
            _, _, _, _, pred, test_labels = test(args, reserved, model.to(device), criterion, thresholds=th,
                                                 return_predictions=True, cache=cache)
            fold_predictions.append(pred)
        combined = np.squeeze(np.asarray(mode(np.stack(fold_predictions, axis=0), axis=0, keepdims=True).mode[0]))
        if args.num_classes > 1:
            print("Hamming Loss:", hamming_loss(test_labels, combined))
            print("Subset Accuracy (Exact Match Ratio):", accuracy_score(test_labels, combined))
        else:
            print("Accuracy:", accuracy_score(test_labels, combined))
            print("Balanced Accuracy:", balanced_accuracy_score(test_labels, combined))
        os.makedirs("test", exist_ok=True)
        with open("test/test_list.json", "w") as f:
            json.dump(list(reserved), f)
        for i, (model, th) in enumerate(fold_models):
            torch.save(model.state_dict(), f"test/mil_weights_fold_{i}.pth")
            with open(f"test/mil_threshold_fold_{i}.json", "w") as f:
                json.dump([float(x) for x in th], f)
    else:
        raise ValueError(f"unknown --eval_scheme {args.eval_scheme}")
    if fold_results:
        print(f"Final results: Mean Accuracy: {np.mean([r[0] for r in fold_results])}")
        for i, m in enumerate(np.mean(np.array([r[1] for r in fold_results]), axis=0)):
            print(f"Class {i}: Mean AUC = {m:.4f}")
    return fold_results


# ---- classical MIL benchmarks (train_mil.py) -------------------------------------------------
def parse_mil_file(path):
    """train_mil.py:17-35 — lines 'inst:bag:label idx:val idx:val ...' (svmlight-like).
    Returns (features [n_inst, n_feat] float, bag_id [n_inst], label [n_inst])."""
    rows, bag_ids, labels = [], [], []
    n_feat = 0
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            head, *pairs = line.split()
            _inst, bag, lab = head.split(":")
            kv = [(int(p.split(":")[0]), float(p.split(":")[1])) for p in pairs]
            n_feat = max(n_feat, max((k for k, _ in kv), default=0) + 1)
            rows.append(kv)
            bag_ids.append(int(bag))
            labels.append(1.0 if float(lab) > 0 else 0.0)
    X = np.zeros((len(rows), n_feat), np.float32)
    for i, kv in enumerate(rows):
        for k, v in kv:
            X[i, k] = v
    return X, np.asarray(bag_ids), np.asarray(labels, np.float32)


def group_bags(X, bag_ids, labels):
    """train_mil.py:143-149 — one [n_i, F] array + one bag label per bag id."""
    bags, ys = [], []
    for b in np.unique(bag_ids):
        m = bag_ids == b
        bags.append(X[m])
        ys.append(float(labels[m].max()))
    return bags, np.asarray(ys, np.float32)


def mil_epoch_train(bags, ys, idx, milnet, criterion, optimizer, device):
    """train_mil.py:42-59 (instances of a bag are shuffled, :46)."""
    milnet.train()
    total = 0.0
    for i in idx:
        optimizer.zero_grad()
        x = torch.from_numpy(bags[i][np.random.permutation(len(bags[i]))]).to(device)
        y = torch.tensor([[ys[i]]], device=device)
        classes, bag_prediction, _, _ = milnet(x)
        max_prediction, _ = torch.max(classes, 0)
        loss = 0.5 * criterion(bag_prediction.view(1, -1), y.view(1, -1)) + \
            0.5 * criterion(max_prediction.view(1, -1), y.view(1, -1))
        loss.backward()
        optimizer.step()
        total += loss.item()
    return total / max(1, len(idx))


@torch.no_grad()
def mil_epoch_test(bags, ys, idx, milnet, criterion, device):
    """train_mil.py:61-80."""
    milnet.eval()
    total, preds = 0.0, []
    for i in idx:
        x = torch.from_numpy(bags[i]).to(device)
        y = torch.tensor([[ys[i]]], device=device)
        classes, bag_prediction, _, _ = milnet(x)
        max_prediction, _ = torch.max(classes, 0)
        loss = 0.5 * criterion(bag_prediction.view(1, -1), y.view(1, -1)) + \
            0.5 * criterion(max_prediction.view(1, -1), y.view(1, -1))
        total += loss.item()
        preds.append(float(torch.sigmoid(bag_prediction).squeeze()))  # train_mil.py:77
    return total / max(1, len(idx)), np.asarray(preds)


def five_scores(bag_labels, bag_predictions):
    """train_mil.py:87-97."""
    from sklearn.metrics import precision_recall_fscore_support, roc_auc_score, roc_curve
    fpr, tpr, threshold = roc_curve(bag_labels, bag_predictions, pos_label=1)
    _, _, th = optimal_thresh(fpr, tpr, threshold)
    auc_value = roc_auc_score(bag_labels, bag_predictions)
    hard = (np.asarray(bag_predictions) >= th).astype(int)
    precision, recall, fscore, _ = precision_recall_fscore_support(bag_labels, hard, average="binary", zero_division=0)
    accuracy = 1 - np.count_nonzero(np.asarray(bag_labels).astype(int) - hard) / len(bag_labels)
    return accuracy, auc_value, precision, recall, fscore


def write_synthetic_mil_file(path, n_bags=92, n_inst=476, n_feat=166, n_pos=47, seed=0):
    """A stand-in for musk1norm.svm (a download, download.py:33-37) in the same text format:
    92 bags / 476 instances / 166 features / 47 positive bags.  Positive
    bags carry one 'witness' instance shifted along a fixed direction so the task is learnable."""
    rng = np.random.default_rng(seed)
    sizes = np.full(n_bags, n_inst // n_bags)
    sizes[: n_inst - sizes.sum()] += 1
    pos = np.zeros(n_bags, bool)
    pos[rng.permutation(n_bags)[:n_pos]] = True
    direction = rng.standard_normal(n_feat).astype(np.float32)
    direction /= np.linalg.norm(direction)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    inst = 0
    with open(path, "w") as f:
        for b in range(n_bags):
            X = rng.standard_normal((sizes[b], n_feat)).astype(np.float32)
            if pos[b]:
                X[0] += 6.0 * direction
            for r in X:
                f.write(f"{inst}:{b}:{1 if pos[b] else -1} " + " ".join(f"{k}:{v:.6f}" for k, v in enumerate(r)) + "\n")
                inst += 1
    return path
