"""Host-side loops around the embedder hot path, mirroring the reference's drivers:
compute_feats.compute_feats (:58-82), compute_tree_feats (:84-126), the SimCLR checkpoint
re-keying (:219-233) and attention_map.test (:59-118).  Differences are deliberate and
result-preserving:
  * features stay on the device between batches (one D2H per bag instead of one per batch,
    compute_feats.py:74) and, for attention maps, between embedder and aggregator
    (attention_map.py:75-84 round-trips through numpy);
  * the high-magnification patches of a tree bag are embedded in batches instead of one forward
    per patch (compute_feats.py:106-109) — InstanceNorm is per image, so rows are unchanged;
  * with torch.distributed initialised, a bag's ordered patch list is sharded contiguously over
    ranks and reassembled by one all-gather (dist.py).
"""
import glob
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import dist as ddist


def to_tensor(img):
    """PIL image -> float32 CHW in [0,1] (what torchvision's VF.to_tensor does for 8-bit images,
    compute_feats.py:35-39; no mean/std normalisation)."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.dtype == np.uint8:
        t = torch.from_numpy(np.array(a, copy=True)).permute(2, 0, 1).float().div_(255.0)
    else:
        t = torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).permute(2, 0, 1)
    return t


def patch_position(path):
    """attention_map.py:27-28 — tiles are named '<row>_<col>.<ext>'."""
    stem = os.path.basename(path).split(".")[0].split("_")
    return np.asarray([int(stem[0]), int(stem[1])])


class PatchFiles(Dataset):
    """uint8=False: what the reference's BagDataset + ToTensor yield (float32 CHW, compute_feats.py:21-39).
    uint8=True: the decoded RGB image as uint8 HWC — ToTensor then runs fused in the native stem
    (dsmil_resnet18in_forward_u8), bit-identically, and the H2D copy is 4x smaller."""

    def __init__(self, files, with_position=False, uint8=False):
        self.files = list(files)
        self.with_position = with_position
        self.uint8 = uint8

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        from PIL import Image
        with Image.open(self.files[i]) as im:
            if self.uint8:
                sample = {"input": torch.from_numpy(np.array(im.convert("RGB"), dtype=np.uint8, copy=True))}
            else:
                sample = {"input": to_tensor(im.convert("RGB") if im.mode not in ("RGB", "L") else im)}
        if self.with_position:
            sample["position"] = patch_position(self.files[i])
        return sample


def patch_loader(files, batch_size, num_workers, with_position=False, uint8=False):
    return DataLoader(PatchFiles(files, with_position, uint8), batch_size=batch_size, shuffle=False,
                      num_workers=num_workers, drop_last=False)


def glob_patches(bag_dir, magnification="single"):
    """compute_feats.py:64-67."""
    if magnification in ("single", "low"):
        pats = [os.path.join(bag_dir, "*.jpg"), os.path.join(bag_dir, "*.jpeg")]
    else:
        pats = [os.path.join(bag_dir, "*" + os.sep + "*.jpg"), os.path.join(bag_dir, "*" + os.sep + "*.jpeg")]
    out = []
    for p in pats:
        out += glob.glob(p)
    return out


@torch.no_grad()
def embed_files(i_classifier, files, batch_size=128, num_workers=4, device=None, want_position=False,
                sharded=True):
    """The hot loop of compute_feats.py:70-76 / attention_map.py:69-79.  Returns
    (feats [N,F], classes [N,C]) on `device` (and positions [N,2] if asked).  Sharded over ranks
    when torch.distributed is initialised."""
    device = device or next(i_classifier.parameters()).device
    world, rank = ddist.world_rank() if sharded else (1, 0)
    n_total = len(files)
    lo, hi = ddist.shard_range(n_total, rank, world)
    feats_l, cls_l, pos_l = [], [], []
    # decoded uint8 images go to the GPU as they are when the native ResNet-18-IN stem will take them
    from .modules import resnet_convs_of
    u8 = torch.device(device).type == "cuda" and resnet_convs_of(i_classifier.feature_extractor) is not None
    if hi > lo:
        for batch in patch_loader(files[lo:hi], batch_size, num_workers, want_position, uint8=u8):
            patches = batch["input"].to(device, non_blocking=True)
            if not u8:
                patches = patches.float()
            feats, classes = i_classifier(patches)
            feats_l.append(feats)
            cls_l.append(classes)
            if want_position:
                pos_l.append(batch["position"])
    if feats_l:
        feats, classes = torch.cat(feats_l), torch.cat(cls_l)
    else:
        F = i_classifier.fc.in_features
        feats = torch.zeros((0, F), device=device)
        classes = torch.zeros((0, i_classifier.fc.out_features), device=device)
    if world > 1:
        feats = ddist.all_gather_rows(feats, n_total)
        classes = ddist.all_gather_rows(classes, n_total)
    if want_position:
        pos = np.vstack([patch_position(f) for f in files]) if n_total else np.zeros((0, 2), int)
        return feats, classes, pos
    return feats, classes


def save_feats_csv(feats, path, npy=False):
    """compute_feats.py:80-82 — pandas CSV, header 0..F-1, '%.4f'; with ``npy`` also the exact float32 rows
    as <bag>.npy (the text format quantises to 1e-4 and dominates I/O time)."""
    import pandas as pd
    os.makedirs(os.path.dirname(path), exist_ok=True)
    arr = feats.detach().cpu().numpy() if torch.is_tensor(feats) else np.asarray(feats)
    pd.DataFrame(arr).to_csv(path, index=False, float_format="%.4f")
    if npy:
        np.save(os.path.splitext(path)[0] + ".npy", np.ascontiguousarray(arr, dtype=np.float32))


def _bag_csv_path(save_path, bag_dir):
    parts = bag_dir.rstrip(os.sep).split(os.sep)
    return os.path.join(save_path, parts[-2], parts[-1] + ".csv")


def compute_feats(args, bags_list, i_classifier, save_path=None, magnification="single"):
    """compute_feats.py:58-82."""
    i_classifier.eval()
    _, rank = ddist.world_rank()
    for i, bag in enumerate(bags_list):
        files = glob_patches(bag, magnification)
        feats, _ = embed_files(i_classifier, files, args.batch_size, args.num_workers)
        if rank == 0:
            sys.stdout.write("\r Computed: {}/{} -- {} patches".format(i + 1, len(bags_list), len(files)))
            if len(files) == 0:
                print("No valid patch extracted from: " + bag)
            else:
                save_feats_csv(feats, _bag_csv_path(save_path, bag), npy=getattr(args, "save_npy", False))


@torch.no_grad()
def compute_tree_feats(args, bags_list, embedder_low, embedder_high, save_path=None):
    """compute_feats.py:84-126 — every high-magnification patch is paired with its low-mag parent:
    'cat' -> [high(512) || low(512)], 'fusion' -> high + 0.25*low."""
    embedder_low.eval()
    embedder_high.eval()
    if args.tree_fusion not in ("fusion", "cat"):
        raise NotImplementedError(f"{args.tree_fusion} is not an excepted option for --tree_fusion. "
                                  "This argument accepts 2 options: 'fusion' and 'cat'.")
    _, rank = ddist.world_rank()
    for i, bag in enumerate(bags_list):
        low_files = glob_patches(bag, "low")
        low_feats, _ = embed_files(embedder_low, low_files, args.batch_size, args.num_workers)
        high_files, parent = [], []
        for idx, lp in enumerate(low_files):
            folder = os.path.join(os.path.dirname(lp), os.path.splitext(os.path.basename(lp))[0])
            hp = glob.glob(folder + os.sep + "*.jpg") + glob.glob(folder + os.sep + "*.jpeg")
            high_files += hp
            parent += [idx] * len(hp)
        if rank == 0:
            sys.stdout.write("\r Computed: {}/{} -- {} low / {} high".format(i + 1, len(bags_list), len(low_files), len(high_files)))
        if not high_files:
            if rank == 0:
                print("No valid patch extracted from: " + bag)
            continue
        high_feats, _ = embed_files(embedder_high, high_files, args.batch_size, args.num_workers)
        low_of_high = low_feats.index_select(0, torch.as_tensor(parent, device=low_feats.device))
        tree = high_feats + 0.25 * low_of_high if args.tree_fusion == "fusion" else torch.cat([high_feats, low_of_high], dim=-1)
        if rank == 0:
            save_feats_csv(tree, _bag_csv_path(save_path, bag), npy=getattr(args, "save_npy", False))
    if rank == 0:
        print("\n")


def load_simclr_weights(i_classifier, state_dict_weights):
    """compute_feats.py:223-231: drop the 4 projection-head tensors (l1/l2 of ResNetSimCLR,
    simclr/models/resnet_simclr.py:19-20), then map the remaining tensors BY POSITION onto the
    IClassifier's own keys and load non-strictly.  Returns the re-keyed dict (saved as
    embedder.pth by the caller, compute_feats.py:232-233)."""
    state_dict_weights = OrderedDict(state_dict_weights)
    for _ in range(4):
        state_dict_weights.popitem()
    new_state_dict = OrderedDict()
    for (_k, v), (k0, _v0) in zip(state_dict_weights.items(), i_classifier.state_dict().items()):
        new_state_dict[k0] = v
    i_classifier.load_state_dict(new_state_dict, strict=False)
    return new_state_dict


# ---------------------------------------------------------------------------------------------
# attention maps (attention_map.py:59-118)
# ---------------------------------------------------------------------------------------------
def rescale_intensity01(img):
    """skimage.exposure.rescale_intensity(img, out_range=(0, 1)) for float input."""
    lo, hi = float(img.min()), float(img.max())
    if hi <= lo:
        return np.zeros_like(img, dtype=np.float64)
    return (img - lo) / (hi - lo)


def attention_colormap(A, pos_arr, bag_prediction, thres, colors, class_names=None, bag_name="", log=print):
    """attention_map.py:86-113.  A [N,C] numpy, pos_arr [N,2] (row,col), bag_prediction [C]
    (sigmoid).  Returns the uint8 colour map (32x nearest-neighbour upsampled)."""
    C = A.shape[1]
    class_names = class_names or ["class {}".format(c) for c in range(C)]
    benign, num_pos = True, 0
    colored = None
    for c in range(C):
        if bag_prediction[c] >= thres[c]:
            att = A[:, c]
            num_pos += 1
            layer = att[:, None] * np.asarray(colors[c], dtype=np.float64)[None, :]
            if benign:
                log(bag_name + " is detected as: " + class_names[c])
                colored = layer
            else:
                log("and " + class_names[c])
                colored = colored + layer
            benign = False
    if benign:
        log(bag_name + " is detected as: benign")
        colored = np.zeros((A.shape[0], 3))
    else:
        colored = colored / num_pos
    colored = rescale_intensity01(colored)
    H, W = int(pos_arr[:, 0].max()) + 1, int(pos_arr[:, 1].max()) + 1
    cmap = np.zeros((H, W, 3))
    cmap[pos_arr[:, 0], pos_arr[:, 1]] = colored
    cmap = np.repeat(np.repeat(cmap, 32, axis=0), 32, axis=1)   # transform.resize(order=0)
    return np.clip(np.rint(cmap * 255.0), 0, 255).astype(np.uint8)


@torch.no_grad()
def attention_maps(args, bags_list, milnet, colors=None, rng=None):
    """attention_map.test (:59-118): embed -> aggregate (features never leave the device) ->
    threshold -> colour map PNG (+ optional attention CSV)."""
    from PIL import Image
    milnet.eval()
    rng = rng or np.random
    colors = colors or [rng.choice(range(256), size=3) for _ in range(args.num_classes)]
    _, rank = ddist.world_rank()
    out = []
    for bag in bags_list:
        files = glob.glob(os.path.join(bag, "*." + args.patch_ext))
        if not files:
            continue
        feats, classes, pos_arr = embed_files(milnet.i_classifier, files, args.batch_size, args.num_workers,
                                              want_position=True)
        bag_prediction, A, _ = milnet.b_classifier(feats, classes)
        pred = np.atleast_1d(torch.sigmoid(bag_prediction).squeeze().cpu().numpy())
        cmap = attention_colormap(A.cpu().numpy(), pos_arr, pred, args.thres, colors, args.class_name, bag)
        slide = bag.rstrip(os.sep).split(os.sep)[-1]
        if rank == 0:
            Image.fromarray(cmap).save(os.path.join(args.map_path, slide + ".png"))
            if getattr(args, "export_scores", 0):
                import pandas as pd
                df = pd.DataFrame(A.cpu().numpy())
                df["pos"] = [str(s) for s in pos_arr]
                df.to_csv(os.path.join(args.score_path, slide + ".csv"), index=False)
        out.append((slide, pred, cmap))
    return out
