#!/usr/bin/env python3
"""Drop-in for the reference's compute_feats.py (same flags, folder conventions and CSV
outputs: compute_feats.py:128-262), with the embedder running in libdsmil_hip.so.

WSI/<dataset>/{single|pyramid}/<class>/<slide>/*.jpeg  ->  datasets/<dataset>/<class>/<slide>.csv
plus the per-class index CSVs and the shuffled dataset CSV (compute_feats.py:249-260).
Launch under torchrun to shard each slide's patches over the GPUs of a node.
"""
import argparse
import copy
import glob
import os

import pandas as pd
import torch
import torch.nn as nn
from sklearn.utils import shuffle

import dsmil as mil
from dsmil_wsi_amd import dist as ddist
from dsmil_wsi_amd import pipeline

try:  # torchvision is optional: the in-tree constructor has its names and state_dict order
    import torchvision.models as models
except Exception:  # pragma: no cover
    from dsmil_wsi_amd import resnet as models


def build_parser():
    p = argparse.ArgumentParser(description="Compute TCGA features from SimCLR embedder")
    p.add_argument("--num_classes", default=2, type=int, help="Number of output classes [2]")
    p.add_argument("--batch_size", default=128, type=int, help="Batch size of dataloader [128]")
    p.add_argument("--num_workers", default=4, type=int, help="Number of threads for datalodaer")
    p.add_argument("--gpu_index", type=int, nargs="+", default=(0,), help="GPU ID(s) [0]")
    p.add_argument("--backbone", default="resnet18", type=str, help="Embedder backbone [resnet18]")
    p.add_argument("--norm_layer", default="instance", type=str, help="Normalization layer [instance]")
    p.add_argument("--magnification", default="single", type=str,
                   help="single | tree | high | low")
    p.add_argument("--weights", default=None, type=str, help="Folder of the pretrained weights, simclr/runs/*")
    p.add_argument("--weights_high", default=None, type=str)
    p.add_argument("--weights_low", default=None, type=str)
    p.add_argument("--tree_fusion", default="cat", type=str, help="[cat|fusion]")
    p.add_argument("--dataset", default="TCGA-lung-single", type=str, help="Dataset folder name")
    p.add_argument("--bg_threshold", default=None, type=float,
                   help="new, default off: drop background tiles before embedding — keep a tile iff mean(FIND_EDGES band sums) / "
                        "tile_size^2 > T, the criterion deepzoom_tiler.py:56-61 applies while tiling (its -t, 15); "
                        "single-magnification bags only")
    p.add_argument("--precision", default="fp32", choices=("fp32", "half", "bf16"),
                   help="(new, default fp32 = the parity path) half: fp16 ACTIVATIONS behind the stem, one fp16 MFMA product per "
                        "MAC, f32 accumulation and norm statistics (ResNet-18 / 34 with InstanceNorm; other trunks: fp32 activations, "
                        "one fp16 plane per conv operand): ~2.6e-3 feature error, 2x; bf16: the same trunk on bf16 activations "
                        "(fp32's range): ~2e-2 (ResNet-34: ~1e-2 / ~5e-2); the reference has no such switch")
    p.add_argument("--gpu_decode", action="store_true",
                   help="(new, default off) decode the tiles' JPEG files on the GPU (dsmil_jpeg_decode: baseline JPEGs, bit-identical to "
                        "Pillow; other files take Pillow inside the same call) instead of in --num_workers DataLoader processes "
                        "(compute_feats.py:28,55): a tenth of the PCIe bytes, no decode-bound loader")
    p.add_argument("--save_npy", action="store_true",
                   help="(new) also write each bag's features as float32 <bag>.npy next to the '%%.4f' CSV: lossless "
                        "and ~6x smaller/faster to load than the text detour of compute_feats.py:80-82")
    return p


def build_backbone(args):
    norm = nn.InstanceNorm2d if args.norm_layer == "instance" else nn.BatchNorm2d
    pretrain = args.norm_layer == "batch" and args.weights == "ImageNet"
    ctor = {"resnet18": (getattr(models, "resnet18", None), 512), "resnet34": (getattr(models, "resnet34", None), 512),
            "resnet50": (getattr(models, "resnet50", None), 2048), "resnet101": (getattr(models, "resnet101", None), 2048)}
    fn, num_feats = ctor[args.backbone]
    if fn is None:
        raise ValueError(f"unknown backbone {args.backbone}")
    resnet = fn(pretrained=pretrain, norm_layer=norm)
    for prm in resnet.parameters():
        prm.requires_grad = False
    resnet.fc = nn.Identity()
    return resnet, num_feats


def load_embedder(i_classifier, run, out_name, args, device):
    if run is not None:
        path = os.path.join("simclr", "runs", run, "checkpoints", "model.pth")
    else:
        path = glob.glob("simclr/runs/*/checkpoints/*.pth")[-1]
    new_sd = pipeline.load_simclr_weights(i_classifier, torch.load(path, map_location=device))
    if ddist.world_rank()[1] == 0:
        os.makedirs(os.path.join("embedder", args.dataset), exist_ok=True)
        torch.save(new_sd, os.path.join("embedder", args.dataset, out_name))
    print("Use pretrained features.")


def main(argv=None):
    args = build_parser().parse_args(argv)
    world = ddist.init_from_env()
    if world == 1:
        os.environ["CUDA_VISIBLE_DEVICES"] = ",".join(str(x) for x in tuple(args.gpu_index))
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    resnet, num_feats = build_backbone(args)
    imagenet = "ImageNet" in (args.weights, args.weights_high, args.weights_low)
    if imagenet and args.norm_layer != "batch":
        raise ValueError("Please use batch normalization for ImageNet feature")
    if args.magnification == "tree" and args.weights_high is not None and args.weights_low is not None:
        ic_h = mil.IClassifier(resnet, num_feats, output_class=args.num_classes).to(device)
        ic_l = mil.IClassifier(copy.deepcopy(resnet), num_feats, output_class=args.num_classes).to(device)
        ic_h.embed_precision = ic_l.embed_precision = args.precision
        if imagenet:
            print("Use ImageNet features.")
        else:
            load_embedder(ic_h, args.weights_high, "embedder-high.pth", args, device)
            load_embedder(ic_l, args.weights_low, "embedder-low.pth", args, device)
    else:
        i_classifier = mil.IClassifier(resnet, num_feats, output_class=args.num_classes).to(device)
        i_classifier.embed_precision = args.precision
        if imagenet:
            print("Use ImageNet features.")
        else:
            load_embedder(i_classifier, args.weights, "embedder.pth", args, device)
    sub = "pyramid" if args.magnification in ("tree", "low", "high") else "single"
    bags_list = sorted(glob.glob(os.path.join("WSI", args.dataset, sub, "*", "*")))
    feats_path = os.path.join("datasets", args.dataset)
    os.makedirs(feats_path, exist_ok=True)
    if args.bg_threshold is not None and args.magnification == "tree":
        raise ValueError("--bg_threshold filters single-magnification bags (a pyramid bag's rows are tied to its tile tree)")
    pipeline.GPU_DECODE[0] = bool(args.gpu_decode)
    try:
        if args.magnification == "tree":
            pipeline.compute_tree_feats(args, bags_list, ic_l, ic_h, feats_path)
        else:
            pipeline.compute_feats(args, bags_list, i_classifier, feats_path, args.magnification)
    finally:
        pipeline.GPU_DECODE[0] = False
    if ddist.world_rank()[1] == 0:  # compute_feats.py:249-260
        all_df = []
        for i, item in enumerate(sorted(glob.glob(os.path.join("datasets", args.dataset, "*" + os.path.sep)))):
            bag_df = pd.DataFrame(glob.glob(os.path.join(item, "*.csv")))
            bag_df["label"] = i
            bag_df.to_csv(os.path.join("datasets", args.dataset, item.split(os.path.sep)[2] + ".csv"), index=False)
            all_df.append(bag_df)
        bags_path = shuffle(pd.concat(all_df, axis=0, ignore_index=True))
        bags_path.to_csv(os.path.join("datasets", args.dataset, args.dataset + ".csv"), index=False)


if __name__ == "__main__":
    main()
