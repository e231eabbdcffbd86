#!/usr/bin/env python3
"""Drop-in for the reference's attention_map.py (flags: attention_map.py:121-137): embed a
slide's tiles, aggregate, threshold, paint the attention colour map."""
import argparse
import glob
import os
import warnings

import torch
import torch.nn as nn

import dsmil as mil
from dsmil_wsi_amd import pipeline

try:
    import torchvision.models as models
except Exception:  # pragma: no cover
    from dsmil_wsi_amd import resnet as models


def build_parser():
    p = argparse.ArgumentParser(description="Testing workflow includes attention computing and color map production")
    p.add_argument("--num_classes", type=int, default=2)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--feats_size", type=int, default=512)
    p.add_argument("--thres", nargs="+", type=float, default=[0.7371, 0.2752])
    p.add_argument("--class_name", nargs="+", type=str, default=None)
    p.add_argument("--embedder_weights", type=str, default="test/weights/embedder.pth")
    p.add_argument("--aggregator_weights", type=str, default="test/weights/aggregator.pth")
    p.add_argument("--bag_path", type=str, default="test/patches")
    p.add_argument("--patch_ext", type=str, default="jpg")
    p.add_argument("--map_path", type=str, default="test/output")
    p.add_argument("--export_scores", type=int, default=0)
    p.add_argument("--score_path", type=str, default="test/score")
    # new (BASELINE configs[4]): multi-scale maps over the pyramid layout compute_feats.py --magnification tree reads
    p.add_argument("--magnification", type=str, default="single", help="single | tree (low tiles + one folder of "
                   "high-magnification children per tile; needs --embedder_weights_low / --embedder_weights_high and "
                   "an aggregator trained on the 1024-d tree features, --feats_size 1024)")
    p.add_argument("--embedder_weights_low", type=str, default=None)
    p.add_argument("--embedder_weights_high", type=str, default=None)
    p.add_argument("--tree_fusion", type=str, default="cat", help="[cat|fusion]")
    p.add_argument("--gpu_decode", action="store_true", help="(new, default off) decode the tiles' JPEG files on the GPU "
                   "(dsmil_jpeg_decode: bit-identical to Pillow for baseline JPEGs, Pillow for everything else) instead of in "
                   "DataLoader workers (attention_map.py:69-79)")
    return p


def build_milnet(args, device):
    if args.embedder_weights == "ImageNet":
        print("Use ImageNet features")
        resnet = models.resnet18(pretrained=True, norm_layer=nn.BatchNorm2d)
    else:
        resnet = models.resnet18(pretrained=False, norm_layer=nn.InstanceNorm2d)
    for prm in resnet.parameters():
        prm.requires_grad = False
    resnet.fc = nn.Identity()
    i_classifier = mil.IClassifier(resnet, args.feats_size, output_class=args.num_classes)
    b_classifier = mil.BClassifier(input_size=args.feats_size, output_class=args.num_classes)
    milnet = mil.MILNet(i_classifier, b_classifier).to(device)
    if args.embedder_weights != "ImageNet":
        pipeline.load_simclr_weights(milnet.i_classifier, torch.load(args.embedder_weights, map_location=device))
    sd = torch.load(args.aggregator_weights, map_location=device)
    sd["i_classifier.fc.weight"] = sd["i_classifier.fc.0.weight"]     # attention_map.py:163-164
    sd["i_classifier.fc.bias"] = sd["i_classifier.fc.0.bias"]
    milnet.load_state_dict(sd, strict=False)
    return milnet


def build_tree_models(args, device):
    """Two ResNet-18-IN embedders (low / high magnification, compute_feats.py:198-209) and the aggregator the
    reference trains on tree features: MILNet(FCLayer(feats_size), BClassifier(feats_size)) (train_tcga.py:236-238)."""
    embs = []
    for path in (args.embedder_weights_low, args.embedder_weights_high):
        resnet = models.resnet18(pretrained=False, norm_layer=nn.InstanceNorm2d)
        for prm in resnet.parameters():
            prm.requires_grad = False
        resnet.fc = nn.Identity()
        ic = mil.IClassifier(resnet, 512, output_class=args.num_classes).to(device)
        pipeline.load_simclr_weights(ic, torch.load(path, map_location=device))
        embs.append(ic.eval())
    milnet = mil.MILNet(mil.FCLayer(in_size=args.feats_size, out_size=args.num_classes),
                        mil.BClassifier(input_size=args.feats_size, output_class=args.num_classes)).to(device)
    milnet.load_state_dict(torch.load(args.aggregator_weights, map_location=device), strict=True)
    return embs[0], embs[1], milnet


def main(argv=None):
    warnings.filterwarnings("ignore")
    args = build_parser().parse_args(argv)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    emb_low = emb_high = None
    if args.magnification == "tree":
        if not (args.embedder_weights_low and args.embedder_weights_high):
            raise ValueError("--magnification tree needs --embedder_weights_low and --embedder_weights_high")
        emb_low, emb_high, milnet = build_tree_models(args, device)
    else:
        milnet = build_milnet(args, device)
    bags_list = glob.glob(os.path.join(args.bag_path, "*"))
    os.makedirs(args.map_path, exist_ok=True)
    if args.export_scores:
        os.makedirs(args.score_path, exist_ok=True)
    if args.class_name is None:
        args.class_name = ["class {}".format(c) for c in range(args.num_classes)]
    if len(args.thres) != args.num_classes:
        raise ValueError("Number of thresholds does not match classes.")
    pipeline.GPU_DECODE[0] = bool(args.gpu_decode)
    try:
        pipeline.attention_maps(args, bags_list, milnet, embedder_low=emb_low, embedder_high=emb_high)
    finally:
        pipeline.GPU_DECODE[0] = False


if __name__ == "__main__":
    main()
