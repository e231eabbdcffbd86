#!/usr/bin/env python3
"""Drop-in for the reference's train_tcga.py (same flags and on-disk conventions,
train_tcga.py:199-432): aggregator training/evaluation on pre-computed feature CSVs, with the
per-bag forward in libdsmil_hip.so."""
import argparse
import os

import pandas as pd
import torch


def build_parser():
    p = argparse.ArgumentParser(description="Train DSMIL on 20x patch features learned by SimCLR")
    p.add_argument("--num_classes", default=2, type=int, help="Number of output classes [2]")
    p.add_argument("--feats_size", default=512, type=int, help="Dimension of the feature size [512]")
    p.add_argument("--lr", default=0.0001, type=float, help="Initial learning rate [0.0001]")
    p.add_argument("--num_epochs", default=50, type=int, help="Number of total training epochs")
    p.add_argument("--stop_epochs", default=10, type=int, help="Early-stop patience [10]")
    p.add_argument("--gpu_index", type=int, nargs="+", default=(0,), help="GPU ID(s) [0]")
    p.add_argument("--weight_decay", default=1e-3, type=float, help="Weight decay [1e-3]")
    p.add_argument("--dataset", default="TCGA-lung-default", type=str, help="Dataset folder name")
    p.add_argument("--split", default=0.2, type=float, help="Training/Validation split [0.2]")
    p.add_argument("--model", default="dsmil", type=str, help="MIL model [dsmil]")
    p.add_argument("--dropout_patch", default=0, type=float, help="Patch dropout rate [0]")
    p.add_argument("--dropout_node", default=0, type=float, help="Bag classifier dropout rate [0]")
    p.add_argument("--non_linearity", default=1, type=float, help="Additional nonlinear operation [0]")
    p.add_argument("--average", type=bool, default=False, help="Average max-pooling and bag scores")
    p.add_argument("--eval_scheme", default="5-fold-cv", type=str,
                   help="[5-fold-cv | 5-fold-cv-standalone-test | 5-time-train+valid+test ]")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    print(args.eval_scheme)
    os.environ["CUDA_VISIBLE_DEVICES"] = ",".join(str(x) for x in tuple(args.gpu_index))
    if args.model == "dsmil":
        import dsmil as mil
    elif args.model == "abmil":
        import abmil as mil  # not shipped by the reference either (.gitignore:17)
    else:
        raise ValueError(args.model)
    from dsmil_wsi_amd import training
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if args.dataset == "TCGA-lung-default":
        bags_csv = "datasets/tcga-dataset/TCGA.csv"
    else:
        bags_csv = os.path.join("datasets", args.dataset, args.dataset + ".csv")
    training.generate_pt_files(args, pd.read_csv(bags_csv))
    training.run_eval_scheme(args, mil, device)


if __name__ == "__main__":
    main()
