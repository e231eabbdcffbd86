#!/usr/bin/env python3
"""Drop-in for the reference's train_mil.py (flags: train_mil.py:112-120): DSMIL on the classical
MIL benchmarks (MUSK1/2, Elephant, Fox, Tiger), k-fold cross validation.  BASELINE config 0: runs
on CPU when no GPU is present (the reference hard-codes .cuda(), train_mil.py:47,49,171)."""
import argparse
import sys

import numpy as np
import torch
import torch.nn as nn

DATA = {"musk1": ("datasets/mil_dataset/Musk/musk1norm.svm", 166),
        "musk2": ("datasets/mil_dataset/Musk/musk2norm.svm", 166),
        "elephant": ("datasets/mil_dataset/Elephant/data_100x100.svm", 230),
        "fox": ("datasets/mil_dataset/Fox/data_100x100.svm", 230),
        "tiger": ("datasets/mil_dataset/Tiger/data_100x100.svm", 230)}


def build_parser():
    p = argparse.ArgumentParser(description="Train DSMIL on classfical MIL datasets")
    p.add_argument("--datasets", default="musk1", type=str, help="musk1, musk2, elephant, fox, tiger [musk1]")
    p.add_argument("--lr", default=0.0002, type=float, help="Initial learning rate [0.0002]")
    p.add_argument("--num_epoch", default=40, type=int, help="Number of total training epochs [40]")
    p.add_argument("--cv_fold", default=10, type=int, help="Number of cross validation fold [10]")
    p.add_argument("--weight_decay", default=5e-3, type=float, help="Weight decay [5e-3]")
    p.add_argument("--model", default="dsmil", type=str, help="MIL model [dsmil]")
    p.add_argument("--data_file", default=None, type=str, help="override the dataset file path")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.model == "dsmil":
        import dsmil as mil
    else:
        import abmil as mil
    from dsmil_wsi_amd import training as T
    path, args.num_feats = DATA[args.datasets]
    X, bag_ids, labels = T.parse_mil_file(args.data_file or path)
    X = X[:, :args.num_feats]
    if X.shape[1] < args.num_feats:
        X = np.pad(X, ((0, 0), (0, args.num_feats - X.shape[1])))
    bags, ys = T.group_bags(X, bag_ids, labels)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    order = np.random.permutation(len(bags))
    n = int(len(order) / args.cv_fold)
    chunks = [order[i:i + n] for i in range(0, len(order), n)]          # train_mil.py:99-104
    acs = []
    print("Dataset: " + args.datasets)
    for k in range(args.cv_fold):
        print("Start %d-fold cross validation: fold %d " % (args.cv_fold, k))
        test_idx = chunks[k]
        train_idx = np.concatenate([c for j, c in enumerate(chunks) if j != k])
        milnet = mil.MILNet(mil.FCLayer(args.num_feats, 1),
                            mil.BClassifier(input_size=args.num_feats, output_class=1)).to(device)
        pos = float(ys[train_idx].sum())
        criterion = nn.BCEWithLogitsLoss(torch.tensor((len(train_idx) - pos) / max(pos, 1.0), device=device))
        optimizer = torch.optim.Adam(milnet.parameters(), lr=args.lr, betas=(0.5, 0.9), weight_decay=args.weight_decay)
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, args.num_epoch, 0)
        optimal_ac = 0
        for epoch in range(args.num_epoch):
            train_loss = T.mil_epoch_train(bags, ys, train_idx, milnet, criterion, optimizer, device)
            test_loss, preds = T.mil_epoch_test(bags, ys, test_idx, milnet, criterion, device)
            if len(np.unique(ys[test_idx])) > 1:
                accuracy, auc_value, precision, recall, fscore = T.five_scores(ys[test_idx], preds)
            else:
                accuracy = float(np.mean((preds >= 0.5) == (ys[test_idx] > 0)))
                auc_value = precision = recall = fscore = float("nan")
            sys.stdout.write("\r Epoch [%d/%d] train loss: %.4f, test loss: %.4f, accuracy: %.4f, aug score: %.4f, "
                             "precision: %.4f, recall: %.4f, fscore: %.4f " %
                             (epoch + 1, args.num_epoch, train_loss, test_loss, accuracy, auc_value, precision, recall, fscore))
            optimal_ac = max(accuracy, optimal_ac)
            scheduler.step()
        print("\n Optimal accuracy: %.4f " % optimal_ac)
        acs.append(optimal_ac)
    print("Cross validation accuracy mean: %.4f, std %.4f " % (np.mean(acs), np.std(acs)))
    return acs


if __name__ == "__main__":
    main()
