"""End-to-end host plumbing on CPU with tiny synthetic datasets: compute_feats.py (single + tree),
train_tcga.py, train_mil.py (BASELINE config 0: MUSK1-format, CPU), attention_map.py."""
import collections
import glob
import zlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

import dsmil
import resnet_oracle as ro

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _jpeg(path, seed, size=64):
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)).save(path, quality=95)


def _simclr_checkpoint(path, seed):
    w = ro.make_weights(seed=seed)
    sd = collections.OrderedDict(("features." + k, v) for k, v in w.items())
    sd["l1.weight"] = torch.zeros(512, 512); sd["l1.bias"] = torch.zeros(512)
    sd["l2.weight"] = torch.zeros(256, 512); sd["l2.bias"] = torch.zeros(256)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(sd, path)
    return w


@pytest.fixture()
def workdir(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "")
    return tmp_path


def test_compute_feats_single_matches_oracle(workdir):
    import compute_feats as cf
    w = _simclr_checkpoint("simclr/runs/r0/checkpoints/model.pth", 31)
    for cls in ("0_normal", "1_tumor"):
        for slide in ("s1", "s2"):
            for i in range(3):
                _jpeg(f"WSI/toy/single/{cls}/{slide}/{i}_{i + 1}.jpeg", zlib.crc32(f'{cls}/{slide}/{i}'.encode()) % 10000)
    cf.main(["--dataset", "toy", "--weights", "r0", "--batch_size", "2", "--num_workers", "0", "--num_classes", "1",
             "--save_npy"])
    import pandas as pd
    csvs = sorted(glob.glob("datasets/toy/*/*.csv"))
    assert len(csvs) == 4
    df = pd.read_csv("datasets/toy/toy.csv")
    assert len(df) == 4 and set(df["label"]) == {0, 1}
    assert os.path.exists("embedder/toy/embedder.pth")
    # rows must equal the oracle's features for the same files in glob order, to the CSV quantum
    from dsmil_wsi_amd.pipeline import PatchFiles, glob_patches
    bag = "WSI/toy/single/1_tumor/s2"
    files = glob_patches(bag, "single")
    x = torch.stack([PatchFiles(files)[i]["input"] for i in range(len(files))])
    with torch.no_grad():
        ref = ro.resnet18_in_features(x, w).numpy()
    got = pd.read_csv("datasets/toy/1_tumor/s2.csv").to_numpy()
    assert got.shape == (3, 512)
    np.testing.assert_allclose(got, ref, atol=8e-5)
    exact = np.load("datasets/toy/1_tumor/s2.npy")           # --save_npy: the unquantised rows
    assert exact.dtype == np.float32 and exact.shape == (3, 512)
    np.testing.assert_allclose(exact, ref, atol=3e-5)
    np.testing.assert_allclose(got, exact, atol=5.1e-5)  # half the '%.4f' CSV quantum + fp32 order effects


def test_compute_feats_tree_concat(workdir):
    import compute_feats as cf
    _simclr_checkpoint("simclr/runs/hi/checkpoints/model.pth", 41)
    _simclr_checkpoint("simclr/runs/lo/checkpoints/model.pth", 42)
    for li in range(2):
        _jpeg(f"WSI/toy2/pyramid/0_a/s1/{li}_0.jpeg", 100 + li)
        for hi in range(3):
            _jpeg(f"WSI/toy2/pyramid/0_a/s1/{li}_0/{hi}_1.jpeg", 200 + 10 * li + hi)
    cf.main(["--dataset", "toy2", "--magnification", "tree", "--weights_high", "hi", "--weights_low", "lo",
             "--batch_size", "4", "--num_workers", "0"])
    import pandas as pd
    got = pd.read_csv("datasets/toy2/0_a/s1.csv").to_numpy()
    assert got.shape == (6, 1024)
    # the low-magnification half repeats within a parent's children
    assert np.allclose(got[0, 512:], got[1, 512:]) and not np.allclose(got[0, :512], got[1, :512])


def test_train_tcga_runs_and_learns(workdir):
    import pandas as pd
    import train_tcga as tt
    rng = np.random.default_rng(0)
    os.makedirs("datasets/toy3/c0", exist_ok=True)
    os.makedirs("datasets/toy3/c1", exist_ok=True)
    rows = []
    direction = rng.standard_normal(32).astype(np.float32)
    for b in range(20):
        lab = b % 2
        X = rng.standard_normal((12 + b, 32)).astype(np.float32)
        if lab:
            X[:3] += 3.0 * direction
        p = f"datasets/toy3/c{lab}/bag{b}.csv"
        pd.DataFrame(X).to_csv(p, index=False, float_format="%.4f")
        rows.append((p, lab))
    pd.DataFrame(rows, columns=["0", "label"]).to_csv("datasets/toy3/toy3.csv", index=False)
    tt.main(["--dataset", "toy3", "--num_classes", "1", "--feats_size", "32", "--num_epochs", "6",
             "--lr", "0.002", "--eval_scheme", "5-fold-cv"])
    assert glob.glob("weights/*/fold_0_*.pth") and glob.glob("weights/*/fold_0_*.json")
    sd = torch.load(sorted(glob.glob("weights/*/fold_0_*.pth"))[0])
    assert list(sd.keys())[0] == "i_classifier.fc.0.weight"


def test_train_mil_musk_format_on_cpu(workdir):
    import train_mil as tm
    from dsmil_wsi_amd import training as T
    path = T.write_synthetic_mil_file("datasets/mil_dataset/Musk/musk1norm.svm")
    X, bag_ids, labels = T.parse_mil_file(path)
    assert X.shape == (476, 166) and len(np.unique(bag_ids)) == 92
    bags, ys = T.group_bags(X, bag_ids, labels)
    assert int(ys.sum()) == 47
    np.random.seed(0)
    torch.manual_seed(0)
    acs = tm.main(["--datasets", "musk1", "--num_epoch", "3", "--cv_fold", "4"])
    assert len(acs) == 4 and all(0.0 <= a <= 1.0 for a in acs)


def test_attention_map_end_to_end(workdir):
    import attention_map as am
    from util import state_dict_from_npz
    from conftest import load_weights
    w = ro.make_weights(seed=51)
    emb = collections.OrderedDict(w)
    for n in ("l1.weight", "l1.bias", "l2.weight", "l2.bias"):
        emb[n] = torch.zeros(1)
    os.makedirs("test/weights", exist_ok=True)
    torch.save(emb, "test/weights/embedder.pth")
    torch.save(state_dict_from_npz(load_weights("tcga")), "test/weights/aggregator.pth")
    for r in range(2):
        for c in range(3):
            _jpeg(f"test/patches/slideA/{r}_{c}.jpg", 300 + 10 * r + c)
    am.main(["--num_workers", "0", "--thres", "0.0", "0.0", "--export_scores", "1", "--batch_size", "4"])
    from PIL import Image
    img = np.asarray(Image.open("test/output/slideA.png"))
    assert img.shape == (64, 96, 3)
    import pandas as pd
    sc = pd.read_csv("test/score/slideA.csv")
    assert sc.shape == (6, 3) and abs(sc["0"].sum() - 1.0) < 1e-4


@pytest.mark.gpu
def test_entry_points_on_gpu_use_native_path(tmp_path, monkeypatch):
    """compute_feats.py, attention_map.py and train_tcga.py on a real GPU: same tiny datasets, the
    modules run in libdsmil_hip.so (a missing library would raise, there is no GPU fallback)."""
    monkeypatch.chdir(tmp_path)
    import compute_feats as cf
    import pandas as pd
    import train_tcga as tt
    from dsmil_wsi_amd.pipeline import PatchFiles, glob_patches
    w = _simclr_checkpoint("simclr/runs/r0/checkpoints/model.pth", 31)
    for slide in ("s1", "s2"):
        for i in range(5):
            _jpeg(f"WSI/toy/single/0_x/{slide}/{i}_{i + 1}.jpeg", zlib.crc32(f'{slide}/{i}'.encode()) % 10000, size=224)
    cf.main(["--dataset", "toy", "--weights", "r0", "--batch_size", "4", "--num_workers", "0"])
    files = glob_patches("WSI/toy/single/0_x/s2", "single")
    x = torch.stack([PatchFiles(files)[i]["input"] for i in range(len(files))])
    with torch.no_grad():
        ref = ro.resnet18_in_features(x.double(), {k: v.double() for k, v in w.items()}).numpy()
    got = pd.read_csv("datasets/toy/0_x/s2.csv").to_numpy()
    np.testing.assert_allclose(got, ref, atol=1.6e-4)   # 1e-4 parity + the CSV's %.4f quantum
    # aggregator training on the GPU
    rng = np.random.default_rng(0)
    os.makedirs("datasets/toy3/c0", exist_ok=True)
    os.makedirs("datasets/toy3/c1", exist_ok=True)
    rows = []
    direction = rng.standard_normal(512).astype(np.float32)
    for b in range(10):
        lab = b % 2
        X = rng.standard_normal((200 + 17 * b, 512)).astype(np.float32)
        if lab:
            X[:5] += 2.0 * direction
        p = f"datasets/toy3/c{lab}/bag{b}.csv"
        pd.DataFrame(X).to_csv(p, index=False, float_format="%.4f")
        rows.append((p, lab))
    pd.DataFrame(rows, columns=["0", "label"]).to_csv("datasets/toy3/toy3.csv", index=False)
    tt.main(["--dataset", "toy3", "--num_classes", "2", "--num_epochs", "3", "--lr", "0.001"])
    assert glob.glob("weights/*/fold_*_*.pth")   # a fold whose 4 test bags score 0 saves nothing
