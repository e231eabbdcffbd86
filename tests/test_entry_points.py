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
    # (this test is not GPU-marked: it checks the script on the CPU module path; run on a GPU box — without `-m "not gpu"` — the
    # script takes the HIP embedder, whose parity bar is 1e-4, under the same CSV quantum)
    on_gpu = torch.cuda.is_available()
    np.testing.assert_allclose(got, ref, atol=1.6e-4 if on_gpu else 8e-5)
    exact = np.load("datasets/toy/1_tumor/s2.npy")           # --save_npy: the unquantised rows
    assert exact.dtype == np.float32 and exact.shape == (3, 512)
    np.testing.assert_allclose(exact, ref, atol=1e-4 if on_gpu else 3e-5)
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


def test_attention_map_multiscale_from_files(workdir):
    """attention_map.py --magnification tree (new): low tiles + a folder of high children per tile -> [high || low]
    1024-d tree features (compute_feats.py:113-114) -> MILNet(FCLayer(1024), BClassifier(1024)) -> map at the high
    tiles' positions.  The attention CSV must equal the product modules applied to the same tree features."""
    import attention_map as am
    import pandas as pd
    from util import state_dict_from_npz
    from conftest import load_weights
    from dsmil_wsi_amd import pipeline as pl
    for name, seed in (("low", 61), ("high", 62)):
        emb = collections.OrderedDict(ro.make_weights(seed=seed))
        for n in ("l1.weight", "l1.bias", "l2.weight", "l2.bias"):
            emb[n] = torch.zeros(1)
        os.makedirs("test/weights", exist_ok=True)
        torch.save(emb, f"test/weights/embedder_{name}.pth")
    torch.save(state_dict_from_npz(load_weights("tree")), "test/weights/aggregator_tree.pth")
    for lr in range(2):
        for lc in range(2):
            _jpeg(f"test/pyr/slideT/{lr}_{lc}.jpg", 400 + 10 * lr + lc)
            for hr in range(2):
                for hc in range(2):
                    _jpeg(f"test/pyr/slideT/{lr}_{lc}/{2 * lr + hr}_{2 * lc + hc}.jpg", 500 + 100 * lr + 40 * lc + 2 * hr + hc)
    am.main(["--magnification", "tree", "--feats_size", "1024", "--embedder_weights_low", "test/weights/embedder_low.pth",
             "--embedder_weights_high", "test/weights/embedder_high.pth", "--aggregator_weights",
             "test/weights/aggregator_tree.pth", "--bag_path", "test/pyr", "--num_workers", "0", "--thres", "0.0", "0.0",
             "--export_scores", "1", "--batch_size", "8"])
    from PIL import Image
    img = np.asarray(Image.open("test/output/slideT.png"))
    assert img.shape == (4 * 32, 4 * 32, 3)
    sc = pd.read_csv("test/score/slideT.csv")
    assert sc.shape == (16, 3) and abs(sc["0"].sum() - 1.0) < 1e-4 and abs(sc["1"].sum() - 1.0) < 1e-4
    # the same through the library functions: tree features -> product MILNet on CPU
    args = am.build_parser().parse_args(["--magnification", "tree", "--feats_size", "1024",
                                         "--embedder_weights_low", "test/weights/embedder_low.pth",
                                         "--embedder_weights_high", "test/weights/embedder_high.pth",
                                         "--aggregator_weights", "test/weights/aggregator_tree.pth"])
    lo, hi, net = am.build_tree_models(args, torch.device("cpu"))
    tree, files, n_low = pl.tree_feats_of_bag("test/pyr/slideT", lo, hi, "cat", 8, 0, ext=("jpg",))
    assert tree.shape == (16, 1024) and n_low == 4
    with torch.no_grad():
        _, _, A, _ = net.eval()(tree)
    order = [str(pl.patch_position(f)) for f in files]
    assert list(sc["pos"]) == order
    np.testing.assert_allclose(sc[["0", "1"]].to_numpy(), A.numpy(), atol=1e-6)


def test_pyramid_tiles_of_a_synthetic_slide():
    """configs[4] data generator: every low tile covers 4 x 4 high tiles; children are listed parent-major and
    row-major inside the parent (the walk of compute_feats.py:98-109)."""
    from dsmil_wsi_amd import pipeline as pl
    g = torch.Generator().manual_seed(0)
    T = 16
    wsi = torch.randint(0, 256, (2 * T * 4, 3 * T * 4, 3), generator=g, dtype=torch.uint8)
    low, high, parent, pos = pl.pyramid_tiles(wsi, tile=T, factor=4)
    assert low.shape == (6, T, T, 3) and high.shape == (96, T, T, 3)
    assert parent.tolist() == [i // 16 for i in range(96)]
    for k in (0, 5, 37, 95):
        r, c = pos[k].tolist()
        assert torch.equal(high[k], wsi[r * T:(r + 1) * T, c * T:(c + 1) * T])
        li = int(parent[k])
        ly, lx = divmod(li, 3)
        assert (r // 4, c // 4) == (ly, lx)
        box = wsi[ly * 4 * T:(ly + 1) * 4 * T, lx * 4 * T:(lx + 1) * 4 * T].view(T, 4, T, 4, 3).double().mean((1, 3))
        assert torch.equal(low[li], torch.floor(box + 0.5).to(torch.uint8))
    assert len({tuple(p) for p in pos.tolist()}) == 96
    # a rank's range of low tiles (dist.shard_range) = the slice of the whole; the copy moves 8-byte words when a tile row is
    # a multiple of 8 bytes (T = 16: 48) and bytes otherwise (T = 14: 42) — same tiles
    l2, h2, p2, q2 = pl.pyramid_tiles(wsi, tile=T, factor=4, lo=1, hi=5)
    assert torch.equal(l2, low[1:5]) and torch.equal(h2, high[16:80]) and torch.equal(q2, pos[16:80]) and p2.tolist() == [i // 16 for i in range(64)]
    T2 = 14
    w2 = torch.randint(0, 256, (2 * T2 * 4, 3 * T2 * 4, 3), generator=g, dtype=torch.uint8)
    _, h3, _, q3 = pl.pyramid_tiles(w2, tile=T2, factor=4)
    for k in (0, 41, 95):
        r, c = q3[k].tolist()
        assert torch.equal(h3[k], w2[r * T2:(r + 1) * T2, c * T2:(c + 1) * T2])


@pytest.mark.gpu
def test_multiscale_end_to_end_on_gpu():
    """BASELINE configs[4] on the device: seeded uint8 slide array -> tiles at two magnifications -> both embedders
    (uint8 ingest, native trunks) -> [high || low] -> MILNet(feats_size=1024) (native aggregator) -> attention map.
    Checks: tree features vs the fp64 embedder oracle on the same tiles; aggregator outputs vs the fp64 aggregator
    oracle ON THE GATHERED FEATURES; the colour map vs pipeline.attention_colormap applied to the oracle's attention."""
    import agg_oracle as orc
    from conftest import load_weights
    from util import build_net
    from dsmil_wsi_amd import pipeline as pl
    from dsmil_wsi_amd.resnet import resnet18
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(4)
    wsi = torch.randint(0, 256, (2 * 896, 2 * 896, 3), generator=g, dtype=torch.uint8)     # 4 low / 64 high tiles

    def emb(seed):
        res = resnet18(norm_layer=nn.InstanceNorm2d)
        res.fc = nn.Identity()
        w = ro.make_weights(seed=seed)
        res.load_state_dict(w, strict=True)
        for p_ in res.parameters():
            p_.requires_grad = False
        return dsmil.IClassifier(res, 512, output_class=2).eval().to(dev), w
    (e_lo, w_lo), (e_hi, w_hi) = emb(71), emb(72)
    net = build_net("tree", dev)
    colors = [np.array([255, 40, 0]), np.array([0, 90, 255])]
    out = pl.multiscale_attention_map(wsi.to(dev), e_lo, e_hi, net, [0.0, 0.0], colors, batch_size=16)
    assert out["feats"].shape == (64, 1024) and out["A"].shape == (64, 2) and out["cmap"].shape == (8 * 32, 8 * 32, 3)
    # embedder halves vs the fp64 oracle on the very tiles the pipeline cut
    low, high, parent, pos = pl.pyramid_tiles(wsi)
    to_f = lambda t: t.permute(0, 3, 1, 2).double().div(255)
    with torch.no_grad():
        f_lo = ro.resnet18_in_features(to_f(low), {k: v.double() for k, v in w_lo.items()})
        f_hi = ro.resnet18_in_features(to_f(high), {k: v.double() for k, v in w_hi.items()})
    tree_ref = torch.cat([f_hi, f_lo[parent]], dim=1).numpy()
    feats = out["feats"].cpu().numpy()
    np.testing.assert_allclose(feats, tree_ref, atol=1e-4, rtol=1e-4)
    assert np.array_equal(out["pos"].cpu().numpy(), pos.numpy())
    # aggregator vs the fp64 oracle on the gathered features
    r = orc.milnet_forward(feats, load_weights("tree"), dtype="f64")
    np.testing.assert_allclose(out["classes"].cpu().numpy(), r[0], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(out["pred"].cpu().numpy(), r[1], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(out["A"].cpu().numpy(), r[2], atol=1e-6, rtol=1e-3)
    np.testing.assert_allclose(out["B"].cpu().numpy(), r[3], atol=1e-4, rtol=1e-5)
    # map bytes vs the host colour-map function on the oracle's attention (a byte may differ where rint() sits on .5)
    prob = 1.0 / (1.0 + np.exp(-np.asarray(r[1], np.float64).ravel()))
    cm = pl.attention_colormap(np.asarray(r[2], np.float64), pos.numpy(), prob, [0.0, 0.0], colors)
    assert cm.shape == out["cmap"].shape
    assert np.abs(cm.astype(int) - out["cmap"].astype(int)).max() <= 1


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


def _oracle_map(feat_fn, files, pos, agg_tag, thres, colors):
    """Attention colour map the fp64 oracles give for the same decoded tiles: embedder oracle -> aggregator oracle ->
    pipeline.attention_colormap (the host restatement of attention_map.py:86-113)."""
    import agg_oracle as orc
    from conftest import load_weights
    from dsmil_wsi_amd import pipeline as pl
    feats = feat_fn(files)
    r = orc.milnet_forward(feats, load_weights(agg_tag), dtype="f64")
    prob = 1.0 / (1.0 + np.exp(-np.asarray(r[1], np.float64).ravel()))
    return pl.attention_colormap(np.asarray(r[2], np.float64), pos, prob, thres, colors, log=lambda *_: None), r


@pytest.mark.gpu
def test_attention_map_scripts_on_gpu(tmp_path, monkeypatch):
    """attention_map.py itself on the GPU, from files: single scale (attention_map.py:59-118: embed -> aggregate ->
    threshold -> PNG + score CSV) and --magnification tree.  The PNG must equal the colour map that the fp64 embedder
    oracle + the fp64 aggregator oracle + attention_colormap give for the same decoded tiles (a byte may differ where
    rint() sits on .5), the score CSV the oracle's attention."""
    monkeypatch.chdir(tmp_path)
    import attention_map as am
    import pandas as pd
    from PIL import Image
    from util import state_dict_from_npz
    from conftest import load_weights
    from dsmil_wsi_amd import pipeline as pl
    from dsmil_wsi_amd.pipeline import PatchFiles
    assert torch.cuda.is_available()
    os.makedirs("test/weights", exist_ok=True)
    ws = {}
    for name, seed in (("embedder", 51), ("embedder_low", 61), ("embedder_high", 62)):
        ws[name] = ro.make_weights(seed=seed)
        emb = collections.OrderedDict(ws[name])
        for n in ("l1.weight", "l1.bias", "l2.weight", "l2.bias"):
            emb[n] = torch.zeros(1)
        torch.save(emb, f"test/weights/{name}.pth")
    torch.save(state_dict_from_npz(load_weights("tcga")), "test/weights/aggregator.pth")
    torch.save(state_dict_from_npz(load_weights("tree")), "test/weights/aggregator_tree.pth")

    def feats_of(wname):
        def f(files):
            x = torch.stack([PatchFiles(files)[i]["input"] for i in range(len(files))]).double()
            with torch.no_grad():
                return ro.resnet18_in_features(x, {k: v.double() for k, v in ws[wname].items()}).numpy()
        return f

    # ---- single scale
    for r in range(3):
        for c in range(4):
            _jpeg(f"test/patches/slideA/{r}_{c}.jpg", 300 + 10 * r + c, size=224)
    np.random.seed(7)
    am.main(["--num_workers", "0", "--thres", "0.0", "0.0", "--export_scores", "1", "--batch_size", "5"])
    np.random.seed(7)
    colors = [np.random.choice(range(256), size=3) for _ in range(2)]
    sc = pd.read_csv("test/score/slideA.csv")
    files = sorted(glob.glob("test/patches/slideA/*.jpg"), key=lambda f: list(sc["pos"]).index(str(pl.patch_position(f))))
    pos = np.vstack([pl.patch_position(f) for f in files])
    cm, r_ = _oracle_map(feats_of("embedder"), files, pos, "tcga", [0.0, 0.0], colors)
    img = np.asarray(Image.open("test/output/slideA.png"))
    assert img.shape == cm.shape == (3 * 32, 4 * 32, 3)
    assert np.abs(img.astype(int) - cm.astype(int)).max() <= 1
    np.testing.assert_allclose(sc[["0", "1"]].to_numpy(), r_[2], atol=1e-6, rtol=2e-3)

    # ---- two scales: 2 x 2 low tiles, 2 x 2 high children each
    for lr in range(2):
        for lc in range(2):
            _jpeg(f"test/pyr/slideT/{lr}_{lc}.jpg", 400 + 10 * lr + lc, size=224)
            for hr in range(2):
                for hc in range(2):
                    _jpeg(f"test/pyr/slideT/{lr}_{lc}/{2 * lr + hr}_{2 * lc + hc}.jpg", 500 + 100 * lr + 40 * lc + 2 * hr + hc, size=224)
    np.random.seed(8)
    am.main(["--magnification", "tree", "--feats_size", "1024", "--embedder_weights_low", "test/weights/embedder_low.pth",
             "--embedder_weights_high", "test/weights/embedder_high.pth", "--aggregator_weights",
             "test/weights/aggregator_tree.pth", "--bag_path", "test/pyr", "--num_workers", "0", "--thres", "0.0", "0.0",
             "--export_scores", "1", "--batch_size", "8"])
    np.random.seed(8)
    colors = [np.random.choice(range(256), size=3) for _ in range(2)]
    sc = pd.read_csv("test/score/slideT.csv")
    assert sc.shape == (16, 3)
    highs = glob.glob("test/pyr/slideT/*/*.jpg")
    highs = sorted(highs, key=lambda f: list(sc["pos"]).index(str(pl.patch_position(f))))
    pos = np.vstack([pl.patch_position(f) for f in highs])

    def tree_feats(files):
        lows = [os.path.dirname(f) + ".jpg" for f in files]
        return np.concatenate([feats_of("embedder_high")(files), feats_of("embedder_low")(lows)], axis=1)
    cm, r_ = _oracle_map(tree_feats, highs, pos, "tree", [0.0, 0.0], colors)
    img = np.asarray(Image.open("test/output/slideT.png"))
    assert img.shape == cm.shape == (4 * 32, 4 * 32, 3)
    assert np.abs(img.astype(int) - cm.astype(int)).max() <= 1
    np.testing.assert_allclose(sc[["0", "1"]].to_numpy(), r_[2], atol=1e-6, rtol=2e-3)


def _bg_dataset(root, size):
    """Five tiles per slide: noise (tissue), flat grey (background), a faint ramp (background), noise, flat white."""
    from PIL import Image
    rng = np.random.default_rng(77)
    os.makedirs(root, exist_ok=True)
    tiles = [rng.integers(0, 256, (size, size, 3), dtype=np.uint8), np.full((size, size, 3), 228, np.uint8),
             np.repeat((np.arange(size)[None, :, None] * 40 // size + 100).astype(np.uint8), size, axis=0).repeat(3, axis=2),
             rng.integers(0, 256, (size, size, 3), dtype=np.uint8), np.full((size, size, 3), 255, np.uint8)]
    for i, t in enumerate(tiles):
        Image.fromarray(t).save(os.path.join(root, f"{i}_{i}.jpeg"), quality=95)


def _pil_keep(files, thr):
    """deepzoom_tiler.py:56-61 itself, on the decoded files: mean(ImageStat(FIND_EDGES).sum) / tile_size^2 > thr."""
    from PIL import Image, ImageFilter, ImageStat
    out = []
    for f in files:
        with Image.open(f) as im:
            im = im.convert("RGB")
            edge = np.mean(ImageStat.Stat(im.filter(ImageFilter.FIND_EDGES)).sum) / (im.size[0] ** 2)
        out.append(edge > thr)
    return np.array(out)


def _check_bg_filter(size, atol):
    import compute_feats as cf
    import pandas as pd
    from dsmil_wsi_amd.pipeline import PatchFiles, glob_patches
    w = _simclr_checkpoint("simclr/runs/r0/checkpoints/model.pth", 31)
    _bg_dataset("WSI/bg/single/0_x/s1", size)
    cf.main(["--dataset", "bg", "--weights", "r0", "--batch_size", "2", "--num_workers", "0", "--bg_threshold", "15", "--save_npy"])
    files = glob_patches("WSI/bg/single/0_x/s1", "single")
    keep = _pil_keep(files, 15.0)
    assert len(files) == 5 and 2 <= keep.sum() < 5 and not keep.all()     # the noise tiles stay, the flat tiles go
    got = np.load("datasets/bg/0_x/s1.npy")
    assert got.shape == (int(keep.sum()), 512)                       # background rows are absent, order preserved
    kept = [f for f, k in zip(files, keep) if k]
    x = torch.stack([PatchFiles(kept)[i]["input"] for i in range(len(kept))])
    with torch.no_grad():
        ref = ro.resnet18_in_features(x.double(), {k: v.double() for k, v in w.items()}).numpy()
    np.testing.assert_allclose(got, ref, atol=atol)


@pytest.mark.gpu
def test_compute_feats_reduced_precision_flags_on_gpu(tmp_path, monkeypatch):
    """compute_feats.py --precision half | bf16 (new, default fp32): the opt-in 16-bit-activation trunk (csrc/resnet_b16.h) behind
    the reference's command line.  The rows of the default run are the reference's features (1e-4 bar, elsewhere); `half` must
    stay within 5e-3 of them, `bf16` within 5e-2 — the bars stated in include/dsmil_hip.h — and both must differ from them."""
    import zlib
    import glob
    monkeypatch.chdir(tmp_path)
    import compute_feats as cf
    _simclr_checkpoint("simclr/runs/r0/checkpoints/model.pth", 31)
    for i in range(10):
        _jpeg(f"WSI/toy/single/0_x/s1/{i}_{i + 1}.jpeg", zlib.crc32(f"p/{i}".encode()) % 10000, size=224)
    rows = {}
    for prec in ("fp32", "half", "bf16"):
        cf.main(["--dataset", "toy", "--weights", "r0", "--batch_size", "4", "--num_workers", "0", "--save_npy", "--precision", prec])
        (f,) = glob.glob("datasets/toy/0_x/*.npy")
        rows[prec] = np.load(f)
        os.remove(f)
    assert rows["fp32"].shape == (10, 512)
    dh, db = np.abs(rows["half"] - rows["fp32"]).max(), np.abs(rows["bf16"] - rows["fp32"]).max()
    assert 1e-6 < dh < 5e-3 and 1e-4 < db < 5e-2, (dh, db)


def test_compute_feats_background_filter_matches_pil_decisions(workdir):
    """compute_feats.py --bg_threshold (new, default off): the tilers' FIND_EDGES criterion (deepzoom_tiler.py:56-61)
    applied to the decoded tiles before embedding; the decisions equal PIL's own, the kept rows equal the oracle's."""
    _check_bg_filter(64, 1e-4)


@pytest.mark.gpu
def test_compute_feats_background_filter_on_gpu(tmp_path, monkeypatch):
    """The same on the GPU: dsmil_tile_stats (exact integer band sums) decides, the native trunk embeds the kept tiles."""
    monkeypatch.chdir(tmp_path)
    _check_bg_filter(224, 1e-4)


def test_train_progress_line_reports_every_bag_in_order(capsys):
    """training.train writes train_tcga.py:75's progress line for EVERY bag, in order, with that bag's loss — through
    training.LossReadback (each loss is read one step late, so a step never stalls on the host read; CPU: same code path with
    plain floats) — and returns the mean of exactly those losses."""
    import argparse
    import re
    import numpy as np
    import torch
    import dsmil as mil
    from dsmil_wsi_amd import training as T
    rng = np.random.default_rng(3)
    bags = [torch.from_numpy(np.concatenate([rng.standard_normal((20 + b, 16)).astype(np.float32),
                                             np.full((20 + b, 1), b % 2, np.float32)], 1)) for b in range(5)]
    args = argparse.Namespace(feats_size=16, num_classes=1, dropout_patch=0.0, dropout_node=0.0, non_linearity=1,
                              lr=1e-3, weight_decay=1e-4, num_epochs=1, average=False)
    torch.manual_seed(0)
    np.random.seed(0)
    net, crit, opt, sched = T.init_model(args, mil, torch.device("cpu"))
    mean_loss = T.train(args, bags, net, crit, opt, log=True)
    out = capsys.readouterr().out
    lines = re.findall(r"Training bag \[(\d+)/5\] bag loss: ([0-9.]+)", out)
    assert [int(i) for i, _ in lines] == [0, 1, 2, 3, 4], out
    assert abs(sum(float(v) for _, v in lines) / 5 - mean_loss) < 1e-3
    rb = T.LossReadback("cpu")
    assert rb.push(torch.tensor(1.5)) is None and rb.push(torch.tensor(2.5)) == 1.5 and rb.flush() == 2.5


@pytest.mark.parametrize("C", [1, 2])
def test_test_loop_matches_a_per_bag_host_loop(C, capsys):
    """training.test keeps losses / labels / predictions on the device until the loop ends; result and progress lines equal
    the reference's per-bag host reads (train_tcga.py:85-132 written out plainly here)."""
    import argparse
    import re
    import numpy as np
    import torch
    import dsmil as mil
    from dsmil_wsi_amd import training as T
    rng = np.random.default_rng(7)
    bags = []
    for b in range(6):
        X = rng.standard_normal((15 + 3 * b, 16)).astype(np.float32)
        lab = np.zeros((X.shape[0], C), np.float32)
        lab[:, b % C] = 1.0 if C > 1 else float(b % 2)
        bags.append(torch.from_numpy(np.concatenate([X, lab], 1)))
    args = argparse.Namespace(feats_size=16, num_classes=C, dropout_patch=0.0, dropout_node=0.0, non_linearity=1,
                              lr=1e-3, weight_decay=1e-4, num_epochs=1, average=True)
    torch.manual_seed(1)
    net, crit, opt, sched = T.init_model(args, mil, torch.device("cpu"))
    loss, score, aucs, th = T.test(args, bags, net, crit, log=True)
    out = capsys.readouterr().out
    lines = re.findall(r"Testing bag \[(\d+)/6\] bag loss: ([0-9.]+)", out)
    assert [int(i) for i, _ in lines] == list(range(6))
    # the same loop with the reference's host reads
    net.eval()
    cache = T.BagCache(torch.device("cpu"))
    tot, labels, preds = 0.0, [], []
    with torch.no_grad():
        for item in bags:
            x, y = cache.get(item, 16)
            l, bp, mp = T.bag_loss(net, crit, x, y, None)
            tot += l.item()
            labels.append(y.squeeze().numpy().astype(int))
            preds.append((torch.sigmoid(mp) + torch.sigmoid(bp)).squeeze().numpy())
    assert abs(tot / 6 - loss) < 1e-6
    assert abs(sum(float(v) for _, v in lines) / 6 - loss) < 1e-3
    ref_auc, _, ref_th = T.multi_label_roc(np.array(labels), np.array(preds), C, log=False)
    assert np.allclose(aucs, ref_auc) and np.allclose(th, ref_th)
