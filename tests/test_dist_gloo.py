"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: contiguous patch sharding + ONE
all-gather of feature rows reproduces the unsharded bag bit for bit; bag round-robin covers every
bag exactly once."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import torch.nn as nn
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dsmil
    import resnet_oracle as ro
    from dsmil_wsi_amd import dist as dd
    from dsmil_wsi_amd.resnet import resnet18
    from inputs import make_patches
    from util import build_net
    torch.set_num_threads(2)
    res = resnet18(norm_layer=nn.InstanceNorm2d)
    res.fc = nn.Identity()
    res.load_state_dict(ro.make_weights(seed=21), strict=True)
    ic = dsmil.IClassifier(res, 512, output_class=2).eval()
    torch.manual_seed(0)
    with torch.no_grad():
        ic.fc.weight.normal_(0, 0.1)
        ic.fc.bias.zero_()
    x = torch.from_numpy(make_patches(31, n_total, 64, 64))       # the slide's ordered patches

    def embed(lo, hi):
        with torch.no_grad():
            return ic(x[lo:hi])[0]
    feats = dd.embed_rows_sharded(embed, n_total)
    with torch.no_grad():
        full = ic(x)[0]
    same = bool(torch.equal(feats, full))
    # aggregate the gathered bag on every rank; all ranks must agree
    net = build_net("tcga")
    with torch.no_grad():
        pred = net.b_classifier(feats, net.i_classifier(feats)[1])[0]
    gathered = [torch.empty_like(pred) for _ in range(world)]
    dist.all_gather(gathered, pred)
    agree = all(torch.equal(g, pred) for g in gathered)
    bags = dd.shard_bags(7, rank, world)
    q.put((rank, same, agree, tuple(feats.shape), bags, dd.shard_range(n_total, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [5, 8])
def test_sharded_embed_all_gather_equals_unsharded(n_total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    for rank, same, agree, shape, bags, rng in res:
        assert same, "gathered feature rows differ from the unsharded embedding"
        assert agree, "ranks disagree on the aggregated bag"
        assert shape == (n_total, 512)
    assert sorted(res[0][4] + res[1][4]) == list(range(7))
    assert res[0][5][1] == res[1][5][0] and res[0][5][0] == 0 and res[1][5][1] == n_total


def test_shard_range_partitions_exactly():
    import dsmil  # noqa: F401  (registers the dsmil_wsi_amd package)
    from dsmil_wsi_amd.dist import shard_range
    for n in (0, 1, 7, 10000, 100003):
        for w in (1, 2, 3, 8):
            ranges = [shard_range(n, r, w) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def _shard_worker(rank, world, port, n_total, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dsmil  # noqa: F401
    from dsmil_wsi_amd import dist as dd
    from inputs import make_bag
    from util import build_net
    torch.set_num_threads(2)
    net = build_net("tcga")
    x = torch.from_numpy(make_bag(77, n_total, 512))
    lo, hi = dd.shard_range(n_total, rank, world)
    classes, pred, A, B, idx = dd.sharded_bag_forward(net, x[lo:hi], lo)
    with torch.no_grad():
        ref = net(x)
    ok = (torch.allclose(classes, ref[0][lo:hi], atol=1e-6) and torch.allclose(pred, ref[1], atol=1e-5)
          and torch.allclose(A, ref[2][lo:hi], atol=1e-7, rtol=1e-4) and torch.allclose(B, ref[3], atol=1e-5)
          and torch.equal(idx, torch.argmax(ref[0], dim=0)))
    q.put((rank, bool(ok), float((A.sum(0)).sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 301), (3, 1000), (3, 2)])   # (3, 2): one rank holds no rows
def test_instance_sharded_bag_equals_unsharded(world, n_total):
    """ONE bag spread over the ranks by rows: two exchanges of C*(2+K) floats reproduce MILNet.forward
    of the whole bag (local slices of classes/A, identical pred/B/idx on every rank)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert abs(sum(s for _, _, s in res) - 2.0) < 1e-4   # the local attention slices sum to 1 per class


def test_global_row_index_survives_the_float_message_beyond_2_pow_24():
    """The critical row's GLOBAL index travels inside the float message as raw int64 bits (two fp32 lanes): a row
    offset above 2^24 (a float32 VALUE would round it) comes back exactly.  Single process, injected gather."""
    import dsmil  # noqa: F401
    from dsmil_wsi_amd import dist as dd
    from inputs import make_bag
    from util import build_net
    net = build_net("tcga")
    x = torch.from_numpy(make_bag(3, 50, 512))
    off = (1 << 24) + 12345   # not representable in float32
    with torch.no_grad():
        ref = net(x)
    out = dd.sharded_bag_forward(net, x, off, gather=lambda t, group=None: [t])
    assert torch.equal(out[4], torch.argmax(ref[0], dim=0) + off)
    assert torch.allclose(out[1], ref[1], atol=1e-5)


def _ms_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import torch.nn as nn
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dsmil
    import resnet_oracle as ro
    from dsmil_wsi_amd import pipeline as pl
    from dsmil_wsi_amd.resnet import resnet18
    torch.set_num_threads(2)

    def emb(seed):
        res = resnet18(norm_layer=nn.InstanceNorm2d)
        res.fc = nn.Identity()
        res.load_state_dict(ro.make_weights(seed=seed), strict=True)
        return dsmil.IClassifier(res, 512, output_class=2).eval()
    e_lo, e_hi = emb(81), emb(82)
    T = 64   # 64 -> 2x2 in layer4 (torch's instance_norm refuses a 1x1 map)
    g = torch.Generator().manual_seed(6)
    wsi = torch.randint(0, 256, (1 * T * 4, 3 * T * 4, 3), generator=g, dtype=torch.uint8)   # 3 low / 48 high tiles
    tree, pos = pl.multiscale_bag(wsi, e_lo, e_hi, "cat", tile=T, batch_size=8)              # sharded 2 + 1 low tiles
    low, high, parent, pos_all = pl.pyramid_tiles(wsi, T)
    with torch.no_grad():
        f_lo, _ = e_lo(low)
        f_hi, _ = e_hi(high)
    full = torch.cat([f_hi, f_lo[parent]], dim=1)
    q.put((rank, bool(torch.allclose(tree, full, atol=5e-5, rtol=1e-4)), bool(torch.equal(pos, pos_all)), tuple(tree.shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_multiscale_bag_sharded_by_low_tile_equals_unsharded():
    """configs[4] over 2 ranks: low tiles (with their 16 children) are cut contiguously over the ranks, the
    [high || low] concatenation happens locally, ONE all-gather of tree rows: the unsharded bag (up to the batch-
    composition effects of torch's CPU convolutions; the native kernels are batch-independent, test_resnet_gpu.py)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ms_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, same_pos, shape in res:
        assert same and same_pos and shape == (48, 1024)


def _packed_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dsmil  # noqa: F401
    from dsmil_wsi_amd import dist as dd
    sizes = [5, 0, 3][:world]                      # ragged shards, one of them empty
    g = torch.Generator().manual_seed(3)
    feats_all = torch.randn(sum(sizes), 16, generator=g)
    cls_all = torch.randn(sum(sizes), 2, generator=g)
    pos_all = torch.randint(-2**40, 2**40, (sum(sizes), 2), generator=g, dtype=torch.int64)   # beyond float32 / int32 range
    lo = sum(sizes[:rank])
    sl = slice(lo, lo + sizes[rank])
    f, c, p_ = dd.all_gather_packed([feats_all[sl], cls_all[sl], pos_all[sl]], sizes)
    ok = torch.equal(f, feats_all) and torch.equal(c, cls_all) and torch.equal(p_, pos_all) and p_.dtype == torch.int64
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_one_packed_collective_carries_rows_logits_and_int64_positions():
    """dist.all_gather_packed: feature rows, instance logits and int64 grid positions of a slide in ONE collective (the
    int64 columns travel bit-cast in float lanes), ragged shards incl. an empty rank, world 3."""
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_packed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True), (2, True)]
