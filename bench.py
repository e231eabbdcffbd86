#!/usr/bin/env python3
"""bench.py — throughput of the two DSMIL hot paths on MI355X: the aggregator (BASELINE.json
configs[1], the headline `value`) and the ResNet-18-IN patch embedder (`embedder` sub-object).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (dsmil_agg_forward: instance logits + critical-instance
arg-max, query MLP on f32 MFMA, attention softmax over instances, weighted value sum, Conv1d bag
head) over one batch of --bags synthetic 10 000 x 512 fp32 bags that are already resident in
HBM.  The batch (default 64 distinct bags = 1.31 GB) is larger than the 256 MiB Infinity Cache,
so every step streams its features from HBM.  Bags are independent units: with N ranks each rank
owns its own --bags bags (weak scaling), there is no data-path collective.

The embedder leg (same run, reported under "embedder") times IClassifier over --patches synthetic
224x224 patches per rank per step (ResNet-18 + InstanceNorm on f32 MFMA, then Linear(512,C));
with N ranks each rank embeds its own shard of the slide and ONE RCCL all-gather of the
[--patches, 512] feature rows follows inside the timed step.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_query_attend), timed
live with HIP events on its launch stream inside the library; `cpu_baseline` is the numpy oracle
(oracle/agg_oracle.py, a port of the reference arithmetic) on this box's host cores over a
bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md:42: ~2.5 PF dense bf16 MFMA (no sparsity)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: exact-f32 MFMA (no xf32 on gfx950)
PEAK_HBM_GBS = 8000.0
Q = 128


def flops_per_bag(N, K, C):
    """SURVEY.md §8(d) / BASELINE.md §2 algorithmic FLOPs of one bag."""
    return 2 * N * K * C + 2 * N * K * Q + 2 * N * Q * Q + 2 * C * (K * Q + Q * Q) + 2 * N * Q * C + 2 * N * K * C + 2 * C * C * K


def attend_flops_per_bag(N, K, C):
    """What the dominant kernel itself does per bag: query MLP, scores, weighted value sum."""
    return 2 * N * K * Q + 2 * N * Q * Q + 2 * N * Q * C + 2 * N * K * C


def bytes_per_bag(N, K, C, s=4):
    return N * K * s + (K * Q + Q + Q * Q + Q + C * K + C + C * C * K + C) * s + (2 * N * C + C * K + C) * 4


def cpu_baseline(weights, N, K, C, budget_s):
    """The oracle (a numpy port of dsmil.py) on the host cores, bounded to ~budget_s seconds."""
    import agg_oracle as orc
    from inputs import make_bag
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    bags = [make_bag(50 + i, N, K) for i in range(4)]
    for b in bags[:2]:
        orc.milnet_forward(b, weights)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        orc.milnet_forward(bags[n % len(bags)], weights)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 2000:
            break
    return {"value": round(n / el, 2), "unit": "bags/s", "cores": int(threads), "kind": "port",
            "sample": f"{n} forwards of a {N}x{K} fp32 bag (C={C}) by oracle/agg_oracle.py (numpy/BLAS) in {el:.1f}s"}


def _pmc(name, key):
    """HBM bytes from the committed PMC summary (profiles/): (2*FETCH_SIZE + WRITE_SIZE), or None."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(path)).get(key)
    except Exception:
        return None


FLOPS_PER_PATCH = 3627122688          # 2 x 1 813 561 344 MAC, 20 convs
STEM_FLOPS_PER_PATCH = 2 * 12544 * 64 * 147


def embedder_cpu_baseline(budget_s):
    """oracle/resnet_oracle.py (torch CPU ops, fp32) on the host cores, bounded."""
    import resnet_oracle as ro
    from inputs import make_patches
    w = ro.make_weights(seed=11)
    x = torch.from_numpy(make_patches(7, 8))
    threads = torch.get_num_threads()
    with torch.no_grad():
        ro.resnet18_in_features(x[:2], w)
        n, t0 = 0, time.perf_counter()
        while True:
            ro.resnet18_in_features(x, w)
            n += x.shape[0]
            el = time.perf_counter() - t0
            if el >= budget_s:
                break
    return {"value": round(n / el, 2), "unit": "patches/s", "cores": int(threads), "kind": "port",
            "sample": f"{n} patches (batches of 8, 224x224) through oracle/resnet_oracle.py (torch CPU fp32) in {el:.1f}s"}


def embedder_leg(args, dev, rank, world, dist, L):
    import torch.nn as nn
    import dsmil
    import resnet_oracle as ro
    from dsmil_wsi_amd.resnet import resnet18
    from conftest import load_weights
    res = resnet18(pretrained=False, norm_layer=nn.InstanceNorm2d)
    for p in res.parameters():
        p.requires_grad = False
    res.fc = nn.Identity()
    res.load_state_dict(ro.make_weights(seed=11), strict=True)
    wt = load_weights("tcga")
    ic = dsmil.IClassifier(res, 512, output_class=2)
    with torch.no_grad():
        ic.fc.weight.copy_(torch.from_numpy(wt["fc_w"]))
        ic.fc.bias.copy_(torch.from_numpy(wt["fc_b"]))
    ic = ic.to(dev).eval()
    Bp = args.patches
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    x = torch.rand((Bp, 3, 224, 224), generator=g, device=dev, dtype=torch.float32)
    gathered = torch.empty((world * Bp, 512), device=dev) if world > 1 else None

    def step():
        with torch.no_grad():
            feats, c = ic(x)
        if world > 1:
            dist.all_gather_into_tensor(gathered, feats)
        return feats

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        feats = step()
    fence()
    L.dsmil_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        feats = step()
    fence()
    dt = time.perf_counter() - t0
    tot_ms, launches = ctypes.c_double(0), ctypes.c_int64(0)
    L.dsmil_profile_collect(1, ctypes.byref(tot_ms), ctypes.byref(launches))
    L.dsmil_profile_enable(0)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(feats).all()
    value = world * Bp * args.steps / dt
    conv_flops = (FLOPS_PER_PATCH - STEM_FLOPS_PER_PATCH) * Bp * args.steps
    ach = conv_flops / (tot_ms.value * 1e-3) / 1e12 if tot_ms.value > 0 else None
    return {"metric": "patches/sec embedded (ResNet-18-IN, 224x224, bs=%d)" % Bp, "value": round(value, 1),
            "unit": "patches/s", "ms_per_step": round(dt / args.steps * 1e3, 3), "dtype": "f32",
            "config": {"workload": f"IClassifier(ResNet-18 InstanceNorm, fc=Identity)+Linear(512,2), {Bp} synthetic "
                                   f"224x224 patches per GPU per step, kaiming(seed 11) weights",
                       "collective": "all_gather_into_tensor([%d,512] f32) per step" % Bp if world > 1 else "none"},
            "roofline": {"kernel": "k_conv_wino_s3 (13 Winograd convs, bf16 MFMA over exact 3-plane cuts) + k_conv "
                                   "(6 direct convs, f32 MFMA) per forward", "bound": "mfma",
                         "achieved": round(ach, 2) if ach else None, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4) if ach else None,
                         "traffic": _pmc("pmc_k_conv.json", "hbm_bytes_per_forward"),
                         "kernel_ms_total": round(tot_ms.value, 3), "launches": int(launches.value),
                         "alg_flops_total": conv_flops,
                         "whole_path_frac_of_roofline": round(value / world / (PEAK_F32_MFMA_TFLOPS * 1e12 / FLOPS_PER_PATCH), 4)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bags", type=int, default=64, help="bags per rank per step")
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--feats", type=int, default=512)
    ap.add_argument("--weights", default="c16", choices=["c16", "tcga"],
                    help="c16: Camelyon16 aggregator (C=1, BASELINE configs[1]); tcga: C=2")
    ap.add_argument("--patches", type=int, default=256, help="patches per rank per embedder step")
    ap.add_argument("--workload", default="both", choices=["both", "aggregator", "embedder"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="aggregator leg: f32 (BASELINE configs[1], the headline) or bf16 storage (configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-bag", action="store_true",
                    help="skip the single-bag latency probe (profiling runs: keeps per-kernel averages clean)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import dsmil  # noqa: F401
    import dsmil_wsi_amd._native as nat
    import dsmil_wsi_amd.ops as ops
    from conftest import load_weights

    L = nat.lib()
    run_agg = args.workload in ("both", "aggregator")
    wnp = load_weights(args.weights)
    N, K, nb = args.rows, args.feats, args.bags
    C = wnp["fc_w"].shape[0]
    dt, tot_ms, launches = 1.0, ctypes.c_double(0), ctypes.c_int64(0)
    single_ms = None
    if run_agg:
        w = {k: torch.from_numpy(v).to(dev) for k, v in wnp.items()}
        if K != wnp["fc_w"].shape[1]:
            raise SystemExit("--feats must match the weight file (512)")
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        feats = torch.randn((nb * N, K), generator=g, device=dev, dtype=torch.float32)
        if args.dtype == "bf16":
            feats = feats.to(torch.bfloat16)
        lengths = [N] * nb
        offsets = ops.offsets_tensor(lengths, dev)

        def step():
            return ops.agg_forward(feats, lengths, w, offsets=offsets)

        def fence():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(max(1, args.warmup)):
            out = step()
        fence()
        L.dsmil_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        fence()
        dt = time.perf_counter() - t0
        L.dsmil_profile_collect(0, ctypes.byref(tot_ms), ctypes.byref(launches))
        L.dsmil_profile_enable(0)
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # single-bag latency (SURVEY 8d config 2 asks for it beside the batched rate): one
        # MILNet.forward-sized call per iteration, outside the timed region, rank 0's number is reported
        if not args.no_single_bag:
            one = feats[:N]
            for _ in range(5):
                ops.agg_forward(one, [N], w)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(50):
                ops.agg_forward(one, [N], w)
            torch.cuda.synchronize()
            single_ms = (time.perf_counter() - t1) / 50 * 1e3
        # sanity: outputs finite, attention sums to 1 per bag (cheap, outside the timed region)
        A = out[2]
        s = A.view(nb, N, C).sum(1)
        if not os.environ.get("DSMIL_EXPT"):
            assert torch.isfinite(out[1]).all() and torch.allclose(s, torch.ones_like(s), atol=1e-4)
        del feats, out, A, s
        torch.cuda.empty_cache()

    emb = None
    if args.workload in ("both", "embedder"):
        emb = embedder_leg(args, dev, rank, world, dist, L)

    if rank == 0:
        bags_total = world * nb * args.steps
        value = bags_total / dt
        kern_ms = tot_ms.value / max(1, launches.value)
        fl = attend_flops_per_bag(N, K, C) * nb
        achieved = fl / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None
        traffic = _pmc("pmc_k_query_attend.json", "hbm_bytes_per_launch") if (nb, N, K) == (64, 10000, 512) else None
        form = int(L.dsmil_agg_mlp_form())
        bf16 = args.dtype == "bf16"
        line = {
            "metric": "bags/sec aggregated (10kx512)", "value": round(value, 1), "unit": "bags/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"DSMIL aggregator forward (FCLayer+BClassifier), {args.weights} weights C={C}, "
                                   f"{nb} bags x {N} x {K} {'bf16 storage, f32 accumulate' if bf16 else 'fp32'} per GPU per step, HBM-resident",
                       "bags_per_step_per_gpu": nb, "rows": N, "feats": K, "classes": C,
                       "tile_rows": int(L.dsmil_agg_tile_rows(nb, nb * N)), "parallelism": f"bag-sharded x{world}",
                       "single_bag_forward_ms": round(single_ms, 4) if single_ms is not None else None},
            "roofline": {"kernel": "k_query_attend" + ("_split" if form else ""), "bound": "mfma",
                         "achieved": round(achieved, 2) if achieved else None,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4) if achieved else None,
                         "mfma_form": {0: "v_mfma_f32_32x32x2_f32", 9: "bf16 MFMA, exact 3-plane cut, 9 plane products",
                                       6: "bf16 MFMA, exact 3-plane cut, 6 plane products"}[form],
                         # the pipe the kernel actually runs on: bf16 dense peak / plane products per MAC
                         "peak_executed_form": round(PEAK_BF16_MFMA_TFLOPS / form, 1) if form else PEAK_F32_MFMA_TFLOPS,
                         "frac_executed_form": (round(achieved / (PEAK_BF16_MFMA_TFLOPS / form if form else PEAK_F32_MFMA_TFLOPS), 4)
                                                if achieved else None),
                         "traffic": traffic, "kernel_ms": round(kern_ms, 4), "launches": int(launches.value),
                         "alg_flops_per_launch": fl,
                         "whole_path_frac_of_roofline": round(
                             value / world / (1.0 / max(flops_per_bag(N, K, C) / (PEAK_F32_MFMA_TFLOPS * 1e12),
                                                             bytes_per_bag(N, K, C) / (PEAK_HBM_GBS * 1e9))), 4)},
        }
        if bf16:
            # bf16 storage: the MLP runs on bf16 MFMA (0.7 us/bag at 2.5 PF) and the feature stream
            # (10.5 MB/bag) binds -> HBM roofline for the same dominant kernel
            by = bytes_per_bag(N, K, C, s=2) * nb
            gbs = by / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
            line["roofline"] = {"kernel": "k_query_attend_bf16", "bound": "hbm", "achieved": round(gbs, 1) if gbs else None,
                                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4) if gbs else None,
                                "traffic": None, "kernel_ms": round(kern_ms, 4), "launches": int(launches.value),
                                "alg_bytes_per_launch": by,
                                "whole_path_frac_of_roofline": round(
                                    value / world / (1.0 / max(flops_per_bag(N, K, C) / (PEAK_BF16_MFMA_TFLOPS * 1e12),
                                                                    bytes_per_bag(N, K, C, s=2) / (PEAK_HBM_GBS * 1e9))), 4)}
        if emb is not None:
            line["embedder"] = emb
        if not run_agg:   # embedder-only run (profiling): promote the embedder leg to the top level
            line = dict(emb, n_gpus=world, steps=args.steps, warmup=args.warmup, higher_is_better=True,
                        scaling="weak", vs_baseline=None, data="synthetic")
            emb = None
        if not args.no_cpu_baseline and run_agg and world == 1 and not bf16:   # CPU baselines: rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(wnp, N, K, C, args.cpu_seconds)
            if emb is not None:
                emb["cpu_baseline"] = embedder_cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
