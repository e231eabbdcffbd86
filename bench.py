#!/usr/bin/env python3
"""bench.py — throughput of the DSMIL aggregator hot path on MI355X (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (dsmil_agg_forward: instance logits + critical-instance
arg-max, query MLP on f32 MFMA, attention softmax over instances, weighted value sum, Conv1d bag
head) over one batch of --bags synthetic 10 000 x 512 fp32 bags that are already resident in
HBM.  The batch (default 64 distinct bags = 1.31 GB) is larger than the 256 MiB Infinity Cache,
so every step streams its features from HBM.  Bags are independent units: with N ranks each rank
owns its own --bags bags (weak scaling), there is no data-path collective.

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
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    wnp = load_weights(args.weights)
    w = {k: torch.from_numpy(v).to(dev) for k, v in wnp.items()}
    N, K, nb = args.rows, args.feats, args.bags
    C = wnp["fc_w"].shape[0]
    if K != wnp["fc_w"].shape[1]:
        raise SystemExit("--feats must match the weight file (512)")
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    feats = torch.randn((nb * N, K), generator=g, device=dev, dtype=torch.float32)
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
    L = nat.lib()
    L.dsmil_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    tot_ms, launches = ctypes.c_double(0), ctypes.c_int64(0)
    L.dsmil_profile_collect(ctypes.byref(tot_ms), ctypes.byref(launches))
    L.dsmil_profile_enable(0)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # sanity: outputs are finite and attention sums to 1 per bag (cheap, outside the timed region)
    A = out[2]
    s = A.view(nb, N, C).sum(1)
    assert torch.isfinite(out[1]).all() and torch.allclose(s, torch.ones_like(s), atol=1e-4)

    if rank == 0:
        bags_total = world * nb * args.steps
        value = bags_total / dt
        kern_ms = tot_ms.value / max(1, launches.value)
        fl = attend_flops_per_bag(N, K, C) * nb
        achieved = fl / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_k_query_attend.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "bags/sec aggregated (10kx512)", "value": round(value, 1), "unit": "bags/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"DSMIL aggregator forward (FCLayer+BClassifier), {args.weights} weights C={C}, "
                                   f"{nb} bags x {N} x {K} fp32 per GPU per step, HBM-resident",
                       "bags_per_step_per_gpu": nb, "rows": N, "feats": K, "classes": C,
                       "tile_rows": int(L.dsmil_agg_tile_rows(nb, nb * N)), "parallelism": f"bag-sharded x{world}"},
            "roofline": {"kernel": "k_query_attend", "bound": "mfma", "achieved": round(achieved, 2) if achieved else None,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4) if achieved else None,
                         "traffic": traffic, "kernel_ms": round(kern_ms, 4), "launches": int(launches.value),
                         "alg_flops_per_launch": fl,
                         "whole_path_frac_of_roofline": round(
                             value / world / (1.0 / max(flops_per_bag(N, K, C) / (PEAK_F32_MFMA_TFLOPS * 1e12),
                                                             bytes_per_bag(N, K, C) / (PEAK_HBM_GBS * 1e9))), 4)},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wnp, N, K, C, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
