#!/usr/bin/env python3
"""bench.py — throughput of the DSMIL hot paths on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: re-launches itself as N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Headline (`value`): bags/s of the aggregator on BASELINE.json configs[1] — Camelyon16 weights (C=1), fp32, bags of
10 000 x 512.  One "step" = `passes_per_step` passes of the hot path (dsmil_agg_forward: instance logits +
critical-instance arg-max, query MLP on MFMA, attention softmax over instances, weighted value sum, Conv1d bag head)
over one batch of --bags synthetic bags already resident in HBM; the batch (64 bags = 1.31 GB) is larger than the
256 MiB Infinity Cache, so every pass streams its features from HBM.  `passes_per_step` is chosen after the warm-up so
that the timed region (EXACTLY --steps steps between barrier + synchronize on both sides, max over ranks) lasts at
least --min-seconds (1 s): a 22 ms region cannot be corroborated from outside.  Bags are independent units: with N
ranks each rank owns its own --bags bags (weak scaling), no data-path collective.

Sub-objects of the same JSON line (each measured the same way):
  aggregator_bf16  BASELINE configs[2]: TCGA weights (C=2), bf16 feature storage, f32 accumulate
  embedder         configs[3]: IClassifier(ResNet-18-IN) over --patches synthetic 224x224 patches per rank per step;
                   with N ranks ONE RCCL all-gather of the [--patches,512] rows follows inside the step (weak)
  train_c1/_c2     SURVEY §8 a9: one train_tcga.py:60-75 step (forward, two BCEs, backward, Adam, loss.item()) per 10 000 x 512
                   bag, C = 1 (c16 weights) and C = 2 (tcga weights): bags/s and the GPU time of forward+loss / backward / Adam
  slide_h2d        the `slide` leg below starting from PINNED HOST tiles (compute_feats.py:69-75: H2D per batch, overlapped
                   with the other streams' convs; D2H of rows + logits once per slide)
  slide            configs[3] per-slide strong scaling: a slide of --slide-patches uint8 tiles
                   cut contiguously over the ranks, embedded in batches, ONE all-gather of feature rows, aggregated
  slide_100k       the same with 100 000 tiles (2 slides timed)
  decode           SURVEY §8f N3: 8 192 baseline-JPEG tiles (224x224, quality 70) decoded on the device per launch sequence,
                   beside Pillow (the reference's decoder) on the host's cores
  slide_jpeg       the `slide` leg starting from JPEG files in host memory: device decode on side streams under the embedding
  slide_jpeg_half  the same on the opt-in fp16-activation trunk (the decodes have half the time to hide in)
  feats_csv        the bag's feature file (compute_feats.py:80-82) through dsmil_csv_format_f32, beside pandas (host only)
  e2e              configs[4]: synthetic two-level WSI -> tiles -> two embedders -> [high||low] -> MILNet(1024) ->
                   attention map, sharded by low tile, one all-gather of tree rows

Passes are independent (different batches of bags / patches in a real job), so they are dealt round-robin to --streams
HIP streams (ops.StreamPool, default 3: one workspace per stream): the HBM-bound logits pass of one batch runs under the
MFMA-bound attend kernel of another, and the few-round kernels of one embedder forward fill each other's tails.
`--streams 1` keeps one pass in flight.

`roofline` is for the dominant kernel of the headline leg (k_attend_f3), timed live with HIP events on its
launch stream inside the library.  With several streams a launch's start-to-end interval inside the timed region also
contains the kernels co-running with it, so the kernel's OWN duration is measured in a second region right after the
timed one (same inputs, one pass in flight, 300 passes); the in-region interval is reported beside it.
`cpu_baseline` is the same forward on this box's host cores.  Prints ONE JSON line (rank 0).  The line is compact by default
(--verbose adds the explanatory strings and the committed per-kernel tables); `config.legs` and the trailing `summary` key
repeat every leg's value and roofline fraction so that a truncated stdout tail still carries them.
"""
import argparse
import ctypes
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:     # the `dsmil` shim; inputs and example weights come from the package (dsmil-wsi_amd/synthetic.py),
    sys.path.insert(0, ROOT)  # nothing under tests/ is imported; oracle/ only inside cpu_baseline_embedder

PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA (no sparsity)
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: exact-f32 MFMA (no xf32 on gfx950)
PEAK_HBM_GBS = 8000.0
Q = 128


def flops_per_bag(N, K, C):
    """SURVEY.md §8(d) / BASELINE.md §2 algorithmic FLOPs of one bag."""
    return 2 * N * K * C + 2 * N * K * Q + 2 * N * Q * Q + 2 * C * (K * Q + Q * Q) + 2 * N * Q * C + 2 * N * K * C + 2 * C * C * K


def attend_flops_per_bag(N, K, C):
    """What the dominant kernel itself does per bag: query MLP, scores, weighted value sum."""
    return 2 * N * K * Q + 2 * N * Q * Q + 2 * N * Q * C + 2 * N * K * C


def mlp_flops_per_bag(N, K):
    """The part of it that runs on the matrix pipe (the two GEMMs of the query MLP)."""
    return 2 * N * K * Q + 2 * N * Q * Q


def bytes_per_bag(N, K, C, s=4):
    return N * K * s + (K * Q + Q + Q * Q + Q + C * K + C + C * C * K + C) * s + (2 * N * C + C * K + C) * 4


FLOPS_PER_PATCH = 3627122688          # 2 x 1 813 561 344 MAC, 20 convs
STEM_FLOPS_PER_PATCH = 2 * 12544 * 64 * 147
# direct-form FLOPs per patch by conv class (ResNet-18 @224): 13 stride-1 3x3 convs (Winograd), 3 stride-2 3x3 + 3 1x1
WINO_FLOPS_PER_PATCH = 2 * (4 * 3136 * 64 * 64 * 9 + 3 * 784 * 128 * 128 * 9 + 3 * 196 * 256 * 256 * 9 + 3 * 49 * 512 * 512 * 9)
DIRECT_FLOPS_PER_PATCH = FLOPS_PER_PATCH - STEM_FLOPS_PER_PATCH - WINO_FLOPS_PER_PATCH
# algorithmic HBM traffic of one 224x224 patch through ResNet-18-IN with fp32 NHWC activations: every conv reads its input
# and writes its raw output once, the max-pool and the 8 residual kernels read their operands and write their result once
_A = {56: 56 * 56 * 64 * 4, 28: 28 * 28 * 128 * 4, 14: 14 * 14 * 256 * 4, 7: 7 * 7 * 512 * 4}
ACT_BYTES_PER_PATCH = (3 * 224 * 224 * 4 + 112 * 112 * 64 * 4          # stem
                       + 112 * 112 * 64 * 4 + _A[56]                   # IN + ReLU + max-pool
                       + 4 * 2 * _A[56] + 2 * 3 * _A[56]               # layer 1: 4 convs, 2 residual kernels
                       + sum((_A[p] + _A[h]) * 2 + 3 * 2 * _A[h] + 2 * 3 * _A[h]   # layers 2-4: strided conv + downsample,
                             for p, h in ((56, 28), (28, 14), (14, 7))))           # 3 stride-1 convs, 2 residual kernels


KERNEL_TABLE = "r06_kernel_table.json"   # the latest committed per-kernel table (tools/kernel_table.py)


def _kernel_table(leg):
    """Per-kernel rows of the committed profile of this leg (profiles/r03_kernel_table.json, tools/kernel_table.py): name,
    launches per pass, average ms, counter HBM bytes, algorithmic bytes / executed MFMA FLOPs, fraction of the bounding peak —
    measured under rocprofv3 with --streams 1 on the profiling box, NOT in this run (live numbers: `roofline`)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", KERNEL_TABLE)))
        return {"source": t["source"], "rows": t[leg]} if leg in t else None
    except Exception:
        return None


def _pmc(name, key):
    """HBM bytes from the committed PMC summary (profiles/): (2*FETCH_SIZE + WRITE_SIZE), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name))).get(key)
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no torchrun around it becomes N ranks
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_self_launch(args):
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


class Ctx:
    """Per-process bench context: device, ranks, the native library."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
        if self.local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {self.rank}: no GPU {self.local_rank} on this node ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.dist = None
        self.collectives = self.world > 1 or args.force_collective   # do the legs run their RCCL collective?
        if self.collectives:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:      # --force-collective on a bare `python bench.py`: a one-rank group
                os.environ.update({"MASTER_PORT": str(_free_port()), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
            dist.init_process_group("nccl", device_id=self.dev)
            assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
            self.dist = dist
        import dsmil  # noqa: F401
        import dsmil_wsi_amd._native as nat
        if args.force_collective:
            import dsmil_wsi_amd.dist as dd
            dd.force_collective(True)
        import dsmil_wsi_amd.ops as ops
        self.L = nat.lib()
        # independent passes are dealt round-robin to --streams HIP streams (ops.StreamPool, one workspace per stream)
        self.pool = ops.StreamPool(args.streams, self.dev) if args.streams > 1 else None

    def run(self, fn, *a, **k):
        return self.pool.run(fn, *a, **k) if self.pool is not None else fn(*a, **k)

    def kernel_alone(self, fn, channel, passes=6):
        """Average per-launch time of the channel's kernel with ONE pass in flight (no co-running stream)."""
        torch = self.torch
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        self.L.dsmil_profile_enable(1)
        for _ in range(passes):
            fn()
        torch.cuda.synchronize()
        tot_ms, launches = ctypes.c_double(0), ctypes.c_int64(0)
        self.L.dsmil_profile_collect(channel, ctypes.byref(tot_ms), ctypes.byref(launches))
        self.L.dsmil_profile_enable(0)
        return tot_ms.value / max(1, launches.value), tot_ms.value / passes

    def fence(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], device=self.dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, step, steps, warmup, min_seconds, channel=None, fixed_passes=None):
        """Warm up, pick passes_per_step so the region lasts >= min_seconds, then time EXACTLY `steps` steps
        (each = passes_per_step calls of `step`) between fences; returns (seconds [max over ranks], passes_per_step,
        kernel_ms_total, launches) — the last two from the library's HIP-event channel when given."""
        torch = self.torch
        for _ in range(max(1, warmup)):
            step()
        self.fence()
        if fixed_passes is not None:
            inner = fixed_passes
        else:
            t0 = time.perf_counter()
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            t_pass = self.max_over_ranks((time.perf_counter() - t0) / 4)
            # + 25 %: the probe runs cold next to the pipelined region, and the floor is a floor
            inner = max(1, int(math.ceil(1.25 * min_seconds / max(1e-9, steps * t_pass))))
            if channel is not None:   # the library's event ring holds 4096 launches per channel
                inner = max(1, min(inner, 4000 // max(1, steps * self._launches_per_pass(step, channel))))
        self.fence()
        if channel is not None:
            self.L.dsmil_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(steps):
            for _ in range(inner):
                step()
        self.fence()
        dt = self.max_over_ranks(time.perf_counter() - t0)
        tot_ms, launches = ctypes.c_double(0), ctypes.c_int64(0)
        if channel is not None:
            self.L.dsmil_profile_collect(channel, ctypes.byref(tot_ms), ctypes.byref(launches))
            self.L.dsmil_profile_enable(0)
        return dt, inner, tot_ms.value, int(launches.value)

    def _launches_per_pass(self, step, channel):
        self.L.dsmil_profile_enable(1)
        step()
        self.torch.cuda.synchronize()
        tot_ms, launches = ctypes.c_double(0), ctypes.c_int64(0)
        self.L.dsmil_profile_collect(channel, ctypes.byref(tot_ms), ctypes.byref(launches))
        self.L.dsmil_profile_enable(0)
        return max(1, int(launches.value))


# ---------------------------------------------------------------------------------------------------------------
# aggregator legs (configs[1] fp32 headline, configs[2] bf16 storage)
# ---------------------------------------------------------------------------------------------------------------
def aggregator_leg(cx, weights_tag, dtype, single_bag=True):
    torch, args, dev = cx.torch, cx.args, cx.dev
    import dsmil_wsi_amd.ops as ops
    from dsmil_wsi_amd.synthetic import load_weights
    wnp = load_weights(weights_tag)
    N, K, nb = args.rows, args.feats, args.bags
    C = wnp["fc_w"].shape[0]
    if K != wnp["fc_w"].shape[1]:
        raise SystemExit("--feats must match the weight file (512)")
    w = {k: torch.from_numpy(v).to(dev) for k, v in wnp.items()}
    g = torch.Generator(device=dev).manual_seed(1234 + cx.rank)
    bf16 = dtype == "bf16"
    # streams of THIS leg: the bf16 pass wants exactly one other batch's logits / q_max / combine kernels under each
    # persistent attend launch (the co-resident forms of round 6, dsmil_agg_logits_form): two streams; a third batch only
    # queues a second attend launch behind the first (measured: 238 k bags/s on two, 229 k on three, distinct batches)
    n_streams = args.streams_bf16 if (bf16 and args.streams > 1) else args.streams
    if n_streams == args.streams or cx.pool is None:
        pool = cx.pool
    elif n_streams > 1 and n_streams < len(cx.pool.streams):
        # the first n_streams streams of the run's pool, not new ones: every stream a process creates shifts the hardware queue
        # the next one lands on (a later leg's side stream then shares a queue with its conv kernels: slide_jpeg 59 k -> 55 k)
        import copy
        pool = copy.copy(cx.pool)
        pool.streams = cx.pool.streams[:n_streams]
        pool._i = 0
    else:
        pool = ops.StreamPool(n_streams, dev) if n_streams > 1 else None
    # one DISTINCT batch per stream: passes in flight together are different batches of bags in a real job, so no pass
    # may ride another's cache fills (3 x 1.31 GB fp32)
    n_batches = max(1, args.streams)
    batches = []
    for _ in range(n_batches):
        f = torch.randn((nb * N, K), generator=g, device=dev, dtype=torch.float32)
        batches.append(f.to(torch.bfloat16) if bf16 else f)
        del f
    feats = batches[0]
    lengths = [N] * nb
    offsets = ops.offsets_tensor(lengths, dev)
    out = []
    turn = [0]

    def one():
        turn[0] += 1
        return ops.agg_forward(batches[turn[0] % n_batches], lengths, w, offsets=offsets)

    def step():
        out[:] = pool.run(one) if pool is not None else one()

    dt, inner, kern_ms_tot, launches = cx.timed(step, args.steps, args.warmup, args.min_seconds, channel=0)
    kern_alone_ms, _ = cx.kernel_alone(one, 0, passes=300)
    # the same passes with ONE in flight (no stream pool), beside the pooled headline
    value_1s = None
    if pool is not None:
        pool.join()
        def step1():
            out[:] = one()
        dt1, inner1, _, _ = cx.timed(step1, max(2, args.steps // 4), 1, args.min_seconds / 4)
        value_1s = cx.world * nb * inner1 * max(2, args.steps // 4) / dt1
    single_ms = None
    if single_bag:   # one MILNet.forward-sized call per iteration, outside the timed region
        one = feats[:N]
        for _ in range(5):
            ops.agg_forward(one, [N], w)
        torch.cuda.synchronize()
        rounds_ms = []
        for _ in range(5):   # five rounds of 50 calls (host-inclusive: boxes differ in host speed); the MEDIAN round is the figure,
            t1 = time.perf_counter()   # the best round is kept beside it
            for _ in range(50):
                ops.agg_forward(one, [N], w)
            torch.cuda.synchronize()
            rounds_ms.append((time.perf_counter() - t1) / 50 * 1e3)
        single_ms = sorted(rounds_ms)[len(rounds_ms) // 2]
        single_best_ms = min(rounds_ms)
    A = out[2]
    s = A.view(nb, N, C).sum(1)
    if not os.environ.get("DSMIL_EXPT"):   # ablation runs of experiment builds compute garbage on purpose
        assert torch.isfinite(out[1]).all() and torch.allclose(s, torch.ones_like(s), atol=1e-4), "attention does not sum to 1"
    del feats, out[:], A, s
    batches.clear()
    torch.cuda.empty_cache()

    world = cx.world
    value = world * nb * inner * args.steps / dt
    kern_ms = kern_ms_tot / max(1, launches)
    form = int(cx.L.dsmil_agg_mlp_form())
    line = {"metric": "bags/sec aggregated (10kx512)", "value": round(value, 1), "unit": "bags/s",
            "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_pass": round(dt / (args.steps * inner) * 1e3, 4),
            "dtype": dtype,
            "config": {"workload": f"DSMIL aggregator forward (FCLayer+BClassifier), {weights_tag} weights C={C}, "
                                   f"{nb} bags x {N} x {K} {'bf16 storage, f32 accumulate' if bf16 else 'fp32'} per GPU per pass, "
                                   f"HBM-resident", "passes_per_step": inner, "bags_per_pass_per_gpu": nb, "rows": N,
                       "feats": K, "classes": C, "tile_rows": int(cx.L.dsmil_agg_tile_rows(nb, nb * N)),
                       "parallelism": f"bag-sharded x{world}", "timed_region_s": round(dt, 3),
                       "streams": n_streams, "distinct_batches": n_batches,
                       "logits_form": int(cx.L.dsmil_agg_logits_form(-1)),
                       "value_one_stream": round(value_1s, 1) if value_1s else None,
                       "single_bag_forward_ms": round(single_ms, 4) if single_ms is not None else None,
                       "single_bag_forward_ms_is": "median of five rounds of 50 back-to-back forwards of one 10 000-row bag, host-inclusive" if single_ms is not None else None,
                       "single_bag_forward_best_round_ms": round(single_best_ms, 4) if single_ms is not None else None,
                       # what the fp32 summation order of the persistent batch kernels is tied to (a constant since round 6:
                       # results do not depend on the box), and the box itself
                       "device_cus": int(cx.L.dsmil_device_cus()), "persistent_grid": int(cx.L.dsmil_agg_persistent_grid(-1))}}
    # The PASS reads the features twice (k_logits_stream: the instance arg-max of dsmil.py:51-53 must be known before any
    # attention score; then the attend kernel): against the single-read bytes of SURVEY §8(d) a pass cannot exceed 0.5 of the HBM
    # roof.  Both fractions of the whole path, on the HBM roof alone:
    by1 = bytes_per_bag(N, K, C, s=2 if bf16 else 4)
    two_read = {"whole_path_frac_of_hbm_roofline": round(value / world * by1 / (PEAK_HBM_GBS * 1e9), 4),
                "two_read_frac": round(value / world * (by1 + N * K * (2 if bf16 else 4)) / (PEAK_HBM_GBS * 1e9), 4),
                "two_read_frac_is": "whole-path bags/s x (algorithmic bytes + the second read of the features) / 8 TB/s: the pass is k_logits_stream (arg-max over ALL rows of a bag) and then the attend kernel, so every feature byte crosses HBM twice; whole_path_frac_of_hbm_roofline prices the same rate against ONE read and cannot exceed 0.5"}
    if bf16:
        # bf16 storage: the MLP runs on bf16 MFMA (0.7 us/bag at 2.5 PF) and the feature stream (10.5 MB/bag) binds
        by = bytes_per_bag(N, K, C, s=2) * nb
        kern_region_ms, kern_ms = kern_ms, (kern_alone_ms if cx.pool is not None else kern_ms)
        gbs = by / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
        t_roof = max(flops_per_bag(N, K, C) / (PEAK_BF16_MFMA_TFLOPS * 1e12), bytes_per_bag(N, K, C, s=2) / (PEAK_HBM_GBS * 1e9))
        line["roofline"] = {"kernel": "k_attend_bf16_res", "bound": "hbm", "achieved": round(gbs, 1) if gbs else None,
                            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4) if gbs else None,
                            "traffic": _pmc("pmc_k_attend_bf16_res.json", "hbm_bytes_per_launch") if (nb, N, K) == (64, 10000, 512) else None,
                            "kernel_ms": round(kern_ms, 4), "launches": launches, "alg_bytes_per_launch": by,
                            "kernel_ms_in_timed_region": round(kern_region_ms, 4),
                            "kernel_ms_is": "average HIP-event duration of the kernel with ONE pass in flight (300 passes on one stream right after the timed region, same inputs): the kernel's own time. Inside the timed region config.streams passes overlap, so a launch's start-to-end interval (kernel_ms_in_timed_region) also contains the co-running streams' kernels and is not a measure of the kernel",
                            "whole_path_frac_of_roofline": round(value / world * t_roof, 4)}
        line["roofline"].update(two_read)
        return line
    fl = attend_flops_per_bag(N, K, C) * nb
    kern_region_ms, kern_ms = kern_ms, (kern_alone_ms if cx.pool is not None else kern_ms)
    achieved = fl / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None
    # The peak that bounds the kernel AS EXECUTED: forms 6 / 9 run every fp32 MAC as 6 / 9 bf16 plane products on the
    # bf16 matrix pipe, so the bound is (bf16 dense peak) / (plane products per MAC) in algorithmic (fp32) FLOP/s;
    # form 0 runs on the f32 MFMA pipe.  `frac` is therefore the utilisation of the pipe the kernel runs on (<= 1);
    # the fp32-MFMA figure of SURVEY §8(d) is kept beside it (it is not a bound for the split forms).
    peak_exec = PEAK_BF16_MFMA_TFLOPS / form if form else PEAK_F32_MFMA_TFLOPS
    t_roof_f32 = max(flops_per_bag(N, K, C) / (PEAK_F32_MFMA_TFLOPS * 1e12), bytes_per_bag(N, K, C) / (PEAK_HBM_GBS * 1e9))
    t_roof_exec = max(flops_per_bag(N, K, C) / (peak_exec * 1e12), bytes_per_bag(N, K, C) / (PEAK_HBM_GBS * 1e9))
    batch_form = int(cx.L.dsmil_agg_batch_form(-1))
    f2 = (form == 6 and batch_form in (1, 2) and int(cx.L.dsmil_agg_tile_rows(nb, nb * N)) == 128
          and K % 128 == 0 and K <= 512)
    f2_kernel = "k_attend_f3" if (batch_form == 2 and C <= 2) else "k_attend_f2"   # (bench weights: the two-layer query)
    if f2:
        # k_attend_f2 / k_attend_f3 (round 5): every feature byte read once, the query MLP as THREE fp16 plane products per fp32 MAC.  The
        # executed matrix work of a launch (3 x algorithmic FLOPs at the 2.5 PF f16 / bf16 dense rate) is shorter than its
        # algorithmic bytes at 8 TB/s, so HBM is the roof that binds this kernel; the matrix-pipe figure is kept beside it.
        by = bytes_per_bag(N, K, C) * nb
        gbs = by / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
        peak3 = PEAK_BF16_MFMA_TFLOPS / 3
        t_roof_exec = max(flops_per_bag(N, K, C) / (peak3 * 1e12), bytes_per_bag(N, K, C) / (PEAK_HBM_GBS * 1e9))
        line["roofline"] = {
            "kernel": f2_kernel, "bound": "hbm", "achieved": round(gbs, 1) if gbs else None, "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4) if gbs else None,
            "bound_is": "algorithmic bytes / 8 TB/s = %.3f ms per launch > 3 x algorithmic FLOPs / 2.5 PF = %.3f ms (fp16 MFMA over two-plane cuts, three plane products per fp32 MAC)"
                        % (by / (PEAK_HBM_GBS * 1e9) * 1e3, 3 * fl / (PEAK_BF16_MFMA_TFLOPS * 1e12) * 1e3),
            "mfma_form": "fp16 MFMA, two-plane cut of row-scaled operands, 3 plane products",
            "mfma_achieved_tflops": round(achieved, 2) if achieved else None, "mfma_peak_tflops": round(peak3, 1),
            "mfma_frac": round(achieved / peak3, 4) if achieved else None,
            "frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4) if achieved else None,
            "traffic": _pmc("pmc_%s.json" % f2_kernel, "hbm_bytes_per_launch") if (nb, N, K) == (64, 10000, 512) else None,
            "alg_bytes_per_launch": by, "kernel_ms": round(kern_ms, 4), "launches": launches, "alg_flops_per_launch": fl,
            "kernel_ms_in_timed_region": round(kern_region_ms, 4),
            "kernel_ms_is": "average HIP-event duration of the kernel with ONE pass in flight (300 passes on one stream right after the timed region, same inputs): the kernel's own time. Inside the timed region config.streams passes overlap, so a launch's start-to-end interval (kernel_ms_in_timed_region) also contains the co-running streams' kernels and is not a measure of the kernel",
            "whole_path_frac_of_roofline": round(value / world * t_roof_f32, 4),
            "whole_path_roofline_is": "SURVEY §8(d): max(bytes / 8 TB/s, FLOPs / 157.3 TF f32 MFMA) per bag — a figure of the f32-MFMA form, which this kernel no longer executes (> 1 is possible)",
            "whole_path_frac_of_executed_form_roofline": round(value / world * t_roof_exec, 4)}
        line["roofline"].update(two_read)
        return line
    line["roofline"] = {
        "kernel": "k_query_attend_split" if form else "k_query_attend", "bound": "mfma",
        "achieved": round(achieved, 2) if achieved else None, "peak": round(peak_exec, 1), "unit": "TFLOP/s",
        "frac": round(achieved / peak_exec, 4) if achieved else None,
        "peak_is": ("bf16 dense MFMA peak 2500 TF / %d plane products per fp32 MAC (exact 3-plane cuts, f32 accumulate)" % form)
                   if form else "f32 MFMA peak (v_mfma_f32_32x32x2_f32)",
        "mfma_form": {0: "v_mfma_f32_32x32x2_f32", 9: "bf16 MFMA, exact 3-plane cut, 9 plane products",
                      6: "bf16 MFMA, exact 3-plane cut, 6 largest plane products"}[form],
        "frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4) if achieved else None,
        "traffic": _pmc("pmc_k_query_attend.json", "hbm_bytes_per_launch") if (nb, N, K) == (64, 10000, 512) else None,
        "alg_bytes_per_launch": bytes_per_bag(N, K, C) * nb,
        "kernel_ms": round(kern_ms, 4), "launches": launches, "alg_flops_per_launch": fl,
        "kernel_ms_in_timed_region": round(kern_region_ms, 4),
        "kernel_ms_is": "average HIP-event duration of the kernel with ONE pass in flight (300 passes on one stream right after the timed region, same inputs): the kernel's own time. Inside the timed region config.streams passes overlap, so a launch's start-to-end interval (kernel_ms_in_timed_region) also contains the co-running streams' kernels and is not a measure of the kernel",
        "whole_path_frac_of_roofline": round(value / world * t_roof_f32, 4),
        "whole_path_roofline_is": "SURVEY §8(d): max(bytes / 8 TB/s, FLOPs / 157.3 TF f32 MFMA) per bag",
        "whole_path_frac_of_executed_form_roofline": round(value / world * t_roof_exec, 4)}
    line["roofline"].update(two_read)
    return line


def _usable_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a container often sees
    every core of the host in os.cpu_count() while being allowed a fraction of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, n)


def _best_threads(fn, budget_s=4.0):
    """torch CPU thread count that runs `fn` fastest on this box (asking for every visible core can be 50x slower
    than the right number when the container is throttled or the op does not scale): a short trial per candidate."""
    import torch
    cores = _usable_cores()
    cands = sorted({c for c in (cores, cores // 2, 64, 32, 16, 8) if 1 <= c <= cores}, reverse=True)
    best, best_t = cands[-1], float("inf")
    per = budget_s / len(cands)
    for c in cands:
        torch.set_num_threads(c)
        fn()
        n, t0 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            el = time.perf_counter() - t0
            if el >= per or n >= 50:
                break
        if el / n < best_t:
            best, best_t = c, el / n
    torch.set_num_threads(best)
    return best


def cpu_baseline_aggregator(weights_tag, N, K, budget_s):
    """The reference's forward on the host cores: the product's own CPU module path (dsmil-wsi_amd/modules.py
    `_forward_cpu` + FCLayer), op for op the torch sequence of dsmil.py:6-12,46-62 (and checked against vectors the
    reference produced, tests/test_cpu_module_golden.py), all cores."""
    import torch
    from dsmil_wsi_amd.synthetic import build_net, make_bag
    net = build_net(weights_tag, "cpu")
    bags = [torch.from_numpy(make_bag(50 + i, N, K)) for i in range(4)]
    with torch.no_grad():
        _best_threads(lambda: net(bags[0]))
        for b in bags[:3]:
            net(b)
        n, t0 = 0, time.perf_counter()
        while True:
            net(bags[n % len(bags)])
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 5000:
                break
    C = net.i_classifier.fc[0].out_features
    return {"value": round(n / el, 2), "unit": "bags/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "cores_visible": os.cpu_count(),
            "sample": f"{n} forwards of a {N}x{K} fp32 bag (C={C}) through the torch-CPU module path (op for op "
                      f"dsmil.py:46-62; reference sources are not on this box) in {el:.1f}s, at the thread count that ran "
                      f"fastest in a short trial"}


# ---------------------------------------------------------------------------------------------------------------
# embedder legs
# ---------------------------------------------------------------------------------------------------------------
def _build_iclassifier(cx, seed=11, C=2):
    import torch.nn as nn
    import dsmil
    from dsmil_wsi_amd.synthetic import load_weights, make_resnet18_weights   # seeded kaiming init
    from dsmil_wsi_amd.resnet import resnet18
    torch = cx.torch
    res = resnet18(pretrained=False, norm_layer=nn.InstanceNorm2d)
    for p in res.parameters():
        p.requires_grad = False
    res.fc = nn.Identity()
    res.load_state_dict(make_resnet18_weights(seed=seed), strict=True)
    wt = load_weights("tcga")
    ic = dsmil.IClassifier(res, 512, output_class=C)
    with torch.no_grad():
        ic.fc.weight.copy_(torch.from_numpy(wt["fc_w"][:C]))
        ic.fc.bias.copy_(torch.from_numpy(wt["fc_b"][:C]))
    return ic.to(cx.dev).eval()


def embedder_leg(cx, precision="fp32"):
    """precision = "half": the OPT-IN reduced-precision trunk (IClassifier.embed_precision, dsmil_resnet_forward_ex precision 1:
    every conv operand rounded to one fp16 plane, f32 accumulation, fp32 norms) — its own leg, never the headline embedder."""
    torch, args, dev, world, dist = cx.torch, cx.args, cx.dev, cx.world, cx.dist
    ic = _build_iclassifier(cx)
    ic.embed_precision = precision
    Bp = args.patches
    g = torch.Generator(device=dev).manual_seed(7 + cx.rank)
    xs = [torch.rand((Bp, 3, 224, 224), generator=g, device=dev, dtype=torch.float32) for _ in range(max(1, args.streams))]
    keep = []
    turn = [0]

    def one():
        turn[0] += 1
        with torch.no_grad():
            feats, c = ic(xs[turn[0] % len(xs)])    # a distinct batch per stream
        if cx.collectives:
            gathered = torch.empty((world * Bp, 512), device=dev)
            dist.all_gather_into_tensor(gathered, feats)
        return feats

    def step():
        keep[:] = [cx.run(one)]

    dt, inner, kern_ms_tot, launches = cx.timed(step, args.steps, args.warmup, args.min_seconds, channel=1)
    _, conv_alone_ms = cx.kernel_alone(one, 1, passes=40)
    if not os.environ.get("DSMIL_WINO_EXPT"):
        assert torch.isfinite(keep[0]).all()
    passes = args.steps * inner
    value = world * Bp * passes / dt
    # ideal time of the conv kernels on the pipes they run on: Winograd convs = direct FLOPs / 2.25 x 9 plane products
    # on the bf16 pipe; direct convs on the f32 MFMA pipe (or 9 products on the bf16 pipe when built that way)
    wp_, dp_ = ctypes.c_int32(0), ctypes.c_int32(0)
    cx.L.dsmil_resnet_mfma_forms(ctypes.byref(wp_), ctypes.byref(dp_))
    wino_np, direct_np = int(wp_.value), int(dp_.value)
    if precision == "half":
        wino_np = direct_np = 1
    forms = {"wino_products": wino_np, "direct_products": direct_np}
    t_wino = WINO_FLOPS_PER_PATCH / 2.25 * (wino_np / (PEAK_BF16_MFMA_TFLOPS * 1e12) if wino_np else 1 / (PEAK_F32_MFMA_TFLOPS * 1e12))
    t_direct = DIRECT_FLOPS_PER_PATCH * (direct_np / (PEAK_BF16_MFMA_TFLOPS * 1e12) if direct_np else 1 / (PEAK_F32_MFMA_TFLOPS * 1e12))
    # conv-kernel time: with several streams the in-region event intervals overlap each other (they contain the other
    # streams' kernels), so the kernels' own time is taken from 40 forwards with one in flight, right after the region
    kern_region_ms_tot = kern_ms_tot
    if cx.pool is not None:
        kern_ms_tot = conv_alone_ms * passes
    t_stem = STEM_FLOPS_PER_PATCH * (direct_np / (PEAK_BF16_MFMA_TFLOPS * 1e12) if direct_np else 1 / (PEAK_F32_MFMA_TFLOPS * 1e12))
    t_exec_patch = max(t_wino + t_direct + t_stem, ACT_BYTES_PER_PATCH / (PEAK_HBM_GBS * 1e9))
    kern_s = kern_ms_tot * 1e-3
    conv_flops = (FLOPS_PER_PATCH - STEM_FLOPS_PER_PATCH) * Bp * passes
    ach = conv_flops / kern_s / 1e12 if kern_s > 0 else None
    frac_pipe = (t_wino + t_direct) * Bp * passes / kern_s if kern_s > 0 else None
    return {"metric": "patches/sec embedded (ResNet-18-IN, 224x224, bs=%d)%s" % (Bp, ", OPT-IN half-precision operands" if precision == "half" else ""),
            "value": round(value, 1),
            "unit": "patches/s", "ms_per_step": round(dt / args.steps * 1e3, 3), "ms_per_pass": round(dt / passes * 1e3, 3),
            "dtype": "f16 operands (one plane, RNE), f32 accumulate, f32 activations and norms" if precision == "half" else "f32",
            **({"tolerance": "feature error <= 5e-3 abs against the fp64 restatement (measured ~2e-3; tests/test_resnet_gpu.py, "
                             "tools/form_error_study.py f16x1) — NOT the 1e-4 parity bar of the `embedder` leg"} if precision == "half" else {}),
            "config": {"workload": f"IClassifier(ResNet-18 InstanceNorm, fc=Identity)+Linear(512,2), {Bp} synthetic "
                                   f"224x224 patches per GPU per pass, kaiming(seed 11) weights",
                       "passes_per_step": inner, "timed_region_s": round(dt, 3), "streams": args.streams,
                       "distinct_batches": len(xs),
                       "parity": "unpinned (torchvision absent, the reference ships no embedder vectors); two independent restatements",
                       "collective": "all_gather_into_tensor([%d,512] f32) per pass, %d rank(s)" % (Bp, world) if cx.collectives else "none"},
            "roofline": {"kernel": "conv kernels of one forward: 9 x k_conv_wino_w1 + 4 x k_conv_wino_s3 (Winograd F(2x2,3x3)) + 6 direct "
                                   "convs, all on %s" % ({1: "fp16 MFMA, ONE plane per operand (opt-in reduced precision)",
                                                           3: "fp16 MFMA over two-plane cuts, 3 plane products per fp32 MAC",
                                                           6: "bf16 MFMA over exact 3-plane cuts, 6 plane products",
                                                           9: "bf16 MFMA over exact 3-plane cuts, 9 plane products",
                                                           0: "f32 MFMA"}.get(wino_np, "?")), "bound": "mfma",
                         # algorithmic (direct-form) rate of the conv kernels; NOT compared with a peak: Winograd does
                         # 2.25x fewer multiplies than the direct form, so this rate may exceed the f32 MFMA peak
                         "achieved": round(ach, 2) if ach else None, "unit": "TFLOP/s",
                         # the fraction that IS bounded by 1: ideal time of the executed MFMA work on the pipe each conv
                         # class runs on / measured conv-kernel time
                         "frac": round(frac_pipe, 4) if frac_pipe else None,
                         "frac_is": "sum over conv classes of (executed MFMA FLOPs / peak of the pipe they run on) / kernel time",
                         "peak": PEAK_BF16_MFMA_TFLOPS, "executed_forms": forms,
                         "traffic": _pmc("pmc_k_conv.json", "hbm_bytes_per_forward"),
                         "kernel_ms_total": round(kern_ms_tot, 3), "launches": launches, "alg_flops_total": conv_flops,
                         "conv_ms_per_forward": round(kern_ms_tot / passes, 3),
                         "conv_ms_per_forward_in_timed_region": round(kern_region_ms_tot / passes, 3),
                         "kernel_ms_is": "HIP-event durations of the conv kernels with ONE forward in flight (40 forwards on one stream right after the timed region); inside the region config.streams forwards overlap and a launch's interval contains the co-running kernels (conv_ms_per_forward_in_timed_region)",
                         # the figure the >= 60 % target of BASELINE.json refers to: whole forward vs SURVEY §8(d)'s
                         # direct-form fp32 roofline (3.627 GFLOP/patch at 157.3 TF = 43 368 patches/s)
                         # >= 1 is possible and expected: that roofline prices 3.627 GFLOP of direct-form fp32 MACs on the f32
                         # MFMA pipe, while the forward executes 2.25x fewer multiplies (Winograd) on the bf16 pipe — it is a
                         # RATIO to the reference formulation, not a utilisation
                         "ratio_to_f32_direct_form_roofline": round(value / world / (PEAK_F32_MFMA_TFLOPS * 1e12 / FLOPS_PER_PATCH), 4),
                         # the bound of the forward AS EXECUTED: max(MFMA time of every conv on the pipe it runs on,
                         # activation bytes / 8 TB/s) per patch — this one is <= 1
                         "whole_path_frac_of_roofline": round(value / world * t_exec_patch, 4),
                         "whole_path_roofline_is": "max(executed MFMA FLOPs / pipe peak summed over stem, Winograd and direct convs, "
                                                   "%.1f MB of fp32 NHWC activation traffic / 8 TB/s) = %.2f us per patch"
                                                   % (ACT_BYTES_PER_PATCH / 1e6, t_exec_patch * 1e6)}}


def embedder_16_leg(cx, precision):
    """The OPT-IN 16-bit-ACTIVATION trunk (round 6, csrc/resnet_b16.h): activations stored in 16 bits behind the stem, one 16-bit
    MFMA product per MAC, f32 accumulation and InstanceNorm statistics — BASELINE.md's "1 patch, bf16 MFMA / f32 accumulate" row.
    precision "bf16" (dsmil_resnet_forward_ex precision 2) or "half" (precision 3, fp16 activations: what `--precision half` takes on
    a ResNet-18 / 34 InstanceNorm trunk).  Its own legs and tolerances, never the headline embedder."""
    torch, args, dev, world = cx.torch, cx.args, cx.dev, cx.world
    ic = _build_iclassifier(cx)
    ic.embed_precision = precision
    f16 = precision == "half"
    Bp = args.patches
    g = torch.Generator(device=dev).manual_seed(7 + cx.rank)
    xs = [torch.rand((Bp, 3, 224, 224), generator=g, device=dev, dtype=torch.float32) for _ in range(max(1, args.streams))]
    keep = []
    turn = [0]

    def one():
        turn[0] += 1
        with torch.no_grad():
            feats, c = ic(xs[turn[0] % len(xs)])
        return feats

    def step():
        keep[:] = [cx.run(one)]

    dt, inner, _, launches = cx.timed(step, args.steps, args.warmup, args.min_seconds, channel=1)
    _, conv_alone_ms = cx.kernel_alone(one, 1, passes=40)
    assert torch.isfinite(keep[0]).all()
    passes = args.steps * inner
    value = world * Bp * passes / dt
    conv_flops = (FLOPS_PER_PATCH - STEM_FLOPS_PER_PATCH) * Bp          # direct-form FLOPs of the 19 convs behind the stem, per forward
    frac = conv_flops / (PEAK_BF16_MFMA_TFLOPS * 1e12) / (conv_alone_ms * 1e-3) if conv_alone_ms > 0 else None
    el = "fp16" if f16 else "bf16"
    return {"metric": "patches/sec embedded (ResNet-18-IN, 224x224, bs=%d), OPT-IN %s activations" % (Bp, el),
            "value": round(value, 1), "unit": "patches/s", "ms_per_step": round(dt / args.steps * 1e3, 3),
            "ms_per_pass": round(dt / passes * 1e3, 3),
            "dtype": f"{el} activations and conv operands behind the stem (one {el} MFMA product per MAC), f32 accumulate, f32 InstanceNorm statistics",
            "tolerance": ("feature error <= 5e-3 abs against the fp64 restatement on features of O(1) (measured 2.6e-3 max / 4.6e-4 mean; "
                          "tests/test_resnet_gpu.py)" if f16 else
                          "feature error <= 5e-2 max / 8e-3 mean abs against the fp64 restatement on features of O(1) (measured 2e-2 / "
                          "3e-3; tests/test_resnet_gpu.py)") + " — NOT the 1e-4 parity bar of the `embedder` leg",
            "config": {"workload": f"IClassifier(ResNet-18 InstanceNorm, fc=Identity)+Linear(512,2), {Bp} synthetic 224x224 patches "
                                   f"per GPU per pass, kaiming(seed 11) weights", "passes_per_step": inner,
                       "timed_region_s": round(dt, 3), "streams": args.streams, "distinct_batches": len(xs)},
            "roofline": {"kernel": "the 19 conv launches behind the stem: 13 x k_conv_b16w (3x3 / 1, flat positions, window staging) + 6 x "
                                   f"k_conv_b16g (3x3 / 2, 1x1 / 2), {el} MFMA, one product", "bound": "mfma",
                         "achieved": round(conv_flops / (conv_alone_ms * 1e-3) / 1e12, 2) if conv_alone_ms > 0 else None,
                         "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(frac, 4) if frac else None,
                         "frac_is": "direct-form FLOPs of the 19 convs / 16-bit MFMA peak / their HIP-event time with one forward in flight "
                                    "(the kernels also compute the zero borders of the layout: 3.5 % of the positions at 56 x 56, 31 % at 7 x 7)",
                         "traffic": None, "conv_ms_per_forward": round(conv_alone_ms, 3), "launches": launches,
                         # BASELINE.md's row for this precision: 3.627 GFLOP / 2.5 PFLOP/s = 689 252 patches/s
                         "whole_path_frac_of_roofline": round(value / world / (PEAK_BF16_MFMA_TFLOPS * 1e12 / FLOPS_PER_PATCH), 4)}}


def cpu_baseline_embedder(budget_s):
    """oracle/resnet_oracle.py (the torch-CPU restatement of the torchvision backbone, fp32) on the host cores, bounded."""
    import torch
    if os.path.join(ROOT, "oracle") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import resnet_oracle as ro
    from dsmil_wsi_amd.synthetic import make_patches
    w = ro.make_weights(seed=11)
    x = torch.from_numpy(make_patches(7, 16))
    with torch.no_grad():
        _best_threads(lambda: ro.resnet18_in_features(x, w))
        n, t0 = 0, time.perf_counter()
        while True:
            ro.resnet18_in_features(x, w)
            n += x.shape[0]
            el = time.perf_counter() - t0
            if el >= budget_s:
                break
    return {"value": round(n / el, 2), "unit": "patches/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "cores_visible": os.cpu_count(),
            "sample": f"{n} patches (batches of 16, 224x224) through oracle/resnet_oracle.py (torch CPU fp32) in {el:.1f}s, "
                      f"at the thread count that ran fastest in a short trial"}


def slide_leg(cx, n_patches, n_steps=None, host=False, precision="fp32"):
    """ONE slide of n_patches ordered tiles (uint8 NHWC, resident), cut contiguously over the
    ranks, embedded in batches of --patches, ONE all-gather of the [N_r,512] rows, then the aggregator on the bag.
    Strong scaling: the slide is fixed, per-rank work shrinks with N.
    host=True (`slide_h2d`): the tiles start in PINNED HOST memory, as compute_feats.py:69-75 has them after decode; every
    batch is copied H2D on the stream that embeds it, so the copy engines run under the other streams' convs, and the bag's
    rows + logits go back D2H at the end (the reference's per-batch `.cpu()`, once)."""
    torch, args, dev, world, dist = cx.torch, cx.args, cx.dev, cx.world, cx.dist
    from dsmil_wsi_amd import dist as dd
    from dsmil_wsi_amd import pipeline as pl
    from dsmil_wsi_amd.synthetic import build_net
    ic = _build_iclassifier(cx)
    ic.embed_precision = precision   # "half" / "bf16" (`slide_half`, `slide_bf16`): the opt-in 16-bit-activation trunk (tolerances: the `embedder_half` / `embedder_bf16` legs)
    net = build_net("tcga", dev)
    lo, hi = dd.shard_range(n_patches, cx.rank, world)
    g = torch.Generator(device=dev).manual_seed(99)   # every rank draws the same slide and keeps its rows
    tiles = torch.empty((hi - lo, 224, 224, 3), dtype=torch.uint8, device=dev)
    chunk = 512
    for s in range(0, n_patches, chunk):   # same stream on every rank: rows [lo, hi) are this rank's
        e = min(n_patches, s + chunk)
        blk = torch.randint(0, 256, (e - s, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8)
        a, b = max(s, lo), min(e, hi)
        if b > a:
            tiles[a - lo:b - lo] = blk[a - s:b - s]
    del blk
    if host:
        tiles_dev, tiles = tiles, torch.empty(tiles.shape, dtype=torch.uint8, pin_memory=True)
        tiles.copy_(tiles_dev)
        del tiles_dev
        torch.cuda.empty_cache()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    res = {}
    sizes = [dd.shard_range(n_patches, r, world)[1] - dd.shard_range(n_patches, r, world)[0] for r in range(world)]

    def step():
        with torch.no_grad():
            ev[0].record()
            feats, classes = pl.embed_tiles(ic, tiles, args.patches, streams=args.streams, device=dev)
            ev[1].record()
            if cx.collectives:   # ONE collective per slide: feature rows and instance logits as one [N_r, 512 + C] matrix
                bag, _ = dd.all_gather_packed([feats, classes], sizes)
            else:
                bag = feats
            ev[2].record()
            res["out"] = net(bag)
            if host:   # features and instance logits back to the host, once per slide (compute_feats.py:74 does it per batch)
                res["host"] = (bag.to("cpu", non_blocking=True), res["out"][0].to("cpu", non_blocking=True))
            ev[3].record()

    steps = n_steps if n_steps else max(2, min(args.steps, 5))
    dt, inner, _, _ = cx.timed(step, steps, 1, 0.0, fixed_passes=1)
    torch.cuda.synchronize()
    out = res["out"]
    assert out[2].shape[0] == n_patches and torch.isfinite(out[1]).all()
    return {"metric": "patches/sec, one slide %sembedded%s + gathered + aggregated" % ("copied H2D + " if host else "", " (OPT-IN %s activations)" % ("bf16" if precision == "bf16" else "fp16") if precision != "fp32" else ""),
            "value": round(n_patches * steps / dt, 1),
            "unit": "patches/s", "scaling": "strong", "ms_per_slide": round(dt / steps * 1e3, 3),
            "last_slide_ms": {"embed": round(ev[0].elapsed_time(ev[1]), 3), "all_gather": round(ev[1].elapsed_time(ev[2]), 3),
                              "aggregate": round(ev[2].elapsed_time(ev[3]), 3)},
            "config": {"tiles_start_in": "pinned host memory (H2D per batch on its embed stream, D2H of rows + logits per slide)" if host else "HBM",
                       "h2d_bytes_per_slide": int(tiles.numel()) if host else 0,
                       "workload": f"one slide = {n_patches} uint8 224x224 tiles (ToTensor fused in the stem), contiguous row "
                                   f"shards over {world} rank(s), batches of {args.patches}, one all-gather of [N_r,512] f32, "
                                   f"MILNet(tcga) on the gathered bag", "slides_timed": steps, "rccl_ranks": world,
                       "streams": args.streams, "rows_this_rank": hi - lo,
                       "collectives_per_slide": 1 if cx.collectives else 0,
                       "all_gather_bytes_per_rank": (max(sizes) * (512 + 2) * 4 * world) if cx.collectives else 0,
                       "multi_gpu": "measured on %d rank(s)%s" % (world, "; RCCL exercised with a one-rank group (--force-collective), "
                                    "2/4/8-GPU scaling unmeasured on hardware" if world == 1 and cx.collectives else
                                    ("; unmeasured on hardware beyond one GPU" if world == 1 else ""))}}


def _jpeg_tiles(n, uniq=192, seed=5):
    """n synthetic 224 x 224 tiles as baseline-JPEG files in host memory (bytes): `uniq` distinct tissue-like images (blocky
    structure + noise: ~11 KB per tile at the tiler's quality 70, deepzoom_tiler.py:64,250), repeated in order."""
    import io
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    files = []
    for _ in range(min(uniq, n)):
        base = rng.integers(0, 256, (30, 30, 3)).repeat(8, 0).repeat(8, 1)[:224, :224]
        a = np.clip(base + rng.normal(0, 12, (224, 224, 3)), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(a).save(b, "JPEG", quality=70)
        files.append(b.getvalue())
    return [files[i % len(files)] for i in range(n)]


def decode_leg(cx, n_tiles=8192):
    """dsmil_jpeg_decode alone — n_tiles baseline-JPEG tiles whose compressed bytes and parsed plan are resident
    in HBM -> uint8 NHWC; checked against Pillow's decode of the same files (byte for byte) outside the timed region."""
    torch, args, dev = cx.torch, cx.args, cx.dev
    import io
    import numpy as np
    from PIL import Image
    import dsmil_wsi_amd.ops as ops
    blobs = _jpeg_tiles(n_tiles)
    data, plan, recs = ops.jpeg_parse(blobs)
    assert (recs["status"] == 0).all()
    d_data, d_plan = torch.from_numpy(data).to(dev), torch.from_numpy(plan).to(dev)
    out = torch.empty((n_tiles, 224, 224, 3), dtype=torch.uint8, device=dev)
    status = torch.empty(n_tiles, dtype=torch.int32, device=dev)
    ws = torch.empty(cx.L.dsmil_jpeg_workspace_bytes(n_tiles, 224, 224, int(data.size) - 32), dtype=torch.uint8, device=dev)

    def step():
        rc = cx.L.dsmil_jpeg_decode(d_data.data_ptr(), int(data.size) - 32, d_plan.data_ptr(), n_tiles, 224, 224, out.data_ptr(), status.data_ptr(),
                                    ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
        assert rc == 0

    steps = max(2, min(args.steps, 10))
    dt, inner, _, _ = cx.timed(step, steps, 1, 0.0, fixed_passes=1)
    assert int(status.abs().sum()) == 0
    got = out[:192].cpu().numpy()
    for i in (0, 1, 95, 191):
        assert np.array_equal(got[i], np.array(Image.open(io.BytesIO(blobs[i])).convert("RGB"))), "device decode != Pillow"
    comp = sum(len(b) for b in blobs)
    del out, ws, d_data, d_plan, status
    torch.cuda.empty_cache()
    line = {"metric": "tiles/sec JPEG-decoded on the device (224x224, quality 70, 4:2:0)", "value": round(cx.world * n_tiles * steps / dt, 1),
            "unit": "tiles/s", "ms_per_batch": round(dt / steps * 1e3, 3), "dtype": "u8",
            "config": {"workload": f"{n_tiles} baseline-JPEG tiles per launch sequence (memset + 3 launches), compressed bytes "
                                   f"({comp / n_tiles / 1024:.1f} KiB per tile) and the parsed plan resident in HBM; output uint8 NHWC; "
                                   "bit-identical to Pillow (checked)", "tiles": n_tiles,
                       "compressed_bytes_per_tile": round(comp / n_tiles, 1), "huffman_workgroups": (n_tiles + 1023) // 1024},
            "roofline": {"bound": "latency", "note": "Huffman decoding is serial per stream: one lane per tile, 1 024-lane workgroups "
                         "— the launch holds ceil(tiles / 1024) compute units; not priced against a pipe"}}
    if not args.no_cpu_baseline and cx.world == 1:
        from concurrent.futures import ThreadPoolExecutor
        cores = _usable_cores()

        def dec(b):
            return np.array(Image.open(io.BytesIO(b)).convert("RGB")).shape

        t0 = time.perf_counter()
        n1 = 0
        while time.perf_counter() - t0 < min(3.0, args.cpu_seconds):
            dec(blobs[n1 % 192])
            n1 += 1
        one = n1 / (time.perf_counter() - t0)
        thr = min(cores, 16)
        with ThreadPoolExecutor(thr) as pool:     # (Pillow's decoder releases the GIL)
            t0 = time.perf_counter()
            m = 0
            while time.perf_counter() - t0 < min(5.0, args.cpu_seconds):
                list(pool.map(dec, blobs[:1024]))
                m += 1024
            many = m / (time.perf_counter() - t0)
        line["cpu_baseline"] = {"value": round(many, 1), "unit": "tiles/s", "cores": thr, "kind": "reference",
                                "one_thread": round(one, 1),
                                "sample": f"Pillow (libjpeg-turbo) Image.open(...).convert('RGB') of the same files — the reference's own "
                                          f"decoder, compute_feats.py:28 — {m} decodes on {thr} threads, {n1} on one"}
    return line


def slide_jpeg_leg(cx, n_patches, precision="fp32"):
    """The `slide` leg starting from JPEG FILES IN HOST MEMORY (what compute_feats.py:65-69 globs, read into bytes): device decode
    of 2 048-tile chunks on a side stream under the embedding of the previous chunk (pipeline.embed_jpeg_blobs), one all-gather,
    the aggregator.  Strong scaling: contiguous file shards over the ranks.  precision "half" (`slide_jpeg_half`): the embedder on
    the opt-in fp16-activation trunk (tolerance: the `embedder_half` leg) — the decode has half the time to hide in."""
    torch, args, dev, world = cx.torch, cx.args, cx.dev, cx.world
    from dsmil_wsi_amd import dist as dd
    from dsmil_wsi_amd import pipeline as pl
    from dsmil_wsi_amd.synthetic import build_net
    ic = _build_iclassifier(cx)
    ic.embed_precision = precision
    net = build_net("tcga", dev)
    lo, hi = dd.shard_range(n_patches, cx.rank, world)
    blobs = _jpeg_tiles(n_patches)[lo:hi]
    sizes = [dd.shard_range(n_patches, r, world)[1] - dd.shard_range(n_patches, r, world)[0] for r in range(world)]
    res, stats = {}, {}

    def step():
        with torch.no_grad():
            feats, classes = pl.embed_jpeg_blobs(ic, blobs, args.patches, decode_batch=2048, streams=args.streams, device=dev, stats=stats)
            bag = dd.all_gather_packed([feats, classes], sizes)[0] if cx.collectives else feats
            res["out"] = net(bag)

    steps = max(2, min(args.steps, 5))
    dt, inner, _, _ = cx.timed(step, steps, 1, 0.0, fixed_passes=1)
    torch.cuda.synchronize()
    assert res["out"][2].shape[0] == n_patches and torch.isfinite(res["out"][1]).all() and stats.get("pillow", 0) == 0
    comp = sum(len(b) for b in blobs)
    return {"metric": "patches/sec, one slide of JPEG tiles decoded on the device + embedded + gathered + aggregated" + (", OPT-IN fp16 activations" if precision == "half" else ""),
            "value": round(n_patches * steps / dt, 1), "unit": "patches/s", "scaling": "strong", "ms_per_slide": round(dt / steps * 1e3, 3),
            "config": {"tiles_start_in": "host memory as baseline-JPEG files (bytes)", "h2d_bytes_per_slide": comp,
                       "workload": f"one slide = {n_patches} JPEG tiles (224x224, quality 70), chunks of 2048 decoded by dsmil_jpeg_decode "
                                   f"on a side stream under the previous chunk's embedding (batches of {args.patches}), one all-gather, "
                                   f"MILNet(tcga)", "slides_timed": steps, "rccl_ranks": world, "streams": args.streams,
                       "collectives_per_slide": 1 if cx.collectives else 0}}


def feats_csv_leg(cx, rows=10000, cols=512):
    """The reference's feature file of one bag (compute_feats.py:80-82, `to_csv(float_format='%.4f')`; HOST work, rank 0's cores):
    pipeline.save_feats_csv (dsmil_csv_format_f32: the same bytes) beside pandas formatting the same rows — timed on a tenth of the
    bag and scaled (pandas needs ~5 s for the whole one)."""
    import io
    import tempfile
    import numpy as np
    import pandas as pd
    from dsmil_wsi_amd import pipeline as pl
    rng = np.random.default_rng(12)
    feats = rng.standard_normal((rows, cols)).astype(np.float32)
    n_ref = max(1, rows // 10)
    b = io.StringIO()
    t0 = time.perf_counter()
    pd.DataFrame(feats[:n_ref]).to_csv(b, index=False, float_format="%.4f")
    t_pd = (time.perf_counter() - t0) * rows / n_ref
    assert pl.feats_csv_bytes(feats[:n_ref]) == b.getvalue().encode()
    with tempfile.TemporaryDirectory() as d:
        pl.save_feats_csv(feats, os.path.join(d, "w.csv"))
        ts = []
        for i in range(5):
            t0 = time.perf_counter()
            pl.save_feats_csv(feats, os.path.join(d, f"b{i}.csv"))
            ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    return {"metric": "bags/sec written as the reference's feature CSV (10 000 x 512, '%.4f')", "value": round(1.0 / t, 2), "unit": "bags/s",
            "ms_per_bag": round(t * 1e3, 1), "dtype": "f32 -> decimal text, exact (ties to even)",
            "cpu_baseline": {"value": round(1.0 / t_pd, 3), "unit": "bags/s", "cores": 1, "kind": "reference",
                             "sample": f"pandas DataFrame.to_csv(float_format='%.4f') of {n_ref} x {cols} rows, scaled to {rows}: the reference's own call (compute_feats.py:82)"},
            "config": {"workload": f"one bag = {rows} x {cols} float32 features -> CSV bytes identical to pandas', median of 5",
                       "threads": pl.CSV_THREADS[0], "bytes_per_bag": int(len(pl.feats_csv_bytes(feats[:n_ref])) * rows / n_ref)}}


def e2e_leg(cx, low_grid, precision="fp32"):
    """BASELINE configs[4]: synthetic two-level slide -> attention map (pipeline.multiscale_attention_map).
    precision "half" (`e2e_half`): both embedders on the opt-in fp16-activation trunk (tolerance: the `embedder_half` leg)."""
    torch, args, dev, world = cx.torch, cx.args, cx.dev, cx.world
    import numpy as np
    from dsmil_wsi_amd import pipeline as pl
    from dsmil_wsi_amd.synthetic import build_net
    gy, gx = low_grid
    e_lo, e_hi = _build_iclassifier(cx, seed=11), _build_iclassifier(cx, seed=12)
    e_lo.embed_precision = e_hi.embed_precision = precision
    net = build_net("tree", dev)
    g = torch.Generator(device=dev).manual_seed(2024)
    wsi = torch.randint(0, 256, (gy * 896, gx * 896, 3), generator=g, device=dev, dtype=torch.uint8)
    colors = [np.array([255, 40, 0]), np.array([0, 90, 255])]
    res, tm = {}, {}

    def step():
        res["out"] = pl.multiscale_attention_map(wsi, e_lo, e_hi, net, [0.5, 0.5], colors, batch_size=args.patches, timings=tm, streams=args.streams)

    steps = max(2, min(args.steps, 3))
    dt, _, _, _ = cx.timed(step, steps, 1, 0.0, fixed_passes=1)
    out = res["out"]
    n_high, n_low = gy * gx * 16, gy * gx
    assert out["feats"].shape == (n_high, 1024) and torch.isfinite(out["pred"]).all()
    return {"metric": "slides/sec, multi-scale end to end (tile -> 2-scale embed -> concat -> aggregate -> attention map)" + (", OPT-IN fp16 activations" if precision == "half" else ""),
            "value": round(steps / dt, 4), "unit": "slides/s", "scaling": "strong", "ms_per_slide": round(dt / steps * 1e3, 2),
            "patches_per_s": round((n_high + n_low) * steps / dt, 1),
            "config": {"workload": f"synthetic uint8 slide {gy * 896}x{gx * 896}: {n_low} low tiles + {n_high} high tiles (224x224), "
                                   f"two ResNet-18-IN embedders, [high||low] 1024-d, MILNet(FCLayer(1024,2), BClassifier(1024,2)), "
                                   f"32x colour map replicated on the GPU", "slides_timed": steps, "rccl_ranks": world, "streams": args.streams,
                       "all_gather_s_total": round(tm.get("allgather_s", 0.0), 4), "all_gather_bytes": tm.get("allgather_bytes", 0),
                       "collectives_per_slide": 1 if (world > 1 or args.force_collective) else 0}}


def train_leg(cx, weights_tag):
    """SURVEY §8 a9 / N1 — one `train_tcga.py:60-75` step per bag: zero_grad, forward, 0.5 BCE(bag) + 0.5 BCE(max instance),
    backward, Adam(lr 1e-4, betas (0.5, 0.9), weight_decay 1e-3: train_tcga.py:203,207,241), and the loss `.item()` of the
    progress line (:74) — the step's one host sync.  Bags are 10 000 x 512 fp32, HBM-resident (training.BagCache), a
    different bag every step.  Training is replicas-only across GPUs (bag-level data parallelism would change the per-bag
    SGD semantics): with N ranks the value is N independent replicas."""
    torch, args, dev = cx.torch, cx.args, cx.dev
    from dsmil_wsi_amd import training
    from dsmil_wsi_amd.synthetic import build_net
    N, K = args.rows, args.feats
    net = build_net(weights_tag, dev).train()
    C = net.i_classifier.fc[0].out_features
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.5, 0.9), weight_decay=1e-3)
    crit = torch.nn.BCEWithLogitsLoss()
    g = torch.Generator(device=dev).manual_seed(4321 + cx.rank)
    nbags = 16
    bags = [torch.randn((N, K), generator=g, device=dev) for _ in range(nbags)]
    labels = [torch.zeros((1, C), device=dev) for _ in range(nbags)]
    for i, y in enumerate(labels):
        y[0, i % C] = float((i // C) % 2) if C == 1 else 1.0
    turn = [0]
    last = {}
    fused = training.FusedTrainStep.create(net, crit, opt)   # what training.train (train_tcga.py's loop here) uses
    assert fused is not None, "the reference's model / criterion / optimiser must take the fused step"

    readback = training.LossReadback(dev)

    def step(sync=True):
        i = turn[0] = (turn[0] + 1) % nbags
        loss = fused(bags[i], labels[i])
        if sync == "pipelined":     # what training.train does: every loss reaches the host, read one step late
            prev = readback.push(loss)
            last["loss"] = prev if prev is not None else 0.0
        else:
            last["loss"] = loss.item() if sync else loss

    def step_generic(sync=True):   # the autograd path (any criterion / optimiser): bag_loss -> backward -> optimizer.step
        i = turn[0] = (turn[0] + 1) % nbags
        opt.zero_grad()
        loss, _, _ = training.bag_loss(net, crit, bags[i], labels[i])
        loss.backward()
        opt.step()
        last["loss"] = loss.item() if sync else loss.detach()

    dt, inner, _, _ = cx.timed(lambda: step("pipelined"), args.steps, max(50, args.warmup), args.min_seconds / 2)   # (50 steps: allocator + clocks settled)
    value_pipelined = cx.world * inner * args.steps / dt
    last["loss"] = readback.flush()
    dt4, inner4, _, _ = cx.timed(step, max(2, args.steps // 4), 1, args.min_seconds / 4)
    value = cx.world * inner4 * max(2, args.steps // 4) / dt4
    dt2, inner2, _, _ = cx.timed(lambda: step(False), max(2, args.steps // 4), 1, args.min_seconds / 4)
    value_nosync = cx.world * inner2 * max(2, args.steps // 4) / dt2
    last["loss"] = float(last["loss"])
    fused.sync()
    dt3, inner3, _, _ = cx.timed(step_generic, max(2, args.steps // 4), 1, args.min_seconds / 4)
    value_generic = cx.world * inner3 * max(2, args.steps // 4) / dt3
    # GPU time of a step: HIP events around 100 fused steps enqueued back to back (no host sync in between)
    n_ev = 100
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fused2 = training.FusedTrainStep.create(net, crit, opt)
    for k in range(10):
        fused2(bags[k % nbags], labels[k % nbags])
    e0.record()
    for k in range(n_ev):
        fused2(bags[k % nbags], labels[k % nbags])
    e1.record()
    torch.cuda.synchronize()
    fused2.sync()
    gpu_ms_step = e0.elapsed_time(e1) / n_ev
    assert math.isfinite(float(last["loss"])), "training loss is not finite"
    fl = 3 * flops_per_bag(N, K, C)                      # forward + two backward contractions per forward contraction
    form = int(cx.L.dsmil_agg_mlp_form())
    peak_exec = PEAK_BF16_MFMA_TFLOPS / form if form else PEAK_F32_MFMA_TFLOPS
    t_roof = fl / (peak_exec * 1e12)
    # where the GPU time of a step goes, from the committed rocprofv3 profile of tools/train_fused.py (C = 1; not this run)
    split = None
    tab = _kernel_table("train_step")
    if tab and C == 1:
        ms = {r["kernel"].split("<")[0]: r["ms_per_pass"] for r in tab["rows"]}
        fwd = ("k_logits_stream", "k_attend_hs", "k_finish")
        bwd = ("k_bwd_prep", "k_bwd_rows_hs", "k_bwd_critical", "k_bwd_gh_hs", "k_tn_split", "k_bwd_reduce")
        split = {"source": "profiles/" + KERNEL_TABLE + " (train_step)", "launches": len(tab["rows"]),
                 "weight_planes_and_offsets": round(ms.get("k_train_prologue", 0.0), 4),
                 "forward": round(sum(ms.get(k, 0.0) for k in fwd), 4),
                 "loss_head_backward_adam": round(sum(ms.get(k, 0.0) for k in bwd), 4),
                 "merged_is": "the loss head (and the bag head's last sum) run inside k_bwd_prep, Adam inside k_bwd_reduce"}
    return {"metric": "bags/sec trained (one Adam step per 10kx512 bag)", "value": round(value_pipelined, 1), "unit": "bags/s",
            "ms_per_step_bag": round(1e3 / (value_pipelined / cx.world), 4), "dtype": "f32", "scaling": "replicas",
            "value_blocking_item_per_step": round(value, 1),
            "value_without_per_step_sync": round(value_nosync, 1),
            "value_is": "the product's loop (training.train): every step's loss reaches the host and the progress line of train_tcga.py:75, read one step late through a pinned ring (training.LossReadback); value_blocking_item_per_step: a blocking loss.item() after every step, as train_tcga.py:75 literally writes it; value_without_per_step_sync: no read-back at all",
            "value_generic_autograd_path": round(value_generic, 1),
            "gpu_ms": {"step_enqueued_back_to_back": round(gpu_ms_step, 4), "split_profiled": split},
            "config": {"workload": f"train_tcga.py:60-75 step on MILNet(FCLayer({K},{C}), BClassifier({K},{C})), {weights_tag} weights, "
                                   f"one {N} x {K} fp32 bag per step ({nbags} distinct, HBM-resident), Adam, every step's loss read on the host; "
                                   f"training.FusedTrainStep = one dsmil_agg_train_step call per step",
                       "steps_timed": inner * args.steps, "timed_region_s": round(dt, 3), "classes": C},
            "roofline": {"bound": "mfma", "kernel": "whole step (dsmil_agg_train_step: forward, loss head, backward, Adam)", "unit": "TFLOP/s",
                         "achieved": round(fl * value_pipelined / cx.world / 1e12, 2), "peak": round(peak_exec, 1),
                         "frac": round(t_roof * value_pipelined / cx.world, 4),
                         "alg_flops_per_step": fl, "peak_is": "3 x forward FLOPs on the pipe the forward's MLP executes on (bf16 MFMA / plane products)"}}


def _strip_prose(obj):
    """Drop the explanatory strings (`*_is`, notes) from the printed line: DESIGN.md §5 documents every field."""
    if isinstance(obj, dict):
        for k in [k for k in obj if k.endswith("_is") or k in ("frac_is", "peak_is")]:
            del obj[k]
        for v in obj.values():
            _strip_prose(v)
    elif isinstance(obj, list):
        for v in obj:
            _strip_prose(v)


def _summary(line):
    """Compact per-leg summary (value + the roofline fraction that bounds the leg): stored in config.legs — the driver keeps
    `config` whole — and repeated as the LAST key of the line, so a tail-truncated stdout still carries every leg's number."""
    out = {}

    def put(name, obj):
        if not obj:
            return
        r = obj.get("roofline") or {}
        e = {"value": obj.get("value"), "unit": obj.get("unit")}
        for k in ("frac", "whole_path_frac_of_roofline", "two_read_frac", "kernel_ms", "conv_ms_per_forward"):
            if r.get(k) is not None:
                e[k] = r[k]
        if "ms_per_slide" in obj:
            e["ms_per_slide"] = obj["ms_per_slide"]
        if "ms_per_batch" in obj:
            e["ms_per_batch"] = obj["ms_per_batch"]
        if "ms_per_bag" in obj:
            e["ms_per_bag"] = obj["ms_per_bag"]
            e["pandas_bags_per_s"] = obj["cpu_baseline"]["value"]
        if name == "decode" and obj.get("cpu_baseline"):
            e["pillow_tiles_per_s"] = obj["cpu_baseline"]["value"]
            e["pillow_threads"] = obj["cpu_baseline"]["cores"]
        if "gpu_ms" in obj:
            e["gpu_ms"] = obj["gpu_ms"]
        if obj.get("config", {}).get("single_bag_forward_ms") is not None:
            e["single_bag_forward_ms"] = obj["config"]["single_bag_forward_ms"]
        if obj.get("config", {}).get("value_one_stream") is not None:
            e["value_one_stream"] = obj["config"]["value_one_stream"]
        out[name] = e
    put("aggregator_f32", line if line.get("unit") == "bags/s" else None)
    for k in ("aggregator_bf16", "embedder", "embedder_half", "embedder_bf16", "train_c1", "train_c2", "slide", "slide_half", "slide_bf16", "slide_h2d", "slide_100k", "decode", "slide_jpeg", "e2e", "e2e_half", "slide_jpeg_half", "feats_csv"):
        put(k, line.get(k))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bags", type=int, default=64, help="bags per rank per pass")
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--feats", type=int, default=512)
    ap.add_argument("--patches", type=int, default=256, help="patches per rank per embedder pass (batch size)")
    ap.add_argument("--workload", default="all",
                    help="comma list of aggregator, aggregator_bf16, embedder, embedder_half, embedder_bf16, train, slide, slide_half, slide_bf16, slide_h2d, slide100k, decode, slide_jpeg, e2e, e2e_half, slide_jpeg_half, feats_csv; or all / both (= aggregator,embedder)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="lower bound on each timed region")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams independent passes are dealt to (ops.StreamPool); 1 = one pass in flight")
    ap.add_argument("--streams-bf16", type=int, default=2,
                    help="streams of the aggregator_bf16 leg when --streams > 1 (its co-resident kernels pair one batch's "
                         "logits pass with another's attend kernel)")
    ap.add_argument("--slide-patches", type=int, default=10000)
    ap.add_argument("--e2e-grid", type=int, nargs=2, default=(24, 26), help="low-magnification tile grid of the e2e slide")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the slide / embedder collectives through RCCL even with ONE rank (a one-rank nccl group): "
                         "exercises the multi-GPU code path on a one-GPU box")
    ap.add_argument("--verbose", action="store_true",
                    help="keep the long explanatory strings (*_is) and attach the committed per-kernel tables (profiles/) to the line; "
                         "the default line is compact so that a truncated stdout tail still holds every leg")
    ap.add_argument("--no-single-bag", action="store_true",
                    help="skip the single-bag latency probe (profiling runs: keeps per-kernel averages clean)")
    args = ap.parse_args()
    maybe_self_launch(args)
    wl = {"all": "aggregator,aggregator_bf16,embedder,embedder_half,embedder_bf16,train,slide,slide_half,slide_bf16,slide_h2d,slide100k,decode,slide_jpeg,e2e,e2e_half,slide_jpeg_half,feats_csv", "both": "aggregator,embedder"}.get(args.workload, args.workload)
    wl = [w for w in wl.split(",") if w]
    cx = Ctx(args)
    line = {}
    if "aggregator" in wl:
        line = aggregator_leg(cx, "c16", "f32", single_bag=not args.no_single_bag)
    subs = {}
    if "train" in wl:   # (the latency-bound legs run before the long throughput legs)
        subs["train_c1"] = train_leg(cx, "c16")
        subs["train_c2"] = train_leg(cx, "tcga")
    if "aggregator_bf16" in wl:
        subs["aggregator_bf16"] = aggregator_leg(cx, "tcga", "bf16", single_bag=False)
    if "embedder" in wl:
        subs["embedder"] = embedder_leg(cx)
    if "embedder_half" in wl:   # OPT-IN reduced precision (BASELINE.md §2, row "bf16 MFMA / f32 accumulate"): its own leg and tolerance
        subs["embedder_half"] = embedder_16_leg(cx, "half")
    if "embedder_bf16" in wl:   # OPT-IN bf16-activation trunk (round 6): its own leg and tolerance
        subs["embedder_bf16"] = embedder_16_leg(cx, "bf16")
    if "slide" in wl:
        subs["slide"] = slide_leg(cx, args.slide_patches)
    if "slide_half" in wl:   # the `slide` leg on the opt-in fp16-activation trunk (round 6: what --precision half takes)
        subs["slide_half"] = slide_leg(cx, args.slide_patches, precision="half")
    if "slide_bf16" in wl:   # ... and on bf16 activations
        subs["slide_bf16"] = slide_leg(cx, args.slide_patches, precision="bf16")
    if "slide_h2d" in wl:
        subs["slide_h2d"] = slide_leg(cx, args.slide_patches, host=True)
    if "slide100k" in wl:   # the large slide of SURVEY §8(d) config 4 (15 GB of uint8 tiles over the ranks)
        subs["slide_100k"] = slide_leg(cx, 100000, n_steps=2)
    if "decode" in wl:
        subs["decode"] = decode_leg(cx)
    if "slide_jpeg" in wl:
        subs["slide_jpeg"] = slide_jpeg_leg(cx, args.slide_patches)
    if "e2e" in wl:
        subs["e2e"] = e2e_leg(cx, tuple(args.e2e_grid))
    if "e2e_half" in wl:
        subs["e2e_half"] = e2e_leg(cx, tuple(args.e2e_grid), precision="half")
    if "slide_jpeg_half" in wl:   # (last: every stream a leg creates shifts the hardware queues of the legs behind it)
        subs["slide_jpeg_half"] = slide_jpeg_leg(cx, args.slide_patches, precision="half")
    if "feats_csv" in wl and cx.rank == 0:   # host-only: the reference's feature file of one bag (rank 0's cores)
        subs["feats_csv"] = feats_csv_leg(cx)
    if cx.rank == 0:
        if not line:   # a run without the headline leg (profiling): promote the first sub-object
            k0 = next(iter(subs))
            line = subs.pop(k0)
        line.update({"n_gpus": cx.world, "rccl_ranks": cx.world, "steps": args.steps, "warmup": args.warmup,
                     "higher_is_better": True, "scaling": line.get("scaling", "weak"), "vs_baseline": None, "data": "synthetic"})
        line.update(subs)
        if args.verbose:
            for leg, obj in (("aggregator", line if "aggregator" in wl else None), ("aggregator_bf16", line.get("aggregator_bf16")),
                             ("embedder", line.get("embedder"))):
                kt = _kernel_table(leg)
                if obj is not None and kt is not None:
                    obj["kernels"] = kt
        if not args.no_cpu_baseline and cx.world == 1:   # CPU baselines: rank 0 at N = 1 only
            if "aggregator" in wl:
                line["cpu_baseline"] = cpu_baseline_aggregator("c16", args.rows, args.feats, args.cpu_seconds)
            if "embedder" in line:
                line["embedder"]["cpu_baseline"] = cpu_baseline_embedder(args.cpu_seconds)
        order = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]
        line = {**{k: line[k] for k in order if k in line}, **{k: v for k, v in line.items() if k not in order}}
        if not args.verbose:
            _strip_prose(line)
        summ = _summary(line)
        line.setdefault("config", {})["legs"] = summ     # the driver's parsed record keeps `config` whole
        line["summary"] = summ                           # ... and the tail of stdout keeps the last key
        print(json.dumps(line), flush=True)
    if cx.dist is not None:
        cx.dist.barrier()
        cx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
