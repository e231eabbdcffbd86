#!/usr/bin/env python3
"""Round 6: a short run of the bf16 / fp32 batch pass on S streams for a rocprofv3 --kernel-trace timeline
(which kernels of which stream run beside the persistent attend kernel).   python tools/overlap_run.py bf16|agg S rounds"""
import _path  # noqa: F401
import sys
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops
from dsmil_wsi_amd.synthetic import load_weights

which = sys.argv[1] if len(sys.argv) > 1 else "bf16"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")
w = {k: torch.from_numpy(v).to(dev) for k, v in load_weights("c16" if which == "agg" else "tcga").items()}
nb, N, K = 64, 10000, 512
g = torch.Generator(device=dev).manual_seed(1234)
batches = []
for i in range(3):          # distinct batches, as bench.py deals them: no pass re-reads what the pass in front left in the caches
    f = torch.randn((nb * N, K), generator=g, device=dev)
    batches.append(f.to(torch.bfloat16) if which == "bf16" else f)
lengths = [N] * nb
offsets = ops.offsets_tensor(lengths, dev)
pool = ops.StreamPool(S)
for r in (6, rounds):
    torch.cuda.synchronize()
    for i in range(r):
        fb = batches[i % 3]
        pool.run(lambda: ops.agg_forward(fb, lengths, w, offsets=offsets))
    pool.join()
    torch.cuda.synchronize()
