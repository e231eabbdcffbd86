#!/usr/bin/env python3
"""Round 6: the bf16 batch pass on 1-4 streams over DISTINCT batches (as bench.py deals them), for A/B runs of the co-resident
kernels:   python tools/coresident_ab.py [logits_form 0|1|2]      (DSMIL_NATIVE_LIB / DSMIL_EXPT select experiment builds)"""
import _path  # noqa: F401
import sys
import time
import torch
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops, _native
from dsmil_wsi_amd.synthetic import load_weights

form = int(sys.argv[1]) if len(sys.argv) > 1 else 1
_native.lib().dsmil_agg_logits_form(form)
dev = torch.device("cuda:0")
w = {k: torch.from_numpy(v).to(dev) for k, v in load_weights("tcga").items()}
nb, N, K = 64, 10000, 512
g = torch.Generator(device=dev).manual_seed(1234)
batches = [torch.randn((nb * N, K), generator=g, device=dev).to(torch.bfloat16) for i in range(3)]
lengths = [N] * nb
offsets = ops.offsets_tensor(lengths, dev)
res = []
for S in (1, 2, 3):
    pool = ops.StreamPool(S)
    for r in (24, 240):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(r):
            fb = batches[i % 3]
            pool.run(lambda: ops.agg_forward(fb, lengths, w, offsets=offsets))
        pool.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res.append(f"S{S} {nb * r / dt / 1e3:.1f}k")
print(f"logits_form {form}: " + "  ".join(res), flush=True)
