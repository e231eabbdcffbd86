#!/usr/bin/env python3
"""Experiment: read bandwidth of a repeatedly-read buffer versus its size (is the 256 MiB Infinity Cache serving re-reads?).
Uses the library's own HBM-bound kernel (k_logits_stream through dsmil_fc... no: the bf16 aggregator's logits pass cannot be
called alone), so a plain torch reduction stands in: bf16 sum over the buffer."""
import time
import torch
dev = torch.device("cuda:0")
for mb in (16, 32, 64, 96, 128, 160, 192, 256, 384, 512, 1024):
    n = mb * 1024 * 1024 // 2
    x = torch.ones(n, dtype=torch.bfloat16, device=dev)
    xv = x.view(torch.int32)
    for _ in range(3):
        xv.sum()
    torch.cuda.synchronize()
    reps = max(5, 4096 // mb)
    t0 = time.perf_counter()
    for _ in range(reps):
        xv.sum()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{mb:5d} MB: {dt * 1e6:8.1f} us per pass = {mb * 1.048576e6 / dt / 1e12:6.2f} TB/s", flush=True)
    del x, xv
