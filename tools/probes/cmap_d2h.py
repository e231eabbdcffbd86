"""Where the 18 ms of pipeline.attention_colormap go: upsample on the GPU, pageable / pinned D2H, pinned allocation."""
import time
import numpy as np
import torch

dev = torch.device("cuda:0")
small = torch.randint(0, 256, (96, 104, 3), dtype=torch.uint8)


def T():
    torch.cuda.synchronize()
    return time.perf_counter()


keep = []
for it in range(4):
    t0 = T()
    t = small.to(dev)
    up = t.repeat_interleave(32, dim=0).repeat_interleave(32, dim=1)
    t1 = T()
    a = up.cpu()
    t2 = T()
    host = torch.empty(up.shape, dtype=torch.uint8, pin_memory=True)
    t3 = T()
    host.copy_(up, non_blocking=True)
    t4 = T()
    n = host.numpy()
    b = n.copy()
    t5 = T()
    keep.append(n)      # (as the pipeline's caller does: the previous map is alive while the next is made)
    if len(keep) > 1:
        keep.pop(0)
    print("upsample %.2f  pageable .cpu() %.2f  pinned alloc %.2f  pinned copy %.2f  host memcpy %.2f ms" %
          tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)))
pre = torch.empty(up.shape, dtype=torch.uint8, pin_memory=True)
for it in range(3):
    t0 = T()
    pre.copy_(up, non_blocking=True)
    t1 = T()
    print("reused pinned buffer copy %.2f ms" % (1e3 * (t1 - t0)))
