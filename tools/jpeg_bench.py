#!/usr/bin/env python3
"""Throughput of the device JPEG decoder on synthetic 224 x 224 tiles at the tiler's quality 70 (deepzoom_tiler.py:250), beside
Pillow on the host.     python tools/jpeg_bench.py [n_tiles] [batch]"""
import _path  # noqa: F401
import io
import sys
import time
import numpy as np
import torch
from PIL import Image
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops, _native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rng = np.random.default_rng(0)


def tile(i):
    base = rng.integers(0, 256, (30, 30, 3)).repeat(8, 0).repeat(8, 1)[:224, :224]
    a = np.clip(base + rng.normal(0, 12, (224, 224, 3)), 0, 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", quality=70)
    return b.getvalue()


uniq = [tile(i) for i in range(256)]
blobs = [uniq[i % 256] for i in range(n)]
print(f"{n} tiles, {sum(len(b) for b in blobs) / n / 1024:.1f} KiB per tile")
t0 = time.perf_counter()
for b in blobs[:512]:
    np.array(Image.open(io.BytesIO(b)).convert("RGB"))
dt = time.perf_counter() - t0
print(f"Pillow, one host thread: {512 / dt:.0f} tiles/s")
L = _native.lib()
dev = torch.device("cuda")
for bs in (256, batch, n):
    chunks = [blobs[i:i + bs] for i in range(0, n, bs)]
    # host part (parse + concatenate) and device part timed separately
    t0 = time.perf_counter()
    parsed = [ops.jpeg_parse(c) for c in chunks]
    t_parse = time.perf_counter() - t0
    d = [(torch.from_numpy(p[0]).to(dev), torch.from_numpy(p[1]).to(dev), len(c), int(p[0].size) - 32) for p, c in zip(parsed, chunks)]
    outs = [torch.empty((m, 224, 224, 3), dtype=torch.uint8, device=dev) for _, _, m, _ in d]
    sts = [torch.empty(m, dtype=torch.int32, device=dev) for _, _, m, _ in d]
    ws = torch.empty(L.dsmil_jpeg_workspace_bytes(bs, 224, 224, max(x[3] for x in d)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for (dd, pp, m, nb), o, s in zip(d, outs, sts):
            rc = L.dsmil_jpeg_decode(dd.data_ptr(), nb, pp.data_ptr(), m, 224, 224, o.data_ptr(), s.data_ptr(), ws.data_ptr(), ws.numel(), st)
            assert rc == 0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert all(int(s.abs().sum()) == 0 for s in sts)
    t0 = time.perf_counter()
    full = [ops.jpeg_decode(c, dev) for c in chunks]
    torch.cuda.synchronize()
    t_full = time.perf_counter() - t0
    print(f"batch {bs}: device decode {n / dt:.0f} tiles/s ({dt / len(chunks) * 1e3:.2f} ms per batch); host parse {n / t_parse:.0f} tiles/s; "
          f"ops.jpeg_decode end to end (parse + H2D + decode + status read) {n / t_full:.0f} tiles/s", flush=True)
