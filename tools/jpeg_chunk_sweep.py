#!/usr/bin/env python3
"""Round 6: pipeline.embed_jpeg_blobs over decode-chunk sizes, on the fp32-class trunk and the opt-in fp16-activation trunk
(one 10 000-tile slide from JPEG bytes in host memory), beside the latency of ONE decode of each chunk size (the Huffman
kernel runs one lane per tile: its time is the longest tile stream's, almost independent of the tile count).
python tools/jpeg_chunk_sweep.py"""
import _path  # noqa: F401
import sys
import time
import torch
import bench
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops, pipeline as pl


class A:
    streams = 3
    patches = 256
    gpus = 1
    force_collective = False


sys.argv = ["bench.py"]
cx = type("C", (), {})()
cx.torch, cx.args, cx.dev, cx.rank, cx.world = torch, A, torch.device("cuda", 0), 0, 1
ic = bench._build_iclassifier(cx)
blobs = bench._jpeg_tiles(10000)
dev = cx.dev
sizes = (512, 1024, 2048, 3072, 4096, 5120, 10000)
for n in sizes:
    ops.jpeg_decode(blobs[:n], dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ops.jpeg_decode(blobs[:n], dev)
        torch.cuda.synchronize()
    print(f"decode alone, {n:5d} tiles: {(time.perf_counter() - t0) / 3 * 1e3:6.1f} ms", flush=True)
for prec in ("fp32", "half"):
    ic.embed_precision = prec
    for n in sizes[1:]:
        with torch.no_grad():
            pl.embed_jpeg_blobs(ic, blobs, 256, n, 3, dev)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                pl.embed_jpeg_blobs(ic, blobs, 256, n, 3, dev)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
        print(f"{prec}: decode_batch {n:5d}: {min(ts) * 1e3:6.1f} ms per slide (median {sorted(ts)[1] * 1e3:6.1f}) = {len(blobs) / min(ts):8.0f} patches/s", flush=True)
