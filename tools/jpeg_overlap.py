#!/usr/bin/env python3
"""Where does the time of pipeline.embed_jpeg_blobs go?  Host timestamps around every decode / embed enqueue of one slide."""
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
blobs = bench._jpeg_tiles(10240)
dev = cx.dev
# warm
pl.embed_jpeg_blobs(ic, blobs[:4096], 256, 2048, 3, dev)
torch.cuda.synchronize()
for mode in ("decode only", "embed only", "pipelined"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "decode only":
        for i in range(0, len(blobs), 2048):
            ops.jpeg_decode(blobs[i:i + 2048], dev)
    elif mode == "embed only":
        imgs = ops.jpeg_decode(blobs[:2048], dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(0, len(blobs), 2048):
            pl.embed_tiles(ic, imgs, 256, streams=3, device=dev)
    else:
        pl.embed_jpeg_blobs(ic, blobs, 256, 2048, 3, dev)
    torch.cuda.synchronize()
    print(f"{mode}: {(time.perf_counter() - t0) * 1e3:.1f} ms for {len(blobs)} tiles", flush=True)
# host-side pieces of one decode
t0 = time.perf_counter(); data, plan, recs = ops.jpeg_parse(blobs[:2048]); t1 = time.perf_counter()
d = torch.from_numpy(data).to(dev); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"parse + concat {1e3 * (t1 - t0):.1f} ms, H2D of {data.nbytes / 1e6:.1f} MB pageable {1e3 * (t2 - t1):.1f} ms")
