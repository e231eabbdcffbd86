#!/usr/bin/env python3
"""Round 6: the compute_feats.py bag loop end to end (compute_feats.py:58-82: glob -> decode -> embed -> write the bag's CSV) on
JPEG tiles on disk, device decode, fp16-activation trunk: the reference's pandas write in the loop against dsmil_csv_format_f32
on the writer thread.   python tools/feats_loop_time.py [bags] [tiles_per_bag]"""
import _path  # noqa: F401
import argparse
import os
import sys
import tempfile
import time
import torch
import bench
import dsmil  # noqa: F401
from dsmil_wsi_amd import pipeline as pl

n_bags = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 4000


class A:
    streams = 3
    patches = 256
    gpus = 1
    force_collective = False


sys.argv = ["bench.py"]
cx = type("C", (), {})()
cx.torch, cx.args, cx.dev, cx.rank, cx.world = torch, A, torch.device("cuda", 0), 0, 1
ic = bench._build_iclassifier(cx)
ic.embed_precision = "half"
blobs = bench._jpeg_tiles(n_tiles)
args = argparse.Namespace(batch_size=256, num_workers=8, save_npy=False, bg_threshold=None)
with tempfile.TemporaryDirectory() as root:
    bags = []
    for b in range(n_bags):
        d = os.path.join(root, "WSI", "toy", "single", "0_x", f"s{b}")
        os.makedirs(d)
        for i, blob in enumerate(blobs):
            with open(os.path.join(d, f"{i % 100}_{i // 100}.jpeg"), "wb") as fh:
                fh.write(blob)
        bags.append(d)
    pl.GPU_DECODE[0] = True
    out = os.path.join(root, "datasets", "toy")

    def run(tag):
        pl.compute_feats(args, bags[:1], ic, out)           # warm (file cache, allocator)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pl.compute_feats(args, bags, ic, out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"\n{tag}: {dt / n_bags * 1e3:.0f} ms per bag of {n_tiles} tiles ({n_bags * n_tiles / dt:.0f} patches/s through the whole loop)", flush=True)

    run("dsmil_csv_format_f32 on the writer thread")
    real = pl.FeatWriter.save

    def pandas_in_the_loop(self, feats, path, npy=False):
        import pandas as pd
        os.makedirs(os.path.dirname(path), exist_ok=True)
        pd.DataFrame(feats.detach().cpu().numpy()).to_csv(path, index=False, float_format="%.4f")
    pl.FeatWriter.save = pandas_in_the_loop
    run("pandas to_csv in the loop (the reference's call; round 1-5 of this repo)")
    pl.FeatWriter.save = real
