#!/usr/bin/env python3
"""Round 6: host and stream timeline of pipeline.embed_jpeg_blobs on the fp16-activation trunk — when the host enters and
leaves every _ChunkDecoder.begin / end, when the decode stream reaches each decode and when it is done (HIP events).  This is
what showed the host held inside begin() by a pageable copy queued behind the staging buffer's free-wait.
python tools/jpeg_host_timeline.py"""
import _path  # noqa: F401
import sys
import time
import torch
import bench
import dsmil  # noqa: F401
from dsmil_wsi_amd import ops, pipeline as pl
class A: streams=3; patches=256; gpus=1; force_collective=False
sys.argv=["bench.py"]
cx=type("C",(),{})(); cx.torch,cx.args,cx.dev,cx.rank,cx.world=torch,A,torch.device("cuda",0),0,1
ic=bench._build_iclassifier(cx); ic.embed_precision="half"
blobs=bench._jpeg_tiles(10000); dev=cx.dev
log=[]; T0=[0.0]; base=[None]
ob, oe = pl._ChunkDecoder.begin, pl._ChunkDecoder.end
def begin(self, blobs):
    t0=time.perf_counter()
    ds = self.dss[self.n % len(self.dss)]
    e0=torch.cuda.Event(enable_timing=True); e0.record(ds)
    r=ob(self, blobs)
    e1=torch.cuda.Event(enable_timing=True); e1.record(ds)
    log.append(("begin", self.n-1, t0-T0[0], time.perf_counter()-T0[0], e0, e1))
    return r
def end(self, item):
    t0=time.perf_counter(); r=oe(self,item); log.append(("end", None, t0-T0[0], time.perf_counter()-T0[0], None, None)); return r
pl._ChunkDecoder.begin, pl._ChunkDecoder.end = begin, end
with torch.no_grad():
    pl.embed_jpeg_blobs(ic, blobs, 256, 2048, 3, dev); torch.cuda.synchronize()
    log.clear()
    base[0]=torch.cuda.Event(enable_timing=True); base[0].record(); torch.cuda.synchronize()
    T0[0]=time.perf_counter()
    pl.embed_jpeg_blobs(ic, blobs, 256, 2048, 3, dev); torch.cuda.synchronize()
    print(f"total {1e3*(time.perf_counter()-T0[0]):.1f} ms")
for kind,n,a,b,e0,e1 in log:
    g = f"   GPU: stream reaches it at {base[0].elapsed_time(e0):7.1f} ms, decode done at {base[0].elapsed_time(e1):7.1f} ms" if e0 is not None else ""
    print(f"{kind:5s} {'' if n is None else n}: host {1e3*a:7.1f} -> {1e3*b:7.1f} ms{g}")
