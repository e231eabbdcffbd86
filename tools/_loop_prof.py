import _path, argparse, os, sys, tempfile, time, cProfile, pstats, torch, bench, dsmil
from dsmil_wsi_amd import pipeline as pl
class A: streams=3; patches=256; gpus=1; force_collective=False
sys.argv=["bench.py"]
cx=type("C",(),{})(); cx.torch,cx.args,cx.dev,cx.rank,cx.world=torch,A,torch.device("cuda",0),0,1
ic=bench._build_iclassifier(cx); ic.embed_precision="half"
blobs=bench._jpeg_tiles(4000)
args=argparse.Namespace(batch_size=256,num_workers=8,save_npy=False,bg_threshold=None)
with tempfile.TemporaryDirectory() as root:
    bags=[]
    for b in range(3):
        d=os.path.join(root,"WSI","toy","single","0_x",f"s{b}"); os.makedirs(d)
        for i,blob in enumerate(blobs):
            with open(os.path.join(d,f"{i%100}_{i//100}.jpeg"),"wb") as fh: fh.write(blob)
        bags.append(d)
    pl.GPU_DECODE[0]=True
    out=os.path.join(root,"datasets","toy")
    pl.compute_feats(args,bags[:1],ic,out); torch.cuda.synchronize()
    pr=cProfile.Profile(); pr.enable()
    t0=time.perf_counter(); pl.compute_feats(args,bags,ic,out); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    pr.disable()
    print(f"\n{dt/3*1e3:.0f} ms per bag")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
