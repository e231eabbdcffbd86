"""The multi-GPU code path on ONE GPU: a one-rank `nccl` process group + dist.force_collective(True) sends the slide's
tensors through the REAL RCCL branch of dist.all_gather_rows_sized / all_gather_packed (a world of one otherwise returns
early and never loads RCCL), and `bench.py --gpus 1 --force-collective` runs under torch.distributed.run exactly as the
driver launches N > 1.  2 / 4 / 8-GPU behaviour stays unmeasured here; what this pins is that the collective branch
executes, on its side stream, with the packing and the int64 bit-casts intact."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_SCRIPT = r'''
import os, sys
sys.path[:0] = [os.environ["DSMIL_ROOT"], os.path.join(os.environ["DSMIL_ROOT"], "tests")]
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import dsmil  # noqa
from dsmil_wsi_amd import dist as dd
x = torch.randn(1000, 512, device="cuda"); c = torch.randn(1000, 2, device="cuda")
pos = torch.randint(-2**40, 2**40, (1000, 2), device="cuda", dtype=torch.int64)
assert dd.all_gather_rows(x, 1000) is x                       # a world of one: no collective by default
dd.force_collective(True)
y = dd.all_gather_rows(x, 1000)
assert y is not x and torch.equal(y, x)                       # went through all_gather_into_tensor
f, cc, pp = dd.all_gather_packed([x, c, pos], [1000])
assert torch.equal(f, x) and torch.equal(cc, c) and torch.equal(pp, pos) and pp.dtype == torch.int64
torch.cuda.synchronize()
dist.barrier(); dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
'''


def _env():
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), DSMIL_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_collective_branch_runs_through_rccl_with_one_rank():
    out = subprocess.run([sys.executable, "-c", _SCRIPT], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL_ONE_RANK_OK" in out.stdout, out.stderr[-2000:]


def test_bench_under_torchrun_one_rank_with_forced_collectives():
    """The driver's N > 1 launch line with N = 1: torch.distributed.run -> bench.py --gpus 1, slide leg with the packed
    all-gather forced through RCCL."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "slide",
           "--slide-patches", "1024", "--steps", "2", "--warmup", "1", "--force-collective", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["collectives_per_slide"] == 1 and line["config"]["all_gather_bytes_per_rank"] == 1024 * 514 * 4
    assert line["value"] > 0
