"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm)
or gloo on CPU.  The reference has no distributed code at all; this is the sharding that
BASELINE.json's north_star asks for:

* a slide's ordered patch list is cut into contiguous per-rank row ranges (InstanceNorm is per
  image, so a row does not depend on which rank embedded it), and ONE all-gather of the
  [N_r, 512] feature rows reassembles the bag in the original order on every rank;
* bags are independent units and are dealt round-robin to ranks — no collective on that path.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun-style env vars (no-op for world 1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend)
    return world


def world_rank(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def shard_range(n, rank, world):
    """Contiguous row range [lo, hi) of rank `rank` out of `n` ordered rows."""
    return (rank * n) // world, ((rank + 1) * n) // world


def shard_bags(n_bags, rank, world):
    """Round-robin bag indices for this rank."""
    return list(range(rank, n_bags, world))


def all_gather_rows(local, n_total, group=None):
    """Reassemble [n_total, D] from per-rank contiguous shards (`shard_range` order) with ONE
    collective.  Shards differ by at most one row, so they are padded to the common maximum and
    sent as one equal-size all-gather (RCCL: a single direct all-gather over the xGMI mesh);
    the pad rows are dropped on arrival."""
    world, rank = world_rank(group)
    if world == 1:
        assert local.shape[0] == n_total
        return local
    D = local.shape[1]
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    mx = max(sizes)
    padded = local
    if local.shape[0] < mx:
        padded = torch.zeros((mx, D), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    padded = padded.contiguous()
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world * mx, D), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, padded, group=group)
        parts = [out[r * mx: r * mx + sizes[r]] for r in range(world)]
    else:
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(bufs, padded, group=group)
        parts = [bufs[r][: sizes[r]] for r in range(world)]
    if all(s == mx for s in sizes) and dist.get_backend(group) == "nccl":
        return out
    return torch.cat(parts, dim=0)


def embed_rows_sharded(embed_fn, n_total, group=None):
    """Run `embed_fn(lo, hi) -> [hi-lo, D]` on this rank's row range and all-gather the rows."""
    world, rank = world_rank(group)
    lo, hi = shard_range(n_total, rank, world)
    local = embed_fn(lo, hi)
    return all_gather_rows(local, n_total, group)
