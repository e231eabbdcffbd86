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


# A world of ONE rank needs no exchange, and the functions below return their input untouched in that case — which also
# means that a single-GPU run never loads RCCL.  `force_collective(True)` (or DSMIL_FORCE_COLLECTIVE=1) makes them take
# the real collective branch whenever a process group exists, even with one rank: tests/test_dist_gpu.py and
# `bench.py --force-collective` use it to push tensors through RCCL on a one-GPU box, so that the multi-GPU path has
# been executed before it first meets 8 GPUs.
_FORCE = [os.environ.get("DSMIL_FORCE_COLLECTIVE", "0") not in ("", "0")]


def force_collective(on=True):
    prev = _FORCE[0]
    _FORCE[0] = bool(on)
    return prev


def _skip_collective(world):
    return world == 1 and not (_FORCE[0] and dist.is_available() and dist.is_initialized())


_comm_streams = {}


def _comm_stream(device):
    """One side stream per device for the slide's collective: it is enqueued behind everything the caller's stream has
    submitted (the last embedder batch) and the caller's stream waits for it, so host-side work that follows the
    submission (and any kernels of OTHER streams) overlaps the transfer instead of queueing behind it."""
    key = str(device)
    st = _comm_streams.get(key)
    if st is None:
        st = _comm_streams[key] = torch.cuda.Stream(device=device)
    return st


def all_gather_rows(local, n_total, group=None):
    """Reassemble [n_total, D] from per-rank contiguous shards (`shard_range` order) with ONE
    collective.  Shards differ by at most one row, so they are padded to the common maximum and
    sent as one equal-size all-gather (RCCL: a single direct all-gather over the xGMI mesh);
    the pad rows are dropped on arrival."""
    world, rank = world_rank(group)
    if _skip_collective(world):
        assert local.shape[0] == n_total
        return local
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    return all_gather_rows_sized(local, sizes, group)


def all_gather_rows_sized(local, sizes, group=None):
    """all_gather_rows for arbitrary per-rank row counts (`sizes[r]` rows on rank r, rank order = row order):
    still ONE equal-size collective, shards padded to the largest."""
    world, rank = world_rank(group)
    if _skip_collective(world):
        return local
    assert len(sizes) == world and local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    D, mx = local.shape[1], max(sizes)
    padded = local
    if local.shape[0] < mx:
        padded = torch.zeros((mx, D), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    padded = padded.contiguous()
    if dist.get_backend(group) == "nccl":
        cur = torch.cuda.current_stream(local.device)
        side = _comm_stream(local.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = torch.empty((world * mx, D), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(out, padded, group=group)
        cur.wait_stream(side)
        padded.record_stream(side)
        out.record_stream(cur)
        if all(s == mx for s in sizes):
            return out
        return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([bufs[r][: sizes[r]] for r in range(world)], dim=0)


def all_gather_packed(parts, sizes, group=None):
    """ONE collective for several per-row tensors of a slide (feature rows and instance logits; tree rows and grid
    positions): each [n_r, D_i] part is bit-cast to 32-bit lanes (an int64 column travels as two float lanes, exact),
    the parts are concatenated along the row, all-gathered as one [n_r, sum D_i] matrix and cut apart again.
    Returns the list of gathered tensors with their original dtypes."""
    world, _ = world_rank(group)
    if _skip_collective(world):
        return list(parts)
    lanes, cols = [], []
    for t in parts:
        assert t.dim() == 2 and t.element_size() in (4, 8), (t.shape, t.dtype)
        v = t.contiguous().view(torch.float32) if t.dtype != torch.float32 else t
        lanes.append(v)
        cols.append(v.shape[1])
    packed = all_gather_rows_sized(torch.cat(lanes, dim=1), sizes, group)
    out, c0 = [], 0
    for t, c in zip(parts, cols):
        piece = packed[:, c0:c0 + c].contiguous()
        out.append(piece if t.dtype == torch.float32 else piece.view(t.dtype))
        c0 += c
    return out


def embed_rows_sharded(embed_fn, n_total, group=None):
    """Run `embed_fn(lo, hi) -> [hi-lo, D]` on this rank's row range and all-gather the rows."""
    world, rank = world_rank(group)
    lo, hi = shard_range(n_total, rank, world)
    local = embed_fn(lo, hi)
    return all_gather_rows(local, n_total, group)


# ---------------------------------------------------------------------------------------------
# ONE bag sharded by instances: exchange C*(2+K) floats twice instead of
# all-gathering the [N,K] feature rows
# ---------------------------------------------------------------------------------------------
def _gather_list(t, group=None):
    """All-gather equal-shape tensors into a python list (world 1: [t])."""
    world, _ = world_rank(group)
    if world == 1:
        return [t]
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t.contiguous(), group=group)
    return bufs


def sharded_bag_forward(milnet, feats_local, row_offset, group=None, gather=_gather_list):
    """MILNet.forward (dsmil.py:70-74) for a bag whose instance rows are spread over the ranks:
    this rank holds ``feats_local`` = rows [row_offset, row_offset + n_local) of the bag.

    Returns (classes_local [n_local,C], pred [1,C], A_local [n_local,C], B [1,C,K], idx [C]) — the local
    slices of what the reference returns for the whole bag, plus the bag-wide critical indices;
    ``pred``, ``B`` and ``idx`` are identical on every rank.  Two small exchanges (per class: the
    shard's best logit, its global row index and that feature row; then the shard's softmax
    statistics and un-normalised value sum).  CUDA tensors run dsmil_agg_shard_* natively; CPU tensors
    (gloo tests) run the same arithmetic with torch ops.  ``gather`` is injectable for single-process
    tests that play several ranks in turn."""
    import torch.nn.functional as F
    from . import ops
    ic, bc = milnet.i_classifier, milnet.b_classifier
    if bc.passing_v:
        raise NotImplementedError("instance sharding is implemented for v = Identity (every reference script)")
    x = feats_local
    n_local, K = x.shape
    lin = ic.fc[0]
    w = {k: (v.detach() if v is not None else None) for k, v in bc._weights().items()}
    w["fc_w"], w["fc_b"] = lin.weight.detach(), lin.bias.detach()
    C = w["fcc_w"].shape[0]
    native = x.is_cuda
    ar = torch.arange(C, device=x.device)
    lanes = 8 // x.element_size()   # float lanes that carry one int64 index, bit for bit
    with torch.no_grad():
        # ---- 1. local instance logits and the shard's best row per class.  A rank with NO rows (a bag
        #         smaller than the world size) contributes (-inf, index past every real row, zero row).
        if n_local == 0:
            classes = x.new_zeros((0, C))
            best_val = x.new_full((C,), float("-inf"))
            gbest = torch.full((C,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=x.device)
            rows = x.new_zeros((C, K))
        else:
            if native:
                classes, best_val, best_idx = ops.agg_shard_argmax(x, w, nonlinear=bc.nonlinear)
            else:
                classes = F.linear(x, w["fc_w"], w["fc_b"])
                best_idx = torch.argmax(classes, dim=0)   # lowest index on ties
                best_val = classes[best_idx, ar]
            gbest = best_idx.to(torch.int64) + int(row_offset)
            rows = x[best_idx]                                                # [C,K]
        # one message per class: best logit | global row index (int64 bits in float lanes: exact for any
        # bag size, a float32 VALUE would round above 2^24 rows) | that feature row
        msg = torch.cat([best_val[:, None], gbest.contiguous().view(x.dtype).view(C, lanes), rows], dim=1)
        allmsg = torch.stack(gather(msg, group))                              # [R, C, 1+lanes+K]
        vals_ = allmsg[:, :, 0]
        gidx = allmsg[:, :, 1:1 + lanes].contiguous().view(torch.int64).view(allmsg.shape[0], C)
        # bag-wide winner per class: larger value, then lower global index (dsmil.py:52 + our tie rule)
        best_r = torch.zeros(C, dtype=torch.long, device=x.device)
        for r in range(1, allmsg.shape[0]):
            cur_v = vals_[best_r, ar]
            cur_i = gidx[best_r, ar]
            better = (vals_[r] > cur_v) | ((vals_[r] == cur_v) & (gidx[r] < cur_i))
            best_r = torch.where(better, torch.full_like(best_r, r), best_r)
        crit_rows = allmsg[best_r, ar, 1 + lanes:].contiguous()               # [C,K]
        idx = gidx[best_r, ar]
        # ---- 2. this shard's attention against the bag-wide critical rows
        if n_local == 0:
            A_un = x.new_zeros((0, C))
            ml = torch.stack([x.new_full((C,), float("-inf")), x.new_zeros((C,))], dim=1)
            B_un = x.new_zeros((C, K))
        elif native:
            A_un, ml, B_un = ops.agg_shard_attend(x, w, crit_rows, nonlinear=bc.nonlinear)
        else:
            q = bc.q
            Q = q(x)
            qmax = q(crit_rows)
            s = Q.mm(qmax.t()) / (Q.shape[1] ** 0.5)
            m = s.max(dim=0).values
            A_un = torch.exp(s - m)
            ml = torch.stack([m, A_un.sum(0)], dim=1)
            B_un = A_un.t().mm(x)
        stat = torch.stack(gather(torch.cat([ml, B_un], dim=1), group))       # [R, C, 2+K]
        m_all, l_all, B_all = stat[:, :, 0], stat[:, :, 1], stat[:, :, 2:]
        m = m_all.max(dim=0).values
        wr = torch.exp(m_all - m)                                             # [R, C]
        l = (l_all * wr).sum(0)
        B = (B_all * wr[:, :, None]).sum(0) / l[:, None]                      # [C,K]
        A = A_un * (torch.exp(ml[:, 0] - m) / l)[None, :]
        pred = bc.fcc(B.unsqueeze(0)).view(1, -1)                             # dsmil.py:60-61
    return classes, pred, A, B.unsqueeze(0), idx
