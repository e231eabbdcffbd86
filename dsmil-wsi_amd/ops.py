"""Host-side operators over the C-ABI (include/dsmil_hip.h) for CUDA(HIP) tensors.

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic of the forward
hot path happens in libdsmil_hip.so.  Functions raise if handed CPU tensors — the CPU route of
the nn.Modules (config 0, MUSK1 plumbing) never comes through this file.
"""
import collections
import ctypes

import numpy as np
import torch

from . import _native

Q_DIM = 128

_ws_cache = {}
_off_cache = collections.OrderedDict()


class _LRU:
    """Tiny keyed cache for derived device buffers (packed / folded weights).  Keys carry the device and
    the (data_ptr, _version) of every source tensor; each entry also holds the sources themselves, so a
    freed tensor's address can never come back as a different model's weight while its entry lives.
    A handful of entries: tree mode alternates two embedders, DDP-less multi-device processes hold one
    model per device."""

    def __init__(self, cap=8):
        self.cap = cap
        self.d = collections.OrderedDict()

    def get(self, key):
        v = self.d.get(key)
        if v is not None:
            self.d.move_to_end(key)
        return v

    def put(self, key, value):
        self.d[key] = value
        self.d.move_to_end(key)
        while len(self.d) > self.cap:
            self.d.popitem(last=False)
        return value


def _tkey(t):
    return None if t is None else (t.data_ptr(), t._version)


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_ws_last = [None]


def _workspace(device, nbytes):
    """Scratch for one native call, cached per (device, HIP stream): calls on different streams may run concurrently
    (StreamPool) and must not share partial buffers.  Allocated while that stream is current, so the caching
    allocator ties its lifetime to the stream's work."""
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if len(_ws_cache) >= 16:   # transient streams: forget the oldest entries (their memory returns to the allocator)
            for k in list(_ws_cache)[:8]:
                del _ws_cache[k]
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    _ws_last[0] = buf
    return buf


class _Ready:
    """Marks device data produced on one stream (a packed weight image) so that OTHER streams wait for it before use —
    an event wait enqueued on the consumer's stream, no host sync (the packing of a training step stays asynchronous)."""

    def __init__(self, dev):
        self.ev = None
        if torch.device(dev).type != "cuda":   # host data (the CPU module path folds BatchNorms too): nothing to wait for
            return
        self.sid = int(torch.cuda.current_stream(dev).cuda_stream)
        self.ev = torch.cuda.Event()
        self.ev.record(torch.cuda.current_stream(dev))

    def wait(self, dev, *buffers):
        """Make the current stream wait for the producer; `buffers` (the cached device tensors about to be read) are
        marked as in use on the current stream, so that an entry evicted from the cache while another stream still reads
        it is not handed back to the producing stream's allocator pool early."""
        if self.ev is None:
            return
        cur = torch.cuda.current_stream(dev)
        if int(cur.cuda_stream) != self.sid:
            cur.wait_event(self.ev)
            for b in buffers:
                if b is not None and b.is_cuda:
                    b.record_stream(cur)


class StreamPool:
    """Round-robin dispatch of INDEPENDENT native calls over n HIP streams of one device.

    The aggregator forward is a chain of kernels with complementary bounds — `k_logits_stream` streams the bag at
    HBM speed with idle matrix cores, `k_query_attend_split` is MFMA-bound at half the HBM rate — and a batch of bags
    cannot overlap them (the critical instance must be known before any score).  Two INDEPENDENT batches can: dealt to
    different streams, the logits pass of one runs under the attend kernel of the other (measured on 64 x 10 000 x 512
    bags: 73.4 k bags/s on one stream, 78.2 k on two, 80.0 k on three; round 6, bf16 bags: the persistent attend kernel
    leaves one 80-register slot per SIMD and the logits / q_max / combine kernels are cut to it — 190 k bags/s on one
    stream, 250-260 k on two or three, `dsmil_agg_logits_form`).  Each stream has its own workspace
    (`_workspace`); outputs are allocated on the stream that produces them; `join()` makes the caller's stream wait
    for everything submitted.

        pool = ops.StreamPool(3)
        outs = [pool.run(ops.agg_forward, feats_b, lengths_b, w) for (feats_b, lengths_b) in batches]
        pool.join()
    """

    def __init__(self, n=3, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(max(1, int(n)))]
        self._i = 0

    def run(self, fn, *args, **kwargs):
        s = self.streams[self._i % len(self.streams)]
        self._i += 1
        s.wait_stream(torch.cuda.current_stream(self.device))   # inputs may have been produced on the caller's stream
        with torch.cuda.stream(s):
            return fn(*args, **kwargs)

    def join(self):
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)


def offsets_tensor(lengths, device):
    """Device int64 [n_bags+1] prefix offsets for a tuple of bag lengths (cached: a repeated bag
    shape costs no H2D copy, keeping MILNet.forward free of host syncs)."""
    key = (str(device), tuple(int(n) for n in lengths))
    t = _off_cache.get(key)
    if t is None:
        off = np.zeros(len(key[1]) + 1, np.int64)
        np.cumsum(np.asarray(key[1], np.int64), out=off[1:])
        t = torch.from_numpy(off).to(device)
        _off_cache[key] = t
        if len(_off_cache) > 256:
            _off_cache.popitem(last=False)
    else:
        _off_cache.move_to_end(key)
    return t


def _f32c(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA(HIP) tensor for the native path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def fc_forward(feats, fc_w, fc_b):
    """FCLayer / IClassifier.fc on the GPU: dsmil.py:11,24."""
    feats = _f32c(feats, "feats"); fc_w = _f32c(fc_w, "fc_w"); fc_b = _f32c(fc_b, "fc_b")
    N, K = feats.shape
    C = fc_w.shape[0]
    out = torch.empty((N, C), dtype=torch.float32, device=feats.device)
    if N == 0:
        return out
    with torch.cuda.device(feats.device):
        rc = _native.lib().dsmil_fc_forward(_ptr(feats), N, K, C, _ptr(fc_w), _ptr(fc_b), _ptr(out),
                                            _stream(feats.device))
    _native.check(rc, "dsmil_fc_forward")
    return out


_bf16_cache = _LRU()
_split_cache = _LRU()


def _bf16_params(w, nonlinear, dev):
    """bf16 path: every weight/bias rounded to bf16 (what module.bfloat16() would hold), kept as
    fp32 tensors for the f32-accumulating stages + the packed bf16 MFMA operands.  Cached per
    parameter set (data_ptr, _version); the entry keeps the source tensors alive so that a freed
    parameter's address can never be mistaken for a new one with the same version count."""
    names = ("fc_w", "fc_b", "q0_w", "q0_b", "q2_w", "q2_b", "fcc_w", "fcc_b")
    key = (str(dev), bool(nonlinear)) + tuple(_tkey(w.get(k)) for k in names)
    ent = _bf16_cache.get(key)
    if ent is not None:
        ent[3].wait(dev, ent[1], *[t for t in ent[0].values() if t is not None])
        return ent[0], ent[1]
    r = {k: (w[k].detach().to(torch.bfloat16).to(torch.float32).contiguous() if w.get(k) is not None else None)
         for k in names}
    L = _native.lib()
    K = r["q0_w"].shape[1]
    packed = torch.empty(L.dsmil_agg_packed_bf16_bytes(K), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.dsmil_agg_pack_bf16(_ptr(r["q0_w"]), _ptr(r["q2_w"] if nonlinear else None), K, _ptr(packed),
                                   _stream(dev))
    _native.check(rc, "dsmil_agg_pack_bf16")
    _bf16_cache.put(key, (r, packed, [w.get(k) for k in names], _Ready(dev)))
    return r, packed


def _split_params(q0_w, q2_w, nonlinear, dev):
    """fp32 path, MFMA forms 6 / 9: the query weights cut into three bf16 planes in MFMA-fragment order
    (dsmil_agg_pack_split), prepared once per weight set; None for form 0 (f32 MFMA reads the weights as
    they are)."""
    L = _native.lib()
    if L.dsmil_agg_mlp_form() == 0:
        return None
    key = (str(dev), bool(nonlinear), _tkey(q0_w), _tkey(q2_w) if nonlinear else None)
    ent = _split_cache.get(key)
    if ent is not None:
        ent[2].wait(dev, ent[0])
        return ent[0]
    K = q0_w.shape[1]
    packed = torch.empty(L.dsmil_agg_packed_split_bytes(K, 1 if nonlinear else 0), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.dsmil_agg_pack_split(_ptr(q0_w), _ptr(q2_w if nonlinear else None), K, _ptr(packed), _stream(dev))
    _native.check(rc, "dsmil_agg_pack_split")
    _split_cache.put(key, (packed, [q0_w, q2_w], _Ready(dev)))
    return packed


def _f2_params(q0_w, q2_w, nonlinear, dev):
    """fp32 path, batches of bags (k_attend_f2, csrc/agg_f2.h): the query weights as two fp16 planes of their power-of-two
    scaled values in MFMA-fragment order (dsmil_agg_pack_f2), prepared once per weight set."""
    L = _native.lib()
    key = ("f2", str(dev), bool(nonlinear), _tkey(q0_w), _tkey(q2_w) if nonlinear else None)
    ent = _split_cache.get(key)
    if ent is not None:
        ent[2].wait(dev, ent[0])
        return ent[0]
    K = q0_w.shape[1]
    packed = torch.empty(L.dsmil_agg_packed_f2_bytes(K), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.dsmil_agg_pack_f2(_ptr(q0_w), _ptr(q2_w if nonlinear else None), K, _ptr(packed), _stream(dev))
    _native.check(rc, "dsmil_agg_pack_f2")
    _split_cache.put(key, (packed, [q0_w, q2_w], _Ready(dev)))
    return packed


def _i64c(t, name):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.int64:
        raise RuntimeError(f"{name} must be an int64 CUDA(HIP) tensor")
    return t if t.is_contiguous() else t.contiguous()


def agg_forward(feats, lengths, w, classes_in=None, vals=None, nonlinear=True, offsets=None, row_map=None):
    """dsmil_agg_forward / dsmil_agg_forward_bf16 over a batch of bags stored back to back.

    feats [total,K] fp32 or bf16 CUDA; lengths: python ints (bag sizes); w: dict of CUDA tensors with
    keys fc_w fc_b q0_w q0_b q2_w q2_b fcc_w fcc_b (fc_* may be None when classes_in is given).
    A bf16 `feats` selects the bf16-storage path (BASELINE config 2): weights are rounded to bf16,
    accumulation stays f32, outputs are fp32.
    ``row_map`` (int64 [total], fp32 path): logical row i lives at physical row row_map[i] of feats / vals —
    train_tcga.py:78-83's `feats[random_indices]` without the gathered copy; `lengths` then count LOGICAL rows.
    Returns (classes [total,C], pred [n_bags,C], A [total,C], B [n_bags,C,Kv], idx int64 [n_bags,C]).
    """
    bf16 = feats.dtype == torch.bfloat16
    if not feats.is_cuda:
        raise RuntimeError("feats must be a CUDA(HIP) tensor for the native path")
    if bf16:
        feats = feats if feats.is_contiguous() else feats.contiguous()
    else:
        feats = _f32c(feats, "feats")
    dev = feats.device
    total, K = feats.shape
    row_map = _i64c(row_map, "row_map")
    if row_map is not None:
        if bf16:
            raise ValueError("row_map is implemented for the fp32 path")
        total = int(row_map.numel())
    lengths = [int(n) for n in lengths]
    if sum(lengths) != total:
        raise ValueError(f"bag lengths sum to {sum(lengths)} but feats has {total} rows")
    if any(n <= 0 for n in lengths):
        raise ValueError("every bag needs at least one instance (the reference's sort/index_select "
                         "at dsmil.py:52-53 fails on an empty bag too)")
    n_bags = len(lengths)
    if vals is None:
        vals = feats
    elif bf16:
        vals = vals.to(torch.bfloat16).contiguous()
    else:
        vals = _f32c(vals, "vals")
    Kv = vals.shape[1]
    packed = None
    if bf16:
        w, packed = _bf16_params(w, nonlinear, dev)
    fcc_w = _f32c(w["fcc_w"], "fcc_w")
    C = fcc_w.shape[0]
    if fcc_w.shape[2] != Kv:
        raise ValueError(f"fcc kernel_size {fcc_w.shape[2]} != value width {Kv}")
    if classes_in is not None:
        classes_in = _f32c(classes_in.float(), "classes_in")
        if tuple(classes_in.shape) != (total, C):
            raise ValueError(f"c must be [{total},{C}], got {tuple(classes_in.shape)}")
    keep = [_f32c(w.get(k), k) for k in ("fc_w", "fc_b", "q0_w", "q0_b", "q2_w", "q2_b", "fcc_w", "fcc_b")]
    p = _native.AggParams(*[(t.data_ptr() if t is not None else 0) for t in keep],
                          K, Kv, C, 1 if nonlinear else 0)
    off = offsets if offsets is not None else offsets_tensor(lengths, dev)
    classes = classes_in if classes_in is not None else torch.empty((total, C), dtype=torch.float32, device=dev)
    A = torch.empty((total, C), dtype=torch.float32, device=dev)
    B = torch.empty((n_bags, C, Kv), dtype=torch.float32, device=dev)
    pred = torch.empty((n_bags, C), dtype=torch.float32, device=dev)
    idx = torch.empty((n_bags, C), dtype=torch.int64, device=dev)
    L = _native.lib()
    nbytes = L.dsmil_agg_workspace_bytes(n_bags, total, K, Kv, C)
    ws = _workspace(dev, nbytes)
    with torch.cuda.device(dev):
        if bf16:
            rc = L.dsmil_agg_forward_bf16(_ptr(feats), _ptr(vals), _ptr(off), n_bags, total, max(lengths),
                                          ctypes.byref(p), _ptr(packed), _ptr(classes_in),
                                          _ptr(classes if classes_in is None else None),
                                          _ptr(A), _ptr(B), _ptr(pred), _ptr(idx), _ptr(ws), ws.numel(),
                                          _stream(dev))
        else:
            split = _split_params(keep[2], keep[4], nonlinear, dev)
            # batches in the 128-row-tile regime take k_attend_f3 / k_attend_f2 (K a multiple of 128 up to 512, v = Identity): its weight image
            f2 = None
            if (split is not None and L.dsmil_agg_tile_rows(n_bags, total) == 128 and K % 128 == 0 and K <= 512 and vals is feats
                    and classes_in is None):
                f2 = _f2_params(keep[2], keep[4], nonlinear, dev)
            opts = _native.AggOpts(split.data_ptr() if split is not None else 0,
                                   row_map.data_ptr() if row_map is not None else 0,
                                   f2.data_ptr() if f2 is not None else 0)
            rc = L.dsmil_agg_forward_ex(_ptr(feats), _ptr(vals), _ptr(off), n_bags, total, max(lengths),
                                        ctypes.byref(p), ctypes.byref(opts), _ptr(classes_in),
                                        _ptr(classes if classes_in is None else None),
                                        _ptr(A), _ptr(B), _ptr(pred), _ptr(idx), _ptr(ws), ws.numel(),
                                        _stream(dev))
    _native.check(rc, "dsmil_agg_forward_bf16" if bf16 else "dsmil_agg_forward_ex")
    del keep
    return classes, pred, A, B, idx


class GraphedAggForward:
    """MILNet.forward of ONE bag shape captured into a hipGraph and replayed (the library enqueues on the
    caller's stream, allocates nothing and never synchronises, so the whole 5-launch forward is capturable).  A single
    10 000-row bag is launch-bound (5 dependent launches for ~30 us of device work); replay issues them with one call.

        g = GraphedAggForward(w, n_rows, K)          # captures once (weights are read from `w`'s tensors in place)
        classes, pred, A, B, idx = g(feats)          # copies feats into the static input, replays, returns the
                                                     # static output tensors (valid until the next call)
    Inference only: the capture holds the addresses of the weights and of their packed plane cuts, so the weights
    must stay as they are — build a new object after an optimizer step or a load_state_dict."""

    def __init__(self, w, n_rows, K, nonlinear=True, device=None, dtype=torch.float32):
        dev = torch.device(device) if device is not None else w["q0_w"].device
        self.w, self.n, self.nonlinear, self.dev = w, int(n_rows), nonlinear, dev
        self.x = torch.zeros((self.n, K), dtype=dtype, device=dev)
        self.offsets = offsets_tensor([self.n], dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):   # warm-up outside capture: workspace, packed weights, function attributes
            for _ in range(2):
                agg_forward(self.x, [self.n], w, nonlinear=nonlinear, offsets=self.offsets)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # the captured launches read these buffers by address: keep them alive as long as the graph
        self._keep = [_split_params(_f32c(w["q0_w"], "q0_w"), _f32c(w.get("q2_w"), "q2_w"), nonlinear, dev)
                      if dtype == torch.float32 else _bf16_params(w, nonlinear, dev)]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = agg_forward(self.x, [self.n], w, nonlinear=nonlinear, offsets=self.offsets)
        self._keep.append(_ws_last[0])   # the workspace of the capture stream

    def __call__(self, feats):
        if tuple(feats.shape) != tuple(self.x.shape):   # copy_ would broadcast a wrong-sized bag silently
            raise ValueError(f"captured for a bag of shape {tuple(self.x.shape)}, got {tuple(feats.shape)}")
        self.x.copy_(feats)
        self.graph.replay()
        return self.out   # the STATIC output tensors of the capture: valid until the next call (clone to keep)


def _agg_params(w, K, Kv, nonlinear):
    fcc_w = _f32c(w["fcc_w"], "fcc_w")
    C = fcc_w.shape[0]
    keep = [_f32c(w.get(k), k) for k in ("fc_w", "fc_b", "q0_w", "q0_b", "q2_w", "q2_b", "fcc_w", "fcc_b")]
    p = _native.AggParams(*[(t.data_ptr() if t is not None else 0) for t in keep], K, Kv, C, 1 if nonlinear else 0)
    return p, keep, C


def agg_shard_argmax(feats, w, nonlinear=True):
    """dsmil_agg_shard_argmax: one rank's row range of an instance-sharded bag.
    Returns (classes [rows,C], best_val [C], best_idx [C] shard-local)."""
    feats = _f32c(feats, "feats")
    dev = feats.device
    rows, K = feats.shape
    p, keep, C = _agg_params(w, K, w["fcc_w"].shape[2], nonlinear)
    classes = torch.empty((rows, C), dtype=torch.float32, device=dev)
    best_val = torch.empty((C,), dtype=torch.float32, device=dev)
    best_idx = torch.empty((C,), dtype=torch.int64, device=dev)
    L = _native.lib()
    ws = _workspace(dev, L.dsmil_agg_workspace_bytes(1, rows, K, w["fcc_w"].shape[2], C))
    with torch.cuda.device(dev):
        rc = L.dsmil_agg_shard_argmax(_ptr(feats), rows, ctypes.byref(p), _ptr(classes), _ptr(best_val),
                                      _ptr(best_idx), _ptr(ws), ws.numel(), _stream(dev))
    _native.check(rc, "dsmil_agg_shard_argmax")
    del keep
    return classes, best_val, best_idx


def agg_shard_attend(feats, w, crit_rows, vals=None, nonlinear=True):
    """dsmil_agg_shard_attend: attention of this shard's rows against the bag-wide critical rows.
    Returns (A_unnorm [rows,C] = exp(s - m_shard), ml [C,2] = (m_shard, l_shard), B_unnorm [C,Kv])."""
    feats = _f32c(feats, "feats")
    dev = feats.device
    rows, K = feats.shape
    vals = feats if vals is None else _f32c(vals, "vals")
    Kv = vals.shape[1]
    p, keep, C = _agg_params(w, K, Kv, nonlinear)
    crit_rows = _f32c(crit_rows, "crit_rows")
    if tuple(crit_rows.shape) != (C, K):
        raise ValueError(f"crit_rows must be [{C},{K}], got {tuple(crit_rows.shape)}")
    A = torch.empty((rows, C), dtype=torch.float32, device=dev)
    ml = torch.empty((C, 2), dtype=torch.float32, device=dev)
    B = torch.empty((C, Kv), dtype=torch.float32, device=dev)
    L = _native.lib()
    ws = _workspace(dev, L.dsmil_agg_workspace_bytes(1, rows, K, Kv, C))
    with torch.cuda.device(dev):
        rc = L.dsmil_agg_shard_attend(_ptr(feats), _ptr(vals), rows, ctypes.byref(p), _ptr(crit_rows), _ptr(A),
                                      _ptr(ml), _ptr(B), _ptr(ws), ws.numel(), _stream(dev))
    _native.check(rc, "dsmil_agg_shard_attend")
    del keep
    return A, ml, B


def agg_loss_head(classes, pred, idx, label):
    """dsmil_agg_loss_head: the training objective of one bag (train_tcga.py:67-71) and its logit gradients in one
    launch.  Returns (loss [] , max_pred [C], g_pred [C], g_max [C])."""
    classes = _f32c(classes, "classes"); pred = _f32c(pred.reshape(-1), "pred")
    label = _f32c(label.reshape(-1).to(torch.float32), "label")
    idx = _i64c(idx.reshape(-1), "idx")
    C = classes.shape[1]
    dev = classes.device
    out = torch.empty((1 + 3 * C,), dtype=torch.float32, device=dev)
    loss, max_pred, g_pred, g_max = out[0:1], out[1:1 + C], out[1 + C:1 + 2 * C], out[1 + 2 * C:]
    with torch.cuda.device(dev):
        rc = _native.lib().dsmil_agg_loss_head(_ptr(classes), _ptr(pred), _ptr(idx), _ptr(label), C, _ptr(loss),
                                               _ptr(max_pred), _ptr(g_pred), _ptr(g_max), _stream(dev))
    _native.check(rc, "dsmil_agg_loss_head")
    return loss.reshape(()), max_pred, g_pred, g_max


def agg_backward(feats, w, A, B, idx, g_pred, g_classes=None, g_A=None, g_B=None, vals=None, nonlinear=True,
                 want_g_vals=False, g_max=None, row_map=None):
    """dsmil_agg_backward: parameter gradients of FCLayer + BClassifier for ONE bag (what autograd
    derives for train_tcga.py:67-72).  feats [N,K] fp32 CUDA, w as in agg_forward, A [N,C], B [1,C,Kv],
    idx [1,C] = the forward's outputs; g_* = upstream gradients (None = zero).  Returns a dict with the
    gradient of every key of ``w`` (fc_* only when g_classes or g_max is given, q2_* only when nonlinear) and
    ``vals`` (when want_g_vals).  ``g_max`` [C]: the sparse gradient of max_n classes[n,:] (the training objective's
    instance stream); ``row_map``: see agg_forward (N = its length)."""
    feats = _f32c(feats, "feats")
    dev = feats.device
    N, K = feats.shape
    row_map = _i64c(row_map, "row_map")
    if row_map is not None:
        N = int(row_map.numel())
    vals = feats if vals is None else _f32c(vals, "vals")
    Kv = vals.shape[1]
    fcc_w = _f32c(w["fcc_w"], "fcc_w")
    C = fcc_w.shape[0]
    keep = [_f32c(w.get(k), k) for k in ("fc_w", "fc_b", "q0_w", "q0_b", "q2_w", "q2_b", "fcc_w", "fcc_b")]
    p = _native.AggParams(*[(t.data_ptr() if t is not None else 0) for t in keep],
                          K, Kv, C, 1 if nonlinear else 0)
    A = _f32c(A, "A"); B = _f32c(B, "B")
    idx = idx.contiguous()
    g_pred = _f32c(g_pred.reshape(-1), "g_pred")
    g_classes = _f32c(g_classes, "g_classes"); g_A = _f32c(g_A, "g_A"); g_B = _f32c(g_B, "g_B")
    g_max = _f32c(g_max.reshape(-1), "g_max") if g_max is not None else None
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    out = {"q0_w": new(Q_DIM, K), "q0_b": new(Q_DIM), "fcc_w": new(C, C, Kv), "fcc_b": new(C)}
    if nonlinear:
        out["q2_w"], out["q2_b"] = new(Q_DIM, Q_DIM), new(Q_DIM)
    if g_classes is not None or g_max is not None:
        out["fc_w"], out["fc_b"] = new(C, K), new(C)
    g = _native.AggGrads(*[(out[k].data_ptr() if k in out else 0)
                           for k in ("fc_w", "fc_b", "q0_w", "q0_b", "q2_w", "q2_b", "fcc_w", "fcc_b")])
    g_vals = new(N, Kv) if want_g_vals else None
    L = _native.lib()
    nbytes = L.dsmil_agg_backward_workspace_bytes(N, K, Kv, C)
    ws = _workspace(dev, nbytes)
    with torch.cuda.device(dev):
        rc = L.dsmil_agg_backward_ex(_ptr(feats), _ptr(vals), N, ctypes.byref(p), _ptr(A), _ptr(B), _ptr(idx),
                                     _ptr(g_classes), _ptr(g_max), _ptr(g_pred), _ptr(g_A), _ptr(g_B), ctypes.byref(g),
                                     _ptr(g_vals), _ptr(row_map), _ptr(ws), ws.numel(), _stream(dev))
    _native.check(rc, "dsmil_agg_backward_ex")
    del keep
    if want_g_vals:
        out["vals"] = g_vals
    return out


def agg_train_step(feats, label, params, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay, nonlinear=True,
                   row_map=None, loss_out=None):
    """dsmil_agg_train_step: one train_tcga.py:60-75 step (forward, 0.5 BCE(bag) + 0.5 BCE(max instance), backward,
    Adam) as ONE native call.  ``params`` / ``exp_avg`` / ``exp_avg_sq``: the eight tensors in the order fc_w, fc_b, q0_w,
    q0_b, q2_w, q2_b, fcc_w, fcc_b (None for q2_* of a linear query) — parameters and moments are updated IN PLACE.
    ``step`` = 1-based index of this update.  Returns the loss as a 1-element device tensor (no host sync)."""
    feats = _f32c(feats, "feats")
    dev = feats.device
    rows, K = feats.shape
    row_map = _i64c(row_map, "row_map")
    N = int(row_map.numel()) if row_map is not None else rows
    label = _f32c(label.reshape(-1).to(torch.float32), "label")
    C = int(label.numel())
    for t in list(params) + list(exp_avg) + list(exp_avg_sq):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("agg_train_step: parameters and Adam moments must be contiguous fp32 CUDA tensors")
    ptr = lambda t: (t.data_ptr() if t is not None else 0)
    p = _native.AggParams(*[ptr(t) for t in params], K, K, C, 1 if nonlinear else 0)
    arr = ctypes.c_void_p * 8
    m_arr, v_arr = arr(*[ptr(t) for t in exp_avg]), arr(*[ptr(t) for t in exp_avg_sq])
    st = _native.AdamState(ctypes.cast(m_arr, ctypes.POINTER(ctypes.c_void_p)), ctypes.cast(v_arr, ctypes.POINTER(ctypes.c_void_p)),
                           int(step), float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay))
    loss = loss_out if loss_out is not None else torch.empty((1,), dtype=torch.float32, device=dev)
    L = _native.lib()
    ws = _workspace(dev, L.dsmil_agg_train_step_workspace_bytes(N, K, C, 1 if nonlinear else 0))
    with torch.cuda.device(dev):
        rc = L.dsmil_agg_train_step(_ptr(feats), N, _ptr(row_map), _ptr(label), ctypes.byref(p), ctypes.byref(st), _ptr(loss),
                                    _ptr(ws), ws.numel(), _stream(dev))
    _native.check(rc, "dsmil_agg_train_step")
    return loss


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay):
    """dsmil_adam_step: torch.optim.Adam's update (amsgrad = maximize = False) of up to 8 tensors in one launch, in place."""
    n = len(params)
    arr = ctypes.c_void_p * n
    cast = lambda ts: ctypes.cast(arr(*[t.data_ptr() for t in ts]), ctypes.POINTER(ctypes.c_void_p))
    numel = (ctypes.c_int64 * n)(*[int(t.numel()) for t in params])
    dev = params[0].device
    with torch.cuda.device(dev):
        rc = _native.lib().dsmil_adam_step(n, cast(params), cast(grads), cast(exp_avg), cast(exp_avg_sq), numel, int(step),
                                           float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), _stream(dev))
    _native.check(rc, "dsmil_adam_step")


# ---------------------------------------------------------------------------------------------
# patch embedder (ResNet-18 + InstanceNorm) — compute_feats.py:146-170,211 / dsmil.py:21-25
# ---------------------------------------------------------------------------------------------
_pack_cache = _LRU()


def resnet_conv_shapes(depth):
    """Shapes of the bias-free convs of a torchvision ResNet in state_dict order (mirror of make_arch in
    csrc/resnet_fwd.hip): BasicBlock depth 18 -> 20 tensors, 34 -> 36; Bottleneck depth 50 -> 53, 101 -> 104."""
    nblk = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]
    shapes = [(64, 3, 7, 7)]
    cin = 64
    for l, n in enumerate(nblk):
        c = 64 << l
        for b in range(n):
            if depth >= 50:
                shapes += [(c, cin, 1, 1), (c, c, 3, 3), (4 * c, c, 1, 1)]
                if b == 0:
                    shapes.append((4 * c, cin, 1, 1))
                cin = 4 * c
            else:
                shapes += [(c, cin, 3, 3), (c, c, 3, 3)]
                if l > 0 and b == 0:
                    shapes.append((c, cin, 1, 1))
                cin = c
    return shapes


RESNET18_SHAPES = resnet_conv_shapes(18)
RESNET_DEPTHS = (18, 34, 50, 101)


def resnet_depth_of(convs):
    """18 / 34 / 50 / 101 when the conv list has exactly that architecture's shapes and order, else None."""
    for depth in RESNET_DEPTHS:
        sh = resnet_conv_shapes(depth)
        if len(convs) == len(sh) and all(tuple(w.shape) == t for w, t in zip(convs, sh)):
            return depth
    return None


RESNET_MAX_ABS_WEIGHT = 100.0   # < 65504 / (2^8 * 2.25): see _packed_resnet_weights


def _packed_resnet_weights(convs, depth=18, precision=0):
    """Device buffer with the non-stem conv weights re-laid-out for the kernels (dsmil_resnet_pack:
    Winograd-transformed or [tap][Cout][Cin]).  Cached per weight set; rebuilt when any tensor was
    modified in place (``_version``), re-assigned or moved (``data_ptr``).  The entry keeps the source
    tensors alive: a freed weight's address cannot come back as a different model's weight."""
    dev = convs[0].device
    key = (str(dev), depth, int(precision)) + tuple(_tkey(w) for w in convs)
    ent = _pack_cache.get(key)
    if ent is not None:
        ent[3].wait(dev, ent[0])
        return ent[0]
    L = _native.lib()
    nbytes = L.dsmil_resnet_packed_bytes_ex(depth, int(precision))
    if nbytes == 0:
        raise ValueError(f"precision {precision} is not implemented for a depth-{depth} trunk "
                         "(the bf16-activation trunk: ResNet-18 / 34 with InstanceNorm)")
    buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    keep = [_f32c(w.detach(), "conv weight") for w in convs]
    # The conv operands are cut into fp16 planes of the 2^8-scaled weights (csrc/resnet_fwd.hip, EMB_WSHIFT); a Winograd
    # weight transform grows a 3x3 kernel by at most 2.25x, so |w| must stay below 65504 / (256 * 2.25) = 113.7 or a plane
    # overflows to inf.  Checked ONCE per weight set (this function is cached on the weights' versions): one host read.
    wmax = float(torch.stack([w.abs().amax() for w in keep]).amax())
    if not wmax < RESNET_MAX_ABS_WEIGHT:   # (also catches NaN)
        raise ValueError(f"conv weight magnitude {wmax:g} is outside the native embedder's range (|w| < {RESNET_MAX_ABS_WEIGHT:g}: "
                         "its operands are fp16 planes of the 2^8-scaled weights)")
    arr = (ctypes.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
    with torch.cuda.device(dev):
        rc = L.dsmil_resnet_pack_ex(depth, arr, _ptr(buf), int(precision), _stream(dev))
    _native.check(rc, "dsmil_resnet_pack_ex")
    _pack_cache.put(key, (buf, keep, list(convs), _Ready(dev)))
    return buf


_bn_cache = _LRU()
_bn_checked = _LRU()
_f16_trunk_ok = _LRU()     # weight set -> its fp16-activation trunk produced finite features (checked once)


def _folded_bn(norms, dev):
    """Fold the trunk's eval-mode BatchNorm2d modules into y = (x - m) * r, concatenated in conv order (cached on the
    parameter / buffer versions)."""
    key = (str(dev),) + tuple((id(n), _tkey(n.running_mean), _tkey(n.running_var), _tkey(n.weight), _tkey(n.bias), n.eps)
                              for n in norms)
    hit = _bn_cache.get(key)
    if hit is not None:
        hit[3].wait(dev, hit[0], hit[1])
        return hit[0], hit[1]
    ms, rs = [], []
    for n in norms:
        var = n.running_var.detach().to(dev, torch.float64)
        mean = n.running_mean.detach().to(dev, torch.float64)
        w = n.weight.detach().to(dev, torch.float64) if n.weight is not None else torch.ones_like(var)
        b = n.bias.detach().to(dev, torch.float64) if n.bias is not None else torch.zeros_like(var)
        r = w / torch.sqrt(var + n.eps)
        if bool((r == 0).any()):
            raise NotImplementedError("a BatchNorm channel with weight 0 cannot be folded into (x - m) * r")
        ms.append(mean - b / r)
        rs.append(r)
    m = torch.cat(ms).to(torch.float32).contiguous()
    r = torch.cat(rs).to(torch.float32).contiguous()
    _bn_cache.put(key, (m, r, list(norms), _Ready(dev)))   # the modules stay alive: their ids stay unique
    return m, r


def resnet18in_forward(x, convs, fc_w=None, fc_b=None, bn_norms=None, precision="fp32"):
    """x: [B,3,H,W] fp32 CUDA in [0,1] (what VF.to_tensor yields), OR decoded images as uint8
    [B,H,W,3] CUDA (the /255 + HWC->CHW of to_tensor is then fused into the stem, bit-identically);
    convs: the trunk's conv weights in torchvision state_dict order (20 for ResNet-18, 36 for ResNet-34, 53 for
    ResNet-50, 104 for ResNet-101).
    ``bn_norms``: the eval-mode BatchNorm2d modules of a `--norm_layer batch` trunk, in the same order;
    None = InstanceNorm.  One native launch sequence (dsmil_resnet_forward_ex).
    ``precision``: "fp32" (default: fp32-class, the parity path); "half" (OPT-IN reduced precision, <= 5e-3 feature error, not
    the 1e-4 bar): fp16 ACTIVATIONS behind the stem, one fp16 MFMA product per MAC, f32 accumulation and InstanceNorm statistics
    (round 6, csrc/resnet_b16.h: ResNet-18 / 34 InstanceNorm trunks, ~2.6e-3, 2x the fp32 path) — and where that trunk does not
    apply (frozen BatchNorm, Bottleneck trunks, patches under 64 x 64 or wider than ~1000 pixels, weights whose conv sums leave
    fp16's range) the older form: fp32 activations, every conv operand rounded to one fp16 plane (~2.6e-3, 1.2x); "bf16"
    (OPT-IN): the same trunk on bf16 activations — fp32's range, 8 significant bits: ~2e-2 feature error; include/dsmil_hip.h.
    Returns (feats [B,512 | 2048], classes [B,C] or None)."""
    if precision not in ("fp32", "half", "bf16"):
        raise ValueError("precision must be 'fp32', 'half' or 'bf16'")
    prec = {"fp32": 0, "half": 1, "bf16": 2}[precision]
    if prec == 2 and bn_norms is not None:
        raise ValueError("precision 'bf16' is implemented for InstanceNorm trunks")
    u8 = x.dtype == torch.uint8
    if u8:
        if not x.is_cuda:
            raise RuntimeError("x must be a CUDA(HIP) tensor for the native path")
        if x.dim() != 4 or x.shape[3] != 3:
            raise ValueError(f"uint8 patches must be NHWC [B,H,W,3], got {tuple(x.shape)}")
        x = x if x.is_contiguous() else x.contiguous()
        B, H, W, _ = x.shape
    else:
        x = _f32c(x, "x")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected [B,3,H,W] patches, got {tuple(x.shape)}")
        B, _, H, W = x.shape
    depth = resnet_depth_of(convs)
    if depth is None:
        raise ValueError("conv weights do not have the ResNet-18 / 34 / 50 / 101 shapes and order")
    dev = x.device
    L = _native.lib()
    feats = torch.empty((B, L.dsmil_resnet_feature_dim(depth)), dtype=torch.float32, device=dev)
    if B == 0:
        return feats, (torch.empty((0, fc_w.shape[0]), device=dev) if fc_w is not None else None)
    fc_w = _f32c(fc_w.detach(), "fc_w") if fc_w is not None else None
    fc_b = _f32c(fc_b.detach(), "fc_b") if fc_b is not None else None
    C = fc_w.shape[0] if fc_w is not None else 0
    classes = torch.empty((B, C), dtype=torch.float32, device=dev) if fc_w is not None else None
    conv1 = _f32c(convs[0].detach(), "conv1.weight")
    nbytes = L.dsmil_resnet_workspace_bytes(depth, B, H, W)
    if nbytes == 0:
        raise ValueError(f"unsupported patch size {H}x{W}")
    ws = _workspace(dev, nbytes)
    bn_m = bn_r = None
    if bn_norms is not None:
        bn_m, bn_r = _folded_bn(bn_norms, dev)
        if bn_m.numel() != L.dsmil_resnet_norm_channels(depth):
            raise ValueError("BatchNorm channel counts do not match the trunk")

    def run(p):
        packed = _packed_resnet_weights(convs, depth, p)
        with torch.cuda.device(dev):
            return packed, L.dsmil_resnet_forward_ex(depth, _ptr(x), 1 if u8 else 0, B, H, W, _ptr(conv1), _ptr(packed), _ptr(bn_m),
                                                     _ptr(bn_r), _ptr(fc_w), _ptr(fc_b), C, _ptr(feats), _ptr(classes), _ptr(ws),
                                                     ws.numel(), p, _stream(dev))

    rc = None
    if prec == 1 and bn_norms is None and L.dsmil_resnet_packed_bytes_ex(depth, 3) != 0:
        # "half": the fp16-ACTIVATION trunk (precision 3) where it applies ...
        wkey = ("f16act",) + tuple(_tkey(w) for w in convs)
        ent = _f16_trunk_ok.get(wkey)              # (ok, the weights: an entry keeps its sources alive, so a freed tensor's address
        state = None if ent is None else ent[0]    # cannot come back as another model's)  None: not tried, True / False: checked once
        if state is not False:
            packed, rc = run(3)
            if rc == _native.DSMIL_E_UNSUPPORTED:  # (patch size outside the trunk's range)
                rc = None
            elif rc == 0 and state is None:
                # fp16 activations: the conv sums of THIS weight set must stay inside +-65504 — checked on its first forward
                # (one host read per weight set); a set that leaves the range keeps the one-plane form below
                ok = bool(torch.isfinite(feats).all())
                _f16_trunk_ok.put(wkey, (ok, list(convs)))
                if not ok:
                    rc = None
    if rc is None:   # ... else (and for every other precision) the form the caller named
        packed, rc = run(prec)
    _native.check(rc, "dsmil_resnet_forward_ex")
    if bn_norms is not None:
        # A frozen-BatchNorm trunk has no bound on its activations (InstanceNorm output is bounded by sqrt(H W)); an activation
        # past fp16's 65504 would turn into inf / NaN features.  The FIRST forward of every folded-BatchNorm set is checked
        # (one host read per weight set, none afterwards).
        key = ("bn_finite", bn_m.data_ptr(), bn_m._version, packed.data_ptr())
        if _bn_checked.get(key) is None:
            if not bool(torch.isfinite(feats).all()):
                raise FloatingPointError("the frozen-BatchNorm trunk produced non-finite features: an activation left the fp16 "
                                         "operand range of the native conv kernels (|a| < 65504)")
            _bn_checked.put(key, True)
    return feats, classes


# ---------------------------------------------------------------------------------------------
# background filters of the tilers (deepzoom_tiler.py:56-61, test_crop_single.py:17-24)
# ---------------------------------------------------------------------------------------------
def tile_stats(tiles):
    """dsmil_tile_stats: uint8 NHWC tiles [B,H,W,3] on the device -> int64 [B,4] = per tile the three band sums of
    PIL's FIND_EDGES image and the sum of the ubyte HSV saturation (exact integers)."""
    if not tiles.is_cuda or tiles.dtype != torch.uint8 or tiles.dim() != 4 or tiles.shape[3] != 3:
        raise RuntimeError("tiles must be a CUDA(HIP) uint8 tensor [B,H,W,3]")
    tiles = tiles if tiles.is_contiguous() else tiles.contiguous()
    B, H, W, _ = tiles.shape
    out = torch.empty((B, 4), dtype=torch.int64, device=tiles.device)
    if B == 0:
        return out
    with torch.cuda.device(tiles.device):
        rc = _native.lib().dsmil_tile_stats(_ptr(tiles), B, H, W, _ptr(out), _stream(tiles.device))
    _native.check(rc, "dsmil_tile_stats")
    return out


# ---------------------------------------------------------------------------------------------
# batched baseline-JPEG decode on the device (compute_feats.py:28,107 `Image.open` in DataLoader workers)
# ---------------------------------------------------------------------------------------------
JPEG_REC = np.dtype([("ecs_begin", "<i8"), ("ecs_end", "<i8"), ("width", "<i4"), ("height", "<i4"), ("ncomp", "<i4"),
                     ("hsamp", "<i4"), ("vsamp", "<i4"), ("restart_interval", "<i4"), ("qt", "<i4", 3), ("dc", "<i4", 3),
                     ("ac", "<i4", 3), ("status", "<i4")])   # struct dsmil_jpeg_image


def jpeg_parse(blobs):
    """dsmil_jpeg_parse (HOST): the files of a batch (bytes-likes) -> (data uint8 [total] numpy, plan uint8 numpy, records =
    a structured view of the plan's per-image records: status 0 = decodable on the device, -2 = outside the decoder's scope,
    -1 = not a JPEG)."""
    n = len(blobs)
    if n == 0:
        raise ValueError("no files")
    sizes = np.fromiter((len(b) for b in blobs), dtype=np.int64, count=n)
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum(sizes, out=offsets[1:])
    # (+32: the device reader prefetches 16 bytes past its position; an EOI behind the last file stops a truncated one)
    data = np.empty(int(offsets[-1]) + 32, np.uint8)
    data[int(offsets[-1]):] = 0
    data[int(offsets[-1])], data[int(offsets[-1]) + 1] = 0xFF, 0xD9
    mv = memoryview(data)
    for b, o, s in zip(blobs, offsets[:-1], sizes):
        mv[int(o):int(o + s)] = b
    L = _native.lib()
    plan = np.zeros(L.dsmil_jpeg_plan_bytes(n) + 16, np.uint8)
    a = (-plan.ctypes.data) % 16
    plan = plan[a:a + L.dsmil_jpeg_plan_bytes(n)]
    _native.check(L.dsmil_jpeg_parse(data.ctypes.data, offsets.ctypes.data, n, plan.ctypes.data), "dsmil_jpeg_parse")
    recs = plan[16:16 + JPEG_REC.itemsize * n].view(JPEG_REC)
    return data, plan, recs


def _pil_rgb(blob):
    import io
    from PIL import Image
    with Image.open(io.BytesIO(blob)) as im:
        return np.array(im.convert("RGB"), dtype=np.uint8, copy=True)


class _JpegPending:
    """A device decode in flight (jpeg_decode_begin): the launch is enqueued, its per-file status not yet read."""
    __slots__ = ("out", "blobs", "H", "W", "st_host", "st_np", "ev")


def jpeg_decode_begin(blobs, device, size=None, out=None, before_launch=None):
    """The first half of jpeg_decode: parse on the host, copy, ENQUEUE the device decode on the current stream and the copy of its
    per-file status into pinned memory — no host wait.  `.ev` (None when no file of the batch is decodable on the device) is
    recorded behind both; `.out` is the result tensor, complete once jpeg_decode_end has run.  A caller with several chunks keeps
    a decode or two in flight this way instead of waiting for every chunk's status before it enqueues anything else
    (pipeline._embed_jpeg_chunks).  `before_launch()` runs between the host-to-device copies of the files and the decode launch:
    the place for a stream wait on whoever still reads `out` — in front of the copies it would hold the HOST, the copy of a
    pageable buffer returns only when it has run."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("jpeg_decode needs a CUDA(HIP) device")
    n = len(blobs)
    data, plan, recs = jpeg_parse(blobs)
    n_data = int(data.size) - 32        # (jpeg_parse pads the batch buffer)
    ok = recs["status"] == 0
    if size is None:
        if ok.any():
            j = int(np.argmax(ok))
            size = (int(recs["height"][j]), int(recs["width"][j]))
        else:
            size = _pil_rgb(blobs[0]).shape[:2]
    H, W = int(size[0]), int(size[1])
    on_dev = ok & (recs["height"] == H) & (recs["width"] == W)
    recs["status"][ok & ~on_dev] = -2          # another size: not this batch's launch (the Pillow path below checks the size)
    if out is not None and (not out.is_cuda or out.dtype != torch.uint8 or not out.is_contiguous() or out.numel() < n * H * W * 3):
        raise ValueError("out must be a contiguous uint8 device tensor of at least n * H * W * 3 elements")
    out = (torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev) if out is None
           else out.view(-1)[:n * H * W * 3].view(n, H, W, 3))
    p = _JpegPending()
    p.out, p.blobs, p.H, p.W, p.st_host, p.st_np, p.ev = out, blobs, H, W, None, None, None
    if on_dev.any():
        L = _native.lib()
        status = torch.empty(n, dtype=torch.int32, device=dev)
        d_data = torch.from_numpy(data).to(dev, non_blocking=True)
        d_plan = torch.from_numpy(plan).to(dev, non_blocking=True)
        nbytes = L.dsmil_jpeg_workspace_bytes(n, H, W, n_data)
        ws = _workspace(dev, nbytes)
        if before_launch is not None:
            before_launch()
        with torch.cuda.device(dev):
            rc = L.dsmil_jpeg_decode(_ptr(d_data), n_data, _ptr(d_plan), n, H, W, _ptr(out), _ptr(status), _ptr(ws), ws.numel(), _stream(dev))
        _native.check(rc, "dsmil_jpeg_decode")
        p.st_host = torch.empty(n, dtype=torch.int32, pin_memory=True)
        p.st_host.copy_(status, non_blocking=True)
        p.ev = torch.cuda.Event()
        p.ev.record(torch.cuda.current_stream(dev))
    else:
        if before_launch is not None:
            before_launch()
        p.st_np = recs["status"].copy()
    return p


def jpeg_decode_end(p, stats=None):
    """The second half: wait for the status, decode what the device did not (other formats, other coding processes, streams it
    reports as corrupt) with Pillow and copy those files in on the CURRENT stream.  Returns (out, files redone)."""
    if p.ev is not None:
        p.ev.synchronize()
        st = p.st_host.numpy()
    else:
        st = p.st_np
    redo = np.nonzero(st != 0)[0]
    for i in redo:
        a = _pil_rgb(p.blobs[int(i)])
        if a.shape[:2] != (p.H, p.W):
            raise ValueError(f"file {int(i)} of the batch is {a.shape[1]}x{a.shape[0]}, the batch is {p.W}x{p.H}")
        p.out[int(i)].copy_(torch.from_numpy(a))
    if stats is not None:
        stats["device"] = stats.get("device", 0) + int(len(st) - len(redo))
        stats["pillow"] = stats.get("pillow", 0) + int(len(redo))
    p.blobs = None
    return p.out, int(len(redo))


def jpeg_decode(blobs, device, size=None, stats=None, out=None):
    """A batch of JPEG files (bytes-likes) -> uint8 [n, H, W, 3] on `device`, what
    `np.array(Image.open(f).convert("RGB"))` gives for every file (compute_feats.py:28 + the uint8 half of VF.to_tensor):
    baseline JPEGs are decoded on the device by dsmil_jpeg_decode (bit-identical to Pillow's defaults: islow IDCT, fancy
    upsampling), every other file (progressive, CMYK, a PNG ...) and every stream the device decoder reports as corrupt is
    decoded with Pillow on the host and copied in — the result never depends on which path a file took.
    ``size`` = (H, W) of the batch (default: the first decodable image's); all files must have it.
    ``stats`` (dict, optional) gets the counts {"device": .., "pillow": ..}.
    ``out``: a uint8 device tensor with room for [n, H, W, 3] (a caller's staging buffer); the result is a view of it.
    (= jpeg_decode_begin + jpeg_decode_end, the two halves a pipelining caller uses.)"""
    return jpeg_decode_end(jpeg_decode_begin(blobs, device, size=size, out=out), stats=stats)[0]
