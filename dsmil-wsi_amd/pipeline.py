"""Host-side loops around the embedder hot path, mirroring the reference's drivers:
compute_feats.compute_feats (:58-82), compute_tree_feats (:84-126), the SimCLR checkpoint
re-keying (:219-233) and attention_map.test (:59-118).  Differences are deliberate and
result-preserving:
  * features stay on the device between batches (one D2H per bag instead of one per batch,
    compute_feats.py:74) and, for attention maps, between embedder and aggregator
    (attention_map.py:75-84 round-trips through numpy);
  * the high-magnification patches of a tree bag are embedded in batches instead of one forward
    per patch (compute_feats.py:106-109) — InstanceNorm is per image, so rows are unchanged;
  * with torch.distributed initialised, a bag's ordered patch list is sharded contiguously over
    ranks and reassembled by one all-gather (dist.py).
"""
import glob
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import dist as ddist
from . import ops


def to_tensor(img):
    """PIL image -> float32 CHW in [0,1] (what torchvision's VF.to_tensor does for 8-bit images,
    compute_feats.py:35-39; no mean/std normalisation)."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.dtype == np.uint8:
        t = torch.from_numpy(np.array(a, copy=True)).permute(2, 0, 1).float().div_(255.0)
    else:
        t = torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).permute(2, 0, 1)
    return t


def patch_position(path):
    """attention_map.py:27-28 — tiles are named '<row>_<col>.<ext>'."""
    stem = os.path.basename(path).split(".")[0].split("_")
    return np.asarray([int(stem[0]), int(stem[1])])


class PatchFiles(Dataset):
    """uint8=False: what the reference's BagDataset + ToTensor yield (float32 CHW, compute_feats.py:21-39).
    uint8=True: the decoded RGB image as uint8 HWC — ToTensor then runs fused in the native stem
    (dsmil_resnet18in_forward_u8), bit-identically, and the H2D copy is 4x smaller."""

    def __init__(self, files, with_position=False, uint8=False):
        self.files = list(files)
        self.with_position = with_position
        self.uint8 = uint8

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        from PIL import Image
        with Image.open(self.files[i]) as im:
            if self.uint8:
                sample = {"input": torch.from_numpy(np.array(im.convert("RGB"), dtype=np.uint8, copy=True))}
            else:
                sample = {"input": to_tensor(im.convert("RGB") if im.mode not in ("RGB", "L") else im)}
        if self.with_position:
            sample["position"] = patch_position(self.files[i])
        return sample


def patch_loader(files, batch_size, num_workers, with_position=False, uint8=False):
    return DataLoader(PatchFiles(files, with_position, uint8), batch_size=batch_size, shuffle=False,
                      num_workers=num_workers, drop_last=False)


def glob_patches(bag_dir, magnification="single"):
    """compute_feats.py:64-67."""
    if magnification in ("single", "low"):
        pats = [os.path.join(bag_dir, "*.jpg"), os.path.join(bag_dir, "*.jpeg")]
    else:
        pats = [os.path.join(bag_dir, "*" + os.sep + "*.jpg"), os.path.join(bag_dir, "*" + os.sep + "*.jpeg")]
    out = []
    for p in pats:
        out += glob.glob(p)
    return out


def read_files(paths):
    """The bytes of a group of files -> list of uint8 arrays (bytes-likes: views of ONE buffer), through dsmil_read_files: one
    C call for the whole group, no interpreter time per file (a Python thread pays ~25 us per open / read / close, serialised
    over all threads — more than the device needs for the tile).  Raises what `open` would for a file that cannot be read."""
    from . import _native
    paths = [os.fsencode(p) for p in paths]
    n = len(paths)
    if n == 0:
        return []
    blob = b"\0".join(paths) + b"\0"
    off = np.zeros(n, np.int64)
    np.cumsum([len(p) + 1 for p in paths[:-1]], out=off[1:])
    sizes = np.empty(n, np.int64)
    pb = np.frombuffer(blob, np.uint8)
    L = _native.lib()
    total = L.dsmil_read_files(pb.ctypes.data, off.ctypes.data, n, None, 0, sizes.ctypes.data)
    buf = np.empty(max(1, int(total)) + 4096, np.uint8)       # (+4096: a file that grew between the two calls)
    total = L.dsmil_read_files(pb.ctypes.data, off.ctypes.data, n, buf.ctypes.data, buf.size, sizes.ctypes.data)
    if total < 0:
        _native.check(int(total), "dsmil_read_files")
    bad = np.nonzero(sizes < 0)[0]
    if bad.size:
        with open(paths[int(bad[0])], "rb") as f:             # raises the OSError the Python loader would have raised
            f.read()
        raise OSError(f"cannot read {os.fsdecode(paths[int(bad[0])])}")
    ends = np.cumsum(sizes)
    return [buf[int(e - s_):int(e)] for s_, e in zip(sizes, ends)]


_decode_streams = {}


class _ChunkDecoder:
    """Device JPEG decode of a slide's chunks on a side stream, double-buffered.
    decode(blobs) -> uint8 [n, H, W, 3] (a view of one of two staging buffers), enqueued on the decode stream; the CALLER's
    current stream is made to wait for it.  release() — called when the consumer has ENQUEUED its work on the chunk — marks the
    buffer: the decode that reuses it (two chunks later) waits for that point of the consumer's stream.
    The stream has HIGH priority: (a) its own hardware queue — the runtime multiplexes a process's normal-priority streams over
    four hardware queues, so a fifth stream created behind the embed pool's would share one with conv kernels and the ~11 ms
    decode launch would sit in line behind them (measured: 35.7 k instead of 55.5 k patches/s); (b) the dispatcher places the
    decode's few workgroups as soon as compute units free up."""

    def __init__(self, device, decode_batch, stats=None, nbuf=2, nstreams=1):
        self.dev = torch.device(device)
        pool = _decode_streams.setdefault(str(self.dev), [])
        while len(pool) < nstreams:                       # (created once per process and device, high priority: see above)
            pool.append(torch.cuda.Stream(device=self.dev, priority=-1))
        self.dss = pool[:nstreams]
        self.ds = self.dss[0]
        self.decode_batch, self.stats = decode_batch, stats
        self.bufs, self.free_ev = [None] * nbuf, [None] * nbuf
        self.size, self.k, self.n, self.last = None, 0, 0, None

    def begin(self, blobs):
        """Enqueue the decode of a chunk into the next staging buffer, on the next decode stream; no host wait
        (ops.jpeg_decode_begin)."""
        k = self.k
        self.k = (k + 1) % len(self.bufs)
        ds = self.dss[self.n % len(self.dss)]
        self.n += 1
        def buffer_free():                                # the consumer's work on the chunk that used this buffer (one event, or
            if self.free_ev[k] is not None:               # one per stream) — waited for BEHIND the files' host-to-device copies:
                for e in (self.free_ev[k] if isinstance(self.free_ev[k], list) else [self.free_ev[k]]):   # a pageable copy
                    ds.wait_event(e)                      # queued behind this wait held the host until the buffer was free

        with torch.cuda.stream(ds):                       # (the inputs are host bytes: nothing of the caller's stream to wait for)
            p = ops.jpeg_decode_begin(blobs, self.dev, size=self.size, out=self.bufs[k], before_launch=buffer_free)
            if self.bufs[k] is None and p.out.shape[0] == self.decode_batch:
                self.bufs[k] = p.out                      # (a short last chunk is not worth keeping)
                for d2 in self.dss:                       # (allocated on one decode stream, written by the others later)
                    if d2 is not ds:
                        p.out.record_stream(d2)
            self.size = tuple(p.out.shape[1:3])
        return p, k, ds

    def end(self, item):
        """Wait for a begin()'s status, redo on the host what the device did not decode -> (imgs, event, buffer index)."""
        p, k, ds = item
        with torch.cuda.stream(ds):
            imgs, redone = ops.jpeg_decode_end(p, stats=self.stats)
            ev = p.ev
            if ev is None or redone:                      # host-decoded files were copied in on the side stream just now
                ev = torch.cuda.Event()                   # (behind a later chunk's decode if one is already queued there)
                ev.record(ds)
        return imgs, ev, k

    def decode(self, blobs):
        return self.end(self.begin(blobs))

    def acquire(self, item):
        """Make the caller's current stream wait for a decode()'s result; returns the images."""
        imgs, ev, k = item
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(ev)
        imgs.record_stream(cur)
        return imgs

    def release(self, item):
        e = torch.cuda.Event()
        e.record(torch.cuda.current_stream(self.dev))
        self.free_ev[item[2]] = e

    def finish(self):
        for ds in self.dss:
            torch.cuda.current_stream(self.dev).wait_stream(ds)


@torch.no_grad()
def embed_jpeg_blobs(i_classifier, blobs, batch_size=256, decode_batch=2048, streams=3, device=None, stats=None):
    """A slide's tiles as JPEG FILES IN HOST MEMORY (bytes-likes) -> (feats [N,F], classes [N,C]) on the device: chunks of
    `decode_batch` files are decoded on the device (ops.jpeg_decode) on a side stream into one of two staging buffers while the
    previous chunk is embedded (embed_tiles: batches of `batch_size` over `streams` HIP streams) — the decode's 256-lane
    workgroups hold a handful of compute units, the conv kernels the rest.  compute_feats.py:55-76 with the loader's Pillow
    workers and the 150 KB-per-tile H2D copy replaced by an ~11 KB-per-tile copy and four launches per chunk."""
    dev = torch.device(device) if device is not None else next(i_classifier.parameters()).device
    n = len(blobs)
    F_, C_ = i_classifier.fc.in_features, i_classifier.fc.out_features
    if n == 0:
        return torch.zeros((0, F_), device=dev), torch.zeros((0, C_), device=dev)
    chunks = [blobs[i:i + decode_batch] for i in range(0, n, decode_batch)]
    return _embed_jpeg_chunks(i_classifier, chunks, batch_size, decode_batch, streams, dev, stats)


def _embed_jpeg_chunks(i_classifier, chunks, batch_size, decode_batch, streams, dev, stats=None):
    """embed_jpeg_blobs over a sequence of chunks (lists of bytes-likes; entries may be concurrent.futures-style lazily read:
    anything with .result() is resolved when its chunk is decoded; a result that is a LIST stands for that many files)."""
    def resolve(chunk):
        out = []
        for b in chunk:
            r = b.result() if hasattr(b, "result") else b
            if isinstance(r, list):                       # (a read task's files)
                out.extend(r)
            else:
                out.append(r)
        return out

    pool = stream_pool(dev, streams) if streams > 1 else None
    ahead = max(1, min(DECODE_AHEAD[0], len(chunks)))
    dec = _ChunkDecoder(dev, decode_batch, stats, nbuf=2 if pool is None else ahead + 1, nstreams=1 if pool is None else DECODE_STREAMS[0])
    fl, cl = [], []
    if pool is None:
        cur = dec.decode(resolve(chunks[0]))
        for ci in range(len(chunks)):
            imgs = dec.acquire(cur)
            f, c = embed_tiles(i_classifier, imgs, batch_size, streams=1, device=dev)
            dec.release(cur)
            if ci + 1 < len(chunks):
                cur = dec.decode(resolve(chunks[ci + 1]))
            fl.append(f)
            cl.append(c)
        dec.finish()
        return torch.cat(fl), torch.cat(cl)
    # The batches of ALL chunks go round-robin over the pool's streams, and a pool stream waits for the DECODE of the chunk it
    # is about to read — not for the caller's stream: embed_tiles per chunk joined the pool into the caller's stream and the next
    # chunk's first batches waited for that join, i.e. the pipeline drained at every chunk boundary (8 batches on 3 streams:
    # the last round of a chunk is two batches wide): 53 k -> 61 k patches/s for a 10 000-tile slide (bench.py `slide_jpeg`).
    # SEVERAL decodes in flight (DECODE_AHEAD, on DECODE_STREAMS side streams, one staging buffer more): a decode is ~11 ms of
    # latency whatever the chunk size (one lane per tile in the Huffman kernel; 17-25 ms beside the conv kernels) and the host
    # must read its status before it may hand the chunk on.  With one decode in flight the host sat in that wait and THEN
    # enqueued the chunk's ~500 launches, so the next decode started late and a 16-bit trunk (15 ms per 2 048 tiles) ran out of
    # decoded tiles: 115 ms per 10 000-tile slide against 75 ms from decoded tiles; now 98 ms (the first decode, which nothing
    # can hide, + ~10 ms of the decodes' share of the device).  The fp32-class trunk (31 ms per chunk) never waited: unchanged.
    caller = torch.cuda.current_stream(dev)
    for s_ in pool.streams:
        s_.wait_stream(caller)                            # (weights / earlier work of the caller's stream)
    pending, nxt = [], 0
    while nxt < len(chunks) and len(pending) < ahead:
        pending.append(dec.begin(resolve(chunks[nxt])))
        nxt += 1
    bi = 0
    for ci in range(len(chunks)):
        imgs, ev, k = dec.end(pending.pop(0))
        used = set()
        for lo in range(0, imgs.shape[0], batch_size):
            s_ = pool.streams[bi % len(pool.streams)]
            bi += 1
            if id(s_) not in used:
                s_.wait_event(ev)                         # this chunk's decode
                used.add(id(s_))
            with torch.cuda.stream(s_):
                f, c = i_classifier(imgs[lo:lo + batch_size])
            fl.append(f)
            cl.append(c)
        done = []
        for s_ in pool.streams:                           # the staging buffer is free when every stream is past this chunk
            if id(s_) in used:
                imgs.record_stream(s_)
                e = torch.cuda.Event()
                e.record(s_)
                done.append(e)
        dec.free_ev[k] = done
        if nxt < len(chunks):
            pending.append(dec.begin(resolve(chunks[nxt])))   # chunk ci + 2, into the buffer chunk ci - 1 was embedded from
            nxt += 1
    for s_ in pool.streams:
        caller.wait_stream(s_)
    for ds in dec.dss:
        caller.wait_stream(ds)
    for t in fl + cl:                                     # produced on a pool stream, consumed (and freed) on the caller's
        t.record_stream(caller)
    return torch.cat(fl), torch.cat(cl)


EMBED_STREAMS = [3]     # HIP streams of embed_files' device-decode path (batches dealt round-robin, ops.StreamPool)
DECODE_AHEAD = [3]      # decodes in flight in front of the chunk being embedded (staging buffers: one more; tools/_ahead sweep:
                        # 1 decode stream x 1 ahead 115 ms per 10 000-tile slide on the fp16-activation trunk, 2 x 3: 98 ms)
DECODE_STREAMS = [2]    # decode streams of the pooled path: the Huffman kernel is latency-bound on a handful of compute units
DECODE_BATCH = [2048]   # files per device-decode chunk of embed_files(gpu_decode=True) (tests lower it to walk the double buffer)


# Default of embed_files(gpu_decode=None): the scripts set it from their --gpu_decode flag (compute_feats.py, attention_map.py)
GPU_DECODE = [False]


def gpu_decoded_batches(files, batch_size, device, io_threads=4, decode_batch=2048, stats=None):
    """The loader of compute_feats.py:21-56 (`Image.open` + ToTensor in DataLoader workers) with the decode on the DEVICE: the
    files' bytes are read by a thread pool, `decode_batch` files at a time go through ops.jpeg_decode (baseline JPEGs:
    dsmil_jpeg_decode, bit-identical to Pillow; anything else: Pillow on the host inside the same call) and come back as uint8
    NHWC batches of `batch_size` already on `device` — what PatchFiles(uint8=True) yields after its H2D copy, at a tenth of the
    PCIe bytes and without worker processes.  The next chunk is read and decoded (on the side stream of _ChunkDecoder) when the
    consumer has taken — i.e. enqueued its work on — the last batch of the current one."""
    from concurrent.futures import ThreadPoolExecutor

    read = read_files                                     # (a task per 64 files, one C call per task: embed_files)

    files = list(files)
    if not files:
        return
    dec = _ChunkDecoder(device, decode_batch, stats)
    with ThreadPoolExecutor(max_workers=max(1, int(io_threads))) as pool:
        def start(chunk):
            return [pool.submit(read, chunk[j:j + 64]) for j in range(0, len(chunk), 64)]

        def collect(tasks):
            return [b for t in tasks for b in t.result()]

        chunks = [files[i:i + decode_batch] for i in range(0, len(files), decode_batch)]
        cur = dec.decode(collect(start(chunks[0])))
        for ci in range(len(chunks)):
            pending = start(chunks[ci + 1]) if ci + 1 < len(chunks) else None   # (read while this chunk is consumed)
            imgs = dec.acquire(cur)
            for o in range(0, imgs.shape[0], batch_size):
                yield {"input": imgs[o:o + batch_size]}
            dec.release(cur)                              # the consumer's work on this chunk is enqueued by now
            if pending is not None:
                cur = dec.decode(collect(pending))
        dec.finish()


@torch.no_grad()
def embed_files(i_classifier, files, batch_size=128, num_workers=4, device=None, want_position=False,
                sharded=True, bg_threshold=None, return_keep=False, gpu_decode=None):
    """The hot loop of compute_feats.py:70-76 / attention_map.py:69-79.  Returns
    (feats [N,F], classes [N,C]) on `device` (and positions [N,2] if asked).  Sharded over ranks
    when torch.distributed is initialised.

    ``bg_threshold`` (new, default off): drop background tiles before they are embedded, by the tilers' own criterion —
    keep a tile iff mean(FIND_EDGES band sums) / tile_size^2 > bg_threshold (deepzoom_tiler.py:56-61, `-t`, 15 there).
    The statistics come from dsmil_tile_stats on the decoded uint8 batch (exact integer sums, decisions identical to
    PIL's); rows, logits and positions of dropped tiles are absent from the result (N = tiles kept), as if the tiler
    had not written them.  Still ONE collective per slide: the keep flags travel with the rows.
    ``return_keep``: also return the bool keep mask over `files` (numpy).
    ``gpu_decode`` (new, default off = GPU_DECODE[0]): decode the tiles' JPEG files on the device (gpu_decoded_batches) instead of
    in DataLoader workers; the decoded bytes are identical, so are the features."""
    device = device or next(i_classifier.parameters()).device
    world, rank = ddist.world_rank() if sharded else (1, 0)
    n_total = len(files)
    lo, hi = ddist.shard_range(n_total, rank, world)
    feats_l, cls_l, keep_l = [], [], []
    # decoded uint8 images go to the GPU as they are when the native ResNet-18-IN stem will take them
    from .modules import resnet_convs_of
    u8 = torch.device(device).type == "cuda" and resnet_convs_of(i_classifier.feature_extractor) is not None
    filt = bg_threshold is not None
    F_, C_ = i_classifier.fc.in_features, i_classifier.fc.out_features
    on_gpu = torch.device(device).type == "cuda"

    def embed(patches, keep_idx):
        if keep_idx is not None:
            if keep_idx.numel() == 0:
                return
            patches = patches.index_select(0, keep_idx)
        if patches.shape[0] == 0:
            return
        if not u8:
            patches = patches.permute(0, 3, 1, 2).float().div(255) if filt else patches.float()   # == VF.to_tensor
        feats, classes = i_classifier(patches)
        feats_l.append(feats)
        cls_l.append(classes)

    def finish(p):   # p = (patches, keep [device], keep [host copy in flight], its event)
        patches, keep, keep_h, ev = p
        if ev is not None:
            ev.synchronize()          # recorded one batch ago: the next batch's decode and H2D ran meanwhile
        embed(patches, torch.nonzero(keep_h)[:, 0].to(device, non_blocking=True))

    if gpu_decode is None:
        gpu_decode = GPU_DECODE[0]
    if hi > lo and gpu_decode and u8 and on_gpu and not filt:
        # Round 6: the device-decode path deals its batches to the stream pool (the loop below embeds batch by batch on ONE
        # stream: ~59 k patches/s at bs 256 against 65 k on three) — the files of a chunk are read by a thread pool while the
        # chunk in front is decoded and embedded; same bytes, same batches, same features
        from concurrent.futures import ThreadPoolExecutor

        read = read_files     # one task per 64 files (a future per FILE cost 26 us to submit and as much to wait for — 105 + 150
        #                       ms of a 4 000-tile bag's 318), one C call per task (no interpreter time per file)

        mine = list(files[lo:hi])
        with ThreadPoolExecutor(max_workers=max(1, int(num_workers))) as tp:
            chunks = [[tp.submit(read, mine[j:min(j + 64, i + DECODE_BATCH[0])]) for j in range(i, min(i + DECODE_BATCH[0], len(mine)), 64)]
                      for i in range(0, len(mine), DECODE_BATCH[0])]
            f_, c_ = _embed_jpeg_chunks(i_classifier, chunks, batch_size, DECODE_BATCH[0], EMBED_STREAMS[0], torch.device(device))
        feats_l.append(f_)
        cls_l.append(c_)
    elif hi > lo:
        pending = None
        loader = (gpu_decoded_batches(files[lo:hi], batch_size, device, io_threads=max(1, num_workers), decode_batch=DECODE_BATCH[0])
                  if gpu_decode and u8 and on_gpu
                  else patch_loader(files[lo:hi], batch_size, num_workers, False, uint8=u8 or filt))
        for batch in loader:
            patches = batch["input"].to(device, non_blocking=True)
            if not filt:
                embed(patches, None)
                continue
            # the keep mask is computed on the device and read by the host ONE BATCH LATER, so the loader, the H2D copy and
            # the embedder of the previous batch do not wait for it
            keep = background_keep_mask(patches, edge_threshold=bg_threshold)
            keep_l.append(keep)
            if on_gpu:
                keep_h = torch.empty(keep.shape, dtype=torch.bool, pin_memory=True)
                keep_h.copy_(keep, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:
                keep_h, ev = keep, None
            if pending is not None:
                finish(pending)
            pending = (patches, keep, keep_h, ev)
        if pending is not None:
            finish(pending)
    if feats_l:
        feats, classes = torch.cat(feats_l), torch.cat(cls_l)
    else:
        feats = torch.zeros((0, F_), device=device)
        classes = torch.zeros((0, C_), device=device)
    keep_all = None
    if filt:   # scatter the kept rows back to the shard's tile positions: shard sizes stay the ones every rank knows
        keep_loc = torch.cat(keep_l) if keep_l else torch.zeros(0, dtype=torch.bool, device=device)
        full_f = torch.zeros((hi - lo, F_), device=device)
        full_c = torch.zeros((hi - lo, C_), device=device)
        full_f[keep_loc], full_c[keep_loc] = feats, classes
        feats, classes = full_f, full_c
        keep_all = keep_loc.to(torch.float32)[:, None]
    if sharded and (world > 1 or ddist._FORCE[0]):
        # ONE collective per slide: feature rows and instance logits (and keep flags) travel as one [n_r, F + C (+1)] matrix
        sizes = [ddist.shard_range(n_total, r, world)[1] - ddist.shard_range(n_total, r, world)[0] for r in range(world)]
        parts = ddist.all_gather_packed([feats, classes] + ([keep_all] if filt else []), sizes)
        feats, classes = parts[0], parts[1]
        if filt:
            keep_all = parts[2]
    keep_np = np.ones(n_total, bool)
    if filt:
        kb = keep_all[:, 0] > 0.5
        feats, classes = feats[kb], classes[kb]
        keep_np = kb.cpu().numpy()
    out = (feats, classes)
    if want_position:
        pos = np.vstack([patch_position(f) for f in files]) if n_total else np.zeros((0, 2), int)
        out = out + (pos[keep_np],)
    if return_keep:
        out = out + (keep_np,)
    return out


CSV_THREADS = [8]    # host threads of the feature-file formatter (row blocks; the C function releases the GIL)


def feats_csv_bytes(arr, decimals=4, threads=None):
    """The text `pd.DataFrame(arr).to_csv(index=False, float_format='%.<decimals>f')` produces for a float32 matrix —
    compute_feats.py:80-82 — as bytes, byte for byte: header `0,1,...,F-1`, then the rows from dsmil_csv_format_f32 (exact
    decimal rounding in C, row blocks on `threads` host threads).  5.4 s -> ~0.1 s for a 10 000 x 512 bag."""
    from concurrent.futures import ThreadPoolExecutor
    from . import _native
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    if arr.ndim != 2 or arr.shape[1] == 0:
        raise ValueError("a [rows, cols > 0] matrix is expected")
    rows, cols = arr.shape
    L = _native.lib()
    head = (",".join(str(i) for i in range(cols)) + "\n").encode()
    if rows == 0:
        return head
    threads = max(1, int(CSV_THREADS[0] if threads is None else threads))
    per = max(1, min(512, -(-rows // threads)))

    def block(lo):
        n = min(per, rows - lo)
        cap = n * cols * 64
        buf = np.empty(cap, np.uint8)                     # (uninitialised: a ctypes buffer would be zeroed, 64 B per value)
        w = L.dsmil_csv_format_f32(arr[lo:lo + n].ctypes.data, n, cols, cols, int(decimals), buf.ctypes.data, cap)
        if w < 0:
            _native.check(int(w), "dsmil_csv_format_f32")
        return buf[:w].tobytes()

    los = list(range(0, rows, per))
    if threads == 1 or len(los) == 1:
        parts = [block(lo) for lo in los]
    else:
        with ThreadPoolExecutor(max_workers=threads) as tp:
            parts = list(tp.map(block, los))
    return head + b"".join(parts)


def read_feats_csv(path, threads=None):
    """A feature file written by compute_feats.py:80-82 -> float32 [N, F], what `torch.tensor(pd.read_csv(path).to_numpy(),
    dtype=torch.float32)` gives (train_tcga.py:27-32 + :49): the header line is skipped as pandas consumes it, the data rows
    go through dsmil_csv_parse_f32 in chunks on a few host threads.  None when the file is not a plain numeric table (the
    caller lets pandas read it).  0.93 s -> ~0.1 s for a 10 000 x 512 bag."""
    from concurrent.futures import ThreadPoolExecutor
    from . import _native
    with open(path, "rb") as fh:
        raw = fh.read()
    nl = raw.find(b"\n")
    if nl < 0:
        return None
    cols = raw[:nl].count(b",") + 1
    buf = np.frombuffer(raw, np.uint8)
    n = len(raw)
    threads = max(1, int(CSV_THREADS[0] if threads is None else threads))
    cuts = [nl + 1]
    for t in range(1, threads):                          # chunk boundaries right behind a line break
        want = nl + 1 + (n - nl - 1) * t // threads
        k = raw.find(b"\n", max(want, cuts[-1]))
        if k < 0:
            break
        if k + 1 > cuts[-1]:
            cuts.append(k + 1)
    cuts.append(n)
    L = _native.lib()

    def chunk(i):
        lo, hi = cuts[i], cuts[i + 1]
        if hi <= lo:
            return np.zeros((0, cols), np.float32)
        max_rows = raw.count(b"\n", lo, hi) + 1
        out = np.empty((max_rows, cols), np.float32)
        r = L.dsmil_csv_parse_f32(buf.ctypes.data + lo, hi - lo, cols, out.ctypes.data, max_rows)
        return None if r < 0 else out[:r]

    idx = list(range(len(cuts) - 1))
    if len(idx) == 1:
        parts = [chunk(0)]
    else:
        with ThreadPoolExecutor(max_workers=len(idx)) as tp:
            parts = list(tp.map(chunk, idx))
    if any(p is None for p in parts):
        return None
    return parts[0] if len(parts) == 1 else np.concatenate(parts)


def save_feats_csv(feats, path, npy=False):
    """compute_feats.py:80-82 — the pandas CSV, header 0..F-1, '%.4f' (the same bytes, formatted by feats_csv_bytes; anything
    that is not a float32 matrix goes through pandas itself); with ``npy`` also the exact float32 rows as <bag>.npy
    (the text format quantises to 1e-4)."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    arr = feats.detach().cpu().numpy() if torch.is_tensor(feats) else np.asarray(feats)
    if arr.dtype == np.float32 and arr.ndim == 2 and arr.shape[1] > 0:
        with open(path, "wb") as fh:
            fh.write(feats_csv_bytes(arr))
    else:
        import pandas as pd
        pd.DataFrame(arr).to_csv(path, index=False, float_format="%.4f")
    if npy:
        np.save(os.path.splitext(path)[0] + ".npy", np.ascontiguousarray(arr, dtype=np.float32))


class FeatWriter:
    """The bag files of compute_feats.py:80-82 written BEHIND the loop: one writer thread formats and writes bag i while the
    device embeds bag i + 1 (at most `depth` bags wait: their host copies are what is held).  close() — or leaving the `with`
    block — waits for the files and re-raises a writer's exception; the files are complete when compute_feats returns."""

    def __init__(self, depth=2):
        from concurrent.futures import ThreadPoolExecutor
        self.tp = ThreadPoolExecutor(max_workers=1)
        self.pending, self.depth = [], max(1, int(depth))

    def submit(self, fn, *args, **kwargs):
        """Any file-writing call (the attention maps' PNG encode, attention_map.py:104) behind the loop, in order."""
        while len(self.pending) >= self.depth:
            self.pending.pop(0).result()
        self.pending.append(self.tp.submit(fn, *args, **kwargs))

    def save(self, feats, path, npy=False):
        arr = feats.detach().cpu().numpy() if torch.is_tensor(feats) else np.asarray(feats)   # (the D2H copy: here, in order)
        self.submit(save_feats_csv, arr, path, npy)

    def close(self):
        try:
            for f in self.pending:
                f.result()
        finally:
            self.pending = []
            self.tp.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def _bag_csv_path(save_path, bag_dir):
    parts = bag_dir.rstrip(os.sep).split(os.sep)
    return os.path.join(save_path, parts[-2], parts[-1] + ".csv")


def compute_feats(args, bags_list, i_classifier, save_path=None, magnification="single"):
    """compute_feats.py:58-82."""
    i_classifier.eval()
    _, rank = ddist.world_rank()
    with FeatWriter() as writer:
        for i, bag in enumerate(bags_list):
            files = glob_patches(bag, magnification)
            feats, _ = embed_files(i_classifier, files, args.batch_size, args.num_workers,
                                   bg_threshold=getattr(args, "bg_threshold", None))
            if rank == 0:
                sys.stdout.write("\r Computed: {}/{} -- {} patches".format(i + 1, len(bags_list), int(feats.shape[0])))
                if feats.shape[0] == 0:
                    print("No valid patch extracted from: " + bag)
                else:
                    writer.save(feats, _bag_csv_path(save_path, bag), npy=getattr(args, "save_npy", False))


@torch.no_grad()
def tree_feats_of_bag(bag, embedder_low, embedder_high, tree_fusion="cat", batch_size=128, num_workers=4, ext=("jpg", "jpeg")):
    """One pyramid bag folder -> (tree feats [N_high, 1024 | 512] on the device, high patch files, #low files).
    compute_feats.py:91-114: every high-magnification patch (folder named after its low-magnification parent)
    is paired with its parent: 'cat' -> [high(512) || low(512)], 'fusion' -> high + 0.25*low.  The high patches
    are embedded in batches (the reference runs one forward per patch, :106-109; InstanceNorm is per image)."""
    if tree_fusion not in ("fusion", "cat"):
        raise NotImplementedError(f"{tree_fusion} is not an excepted option for --tree_fusion. "
                                  "This argument accepts 2 options: 'fusion' and 'cat'.")
    low_files = []
    for e in ext:
        low_files += glob.glob(os.path.join(bag, "*." + e))
    low_feats, _ = embed_files(embedder_low, low_files, batch_size, num_workers)
    high_files, parent = [], []
    for idx, lp in enumerate(low_files):
        folder = os.path.join(os.path.dirname(lp), os.path.splitext(os.path.basename(lp))[0])
        hp = []
        for e in ext:
            hp += glob.glob(folder + os.sep + "*." + e)
        high_files += hp
        parent += [idx] * len(hp)
    if not high_files:
        return None, [], len(low_files)
    high_feats, _ = embed_files(embedder_high, high_files, batch_size, num_workers)
    low_of_high = low_feats.index_select(0, torch.as_tensor(parent, device=low_feats.device))
    tree = high_feats + 0.25 * low_of_high if tree_fusion == "fusion" else torch.cat([high_feats, low_of_high], dim=-1)
    return tree, high_files, len(low_files)


@torch.no_grad()
def compute_tree_feats(args, bags_list, embedder_low, embedder_high, save_path=None):
    """compute_feats.py:84-126."""
    embedder_low.eval()
    embedder_high.eval()
    _, rank = ddist.world_rank()
    with FeatWriter() as writer:
        for i, bag in enumerate(bags_list):
            tree, high_files, n_low = tree_feats_of_bag(bag, embedder_low, embedder_high, args.tree_fusion,
                                                        args.batch_size, args.num_workers)
            if rank == 0:
                sys.stdout.write("\r Computed: {}/{} -- {} low / {} high".format(i + 1, len(bags_list), n_low, len(high_files)))
            if tree is None:
                if rank == 0:
                    print("No valid patch extracted from: " + bag)
                continue
            if rank == 0:
                writer.save(tree, _bag_csv_path(save_path, bag), npy=getattr(args, "save_npy", False))
    if rank == 0:
        print("\n")


def load_simclr_weights(i_classifier, state_dict_weights):
    """compute_feats.py:223-231: drop the 4 projection-head tensors (l1/l2 of ResNetSimCLR,
    simclr/models/resnet_simclr.py:19-20), then map the remaining tensors BY POSITION onto the
    IClassifier's own keys and load non-strictly.  Returns the re-keyed dict (saved as
    embedder.pth by the caller, compute_feats.py:232-233)."""
    state_dict_weights = OrderedDict(state_dict_weights)
    for _ in range(4):
        state_dict_weights.popitem()
    new_state_dict = OrderedDict()
    for (_k, v), (k0, _v0) in zip(state_dict_weights.items(), i_classifier.state_dict().items()):
        new_state_dict[k0] = v
    i_classifier.load_state_dict(new_state_dict, strict=False)
    return new_state_dict


# ---------------------------------------------------------------------------------------------
# attention maps (attention_map.py:59-118)
# ---------------------------------------------------------------------------------------------
def rescale_intensity01(img):
    """skimage.exposure.rescale_intensity(img, out_range=(0, 1)) for float input."""
    lo, hi = float(img.min()), float(img.max())
    if hi <= lo:
        return np.zeros_like(img, dtype=np.float64)
    return (img - lo) / (hi - lo)


def attention_colormap(A, pos_arr, bag_prediction, thres, colors, class_names=None, bag_name="", log=print,
                       upsample_device=None):
    """attention_map.py:86-113.  A [N,C] numpy, pos_arr [N,2] (row,col), bag_prediction [C]
    (sigmoid).  Returns the uint8 colour map (32x nearest-neighbour upsampled)."""
    C = A.shape[1]
    class_names = class_names or ["class {}".format(c) for c in range(C)]
    benign, num_pos = True, 0
    colored = None
    for c in range(C):
        if bag_prediction[c] >= thres[c]:
            att = A[:, c]
            num_pos += 1
            layer = att[:, None] * np.asarray(colors[c], dtype=np.float64)[None, :]
            if benign:
                log(bag_name + " is detected as: " + class_names[c])
                colored = layer
            else:
                log("and " + class_names[c])
                colored = colored + layer
            benign = False
    if benign:
        log(bag_name + " is detected as: benign")
        colored = np.zeros((A.shape[0], 3))
    else:
        colored = colored / num_pos
    colored = rescale_intensity01(colored)
    H, W = int(pos_arr[:, 0].max()) + 1, int(pos_arr[:, 1].max()) + 1
    cmap = np.zeros((H, W, 3))
    cmap[pos_arr[:, 0], pos_arr[:, 1]] = colored
    # transform.resize(order=0) x32 then img_as_ubyte: nearest-neighbour upsampling commutes with the per-pixel
    # conversion, so convert at tile resolution and repeat the bytes (32*32 = 1024x less float work)
    small = np.clip(np.rint(cmap * 255.0), 0, 255).astype(np.uint8)
    if upsample_device is not None and torch.device(upsample_device).type == "cuda":
        # the 32 x 32 byte replication on the GPU + one D2H copy of the finished map (np.repeat twice: 18 ms for a
        # 3072 x 3328 map, a broadcast + reshape copy 35 ms)
        # through ONE cached PINNED staging block (a pageable `.cpu()` of the 30 MB map cost 14 of the 18.6 ms of this
        # function); the returned array is a pageable copy of it
        t = torch.from_numpy(small).to(upsample_device)
        up = t.repeat_interleave(32, dim=0).repeat_interleave(32, dim=1)
        host = _pinned_staging(up.numel()).view(up.shape)
        host.copy_(up, non_blocking=True)
        torch.cuda.current_stream(up.device).synchronize()
        return np.array(host.numpy())   # a pageable copy: callers may keep many maps; the ONE pinned block is reused
    return np.repeat(np.repeat(small, 32, axis=0), 32, axis=1)


_pinned = [None]


def _pinned_staging(nbytes):
    """One page-locked uint8 block, grown on demand and reused by every attention_colormap call of the process."""
    if _pinned[0] is None or _pinned[0].numel() < nbytes:
        _pinned[0] = torch.empty(int(nbytes), dtype=torch.uint8, pin_memory=True)
    return _pinned[0][:nbytes]


@torch.no_grad()
def attention_maps(args, bags_list, milnet, colors=None, rng=None, embedder_low=None, embedder_high=None):
    """attention_map.test (:59-118): embed -> aggregate (features never leave the device) ->
    threshold -> colour map PNG (+ optional attention CSV).

    Multi-scale (new: ``args.magnification == "tree"`` with the two embedders): the bag folder holds the
    low-magnification tiles and one sub-folder of high-magnification children per tile (the layout
    compute_tree_feats reads, compute_feats.py:91-101); the [high || low] tree features go through
    ``milnet`` = MILNet(FCLayer(feats_size), BClassifier(feats_size)) and the map is painted at the high tiles'
    positions."""
    from PIL import Image
    milnet.eval()
    rng = rng or np.random
    colors = colors or [rng.choice(range(256), size=3) for _ in range(args.num_classes)]
    _, rank = ddist.world_rank()
    tree = getattr(args, "magnification", "single") == "tree"
    if tree and (embedder_low is None or embedder_high is None):
        raise ValueError("multi-scale attention maps need the low- and the high-magnification embedder")
    out = []
    writer = FeatWriter()        # the PNG encode of a slide's map (0.2 s for a 3 000 x 3 000 map) runs while the next slide is embedded
    try:
        _attention_map_slides(args, bags_list, milnet, embedder_low, embedder_high, tree, colors, rank, writer, out)
    finally:
        writer.close()
    return out


def _save_png(arr, path):
    from PIL import Image
    Image.fromarray(arr).save(path)


def _save_scores(A, pos_arr, path):
    import pandas as pd
    df = pd.DataFrame(A)
    df["pos"] = [str(s) for s in pos_arr]
    df.to_csv(path, index=False)


def _attention_map_slides(args, bags_list, milnet, embedder_low, embedder_high, tree, colors, rank, writer, out):
    for bag in bags_list:
        if tree:
            feats, files, _ = tree_feats_of_bag(bag, embedder_low, embedder_high, getattr(args, "tree_fusion", "cat"),
                                                args.batch_size, args.num_workers, ext=(args.patch_ext,))
            if feats is None:
                continue
            pos_arr = np.vstack([patch_position(f) for f in files])
            _, bag_prediction, A, _ = milnet(feats)
        else:
            files = glob.glob(os.path.join(bag, "*." + args.patch_ext))
            if not files:
                continue
            feats, classes, pos_arr = embed_files(milnet.i_classifier, files, args.batch_size, args.num_workers,
                                                  want_position=True)
            bag_prediction, A, _ = milnet.b_classifier(feats, classes)
        pred = np.atleast_1d(torch.sigmoid(bag_prediction).squeeze().cpu().numpy())
        cmap = attention_colormap(A.cpu().numpy(), pos_arr, pred, args.thres, colors, args.class_name, bag)
        slide = bag.rstrip(os.sep).split(os.sep)[-1]
        if rank == 0:
            writer.submit(_save_png, cmap, os.path.join(args.map_path, slide + ".png"))
            if getattr(args, "export_scores", 0):
                writer.submit(_save_scores, A.cpu().numpy(), pos_arr, os.path.join(args.score_path, slide + ".csv"))
        out.append((slide, pred, cmap))


# ---------------------------------------------------------------------------------------------
# multi-scale end to end (BASELINE configs[4]): WSI array -> tiles at two magnifications -> embed both ->
# [high || low] tree features (compute_feats.py:84-126, 113-114) -> MILNet(feats_size=1024) -> attention map
# (attention_map.py:86-113).  Everything between the decoded slide array and the colour map stays on the
# device; with torch.distributed initialised the LOW tiles (each with its 16 children) are sharded
# contiguously over the ranks, the concatenation happens where both halves were embedded, and ONE all-gather
# of the [N_r, 1024] tree rows precedes aggregation.
# ---------------------------------------------------------------------------------------------
def box_downsample_u8(img, factor=4):
    """[H,W,3] uint8 -> [H/f, W/f, 3] uint8: mean over f x f boxes, round half up — the lower pyramid level of a
    synthetic slide (a real slide's levels come from the scanner / deepzoom_tiler.py:214,226-227; the tiler is
    out of scope, so this is only the data generator of the synthetic two-level WSI)."""
    H, W, C = img.shape
    assert H % factor == 0 and W % factor == 0 and factor * factor * 255 < 32768
    v = img.view(H // factor, factor, W // factor, factor, C)
    s = torch.zeros((H // factor, W // factor, C), dtype=torch.int16, device=img.device)
    for dy in range(factor):           # f*f strided byte reads, one int16 accumulator: no 4x-sized temporary
        for dx in range(factor):
            s += v[:, dy, :, dx, :]
    n = factor * factor
    return ((s + n // 2) // n).to(torch.uint8)


def tile_grid(img, tile=224):
    """[H,W,3] uint8 -> ([gy*gx, tile, tile, 3] uint8 contiguous tiles in row-major grid order, gy, gx)."""
    H, W, C = img.shape
    gy, gx = H // tile, W // tile
    if (tile * C) % 8 == 0 and (W * C) % 8 == 0 and img.is_contiguous() and img.data_ptr() % 8 == 0:   # 8-byte words: pyramid_tiles
        w8 = img.view(H, W * C).view(torch.int64)[: gy * tile, : gx * tile * C // 8]
        t8 = w8.reshape(gy, tile, gx, tile * C // 8).permute(0, 2, 1, 3).reshape(gy * gx, tile, tile * C // 8).contiguous()
        return t8.view(torch.uint8).view(gy * gx, tile, tile, C), gy, gx
    t = img[: gy * tile, : gx * tile].view(gy, tile, gx, tile, C).permute(0, 2, 1, 3, 4)
    return t.reshape(gy * gx, tile, tile, C).contiguous(), gy, gx


def pyramid_tiles(wsi, tile=224, factor=4, lo=0, hi=None):
    """Two-level tiling of a slide array [H,W,3] uint8 (H, W multiples of tile*factor): every low-magnification
    tile covers factor x factor high-magnification tiles (20x / 5x: factor 4, deepzoom_tiler.py:214,226-227).
    For the low tiles [lo, hi) of the row-major low grid (default: all) returns
    (low [n,t,t,3], high [n*f*f,t,t,3] ordered parent-major then row-major inside the parent — the order
    compute_tree_feats walks, compute_feats.py:98-109 —, parent [n*f*f] long (index into `low`),
    pos_high [n*f*f,2] (row, col) in the high-magnification grid).  One gather copy per level."""
    H, W, C = wsi.shape
    assert H % (tile * factor) == 0 and W % (tile * factor) == 0, "slide sides must be multiples of tile*factor"
    gy, gx = H // (tile * factor), W // (tile * factor)
    hi = gy * gx if hi is None else hi
    dev = wsi.device
    li = torch.arange(lo, hi, device=dev)
    ly, lx = li // gx, li % gx
    low_img = box_downsample_u8(wsi, factor)
    low = low_img.view(gy, tile, gx, tile, C).permute(0, 2, 1, 3, 4)[ly, lx].contiguous()
    # [gy, gx, f(row), f(col), tile, tile, C] view of the slide; one copy of this range's tiles.  The copy moves 8-byte words
    # where a tile row (tile * C bytes) and a slide row are multiples of 8 bytes: torch's strided copy works per ELEMENT, and
    # 1.5 G uint8 elements took 6.8 ms for a 10 000-tile slide against 0.5 ms (whole grid) / 1.5 ms (a range) as int64
    n_hi = (hi - lo) * factor * factor
    if (tile * C) % 8 == 0 and (W * C) % 8 == 0 and wsi.is_contiguous() and wsi.data_ptr() % 8 == 0:
        w8 = wsi.view(H, W * C).view(torch.int64)
        h8 = w8.view(gy, factor, tile, gx, factor, tile * C // 8).permute(0, 3, 1, 4, 2, 5)
        h8 = h8.reshape(n_hi, tile, tile * C // 8) if (lo == 0 and hi == gy * gx) else h8[ly, lx].reshape(n_hi, tile, tile * C // 8)
        high = h8.view(torch.uint8).view(n_hi, tile, tile, C)
    else:
        hv = wsi.view(gy, factor, tile, gx, factor, tile, C).permute(0, 3, 1, 4, 2, 5, 6)
        high = hv[ly, lx].reshape(n_hi, tile, tile, C)
    cy = torch.arange(factor, device=dev).repeat_interleave(factor)  # child offsets, row-major
    cx = torch.arange(factor, device=dev).repeat(factor)
    rows = (ly[:, None] * factor + cy[None, :]).reshape(-1)
    cols = (lx[:, None] * factor + cx[None, :]).reshape(-1)
    parent = torch.arange(hi - lo, device=dev).repeat_interleave(factor * factor)
    return low, high, parent, torch.stack([rows, cols], dim=1)


def tile_stats_reference(tiles):
    """CPU restatement (numpy, exact integers / float64) of the two tiler filters' per-tile sums — see
    csrc/tile_filter.hip; used for CPU tensors and as the checker of the HIP kernel.  tiles: uint8 [B,H,W,3]."""
    a = np.asarray(tiles.cpu() if torch.is_tensor(tiles) else tiles).astype(np.int64)
    B, H, W, _ = a.shape
    edge = a.copy()                                   # borders are copied from the input (Pillow Filter.c)
    if H > 2 and W > 2:
        nb = sum(a[:, 1 + dy:H - 1 + dy, 1 + dx:W - 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1))
        edge[:, 1:-1, 1:-1] = np.clip(9 * a[:, 1:-1, 1:-1] - nb, 0, 255)       # 8*c - 8 neighbours
    M, m = a.max(-1), a.min(-1)
    f = a.astype(np.float64) / 255.0                  # skimage img_as_float
    v, delta = f.max(-1), f.max(-1) - f.min(-1)       # rgb2hsv: out_v, delta = ptp
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(delta == 0.0, 0.0, delta / v)
    sat = np.rint(s * 255.0).astype(np.int64)         # img_as_ubyte of a float image in [0,1]
    del M, m
    return np.concatenate([edge.sum((1, 2)), sat.sum((1, 2))[:, None]], axis=1)


def background_keep_mask(tiles, edge_threshold=15, sat_threshold=None):
    """Which tiles the reference's tilers would keep: deepzoom_tiler.py:56-61 (mean of the FIND_EDGES band sums /
    tile_size^2 > edge_threshold, default 15 = `-t`) and / or test_crop_single.py:17-24 (mean ubyte saturation >=
    sat_threshold, 30 at its call site).  Device tiles run dsmil_tile_stats; the decisions are made in float64 from
    the exact integer sums, as numpy does.  Returns a bool tensor [B] on the tiles' device."""
    B, H, W = tiles.shape[0], tiles.shape[1], tiles.shape[2]
    if torch.is_tensor(tiles) and tiles.is_cuda:
        # decided ON the device, no host round trip (the embed loop reads the mask one batch later): the sums are exact
        # integers < 2^53, so float64 sum / 3 / H^2 rounds exactly as numpy's mean(...) / tile_size**2 does (torch.mean would
        # multiply by 1/3 instead of dividing)
        st = ops.tile_stats(tiles).to(torch.float64)
        keep = torch.ones(B, dtype=torch.bool, device=tiles.device)
        if edge_threshold is not None:
            keep &= (st[:, 0] + st[:, 1] + st[:, 2]) / 3.0 / float(H ** 2) > edge_threshold
        if sat_threshold is not None:
            keep &= st[:, 3] / float(H * W) >= sat_threshold
        return keep
    st = tile_stats_reference(tiles)
    keep = np.ones(B, bool)
    if edge_threshold is not None:
        edge = st[:, :3].astype(np.float64).mean(axis=1) / float(H ** 2)   # np.mean(edge) / tile_size**2
        keep &= edge > edge_threshold
    if sat_threshold is not None:
        keep &= st[:, 3].astype(np.float64) / float(H * W) >= sat_threshold
    dev = tiles.device if torch.is_tensor(tiles) else "cpu"
    return torch.from_numpy(keep).to(dev)


_pools = {}


def stream_pool(device, streams):
    """The StreamPool of a device (one per (device, width), created on first use)."""
    key = (str(device), int(streams))
    if key not in _pools:
        _pools[key] = ops.StreamPool(streams, device)
    return _pools[key]


@torch.no_grad()
def embed_tiles(i_classifier, tiles, batch_size=256, streams=3, device=None):
    """IClassifier over uint8 NHWC tiles in batches (compute_feats.py:70-76 without the JPEG decode).  Returns
    (feats [N,F], classes [N,C]) on the device.
    `tiles` resident on the GPU: batches are independent (InstanceNorm is per image), so they are dealt round-robin to
    `streams` HIP streams (ops.StreamPool): the kernels of one forward have few workgroup rounds each and latency-bound
    phases that a second batch fills (41.8 k -> 46.4 k patches/s with two streams, 47.0 k with three, bs 256).
    `tiles` in HOST memory (decoded tiles as the loader hands them over, compute_feats.py:71 `patches.cuda()`): each
    batch's H2D copy is issued non-blocking on the pool stream that will run its forward, in front of it — with pinned
    memory the copy engines move batch i+1 .. i+streams-1 while the convs of batch i run, so PCIe time (38.5 MB per 256
    tiles) hides under compute; `device` names the GPU (default: the classifier's)."""
    fl, cl = [], []
    on_host = not tiles.is_cuda
    dev = (torch.device(device) if device is not None else next(i_classifier.parameters()).device) if on_host else tiles.device
    use_gpu = dev.type == "cuda"
    pool = stream_pool(dev, streams) if (use_gpu and streams > 1 and tiles.shape[0] > batch_size) else None

    def one(lo):
        x = tiles[lo:lo + batch_size]
        if on_host and use_gpu:
            x = x.to(dev, non_blocking=True)    # allocated and filled on the stream that consumes it
        return i_classifier(x)

    # (round 6, measured and not kept: the copies on their OWN stream, 3 / 6 / 12 batches ahead of the forwards — 165-170 ms
    # per 10 000-tile slide against 159 ms with each copy on the stream that consumes it)
    for lo in range(0, tiles.shape[0], batch_size):
        f, c = pool.run(one, lo) if pool is not None else one(lo)
        fl.append(f)
        cl.append(c)
    if pool is not None:
        pool.join()
        cur = torch.cuda.current_stream(dev)
        for t in fl + cl:                               # produced on a pool stream, consumed (and freed) on this one
            t.record_stream(cur)
    if not fl:
        return (torch.zeros((0, i_classifier.fc.in_features), device=dev),
                torch.zeros((0, i_classifier.fc.out_features), device=dev))
    return torch.cat(fl), torch.cat(cl)


def _multiscale_embed_pooled(wsi, embedder_low, embedder_high, tile, factor, lo, hi, batch_size, streams):
    """The two embedding passes of multiscale_bag with every batch — low or high — dealt to ONE stream pool and a
    single join: the high-magnification tiles are gathered from the slide per batch INSIDE the pooled call (the 3 GB
    of copy traffic of a whole-slide gather then runs under other batches' conv kernels instead of in front of them),
    and the few low-magnification batches overlap with the high ones.  Same tiles, same order, same batches as
    pyramid_tiles + embed_tiles (bit-identical features)."""
    H, W, C = wsi.shape
    gy, gx = H // (tile * factor), W // (tile * factor)
    dev = wsi.device
    ch = factor * factor
    li = torch.arange(lo, hi, device=dev)
    ly, lx = li // gx, li % gx
    words = (tile * C) % 8 == 0 and (W * C) % 8 == 0 and wsi.is_contiguous() and wsi.data_ptr() % 8 == 0
    if words:      # the gathers move 8-byte words (pyramid_tiles: torch's indexed copy works per element, 13x slower on bytes)
        hv = wsi.view(H, W * C).view(torch.int64).view(gy, factor, tile, gx, factor, tile * C // 8).permute(0, 3, 1, 4, 2, 5)
    else:
        hv = wsi.view(gy, factor, tile, gx, factor, tile, C).permute(0, 3, 1, 4, 2, 5, 6)   # [gy, gx, f, f, t, t, C]
    pool = stream_pool(dev, streams)
    ppb = max(1, batch_size // ch)                      # parents per high batch: the same 256-tile batches as before

    def low_all():   # ONE pooled call: the 4 ms box filter of the whole slide and the few low batches run beside the high ones
        low_img = box_downsample_u8(wsi, factor)
        low = low_img.view(gy, tile, gx, tile, C).permute(0, 2, 1, 3, 4)[ly, lx].contiguous()
        return [embedder_low(low[a:a + batch_size])[0] for a in range(0, hi - lo, batch_size)]

    def high_batch(a, b):
        t = hv[ly[a:b], lx[a:b]]
        t = t.reshape((b - a) * ch, tile, tile * C // 8).view(torch.uint8) if words else t
        return embedder_high(t.reshape((b - a) * ch, tile, tile, C))[0]

    fl = pool.run(low_all)
    fh = [pool.run(high_batch, a, min(a + ppb, hi - lo)) for a in range(0, hi - lo, ppb)]
    pool.join()
    cur = torch.cuda.current_stream(dev)
    for t in fh + fl:
        t.record_stream(cur)
    cy = torch.arange(factor, device=dev).repeat_interleave(factor)
    cx = torch.arange(factor, device=dev).repeat(factor)
    rows = (ly[:, None] * factor + cy[None, :]).reshape(-1)
    cols = (lx[:, None] * factor + cx[None, :]).reshape(-1)
    parent = torch.arange(hi - lo, device=dev).repeat_interleave(ch)
    return torch.cat(fl), torch.cat(fh), parent, torch.stack([rows, cols], dim=1)


@torch.no_grad()
def multiscale_bag(wsi, embedder_low, embedder_high, tree_fusion="cat", tile=224, factor=4, batch_size=256,
                   timings=None, streams=3):
    """Slide array -> tree features [N_high, 1024] ('cat': [high || low], compute_feats.py:113-114) or
    [N_high, 512] ('fusion': high + 0.25 low, :111-112), plus the high tiles' grid positions [N_high, 2].
    Sharded by LOW tile over the ranks of the default process group; one all-gather of tree rows."""
    if tree_fusion not in ("fusion", "cat"):
        raise NotImplementedError(f"{tree_fusion} is not an excepted option for --tree_fusion. "
                                  "This argument accepts 2 options: 'fusion' and 'cat'.")
    world, rank = ddist.world_rank()
    ch = factor * factor
    L = (wsi.shape[0] // (tile * factor)) * (wsi.shape[1] // (tile * factor))
    lo, hi = ddist.shard_range(L, rank, world)
    if wsi.is_cuda and streams > 1 and hi - lo > 0:
        f_low, f_high, parent, pos = _multiscale_embed_pooled(wsi, embedder_low, embedder_high, tile, factor, lo, hi,
                                                              batch_size, streams)
    else:
        low, high, parent, pos = pyramid_tiles(wsi, tile, factor, lo, hi)   # this rank's low tiles and their children
        f_low, _ = embed_tiles(embedder_low, low, batch_size, streams)
        f_high, _ = embed_tiles(embedder_high, high, batch_size, streams)
    low_of_high = f_low.index_select(0, parent)
    tree = f_high + 0.25 * low_of_high if tree_fusion == "fusion" else torch.cat([f_high, low_of_high], dim=-1)
    if world > 1 or ddist._FORCE[0]:
        # shards are whole low tiles: rows per rank = f*f x its low-tile count; ONE collective carries the tree rows and
        # the (row, col) grid positions the map needs (int64, bit-cast into float lanes)
        import time
        t0 = time.perf_counter()
        sizes = [(ddist.shard_range(L, r, world)[1] - ddist.shard_range(L, r, world)[0]) * ch for r in range(world)]
        tree, pos = ddist.all_gather_packed([tree, pos], sizes)
        if timings is not None:
            timings["allgather_bytes"] = int(sum(sizes)) * (tree.shape[1] * tree.element_size() + pos.shape[1] * pos.element_size())
        if timings is not None:
            if tree.is_cuda:
                torch.cuda.synchronize()
            timings["allgather_s"] = timings.get("allgather_s", 0.0) + time.perf_counter() - t0
    return tree, pos


@torch.no_grad()
def multiscale_attention_map(wsi, embedder_low, embedder_high, milnet, thres, colors, tree_fusion="cat", tile=224,
                             factor=4, batch_size=256, class_names=None, log=lambda *_: None, timings=None, streams=3):
    """configs[4] end to end for ONE slide array: tile -> two-scale embed -> concat -> MILNet(FCLayer(1024, C),
    BClassifier(1024, C)) -> sigmoid(bag logits) vs thresholds -> colour map at high-tile resolution.
    Returns dict(feats, classes, pred, A, B, prob, cmap, pos); only `prob` [C] and the final map leave the device
    (attention_map.py:86-113 does the map on the host too)."""
    feats, pos = multiscale_bag(wsi, embedder_low, embedder_high, tree_fusion, tile, factor, batch_size, timings, streams)
    classes, pred, A, B = milnet(feats)
    prob = np.atleast_1d(torch.sigmoid(pred).squeeze().cpu().numpy())
    cmap = attention_colormap(A.cpu().numpy(), pos.cpu().numpy(), prob, thres, colors, class_names, "slide", log,
                              upsample_device=A.device if A.is_cuda else None)
    return dict(feats=feats, classes=classes, pred=pred, A=A, B=B, prob=prob, cmap=cmap, pos=pos)
