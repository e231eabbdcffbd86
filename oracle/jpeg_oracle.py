"""CPU restatement of baseline-JPEG decoding as the reference's input path performs it — TEST INFRASTRUCTURE ONLY (tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product never does).

The reference decodes every tile with `Image.open(path)` + `VF.to_tensor` (/root/reference/compute_feats.py:28,35-39; the bs = 1
loop at :107) inside DataLoader workers (:55): Pillow's JPEG plugin over libjpeg-turbo — a third-party dependency that is not
under /root/reference (env.yml pins neither; this image carries Pillow 12.2.0 / libjpeg-turbo 3.1.4.1).  What it computes with
Pillow's defaults (dct_method = JDCT_ISLOW, do_fancy_upsampling = TRUE, no draft mode, JCS_YCbCr -> JCS_RGB) is the published
IJG / libjpeg-turbo algorithm restated here in numpy:

  * marker parsing (ITU T.81 Annex B): SOI, APPn/COM skipped, DQT, SOF0 (baseline, 8 bit), DHT, DRI, SOS (one interleaved scan);
  * Huffman decoding of the entropy-coded segment (T.81 Annex F.2.2; byte stuffing FF00, restart markers RSTn);
  * dequantisation + the "islow" integer inverse DCT (jidctint.c: CONST_BITS 13, PASS1_BITS 2, the twelve FIX_* constants),
    level shift and range limit;
  * "fancy" (triangle) chroma upsampling h2v1 / h2v2 (jdsample.c), edge rows / columns replicated as jdmainct.c's context rows do;
  * YCbCr -> RGB with jdcolor.c's 16-bit fixed-point tables.

PINNED: tests/test_jpeg_host.py checks this restatement byte for byte against Pillow's own decode of the committed fixtures
(tests/golden/jpeg_*.jpg + jpeg_golden.npz, written by tests/golden/make_jpeg_golden.py) and of freshly encoded images.
Pure-Python Huffman loops: small images only.
"""
import numpy as np

ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


class Unsupported(ValueError):
    """A JPEG outside the decoder's scope (progressive, arithmetic, 12 bit, CMYK, non-interleaved scans ...)."""


def parse(buf):
    """Markers of one JPEG file (bytes) -> dict(width, height, comps=[(id, h, v, tq, td, ta)], qt={id: [64] natural order},
    dc={id: (counts[16], symbols)}, ac={...}, restart_interval, ecs=(begin, end) of the entropy-coded segment)."""
    b = bytes(buf)
    if len(b) < 4 or b[0] != 0xFF or b[1] != 0xD8:
        raise Unsupported("no SOI")
    pos = 2
    qt, dc, ac = {}, {}, {}
    out = {"restart_interval": 0, "adobe_transform": None}
    comps = None
    while True:
        while pos < len(b) and b[pos] != 0xFF:
            pos += 1
        while pos < len(b) and b[pos] == 0xFF:
            pos += 1
        if pos >= len(b):
            raise Unsupported("no SOS")
        m = b[pos]
        pos += 1
        if m == 0xD8 or (0xD0 <= m <= 0xD7) or m == 0x01:
            continue
        if m == 0xD9:
            raise Unsupported("EOI before SOS")
        L = (b[pos] << 8) | b[pos + 1]
        seg = b[pos + 2:pos + L]
        if m == 0xDB:       # DQT
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                if pq:
                    raise Unsupported("16-bit quantisation table")
                t = np.zeros(64, np.int32)
                t[ZIGZAG] = np.frombuffer(seg[i:i + 64], np.uint8)
                qt[tq] = t
                i += 64
        elif m == 0xC0:     # SOF0
            if seg[0] != 8:
                raise Unsupported("precision")
            out["height"], out["width"] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            n = seg[5]
            comps = [[seg[6 + 3 * i], seg[7 + 3 * i] >> 4, seg[7 + 3 * i] & 15, seg[8 + 3 * i], 0, 0] for i in range(n)]
        elif m in (0xC1, 0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported("not baseline (SOF%d)" % (m - 0xC0))
        elif m == 0xC4:     # DHT
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                counts = list(seg[i + 1:i + 17])
                ns = sum(counts)
                syms = list(seg[i + 17:i + 17 + ns])
                (ac if tc else dc)[th] = (counts, syms)
                i += 17 + ns
        elif m == 0xDD:     # DRI
            out["restart_interval"] = (seg[0] << 8) | seg[1]
        elif m == 0xEE and seg[:5] == b"Adobe":
            out["adobe_transform"] = seg[11]
        elif m == 0xDA:     # SOS
            ns = seg[0]
            if comps is None or ns != len(comps):
                raise Unsupported("non-interleaved scan")
            for i in range(ns):
                cid, t = seg[1 + 2 * i], seg[2 + 2 * i]
                if cid != comps[i][0]:
                    raise Unsupported("scan component order")
                comps[i][4], comps[i][5] = t >> 4, t & 15
            if seg[1 + 2 * ns] != 0 or seg[2 + 2 * ns] != 63 or seg[3 + 2 * ns] != 0:
                raise Unsupported("spectral selection / successive approximation")
            pos += L
            break
        pos += L
    if len(comps) not in (1, 3):
        raise Unsupported("%d components" % len(comps))
    if len(comps) == 3 and out["adobe_transform"] == 0:
        raise Unsupported("Adobe RGB")
    out.update(comps=[tuple(c) for c in comps], qt=qt, dc=dc, ac=ac, ecs=(pos, len(b)))
    return out


def _huff_tables(counts, syms):
    """T.81 Annex C / F.2.2.3: (mincode, maxcode, valptr) per code length 1..16."""
    code, k = 0, 0
    mincode, maxcode, valptr = [0] * 17, [-1] * 17, [0] * 17
    for l in range(1, 17):
        valptr[l] = k
        mincode[l] = code
        code += counts[l - 1]
        k += counts[l - 1]
        maxcode[l] = code - 1 if counts[l - 1] else -1
        code <<= 1
    return mincode, maxcode, valptr, syms


class _Bits:
    def __init__(self, b, pos, end):
        self.b, self.pos, self.end, self.acc, self.n = b, pos, end, 0, 0

    def _fill(self):
        while self.n <= 24:
            if self.pos >= self.end:
                byte = 0
            else:
                byte = self.b[self.pos]
                if byte == 0xFF:
                    nxt = self.b[self.pos + 1] if self.pos + 1 < self.end else 0xD9
                    while nxt == 0xFF and self.pos + 2 < self.end:   # fill bytes (T.81 B.1.1.2)
                        self.pos += 1
                        nxt = self.b[self.pos + 1]
                    if nxt == 0:
                        self.pos += 2
                    else:          # a marker: feed zeros (libjpeg's behaviour at the end of the segment)
                        byte = 0
                else:
                    self.pos += 1
            self.acc = ((self.acc << 8) | byte) & 0xFFFFFFFFFF
            self.n += 8

    def get(self, k):
        if k == 0:
            return 0
        if self.n < k:
            self._fill()
        self.n -= k
        return (self.acc >> self.n) & ((1 << k) - 1)

    def restart(self):
        """Byte-align and step over an RSTn marker."""
        self.acc, self.n = 0, 0
        while self.pos + 1 < self.end and not (self.b[self.pos] == 0xFF and 0xD0 <= self.b[self.pos + 1] <= 0xD7):
            self.pos += 1          # (also steps over fill bytes in front of the marker)
        self.pos += 2


def _decode_sym(bits, tab):
    mincode, maxcode, valptr, syms = tab
    code = 0
    for l in range(1, 17):
        code = (code << 1) | bits.get(1)
        if maxcode[l] >= 0 and code <= maxcode[l] and code >= mincode[l]:
            return syms[valptr[l] + code - mincode[l]]
    raise Unsupported("bad Huffman code")


def _extend(v, s):
    return v if v >= (1 << (s - 1)) else v - (1 << s) + 1


def decode_coefficients(b, hdr):
    """-> per component an int32 array [blocks_y, blocks_x, 64] of quantised coefficients in NATURAL order (padded to MCUs)."""
    comps = hdr["comps"]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    W, H = hdr["width"], hdr["height"]
    mx, my = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    coefs = [np.zeros((my * c[2], mx * c[1], 64), np.int32) for c in comps]
    dct = {k: _huff_tables(*v) for k, v in hdr["dc"].items()}
    act = {k: _huff_tables(*v) for k, v in hdr["ac"].items()}
    bits = _Bits(bytes(b), hdr["ecs"][0], hdr["ecs"][1])
    pred = [0] * len(comps)
    ri, n_mcu = hdr["restart_interval"], 0
    for yy in range(my):
        for xx in range(mx):
            if ri and n_mcu and n_mcu % ri == 0:
                bits.restart()
                pred = [0] * len(comps)
            n_mcu += 1
            for ci, c in enumerate(comps):
                for v in range(c[2]):
                    for h in range(c[1]):
                        blk = coefs[ci][yy * c[2] + v, xx * c[1] + h]
                        s = _decode_sym(bits, dct[c[4]])
                        if s:
                            pred[ci] += _extend(bits.get(s), s)
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = _decode_sym(bits, act[c[5]])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r != 15:
                                    break
                                k += 16
                                continue
                            k += r
                            if k > 63:
                                raise Unsupported("coefficient index")
                            blk[ZIGZAG[k]] = _extend(bits.get(s), s)
                            k += 1
    return coefs


# jidctint.c
_C = dict(F0298=2446, F0390=3196, F0541=4433, F0765=6270, F0899=7373, F1175=9633, F1501=12299, F1847=15137, F1961=16069,
          F2053=16819, F2562=20995, F3072=25172)


def _idct_1d(x, shift):
    """One pass of jpeg_idct_islow over the LAST axis of x (int64 [..., 8]); result descaled by `shift` bits."""
    x = x.astype(np.int64)
    z2, z3 = x[..., 2], x[..., 6]
    z1 = (z2 + z3) * _C["F0541"]
    tmp2 = z1 + z3 * (-_C["F1847"])
    tmp3 = z1 + z2 * _C["F0765"]
    z2, z3 = x[..., 0], x[..., 4]
    tmp0 = (z2 + z3) << 13
    tmp1 = (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = x[..., 7], x[..., 5], x[..., 3], x[..., 1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * _C["F1175"]
    t0, t1, t2, t3 = t0 * _C["F0298"], t1 * _C["F2053"], t2 * _C["F3072"], t3 * _C["F1501"]
    z1, z2, z3, z4 = z1 * (-_C["F0899"]), z2 * (-_C["F2562"]), z3 * (-_C["F1961"]) + z5, z4 * (-_C["F0390"]) + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    r = np.stack([tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3], axis=-1)
    return (r + (1 << (shift - 1))) >> shift


def idct_islow(coefs, q):
    """coefs [..., 64] natural order, q [64] -> uint8 samples [..., 8, 8] (jpeg_idct_islow incl. level shift and range limit)."""
    d = (coefs.astype(np.int64) * q.astype(np.int64)).reshape(coefs.shape[:-1] + (8, 8))
    ws = _idct_1d(np.swapaxes(d, -1, -2), 13 - 2)            # pass 1: columns (the last axis walks a column), scaled by 2^2
    ws = np.swapaxes(ws, -1, -2)
    out = _idct_1d(ws, 13 + 2 + 3)                           # pass 2: rows
    return np.clip(out + 128, 0, 255).astype(np.uint8)


def _planes(coefs, hdr):
    out = []
    for c, cf in zip(hdr["comps"], coefs):
        s = idct_islow(cf, hdr["qt"][c[3]])                 # [by, bx, 8, 8]
        by, bx = s.shape[:2]
        out.append(s.transpose(0, 2, 1, 3).reshape(by * 8, bx * 8))
    return out


def _fancy_h2(p):
    """jdsample.c h2v1_fancy_upsample on rows of p [R, W2] (W2 > 2)."""
    p = p.astype(np.int32)
    R, W2 = p.shape
    o = np.empty((R, 2 * W2), np.int32)
    v = p * 3
    o[:, 0] = p[:, 0]
    o[:, 1] = (v[:, 0] + p[:, 1] + 2) >> 2
    o[:, 2:-2:2] = (v[:, 1:-1] + p[:, :-2] + 1) >> 2
    o[:, 3:-2:2] = (v[:, 1:-1] + p[:, 2:] + 2) >> 2
    o[:, -2] = (v[:, -1] + p[:, -2] + 1) >> 2
    o[:, -1] = p[:, -1]
    return o


def _fancy_h2v2(p):
    """jdsample.c h2v2_fancy_upsample of p [H2, W2] (W2 > 2); rows above the top / below the bottom replicate the edge row."""
    p = p.astype(np.int32)
    H2, W2 = p.shape
    up = np.vstack([p[:1], p[:-1]])       # the nearer neighbour of output row 2r is row r-1
    dn = np.vstack([p[1:], p[-1:]])       # of output row 2r+1: row r+1
    o = np.empty((2 * H2, 2 * W2), np.int32)
    for v, nb in ((0, up), (1, dn)):
        cs = p * 3 + nb                   # column sums
        r = np.empty((H2, 2 * W2), np.int32)
        r[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        r[:, 1] = (cs[:, 0] * 3 + cs[:, 1] + 7) >> 4
        r[:, 2:-2:2] = (cs[:, 1:-1] * 3 + cs[:, :-2] + 8) >> 4
        r[:, 3:-2:2] = (cs[:, 1:-1] * 3 + cs[:, 2:] + 7) >> 4
        r[:, -2] = (cs[:, -1] * 3 + cs[:, -2] + 8) >> 4
        r[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        o[v::2] = r
    return o


def _upsample(p, h, v, hmax, vmax, W, H):
    """Component plane p (padded) with sampling (h, v) -> [H, W] int32 at full resolution."""
    dw, dh = -(-W * h // hmax), -(-H * v // vmax)    # downsampled_width / height of the component
    p = p[:dh, :dw]
    if h == hmax and v == vmax:
        o = p.astype(np.int32)
    elif 2 * h == hmax and v == vmax:
        o = _fancy_h2(p) if dw > 2 else np.repeat(p.astype(np.int32), 2, axis=1)
    elif 2 * h == hmax and 2 * v == vmax:
        o = _fancy_h2v2(p) if dw > 2 else np.repeat(np.repeat(p.astype(np.int32), 2, axis=0), 2, axis=1)
    else:
        raise Unsupported("sampling %dx%d of %dx%d" % (h, v, hmax, vmax))
    return o[:H, :W]


def ycc_to_rgb(y, cb, cr):
    """jdcolor.c ycc_rgb_convert (SCALEBITS 16)."""
    y, cb, cr = y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def decode(buf):
    """bytes of a baseline JPEG -> uint8 [H, W, 3] RGB, as np.array(Image.open(f).convert("RGB")) gives it."""
    hdr = parse(buf)
    planes = _planes(decode_coefficients(buf, hdr), hdr)
    comps = hdr["comps"]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    W, H = hdr["width"], hdr["height"]
    full = [_upsample(p, c[1], c[2], hmax, vmax, W, H) for p, c in zip(planes, comps)]
    if len(full) == 1:
        g = full[0].astype(np.uint8)
        return np.stack([g, g, g], axis=-1)
    return ycc_to_rgb(full[0], full[1], full[2])
