// Batched baseline-JPEG decode of a slide's tiles on the GPU (second sub-item) — byte / integer work.
//
// What it replaces: the reference decodes every tile in DataLoader worker PROCESSES with Pillow (libjpeg-turbo):
//   /root/reference/compute_feats.py:28  `img = Image.open(img)`  (BagDataset.__getitem__, 4 workers: :55, :132)
//   /root/reference/compute_feats.py:107 the bs = 1 loop over high-magnification tiles
//   /root/reference/attention_map.py:69-79 the same loader
// and ships 150 KB of decoded fp32 per 224 x 224 tile over PCIe (:72).  Here the COMPRESSED bytes (10-20 KB per tile at the
// tiler's quality 70, deepzoom_tiler.py:64,250) go to the device and a batch of tiles is decoded there into uint8 NHWC — the
// layout the fused-ingest stem (dsmil_resnet_forward_ex, x_is_u8_nhwc) takes — bit for bit what Pillow produces with its
// defaults (JDCT_ISLOW, fancy upsampling, YCbCr -> RGB); oracle/jpeg_oracle.py restates the algorithm and is pinned to Pillow.
//
// Scope: baseline sequential DCT (SOF0), 8 bit, Huffman, one interleaved scan, 1 or 3 components, luma sampling 1x1 / 2x1 /
// 2x2 with 1x1 chroma (4:4:4, 4:2:2, 4:2:0), restart intervals, any tables (optimised Huffman tables included).  Everything else
// (progressive, arithmetic, 12 bit, CMYK, Adobe RGB, 1x2) gets DSMIL_E_UNSUPPORTED in its image record at PARSE time and is left
// to the caller's Pillow path — never a wrong pixel.
//
// Three launches per batch, all images of a batch the same width x height:
//   k_jpeg_huffman  one LANE per image walks its entropy-coded segment (T.81 F.2.2: 12-bit lookahead table, canonical slow path,
//                   FF00 unstuffing, RSTn) and scatters the non-zero quantised coefficients (int16, zigzag order) into a
//                   zeroed buffer.  Huffman decoding is serial per stream; a slide has 10^4-10^5 streams, which is the
//                   parallelism used: 1 024-lane workgroups (four waves per SIMD hide each other's lookup latency; a batch of
//                   8 192 tiles holds 8 compute units, the embedder's conv kernels of the previous batch keep the rest).
//   k_jpeg_idct     one thread per 8x8 block: dequantise, jidctint.c's integer inverse DCT (CONST_BITS 13, PASS1_BITS 2), level
//                   shift, clamp -> component planes.
//   k_jpeg_color    one thread per output pixel: jdsample.c's triangle ("fancy") chroma upsampling with jdmainct.c's replicated
//                   edge rows, jdcolor.c's fixed-point YCbCr -> RGB, uint8 NHWC out.
// The marker parser (dsmil_jpeg_parse) is host code: it fills one record per image and de-duplicates the tables of the batch
// (a tiler writes the same tables into every tile).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "dsmil_hip.h"

namespace {

constexpr int JP_MAX_QT = DSMIL_JPEG_MAX_TABLES;   // distinct quantisation / Huffman tables per batch
constexpr int JP_MAX_HT = DSMIL_JPEG_MAX_TABLES;
constexpr int JP_LOOK = 12;                        // lookahead bits of the fast Huffman table

struct JpHuff {                 // one Huffman table, device form
    uint16_t look[1 << JP_LOOK];    // (length << 8) | symbol for codes of <= JP_LOOK bits, 0 otherwise
    int32_t maxcode[18];            // largest code of length l (-1: none); [17] = sentinel
    int32_t valoff[18];             // huffval index of the first code of length l minus that code
    uint8_t vals[256];
};
struct JpHeader {
    int32_t n_images, n_qt, n_ht, reserved;
};
// plan blob: JpHeader | dsmil_jpeg_image[n] | uint16 qt[JP_MAX_QT][64] | JpHuff ht[JP_MAX_HT]
__host__ __device__ inline size_t jp_off_images() { return sizeof(JpHeader); }
__host__ __device__ inline size_t jp_off_qt(int n) { return (jp_off_images() + (size_t)n * sizeof(dsmil_jpeg_image) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t jp_off_ht(int n) { return jp_off_qt(n) + (size_t)JP_MAX_QT * 64 * sizeof(uint16_t); }
__host__ __device__ inline size_t jp_plan_bytes(int n) { return jp_off_ht(n) + (size_t)JP_MAX_HT * sizeof(JpHuff); }

static const uint8_t JP_ZIGZAG_H[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// padded geometry of a batch (every image width x height): luma planes padded to whole 16 x 16 MCUs whatever the sampling
struct JpGeom {
    int W, H, Wp, Hp;           // padded to multiples of 16
    int blocks_plane;           // (Hp / 8) * (Wp / 8): 8x8 blocks of a full-resolution plane
    __host__ __device__ JpGeom(int w, int h) : W(w), H(h), Wp((w + 15) & ~15), Hp((h + 15) & ~15), blocks_plane(0) {
        blocks_plane = (Hp >> 3) * (Wp >> 3);
    }
    // per image: coefficients int16 [3 planes][blocks_plane][64]; samples uint8 [3 planes][Hp][Wp]
    __host__ __device__ size_t coef_elems() const { return (size_t)3 * blocks_plane * 64; }
    __host__ __device__ size_t plane_bytes() const { return (size_t)Hp * Wp; }
};

// ---------------------------------------------------------------------------------------------------------------------
// k_jpeg_huffman
// ---------------------------------------------------------------------------------------------------------------------
// SIMT shapes this kernel: the 64 lanes of a wave decode 64 different streams, and a wave executes the UNION of its lanes'
// control paths.  The first form (byte-wise refill loop, 9-bit lookahead + canonical bit-by-bit slow path, zigzag lookup per
// coefficient) ran every rare path in nearly every iteration because SOME lane needed it: ~4 000 cycles per symbol.  Now
//   * one refill point per symbol, at most two byte-group moves, no loop; the stream sits in a 16-byte register queue (`raw` in
//     front, `nxt` prefetched: the global load of the following eight bytes is issued when `raw` is replaced and is not waited
//     for until `raw` runs out again); an 0xFF at the queue's front (stuffed byte or marker) is resolved in registers;
//   * ONE table lookup per symbol for codes of up to JP_LOOK = 12 bits, from LDS; longer codes (rare symbols of the standard
//     tables, < 0.1 % of the stream) take the canonical compare chain;
//   * coefficients are stored in ZIGZAG order (k_jpeg_idct reads them through constant indices): no map lookup per symbol.
struct JpBits {
    const uint8_t* p;       // address of the first byte behind `nxt`
    unsigned long long acc; // the low `n` bits are valid, MSB first
    unsigned long long raw; // the next `rawn` bytes of the stream, first byte in the top byte
    unsigned long long nxt; // the eight bytes behind them (prefetched)
    int n, rawn;
    __device__ __forceinline__ static unsigned long long load8(const uint8_t* q) {
        unsigned long long w;
        __builtin_memcpy(&w, q, 8);
        return __builtin_bswap64(w);
    }
    __device__ __forceinline__ void init(const uint8_t* q) {
        acc = 0; n = 0;
        raw = load8(q); nxt = load8(q + 8); rawn = 8; p = q + 16;
    }
    __device__ __forceinline__ void drop(int k) {       // k <= rawn bytes leave the front of the queue
        raw = k >= 8 ? 0ull : raw << (8 * k);
        rawn -= k;
        if (rawn == 0) { raw = nxt; rawn = 8; nxt = load8(p); p += 8; }
    }
    // one step: bytes from the queue's front into the bit buffer — a whole group while none of them is 0xFF, else one byte
    // with the stuffing / marker rules (a marker stays at the front and feeds zeros: libjpeg's "insufficient data")
    __device__ __forceinline__ void step() {
        int k = (64 - n) >> 3;
        k = k < rawn ? k : rawn;
        const unsigned long long low = k == 8 ? 0ull : (0x0101010101010101ull >> (8 * k));
        const unsigned long long v = ~raw | low;       // a zero byte among the top k <=> an 0xFF among the top k bytes of raw
        if (((v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull) == 0ull) {
            acc = k == 8 ? raw : ((acc << (8 * k)) | (raw >> (64 - 8 * k)));
            n += 8 * k;
            drop(k);
            return;
        }
        unsigned b = (unsigned)(raw >> 56);
        if (b != 0xFFu) { drop(1); }
        else {
            const unsigned nx = rawn >= 2 ? (unsigned)(raw >> 48) & 255u : (unsigned)(nxt >> 56);
            if (nx == 0) {                              // a stuffed FF: both bytes leave
                if (rawn >= 2) drop(2);
                else { drop(1); drop(1); }
            } else {
                b = 0;                                  // a marker: stays
            }
        }
        acc = (acc << 8) | b;
        n += 8;
    }
    __device__ __forceinline__ void fill32() {          // >= 32 valid bits afterwards (a code + its magnitude bits are <= 27)
        while (n < 32) step();                          // (one or two steps; every step moves at least one byte)
    }
    __device__ __forceinline__ unsigned peek(int k) const { return (unsigned)(acc >> (n - k)) & ((1u << k) - 1u); }
    __device__ __forceinline__ void skip(int k) { n -= k; }
    // restart: the bit buffer holds padding (and zeros fed at the marker); the queue's front must be the RSTn marker
    __device__ __forceinline__ bool restart() {
        acc = 0; n = 0;
        const unsigned b0 = (unsigned)(raw >> 56);
        const unsigned b1 = rawn >= 2 ? (unsigned)(raw >> 48) & 255u : (unsigned)(nxt >> 56);
        if (b0 != 0xFFu || b1 < 0xD0u || b1 > 0xD7u) return false;
        if (rawn >= 2) drop(2);
        else { drop(1); drop(1); }
        return true;
    }
};

// one Huffman symbol (T.81 F.2.2.3 with a 12-bit lookahead, jdhuff.c's scheme); the caller has >= 32 bits; -1: no such code
__device__ __forceinline__ int jp_symbol(JpBits& b, const JpHuff* __restrict__ t) {
    const unsigned e = t->look[b.peek(JP_LOOK)];
    if (e) {
        b.skip((int)(e >> 8));
        return (int)(e & 255u);
    }
    const int code16 = (int)b.peek(16);
    int l = JP_LOOK + 1;
    while (l <= 16 && (code16 >> (16 - l)) > t->maxcode[l]) ++l;
    if (l > 16) return -1;
    b.skip(l);
    return t->vals[((code16 >> (16 - l)) + t->valoff[l]) & 255];
}
__device__ __forceinline__ int jp_receive_extend(JpBits& b, int s) {   // s in 1..15, the bits are there
    const int v = (int)b.peek(s);
    b.skip(s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

// Workgroups of JP_HT threads = 16 waves, four per SIMD: a lane's decode is a chain of dependent steps, the waves of a SIMD hide
// each other's latency, and a batch of 8 192 tiles occupies EIGHT compute units instead of one wave on each of 128 — the
// embedder's conv kernels of the previous batch (one 512-register workgroup per CU) keep the other 248.  The batch's Huffman
// tables (a tiler writes the same four into every tile) are staged in LDS when there are at most JP_LDS_HT of them; the lookups
// go through generic pointers either way.
constexpr int JP_HT = 1024;
constexpr int JP_LDS_HT = 4;
__global__ __launch_bounds__(JP_HT) void k_jpeg_huffman(const uint8_t* __restrict__ data, const uint8_t* __restrict__ plan, int n,
                                                        int W, int H, int16_t* __restrict__ coef, int32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) unsigned s_tab[JP_LDS_HT * sizeof(JpHuff) / 4];
    const int n_ht = reinterpret_cast<const JpHeader*>(plan)->n_ht;
    const bool in_lds = n_ht <= JP_LDS_HT;
    if (in_lds) {
        const unsigned* src = reinterpret_cast<const unsigned*>(plan + jp_off_ht(n));
        for (int t = threadIdx.x; t < n_ht * (int)(sizeof(JpHuff) / 4); t += JP_HT) s_tab[t] = src[t];
    }
    __syncthreads();
    const int i = (int)blockIdx.x * JP_HT + threadIdx.x;
    if (i >= n) return;
    const dsmil_jpeg_image* imp = reinterpret_cast<const dsmil_jpeg_image*>(plan + jp_off_images()) + i;
    const int ist = imp->status;
    if (ist != DSMIL_OK) { status[i] = ist; return; }
    const JpHuff* hts = in_lds ? reinterpret_cast<const JpHuff*>(s_tab) : reinterpret_cast<const JpHuff*>(plan + jp_off_ht(n));
    const JpGeom g(W, H);
    int16_t* cimg = coef + (size_t)i * g.coef_elems();
    const int ncomp = imp->ncomp, hs = imp->hsamp, vs = imp->vsamp, ri = imp->restart_interval;
    // (registers, not an indexed copy of the record: a dynamically indexed local array lives in scratch memory)
    const JpHuff* dc0 = hts + imp->dc[0];
    const JpHuff* ac0 = hts + imp->ac[0];
    const JpHuff* dc1 = hts + imp->dc[1];
    const JpHuff* ac1 = hts + imp->ac[1];
    const JpHuff* dc2 = hts + imp->dc[2];
    const JpHuff* ac2 = hts + imp->ac[2];
    const int mx = (W + 8 * hs - 1) / (8 * hs), my = (H + 8 * vs - 1) / (8 * vs);
    JpBits b;
    b.init(data + imp->ecs_begin);
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int st = DSMIL_OK;
    int n_mcu = 0;
    const int bw = g.Wp >> 3;                          // blocks per row of a full-resolution plane (the row stride of every plane)
    const int nblk_mcu = hs * vs + (ncomp == 3 ? 2 : 0);
    for (int yy = 0; yy < my && st == DSMIL_OK; ++yy) {
        for (int xx = 0; xx < mx && st == DSMIL_OK; ++xx) {
            if (ri && n_mcu && (n_mcu % ri) == 0) {
                if (!b.restart()) { st = DSMIL_E_INVALID; break; }
                pred0 = pred1 = pred2 = 0;
            }
            ++n_mcu;
            for (int bi = 0; bi < nblk_mcu && st == DSMIL_OK; ++bi) {
                // block bi of the MCU: the hs x vs luma blocks row by row, then Cb, then Cr
                const int c = bi < hs * vs ? 0 : bi - hs * vs + 1;
                const int v = c ? 0 : bi / hs, h = c ? 0 : bi - v * hs;
                const int cv = c ? 1 : vs, ch = c ? 1 : hs;
                const JpHuff* dct = c == 0 ? dc0 : (c == 1 ? dc1 : dc2);
                const JpHuff* act = c == 0 ? ac0 : (c == 1 ? ac1 : ac2);
                int16_t* blk = cimg + ((size_t)c * g.blocks_plane + (size_t)(yy * cv + v) * bw + (xx * ch + h)) * 64;
                b.fill32();
                int s = jp_symbol(b, dct);
                if (s < 0 || s > 15) { st = DSMIL_E_INVALID; break; }
                int diff = 0;
                if (s) diff = jp_receive_extend(b, s);
                int pr = (c == 0 ? pred0 : (c == 1 ? pred1 : pred2)) + diff;
                if (c == 0) pred0 = pr; else if (c == 1) pred1 = pr; else pred2 = pr;
                if (pr) blk[0] = (int16_t)pr;
                int k = 1;
                while (k < 64) {
                    b.fill32();
                    const int rs = jp_symbol(b, act);
                    if (rs < 0) { st = DSMIL_E_INVALID; break; }
                    const int r = rs >> 4;
                    s = rs & 15;
                    if (s == 0) {
                        if (r != 15) break;            // EOB
                        k += 16;
                        continue;
                    }
                    k += r;
                    if (k > 63) { st = DSMIL_E_INVALID; break; }
                    blk[k] = (int16_t)jp_receive_extend(b, s);      // ZIGZAG position k
                    ++k;
                }
            }
        }
    }
    status[i] = st;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_jpeg_idct — jidctint.c jpeg_idct_islow
// ---------------------------------------------------------------------------------------------------------------------
#define JP_F0298 2446
#define JP_F0390 3196
#define JP_F0541 4433
#define JP_F0765 6270
#define JP_F0899 7373
#define JP_F1175 9633
#define JP_F1501 12299
#define JP_F1847 15137
#define JP_F1961 16069
#define JP_F2053 16819
#define JP_F2562 20995
#define JP_F3072 25172

// natural position -> zigzag position (the coefficient buffer is in zigzag order)
__device__ constexpr int JP_UNZIG[64] = {
    0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// one 8-point pass over x[0..7 * stride] -> o[0..7]; descale by `shift` bits (round to nearest, arithmetic shift)
template <int SHIFT>
__device__ __forceinline__ void jp_idct8(const int (&x)[8], int (&o)[8]) {
    // (32-bit arithmetic suffices: |dequantised coefficient| <= 2^15 * 255 would not, but baseline coefficients are 11-bit
    // values times an 8-bit quantiser < 2^19, times constants < 2^15, summed over <= 4 terms per stage — libjpeg's JLONG is
    // 64 bit on this platform, so the same products are formed in 64 bit here: no assumption at all)
    long long z2 = x[2], z3 = x[6];
    long long z1 = (z2 + z3) * JP_F0541;
    const long long tmp2 = z1 + z3 * (-JP_F1847);
    const long long tmp3 = z1 + z2 * JP_F0765;
    z2 = x[0]; z3 = x[4];
    const long long tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
    const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    long long t0 = x[7], t1 = x[5], t2 = x[3], t3 = x[1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
    long long z4 = t1 + t3;
    const long long z5 = (z3 + z4) * JP_F1175;
    t0 *= JP_F0298; t1 *= JP_F2053; t2 *= JP_F3072; t3 *= JP_F1501;
    z1 *= -JP_F0899; z2 *= -JP_F2562; z3 = z3 * (-JP_F1961) + z5; z4 = z4 * (-JP_F0390) + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    const long long rnd = 1LL << (SHIFT - 1);
    o[0] = (int)((tmp10 + t3 + rnd) >> SHIFT); o[7] = (int)((tmp10 - t3 + rnd) >> SHIFT);
    o[1] = (int)((tmp11 + t2 + rnd) >> SHIFT); o[6] = (int)((tmp11 - t2 + rnd) >> SHIFT);
    o[2] = (int)((tmp12 + t1 + rnd) >> SHIFT); o[5] = (int)((tmp12 - t1 + rnd) >> SHIFT);
    o[3] = (int)((tmp13 + t0 + rnd) >> SHIFT); o[4] = (int)((tmp13 - t0 + rnd) >> SHIFT);
}

// grid (blocks of 256 threads over 3 * blocks_plane, n): thread = one 8x8 block of one plane of one image
__global__ __launch_bounds__(256) void k_jpeg_idct(const uint8_t* __restrict__ plan, int n, int W, int H,
                                                   const int16_t* __restrict__ coef, uint8_t* __restrict__ planes,
                                                   const int32_t* __restrict__ status) {
    const int i = blockIdx.y;
    const JpGeom g(W, H);
    const int t = (int)blockIdx.x * 256 + threadIdx.x;
    if (t >= 3 * g.blocks_plane || status[i] != DSMIL_OK) return;
    const dsmil_jpeg_image* im = reinterpret_cast<const dsmil_jpeg_image*>(plan + jp_off_images()) + i;
    const int c = t / g.blocks_plane, bi = t - c * g.blocks_plane;
    if (c >= im->ncomp) return;
    const int bw = g.Wp >> 3;
    const int by = bi / bw, bx = bi - by * bw;
    // blocks this component really has (its padded extent in whole MCUs)
    const int hs = im->hsamp, vs = im->vsamp;
    const int mx = (W + 8 * hs - 1) / (8 * hs), my = (H + 8 * vs - 1) / (8 * vs);
    const int cw = c ? mx : mx * hs, chh = c ? my : my * vs;
    if (bx >= cw || by >= chh) return;
    const uint16_t* q = reinterpret_cast<const uint16_t*>(plan + jp_off_qt(n)) + (size_t)im->qt[c] * 64;
    const int16_t* src = coef + (size_t)i * g.coef_elems() + ((size_t)c * g.blocks_plane + bi) * 64;
    int ws[8][8];
    // pass 1: columns
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int x[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = (int)src[JP_UNZIG[r * 8 + col]] * (int)q[r * 8 + col];   // (constant indices: the loops are unrolled)
        jp_idct8<13 - 2>(x, o);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r][col] = o[r];
    }
    // pass 2: rows, level shift, range limit
    uint8_t* dst = planes + ((size_t)i * 3 + c) * g.plane_bytes() + (size_t)(by * 8) * g.Wp + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int o[8];
        jp_idct8<13 + 2 + 3>(ws[r], o);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int a = o[k] + 128, b2 = o[k + 4] + 128;
            a = a < 0 ? 0 : (a > 255 ? 255 : a);
            b2 = b2 < 0 ? 0 : (b2 > 255 ? 255 : b2);
            lo |= (unsigned)a << (8 * k);
            hi |= (unsigned)b2 << (8 * k);
        }
        *reinterpret_cast<uint2*>(dst + (size_t)r * g.Wp) = make_uint2(lo, hi);   // (8-byte aligned: Wp and bx * 8 are multiples of 8)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_jpeg_color — jdsample.c fancy upsampling + jdcolor.c ycc_rgb_convert
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int jp_chroma(const uint8_t* __restrict__ P, int Wp, int x, int y, int hs, int vs, int W2, int H2) {
    if (hs == 1 && vs == 1) return P[(size_t)y * Wp + x];
    const int i = x >> 1;
    if (vs == 1) {                                      // h2v1
        const uint8_t* row = P + (size_t)y * Wp;
        if (W2 <= 2) return row[i];
        const int v = 3 * row[i];
        if (x & 1) return i == W2 - 1 ? row[i] : (v + row[i + 1] + 2) >> 2;
        return i == 0 ? row[i] : (v + row[i - 1] + 1) >> 2;
    }
    const int r = y >> 1;                               // h2v2
    if (W2 <= 2) return P[(size_t)r * Wp + i];
    int nb = (y & 1) ? r + 1 : r - 1;                   // the nearer neighbour row, replicated at the image edges (jdmainct.c)
    nb = nb < 0 ? 0 : (nb > H2 - 1 ? H2 - 1 : nb);
    const uint8_t* r0 = P + (size_t)r * Wp;
    const uint8_t* r1 = P + (size_t)nb * Wp;
    const int cs = 3 * r0[i] + r1[i];
    if (x & 1) return i == W2 - 1 ? (cs * 4 + 7) >> 4 : (cs * 3 + 3 * r0[i + 1] + r1[i + 1] + 7) >> 4;
    return i == 0 ? (cs * 4 + 8) >> 4 : (cs * 3 + 3 * r0[i - 1] + r1[i - 1] + 8) >> 4;
}

__global__ __launch_bounds__(256) void k_jpeg_color(const uint8_t* __restrict__ plan, int n, int W, int H,
                                                    const uint8_t* __restrict__ planes, uint8_t* __restrict__ out,
                                                    const int32_t* __restrict__ status) {
    const int i = blockIdx.y;
    if (status[i] != DSMIL_OK) return;
    const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
    if (px >= (long long)W * H) return;
    const int y = (int)(px / W), x = (int)(px - (long long)y * W);
    const dsmil_jpeg_image* im = reinterpret_cast<const dsmil_jpeg_image*>(plan + jp_off_images()) + i;
    const JpGeom g(W, H);
    const uint8_t* P = planes + (size_t)i * 3 * g.plane_bytes();
    const int Y = P[(size_t)y * g.Wp + x];
    uint8_t* o = out + ((size_t)i * H * W + (size_t)px) * 3;
    if (im->ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; return; }
    const int hs = im->hsamp, vs = im->vsamp;
    const int W2 = (W + hs - 1) / hs, H2 = (H + vs - 1) / vs;      // downsampled_width / height of the chroma components
    const int cb = jp_chroma(P + g.plane_bytes(), g.Wp, x, y, hs, vs, W2, H2) - 128;
    const int cr = jp_chroma(P + 2 * g.plane_bytes(), g.Wp, x, y, hs, vs, W2, H2) - 128;
    int r = Y + ((91881 * cr + 32768) >> 16);
    int gg = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    int b = Y + ((116130 * cb + 32768) >> 16);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    gg = gg < 0 ? 0 : (gg > 255 ? 255 : gg);
    b = b < 0 ? 0 : (b > 255 ? 255 : b);
    o[0] = (uint8_t)r; o[1] = (uint8_t)gg; o[2] = (uint8_t)b;
}

// ---------------------------------------------------------------------------------------------------------------------
// host: marker parser
// ---------------------------------------------------------------------------------------------------------------------
struct HtRaw { uint8_t counts[16]; uint8_t vals[256]; int nvals; };

bool build_huff(const HtRaw& r, JpHuff* t) {
    memset(t, 0, sizeof(*t));
    memcpy(t->vals, r.vals, 256);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        t->valoff[l] = k - code;
        const int cnt = r.counts[l - 1];
        if (cnt) {
            if (code + cnt > (1 << l)) return false;
            if (l <= JP_LOOK)
                for (int j = 0; j < cnt; ++j) {
                    const int c0 = (code + j) << (JP_LOOK - l);
                    for (int f = 0; f < (1 << (JP_LOOK - l)); ++f) t->look[c0 + f] = (uint16_t)((l << 8) | r.vals[k + j]);
                }
            t->maxcode[l] = code + cnt - 1;
        } else {
            t->maxcode[l] = -1;
        }
        code = (code + cnt) << 1;
        k += cnt;
    }
    t->maxcode[17] = 0x7fffffff;
    t->maxcode[0] = -1;
    return k <= 256;
}

}  // namespace

extern "C" {

size_t dsmil_jpeg_plan_bytes(int32_t n) { return n > 0 ? jp_plan_bytes(n) : 0; }

size_t dsmil_jpeg_workspace_bytes(int32_t n, int32_t height, int32_t width) {
    if (n <= 0 || height <= 0 || width <= 0 || height > 65535 || width > 65535) return 0;
    const JpGeom g(width, height);
    return (size_t)n * (g.coef_elems() * sizeof(int16_t) + 3 * g.plane_bytes()) + 256;
}

int dsmil_jpeg_parse(const uint8_t* data, const int64_t* offsets, int32_t n, void* plan) {
    if (!data || !offsets || !plan || n <= 0) return DSMIL_E_INVALID;
    uint8_t* pl = (uint8_t*)plan;
    memset(pl, 0, jp_plan_bytes(n));
    JpHeader* hd = (JpHeader*)pl;
    dsmil_jpeg_image* imgs = (dsmil_jpeg_image*)(pl + jp_off_images());
    uint16_t* qts = (uint16_t*)(pl + jp_off_qt(n));
    JpHuff* hts = (JpHuff*)(pl + jp_off_ht(n));
    static thread_local HtRaw ht_seen[JP_MAX_HT];
    hd->n_images = n;
    int n_qt = 0, n_ht = 0;
    for (int i = 0; i < n; ++i) {
        dsmil_jpeg_image& im = imgs[i];
        im.status = DSMIL_E_UNSUPPORTED;
        const uint8_t* b = data + offsets[i];
        const int64_t len = offsets[i + 1] - offsets[i];
        if (len < 4 || b[0] != 0xFF || b[1] != 0xD8) { im.status = DSMIL_E_INVALID; continue; }
        uint16_t qt_img[4][64];
        bool qt_have[4] = {false, false, false, false};
        HtRaw ht_img[2][4];
        bool ht_have[2][4] = {{false, false, false, false}, {false, false, false, false}};
        int cid[3] = {0, 0, 0}, ch[3] = {1, 1, 1}, cv[3] = {1, 1, 1}, ctq[3] = {0, 0, 0}, ctd[3] = {0, 0, 0}, cta[3] = {0, 0, 0};
        int ncomp = 0, W = 0, H = 0, ri = 0, adobe = -1;
        bool sof = false, ok = false, bad = false;
        int64_t pos = 2;
        while (!bad) {
            while (pos < len && b[pos] != 0xFF) ++pos;
            while (pos < len && b[pos] == 0xFF) ++pos;
            if (pos >= len) { bad = true; break; }
            const int m = b[pos++];
            if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
            if (m == 0xD9 || pos + 2 > len) { bad = true; break; }
            const int64_t L = ((int64_t)b[pos] << 8) | b[pos + 1];
            if (L < 2 || pos + L > len) { bad = true; break; }
            const uint8_t* seg = b + pos + 2;
            const int64_t sl = L - 2;
            if (m == 0xDB) {
                int64_t j = 0;
                while (j < sl) {
                    const int pq = seg[j] >> 4, tq = seg[j] & 15;
                    ++j;
                    if (pq || tq > 3 || j + 64 > sl) { bad = true; break; }
                    for (int k = 0; k < 64; ++k) qt_img[tq][JP_ZIGZAG_H[k]] = seg[j + k];
                    qt_have[tq] = true;
                    j += 64;
                }
            } else if (m == 0xC0) {
                if (sl < 6 || seg[0] != 8) { bad = true; break; }
                H = (seg[1] << 8) | seg[2]; W = (seg[3] << 8) | seg[4];
                ncomp = seg[5];
                if ((ncomp != 1 && ncomp != 3) || sl < 6 + 3 * ncomp) { bad = true; break; }
                for (int c = 0; c < ncomp; ++c) { cid[c] = seg[6 + 3 * c]; ch[c] = seg[7 + 3 * c] >> 4; cv[c] = seg[7 + 3 * c] & 15; ctq[c] = seg[8 + 3 * c]; }
                sof = true;
            } else if (m >= 0xC1 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                bad = true;                                       // not baseline Huffman
            } else if (m == 0xC4) {
                int64_t j = 0;
                while (j < sl) {
                    const int tc = seg[j] >> 4, th = seg[j] & 15;
                    if (tc > 1 || th > 3 || j + 17 > sl) { bad = true; break; }
                    HtRaw& r = ht_img[tc][th];
                    memset(&r, 0, sizeof(r));
                    int ns = 0;
                    for (int k = 0; k < 16; ++k) { r.counts[k] = seg[j + 1 + k]; ns += r.counts[k]; }
                    if (ns > 256 || j + 17 + ns > sl) { bad = true; break; }
                    memcpy(r.vals, seg + j + 17, ns);
                    r.nvals = ns;
                    ht_have[tc][th] = true;
                    j += 17 + ns;
                }
            } else if (m == 0xDD) {
                if (sl < 2) { bad = true; break; }
                ri = (seg[0] << 8) | seg[1];
            } else if (m == 0xEE && sl >= 12 && !memcmp(seg, "Adobe", 5)) {
                adobe = seg[11];
            } else if (m == 0xDA) {
                if (!sof || sl < 1 || seg[0] != ncomp || sl < 4 + 2 * ncomp) { bad = true; break; }
                for (int c = 0; c < ncomp; ++c) {
                    if (seg[1 + 2 * c] != cid[c]) bad = true;
                    ctd[c] = seg[2 + 2 * c] >> 4; cta[c] = seg[2 + 2 * c] & 15;
                }
                if (seg[1 + 2 * ncomp] != 0 || seg[2 + 2 * ncomp] != 63 || seg[3 + 2 * ncomp] != 0) bad = true;
                pos += L;
                ok = !bad;
                break;
            }
            pos += L;
        }
        if (!ok) continue;
        if (W <= 0 || H <= 0) continue;
        if (ncomp == 3 && adobe == 0) continue;                   // Adobe RGB: no colour transform
        if (ncomp == 3 && (ch[1] != 1 || cv[1] != 1 || ch[2] != 1 || cv[2] != 1)) continue;
        if (ncomp == 1) { ch[0] = 1; cv[0] = 1; }                 // a lone component is never subsampled (T.81 A.2.2)
        if (!((ch[0] == 1 && cv[0] == 1) || (ch[0] == 2 && cv[0] == 1) || (ch[0] == 2 && cv[0] == 2))) continue;
        bool miss = false;
        for (int c = 0; c < ncomp; ++c)
            if (ctq[c] > 3 || ctd[c] > 3 || cta[c] > 3 || !qt_have[ctq[c]] || !ht_have[0][ctd[c]] || !ht_have[1][cta[c]]) miss = true;
        if (miss) continue;
        // de-duplicate the tables over the batch
        bool full = false;
        for (int c = 0; c < ncomp && !full; ++c) {
            int f = -1;
            for (int k = 0; k < n_qt; ++k) if (!memcmp(qts + (size_t)k * 64, qt_img[ctq[c]], 128)) { f = k; break; }
            if (f < 0) {
                if (n_qt == JP_MAX_QT) { full = true; break; }
                memcpy(qts + (size_t)n_qt * 64, qt_img[ctq[c]], 128);
                f = n_qt++;
            }
            im.qt[c] = f;
            for (int tc = 0; tc < 2 && !full; ++tc) {
                const HtRaw& r = ht_img[tc][tc ? cta[c] : ctd[c]];
                int g = -1;
                for (int k = 0; k < n_ht; ++k) if (!memcmp(&ht_seen[k], &r, sizeof(HtRaw))) { g = k; break; }
                if (g < 0) {
                    if (n_ht == JP_MAX_HT) { full = true; break; }
                    if (!build_huff(r, hts + n_ht)) { full = true; break; }
                    ht_seen[n_ht] = r;
                    g = n_ht++;
                }
                (tc ? im.ac : im.dc)[c] = g;
            }
        }
        if (full) continue;
        im.ecs_begin = offsets[i] + pos;
        im.ecs_end = offsets[i + 1];
        im.width = W; im.height = H; im.ncomp = ncomp; im.hsamp = ch[0]; im.vsamp = cv[0];
        im.restart_interval = ri;
        im.status = DSMIL_OK;
    }
    hd->n_qt = n_qt;
    hd->n_ht = n_ht;
    return DSMIL_OK;
}

int dsmil_jpeg_decode(const uint8_t* data, const void* plan, int32_t n, int32_t height, int32_t width, uint8_t* out_nhwc,
                      int32_t* status, void* ws, size_t ws_bytes, void* stream) {
    if (!data || !plan || !out_nhwc || !status || !ws || n <= 0 || height <= 0 || width <= 0) return DSMIL_E_INVALID;
    if (height > 65535 || width > 65535 || n > 65535) return DSMIL_E_UNSUPPORTED;
    if (((uintptr_t)ws % 256) || ((uintptr_t)plan % 16)) return DSMIL_E_ALIGN;
    if (ws_bytes < dsmil_jpeg_workspace_bytes(n, height, width)) return DSMIL_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const JpGeom g(width, height);
    int16_t* coef = (int16_t*)ws;
    uint8_t* planes = (uint8_t*)ws + (size_t)n * g.coef_elems() * sizeof(int16_t);
    if (hipMemsetAsync(coef, 0, (size_t)n * g.coef_elems() * sizeof(int16_t), st) != hipSuccess) return DSMIL_E_LAUNCH;
    hipLaunchKernelGGL(k_jpeg_huffman, dim3((unsigned)((n + JP_HT - 1) / JP_HT)), dim3(JP_HT), 0, st, data, (const uint8_t*)plan, n, width, height, coef, status);
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    hipLaunchKernelGGL(k_jpeg_idct, dim3((unsigned)((3 * g.blocks_plane + 255) / 256), (unsigned)n), dim3(256), 0, st,
                       (const uint8_t*)plan, n, width, height, coef, planes, status);
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    const long long npx = (long long)width * height;
    hipLaunchKernelGGL(k_jpeg_color, dim3((unsigned)((npx + 255) / 256), (unsigned)n), dim3(256), 0, st, (const uint8_t*)plan, n, width,
                       height, planes, out_nhwc, status);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

}  // extern "C"
