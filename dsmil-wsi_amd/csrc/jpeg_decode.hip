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

constexpr int JP_MAX_QT = DSMIL_JPEG_MAX_QTABLES;  // distinct quantisation / Huffman tables per batch
constexpr int JP_MAX_HT = DSMIL_JPEG_MAX_HTABLES;
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
// k_jpeg_unstuff — the entropy-coded segment without its byte stuffing and restart markers, as big-endian 32-bit words
// ---------------------------------------------------------------------------------------------------------------------
// T.81 B.1.1.5: inside the entropy-coded segment an 0xFF data byte is followed by a stuffed 0x00; RSTn markers (FF D0..D7) stand
// between restart intervals, each interval padded to a byte boundary; any other marker (EOI) ends the segment.  Resolving that
// inside the decoding lanes put a data-dependent branch into every refill.  One 256-thread workgroup per image copies its
// segment once — 0x00 behind 0xFF dropped, RSTn markers dropped (the decoder byte-aligns at a restart: the marker's position
// needs no record), everything from the first other marker on replaced by zeros (libjpeg feeds zeros behind the data too) —
// into a buffer parallel to the batch's file bytes (unstuffing only shrinks: image i's words start at its segment's offset
// rounded up to 4).  Byte j of the result is stored at j ^ 3, so that an aligned 32-bit load yields the stream's next 32 bits,
// first bit in the MSB.  16 zero bytes follow the data (the reader prefetches two words).
constexpr int JP_UT = 256;      // threads
constexpr int JP_UB = 8;        // bytes per thread per round
__global__ __launch_bounds__(JP_UT) void k_jpeg_unstuff(const uint8_t* __restrict__ data, const uint8_t* __restrict__ plan,
                                                        uint8_t* __restrict__ ust) {
    const int i = blockIdx.x;
    const dsmil_jpeg_image* im = reinterpret_cast<const dsmil_jpeg_image*>(plan + jp_off_images()) + i;
    if (im->status != DSMIL_OK) return;
    const long long beg = im->ecs_begin, len = im->ecs_end - im->ecs_begin;
    const uint8_t* src = data + beg;
    uint8_t* dst = ust + ((beg + 3) & ~3LL);
    __shared__ int s_cnt[JP_UT / 64];
    __shared__ int s_stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long written = 0;
    for (long long base = 0; base < len; base += JP_UT * JP_UB) {
        if (tid == 0) s_stop = 0x7fffffff;
        __syncthreads();
        const long long o = base + (long long)tid * JP_UB;
        unsigned char b[JP_UB + 2];                       // b[0] = the byte in front, b[JP_UB + 1] = the byte behind
#pragma unroll
        for (int k = 0; k < JP_UB + 2; ++k) {
            const long long q = o + k - 1;
            b[k] = (q >= 0 && q < len) ? src[q] : (unsigned char)0;
        }
        unsigned keep = 0;
        int stop = 0x7fffffff;
#pragma unroll
        for (int k = 1; k <= JP_UB; ++k) {
            const long long q = o + k - 1;
            if (q >= len) break;
            const unsigned prev = b[k - 1], cur = b[k], nxt = b[k + 1];
            bool kp = true;
            if (prev == 0xFF && cur == 0x00) kp = false;                                   // stuffed zero
            else if (prev == 0xFF && cur >= 0xD0 && cur <= 0xD7) kp = false;               // second byte of RSTn
            else if (cur == 0xFF && nxt >= 0xD0 && nxt <= 0xD7) kp = false;                // first byte of RSTn
            else if (cur == 0xFF && nxt == 0xFF) kp = false;                               // a fill byte in front of a marker (T.81 B.1.1.2)
            else if (cur == 0xFF && nxt != 0x00) { kp = false; if (stop == 0x7fffffff) stop = (int)(q - base); }   // another marker: the end
            if (kp) keep |= 1u << (k - 1);
        }
        if (stop != 0x7fffffff) atomicMin(&s_stop, stop);
        __syncthreads();
        const int stop_at = s_stop;                        // chunk-relative position of the first terminating marker
#pragma unroll
        for (int k = 0; k < JP_UB; ++k)
            if ((int)(o - base) + k >= stop_at) keep &= ~(1u << k);
        int cnt = __builtin_popcount(keep);
        int incl = cnt;                                   // inclusive scan over the workgroup
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
        if (lane == 63) s_cnt[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < JP_UT / 64; ++w) {
            if (w < wave) before += s_cnt[w];
            total += s_cnt[w];
        }
        long long pos = written + before + incl - cnt;
#pragma unroll
        for (int k = 0; k < JP_UB; ++k)
            if (keep & (1u << k)) { dst[pos ^ 3] = b[k + 1]; ++pos; }
        written += total;
        __syncthreads();
        if (stop_at != 0x7fffffff) break;
    }
    // zeros behind the data: the rest of the last word and 16 more bytes
    const long long zend = ((written + 3) & ~3LL) + 16;
    for (long long q = written + tid; q < zend; q += JP_UT) dst[q ^ 3] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_jpeg_huffman
// ---------------------------------------------------------------------------------------------------------------------
// SIMT shapes this kernel: the 64 lanes of a wave decode 64 different streams, and a wave executes the UNION of its lanes'
// control paths.  The first form (byte-wise refill loop with the stuffing rules, 9-bit lookahead + canonical bit-by-bit slow path,
// zigzag lookup per coefficient) ran every rare path in nearly every iteration because SOME lane needed it: ~4 000 cycles per
// symbol.  Now
//   * the stream is pre-unstuffed (k_jpeg_unstuff): the reader is three 32-bit words (two current, one prefetched — the load of
//     the word after next is issued when a word is used up and not waited for until the next one is) and a bit position; a
//     symbol's code AND its magnitude bits (<= 27 bits) come out of ONE 32-bit window, one advance per symbol, no branch;
//   * ONE table lookup per symbol for codes of up to JP_LOOK = 12 bits, from LDS; longer codes (rare symbols of the standard
//     tables, < 0.1 % of the stream) take the canonical compare chain;
//   * coefficients are stored in ZIGZAG order (k_jpeg_idct reads them through constant indices): no map lookup per symbol.
struct JpBits {
    const unsigned* wp;     // the word behind w2
    unsigned w0, w1, w2;    // current word, next word, the one after (prefetched)
    int bp;                 // bits of w0 already consumed, 0..31
    __device__ __forceinline__ void init(const unsigned* q) { w0 = q[0]; w1 = q[1]; w2 = q[2]; wp = q + 3; bp = 0; }
    // the next 32 bits of the stream
    __device__ __forceinline__ unsigned window() const { return bp ? __builtin_amdgcn_alignbit(w0, w1, 32 - bp) : w0; }
    __device__ __forceinline__ void skip(int k) {   // k <= 32
        bp += k;
        if (bp >= 32) { bp -= 32; w0 = w1; w1 = w2; w2 = *wp++; }
    }
    __device__ __forceinline__ void restart() { skip((8 - (bp & 7)) & 7); }   // the next restart interval starts at a byte boundary
};

// one Huffman symbol + its magnitude (T.81 F.2.2.1 / F.2.2.3 with a 12-bit lookahead, jdhuff.c's scheme).  Returns the symbol
// (-1: no such code) and, for its low nibble s > 0, the EXTENDed value of the s bits behind the code in `val`.
template <class TP>   // TP: pointer to the table — LDS address space (ds_read) or generic
__device__ __forceinline__ int jp_symbol(JpBits& b, TP t, int& val) {
    const unsigned win = b.window();
    const unsigned e = t->look[win >> (32 - JP_LOOK)];
    int len, sym;
    if (e) {
        len = (int)(e >> 8);
        sym = (int)(e & 255u);
    } else {
        const int code16 = (int)(win >> 16);
        len = JP_LOOK + 1;
        while (len <= 16 && (code16 >> (16 - len)) > t->maxcode[len]) ++len;
        if (len > 16) return -1;
        sym = t->vals[((code16 >> (16 - len)) + t->valoff[len]) & 255];
    }
    const int s = sym & 15;
    val = 0;
    if (s) {
        const int v = (int)((win << len) >> (32 - s));     // (len + s <= 31: inside the window)
        val = v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    b.skip(len + s);
    return sym;
}

// Workgroups of JP_HT threads = 16 waves, four per SIMD: a lane's decode is a chain of dependent steps, the waves of a SIMD hide
// each other's latency, and a batch of 8 192 tiles occupies EIGHT compute units instead of one wave on each of 128 — the
// embedder's conv kernels of the previous batch (one 512-register workgroup per CU) keep the other 248.  The batch's Huffman
// tables (a tiler writes the same four into every tile) are staged in LDS when there are at most JP_LDS_HT of them; the lookups
// go through generic pointers either way.
// Batches of up to 4 096 tiles take 256-lane workgroups instead (one wave per SIMD on up to 16 compute units): the lane's chain
// then runs at its own latency, 11 ms per 224 x 224 tile instead of 17-25 ms with four waves sharing a SIMD's issue slots.
constexpr int JP_LDS_HT = 4;
typedef const __attribute__((address_space(3))) JpHuff* JpHuffLds;

// one lane = one tile's stream; TP = how the Huffman tables are addressed (LDS: ds_read; generic: global memory)
template <class TP>
__device__ __forceinline__ int jp_decode_lane(TP hts, const dsmil_jpeg_image* imp, const uint8_t* __restrict__ ust, int W, int H,
                                              int16_t* __restrict__ cimg) {
    const JpGeom g(W, H);
    const int ncomp = imp->ncomp, hs = imp->hsamp, vs = imp->vsamp, ri = imp->restart_interval;
    // (registers, not an indexed copy of the record: a dynamically indexed local array lives in scratch memory)
    TP dc0 = hts + imp->dc[0];
    TP ac0 = hts + imp->ac[0];
    TP dc1 = hts + imp->dc[1];
    TP ac1 = hts + imp->ac[1];
    TP dc2 = hts + imp->dc[2];
    TP ac2 = hts + imp->ac[2];
    const int mx = (W + 8 * hs - 1) / (8 * hs), my = (H + 8 * vs - 1) / (8 * vs);
    JpBits b;
    b.init(reinterpret_cast<const unsigned*>(ust + ((imp->ecs_begin + 3) & ~3LL)));
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int st = DSMIL_OK;
    int n_mcu = 0;
    const int bw = g.Wp >> 3;                          // blocks per row of a full-resolution plane (the row stride of every plane)
    const int nblk_mcu = hs * vs + (ncomp == 3 ? 2 : 0);
    for (int yy = 0; yy < my && st == DSMIL_OK; ++yy) {
        for (int xx = 0; xx < mx && st == DSMIL_OK; ++xx) {
            if (ri && n_mcu && (n_mcu % ri) == 0) {
                b.restart();
                pred0 = pred1 = pred2 = 0;
            }
            ++n_mcu;
            for (int bi = 0; bi < nblk_mcu && st == DSMIL_OK; ++bi) {
                // block bi of the MCU: the hs x vs luma blocks row by row, then Cb, then Cr
                const int c = bi < hs * vs ? 0 : bi - hs * vs + 1;
                const int v = c ? 0 : bi / hs, h = c ? 0 : bi - v * hs;
                const int cv = c ? 1 : vs, ch = c ? 1 : hs;
                TP dct = c == 0 ? dc0 : (c == 1 ? dc1 : dc2);
                TP act = c == 0 ? ac0 : (c == 1 ? ac1 : ac2);
                int16_t* blk = cimg + ((size_t)c * g.blocks_plane + (size_t)(yy * cv + v) * bw + (xx * ch + h)) * 64;
                int diff;
                int s = jp_symbol(b, dct, diff);
                if (s < 0 || s > 15) { st = DSMIL_E_INVALID; break; }
                const int pr = (c == 0 ? pred0 : (c == 1 ? pred1 : pred2)) + diff;
                if (c == 0) pred0 = pr; else if (c == 1) pred1 = pr; else pred2 = pr;
                if (pr) blk[0] = (int16_t)pr;
                int k = 1;
                while (k < 64) {
                    int val;
                    const int rs = jp_symbol(b, act, val);
                    if (rs < 0) { st = DSMIL_E_INVALID; break; }
                    const int r = rs >> 4;
                    if ((rs & 15) == 0) {
                        if (r != 15) break;            // EOB
                        k += 16;
                        continue;
                    }
                    k += r;
                    if (k > 63) { st = DSMIL_E_INVALID; break; }
                    blk[k] = (int16_t)val;             // ZIGZAG position k
                    ++k;
                }
            }
        }
    }
    return st;
}

template <int JP_HT>
__global__ __launch_bounds__(JP_HT) void k_jpeg_huffman(const uint8_t* __restrict__ ust, const uint8_t* __restrict__ plan, int n,
                                                        int W, int H, int16_t* __restrict__ coef, int32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) unsigned s_tab[JP_LDS_HT * sizeof(JpHuff) / 4];
    const int n_ht = reinterpret_cast<const JpHeader*>(plan)->n_ht;
    const bool in_lds = n_ht <= JP_LDS_HT;             // (uniform over the launch)
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
    const JpGeom g(W, H);
    int16_t* cimg = coef + (size_t)i * g.coef_elems();
    // two instantiations of the lane's decode: table lookups as ds_read (the usual case: a tiler's four tables) or from global memory
    if (in_lds) status[i] = jp_decode_lane<JpHuffLds>((JpHuffLds)(s_tab), imp, ust, W, H, cimg);
    else status[i] = jp_decode_lane<const JpHuff*>(reinterpret_cast<const JpHuff*>(plan + jp_off_ht(n)), imp, ust, W, H, cimg);
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

// VEC4: one thread = four horizontally adjacent pixels = twelve output bytes as three aligned dwords (W % 4 == 0); the first form
// stored every byte on its own (2.9 ms per 8 192 tiles at 0.7 TB/s).  Otherwise one pixel per thread.
__device__ __forceinline__ unsigned jp_rgb(int Y, int cb, int cr) {
    int r = Y + ((91881 * cr + 32768) >> 16);
    int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    int b = Y + ((116130 * cb + 32768) >> 16);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    b = b < 0 ? 0 : (b > 255 ? 255 : b);
    return (unsigned)r | ((unsigned)g << 8) | ((unsigned)b << 16);
}

template <bool VEC4>
__global__ __launch_bounds__(256) void k_jpeg_color(const uint8_t* __restrict__ plan, int n, int W, int H,
                                                    const uint8_t* __restrict__ planes, uint8_t* __restrict__ out,
                                                    const int32_t* __restrict__ status) {
    const int i = blockIdx.y;
    if (status[i] != DSMIL_OK) return;
    constexpr int PX = VEC4 ? 4 : 1;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long px = t * PX;
    if (px >= (long long)W * H) return;
    const int y = (int)(px / W), x = (int)(px - (long long)y * W);
    const dsmil_jpeg_image* im = reinterpret_cast<const dsmil_jpeg_image*>(plan + jp_off_images()) + i;
    const JpGeom g(W, H);
    const uint8_t* P = planes + (size_t)i * 3 * g.plane_bytes();
    const int hs = im->hsamp, vs = im->vsamp;
    const int W2 = (W + hs - 1) / hs, H2 = (H + vs - 1) / vs;      // downsampled_width / height of the chroma components
    const bool grey = im->ncomp == 1;
    unsigned rgb[PX];
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        const int Y = P[(size_t)y * g.Wp + x + k];
        if (grey) { rgb[k] = (unsigned)Y * 0x010101u; continue; }
        const int cb = jp_chroma(P + g.plane_bytes(), g.Wp, x + k, y, hs, vs, W2, H2) - 128;
        const int cr = jp_chroma(P + 2 * g.plane_bytes(), g.Wp, x + k, y, hs, vs, W2, H2) - 128;
        rgb[k] = jp_rgb(Y, cb, cr);
    }
    uint8_t* o = out + ((size_t)i * H * W + (size_t)px) * 3;
    if constexpr (VEC4) {
        unsigned* o4 = reinterpret_cast<unsigned*>(o);             // (12 px bytes: 4-byte aligned because W % 4 == 0 and `out` is)
        o4[0] = rgb[0] | (rgb[1] << 24);
        o4[1] = (rgb[1] >> 8) | (rgb[2] << 16);
        o4[2] = (rgb[2] >> 16) | (rgb[3] << 8);
    } else {
        o[0] = (uint8_t)rgb[0]; o[1] = (uint8_t)(rgb[0] >> 8); o[2] = (uint8_t)(rgb[0] >> 16);
    }
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

size_t dsmil_jpeg_workspace_bytes(int32_t n, int32_t height, int32_t width, int64_t data_bytes) {
    if (n <= 0 || height <= 0 || width <= 0 || height > 65535 || width > 65535 || data_bytes <= 0) return 0;
    const JpGeom g(width, height);
    // coefficients | component planes | the unstuffed streams (parallel to the file bytes, + slack for alignment and the zeros)
    // | overrun slack: a CORRUPT last stream can make its lane read on past the batch's bytes — at most 4 bytes per coefficient
    // of one image (a symbol is <= 27 bits); the read must stay inside the allocation (what it decodes is garbage either way)
    return (size_t)n * (g.coef_elems() * sizeof(int16_t) + 3 * g.plane_bytes()) + (((size_t)data_bytes + 64 + 255) & ~(size_t)255) +
           g.coef_elems() * 4 + 512;
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
        {   // the file must END with EOI (a few trailing bytes tolerated): a truncated file is Pillow's to judge ("image file is
            // truncated"), not something to decode silently with zeros behind the data
            bool eoi = false;
            for (int64_t q = len - 2; q >= pos && q >= len - 16; --q)
                if (b[q] == 0xFF && b[q + 1] == 0xD9) { eoi = true; break; }
            if (!eoi) { im.status = DSMIL_E_INVALID; continue; }
        }
        if (ncomp == 3 && cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B') continue;   // libjpeg takes such ids for RGB data
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

int dsmil_jpeg_decode(const uint8_t* data, int64_t data_bytes, const void* plan, int32_t n, int32_t height, int32_t width,
                      uint8_t* out_nhwc, int32_t* status, void* ws, size_t ws_bytes, void* stream) {
    if (!data || !plan || !out_nhwc || !status || !ws || n <= 0 || height <= 0 || width <= 0 || data_bytes <= 0) return DSMIL_E_INVALID;
    if (height > 65535 || width > 65535 || n > 65535) return DSMIL_E_UNSUPPORTED;
    if (((uintptr_t)ws % 256) || ((uintptr_t)plan % 16)) return DSMIL_E_ALIGN;
    if (ws_bytes < dsmil_jpeg_workspace_bytes(n, height, width, data_bytes)) return DSMIL_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const JpGeom g(width, height);
    int16_t* coef = (int16_t*)ws;
    uint8_t* planes = (uint8_t*)ws + (size_t)n * g.coef_elems() * sizeof(int16_t);
    uint8_t* ust = planes + (size_t)n * 3 * g.plane_bytes();
    ust += (256 - ((uintptr_t)ust & 255)) & 255;
    if (hipMemsetAsync(coef, 0, (size_t)n * g.coef_elems() * sizeof(int16_t), st) != hipSuccess) return DSMIL_E_LAUNCH;
    hipLaunchKernelGGL(k_jpeg_unstuff, dim3((unsigned)n), dim3(JP_UT), 0, st, data, (const uint8_t*)plan, ust);
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    if (n <= 4096)
        hipLaunchKernelGGL(k_jpeg_huffman<256>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t*)ust, (const uint8_t*)plan, n, width, height, coef, status);
    else
        hipLaunchKernelGGL(k_jpeg_huffman<1024>, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, st, (const uint8_t*)ust, (const uint8_t*)plan, n, width, height, coef, status);
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    hipLaunchKernelGGL(k_jpeg_idct, dim3((unsigned)((3 * g.blocks_plane + 255) / 256), (unsigned)n), dim3(256), 0, st,
                       (const uint8_t*)plan, n, width, height, coef, planes, status);
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    const long long npx = (long long)width * height;
    if (width % 4 == 0 && ((uintptr_t)out_nhwc % 4) == 0)
        hipLaunchKernelGGL(k_jpeg_color<true>, dim3((unsigned)((npx / 4 + 255) / 256), (unsigned)n), dim3(256), 0, st, (const uint8_t*)plan, n,
                           width, height, planes, out_nhwc, status);
    else
        hipLaunchKernelGGL(k_jpeg_color<false>, dim3((unsigned)((npx + 255) / 256), (unsigned)n), dim3(256), 0, st, (const uint8_t*)plan, n,
                           width, height, planes, out_nhwc, status);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

}  // extern "C"
