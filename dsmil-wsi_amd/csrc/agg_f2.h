// agg_f2.h — k_attend_f2: the fp32 query / attend kernel for BATCHES of bags (dsmil.py:46-62 behind the instance logits) with
// every feature byte read ONCE and the query MLP on v_mfma_f32_32x32x16_f16 over TWO-plane fp16 cuts, three products.
//
// Why (round 5).  k_query_attend_split (rounds 2-4) streamed a 128-row tile through LDS for the query MLP on bf16 MFMA over
// exact three-plane cuts (six plane products per fp32 MAC) and then read the whole tile a SECOND time for the value sum
// B = sum_n p[n] x[n,:] (dsmil.py:57): 2.02x its algorithmic bytes at the fabric counters, every round.  With the logits pass
// before it the 64-bag batch moved 4.0 GB per 0.81 ms = 4.9 TB/s — the whole path sat on the HBM roof, not on the matrix
// pipe.  The second read goes away only if the tile stays on chip from the MLP to the value sum, and a 128-row fp32 tile is
// 256 KB.  So:
//   * tile = 64 rows x K <= 512, resident in LDS as the two fp16 planes the MFMA consumes (4 B per element — the size of
//     the fp32 value): cut ONCE by four cutter waves, in MFMA B-fragment order;
//   * the cut: x' = x * 2^e (per ROW, e from the row's max |x| — a by-product of k_logits_stream, so fp16's five exponent
//     bits never see the data's scale), h0 = rne16(x'), h1 = rne16(x' - h0): x' = h0 + h1 (1 + d), |d| <= 2^-24 while h1 is
//     normal, absolute 2^-25 below that (gfx950's f16 MFMA keeps subnormals: tools/probes/f16_denorm.hip).  Weights are cut
//     the same way at pack time (per-tensor power-of-two scale).  Products h0 w0 + h0 w1 + h1 w0 (each exact in fp32: 11 x 11
//     significand bits); the dropped h1 w1 is <= 2^-24 |x w|.  Error class = the six-product bf16 form's (which drops
//     2^-23 |x w|): tools/form_error_study.py measures both through the 20 convs of the embedder (4.5e-7 vs 3.5e-7 max
//     feature error); here tests/test_agg_gpu.py holds it to the same 1e-4 / golden bars.  HALF the MFMAs, 2/3 of the planes;
//   * roles (agg_hs.h's, re-cut): compute wave w owns hidden / query units 32w..32w+31 of all 64 rows (weights of its units
//     straight from L2 into A-operand registers, ring three steps ahead; B fragments from the resident planes); cutter wave
//     4 + j streams rows 16j..16j+15 global -> registers (eight 32-k chunks in flight), scales, cuts, writes planes, one
//     barrier per chunk;
//   * hidden layer: per-row scale again (max over the row's 128 units through LDS), planes exchanged through 16 KiB of LDS
//     in two halves (GEMM-2 steps 0-3 contract the units of waves 0-1, steps 4-7 those of waves 2-3);
//   * value sum on the VALU from the resident planes: v_fma_mix_f32 takes the fp16 halves directly, all eight waves, 16-lane
//     DPP row reductions; the weights carry 1 / 2^e of their row.
// Partials per tile (m, l, B) in slot off0 / 64 + bag + tile, merged by k_finish as for every other attend kernel.
// LDS: planes 128 KiB [k-step 32][plane 2][hi 2][row 64] x 16 B | hidden planes 16 KiB [plane 2][step 4][hi 2][row 64] x 16 B
// | 10.6 KiB scratch (scores, row maxima, per-wave value-sum weights, biases; critical queries, row scales and tile records of
// this and the next tile) = 154.6 KiB: one 512-thread workgroup per CU.
#pragma once
#include "agg_hs.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
union F2Frag {
    unsigned u[4];
    f16x8 v;
    f32x4 f;
    f16x2 h[4];
};

constexpr int F2_BM = 64;                               // rows per tile
constexpr int F2_THREADS = 512;                         // four compute waves + four cutter waves
constexpr int F2_MAXSTEPS = 32;                         // 16-k steps of GEMM 1: K <= 512
constexpr int F2_XP_F4 = F2_MAXSTEPS * 2 * 2 * F2_BM;   // 16-B units of the resident planes (128 KiB)
constexpr int F2_HP_F4 = 2 * 4 * 2 * F2_BM;             // 16-B units of the hidden-plane exchange (16 KiB)
constexpr int F2_SCR = 2720;                            // floats of scratch
constexpr int F2_LDS_BYTES = (F2_XP_F4 + F2_HP_F4) * 16 + F2_SCR * 4;
constexpr int F2_CHUNK_F4 = 4 * 2 * 64;                 // 16-B units per packed weight chunk (one 16-k step): [tile][plane][lane]
constexpr int F2_WRD = 8;                               // weight register ring slots (step s sits in slot s % 8; runs across tiles)
constexpr int F2_WLA = 6;                               // ... requested this many steps ahead
constexpr int F2_STAGGER = 0;                           // start skew per (workgroup mod 8), in s_sleep(1) units of 64 cycles
constexpr int F2_TRAILER_BYTES = 256;                   // behind the chunks: {1 / scale(W1), 1 / scale(W2)}

// The power of two s with m * s in [2^13, 2^14) (fp16's largest finite value is 65504 = 2^16 - 32) and its inverse; 1 for
// m == 0, subnormal or non-finite m (non-finite data stays non-finite through the cut: the outputs are NaN, as the reference's).
__device__ __forceinline__ float f2_scale(float m, float& inv) {
    const int e = (int)((__float_as_uint(m) >> 23) & 0xFFu);
    int se = 267 - e;                          // 2^(13 - (e - 127)), biased
    if (e == 0 || e == 255) se = 127;
    se = se < 2 ? 2 : (se > 252 ? 252 : se);
    inv = __uint_as_float((unsigned)(254 - se) << 23);
    return __uint_as_float((unsigned)se << 23);
}

// cut 8 (scaled) fp32 values into two fp16 planes, round to nearest, packed in k order
__device__ __forceinline__ void split2h(const float (&x)[8], F2Frag (&o)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x2 h = {(_Float16)x[2 * i], (_Float16)x[2 * i + 1]};
        const float r0 = x[2 * i] - (float)h[0], r1 = x[2 * i + 1] - (float)h[1];   // exact
        const f16x2 l = {(_Float16)r0, (_Float16)r1};
        o[0].h[i] = h;
        o[1].h[i] = l;
    }
}

// The same cut of 8 values x[i] * sc with the scale folded in, two instructions per value: v_fma_mixlo/hi_f16 form
// rne16(x sc) and rne16(x sc - h0) directly (x sc is exact: sc is a power of two), writing the fp16 halves in place.  (From the
// C++ above hipcc made a v_mul, half a v_cvt_pk and two v_fma_mix per value; the cutters share their SIMD's issue port with a
// wave of MFMAs, and every instruction there costs ~9 cycles.)
__device__ __forceinline__ void split2h_scaled(const f32x4& x0, const f32x4& x1, float sc, F2Frag (&o)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = i < 2 ? x0[2 * i] : x1[2 * i - 4], b = i < 2 ? x0[2 * i + 1] : x1[2 * i - 3];
        unsigned h, l;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "v"(sc));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "v"(sc));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(sc), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(sc), "v"(h));
        o[0].u[i] = h;
        o[1].u[i] = l;
    }
}

// (wave_sum / wave_max / dpp_row_sum: agg_common.h)
__device__ __forceinline__ float f2_wave_max(float v) { return wave_max(v); }
__device__ __forceinline__ float f2_wave_sum(float v) { return wave_sum(v); }
__device__ __forceinline__ float f2_row_sum(float v) { return dpp_row_sum(v); }

// chunk c >= 1 opens a group of the cutters' barrier schedule (a barrier follows every chunk of the first half of the tile
// and every c % 4 == 3 behind it)
template <int NK1>
__host__ __device__ constexpr bool f2_group_first(int c) { return c <= NK1 / 2 || c % 4 == 0; }

struct F2Tile {
    int bag;
    long long off0, Nb, row0, slot;
};

// first (bag, tile) item at or behind `item` (step `stride`) whose tile lies inside its bag; evaluated identically by every wave
__device__ __forceinline__ bool f2_fetch(const AttendArgs& a, int tiles_per_bag, int n_items, int stride, int& item, F2Tile& t) {
    while (item < n_items) {
        const int b = item / tiles_per_bag, tile = item - b * tiles_per_bag;
        const int bag = a.bag0 + b;
        const long long off0 = a.offsets[bag];
        const long long Nb = a.offsets[bag + 1] - off0;
        const long long row0 = (long long)tile * F2_BM;
        if (row0 < Nb) {
            t.bag = bag; t.off0 = off0; t.Nb = Nb; t.row0 = row0; t.slot = off0 / F2_BM + bag + tile;
            return true;
        }
        item += stride;
    }
    return false;
}

// Everything behind the query MLP, all eight waves: scores (dsmil.py:55-56), the tile's softmax statistics, the value sum
// (dsmil.py:57) from the resident planes.  Compute waves bring their query tile Qw.  sInv = 1 / row scale of the tile's 64 rows,
// sQ = the critical queries of the first class pair [2][128] (both in LDS, put there by the cutters a tile ahead; later pairs,
// C > 2, are read from memory); sPw = this wave's PRIVATE 128 floats.  Two barriers per class pair: the scores meet in sS (T1),
// then EVERY wave forms the 64 scores and their statistics for itself (lane = row; same values in every wave: fixed order) and
// keeps the value-sum weights it needs in its own LDS strip — no second barrier, no wave waiting for wave 0.
template <int ABL = 0>
__device__ __forceinline__ void f2_tail(const AttendArgs& a, const F2Tile& t, const f32x16 (&Qw)[2], const f32x4* sXp, float* sS,
                                        float* sPw, const float* sQ, const float* sInv, int nks) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int Kv = a.Kv;
    const float scale = 0.08838834764831845f;            // 1/sqrt(128), dsmil.py:56
    for (int c0 = 0; c0 < a.C; c0 += 2) {
        const int c1 = (c0 + 1 < a.C) ? c0 + 1 : c0;
        const bool two = c1 != c0;                        // (block-uniform)
        if (wave < 4) {
            // partial dot products of this wave's 32 query units with the critical query (dsmil.py:55)
            const float* qm0 = c0 == 0 ? sQ + 32 * wave : a.qmax + ((long long)t.bag * a.C + c0) * QD + 32 * wave;
            const float* qm1 = c0 == 0 ? sQ + QD + 32 * wave : a.qmax + ((long long)t.bag * a.C + c1) * QD + 32 * wave;
            float s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(qm0 + 8 * q + 4 * hi);
                const f32x4 u1 = *reinterpret_cast<const f32x4*>(qm1 + 8 * q + 4 * hi);
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s0[g] = fmaf(Qw[g][4 * q + e], u0[e], s0[g]);
                        s1[g] = fmaf(Qw[g][4 * q + e], u1[e], s1[g]);
                    }
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                s0[g] += __shfl_xor(s0[g], 32, 64);       // (lanes l31 and l31 + 32 hold the two halves of a row's dot product)
                s1[g] += __shfl_xor(s1[g], 32, 64);
                if (hi == 0) { sS[(wave * 2 + 0) * F2_BM + 32 * g + l31] = s0[g]; sS[(wave * 2 + 1) * F2_BM + 32 * g + l31] = s1[g]; }
            }
        }
        __syncthreads();                                  // T1
        {
            // lane = row of the tile: scores (fixed summation order), softmax statistics, value-sum weights
            const long long myrow = t.row0 + lane;
            const bool valid = myrow < t.Nb;
            const float rinv = sInv[lane];
            const float s0 = ((sS[0 * F2_BM + lane] + sS[2 * F2_BM + lane]) + (sS[4 * F2_BM + lane] + sS[6 * F2_BM + lane])) * scale;
            const float s1 = two ? ((sS[1 * F2_BM + lane] + sS[3 * F2_BM + lane]) + (sS[5 * F2_BM + lane] + sS[7 * F2_BM + lane])) * scale : s0;
            const float m0 = f2_wave_max(valid ? s0 : -INFINITY), m1 = two ? f2_wave_max(valid ? s1 : -INFINITY) : m0;
            const float p0 = valid ? expf(s0 - m0) : 0.f, p1 = two ? (valid ? expf(s1 - m1) : 0.f) : p0;
            if (wave == 0) {
                const float l0 = f2_wave_sum(p0), l1 = two ? f2_wave_sum(p1) : l0;
                if (valid) {
                    float* o = a.scores + (t.off0 + myrow) * (long long)a.C;
                    o[c0] = s0;
                    if (two) o[c1] = s1;
                }
                if (lane == 0) {
                    float* ml = a.part_ml + (t.slot * a.C + c0) * 2;
                    ml[0] = m0; ml[1] = l0;
                    if (two) { ml[2] = m1; ml[3] = l1; }
                }
            }
            sPw[lane] = p0 * rinv;
            if (two) sPw[F2_BM + lane] = p1 * rinv;
        }
        // ---- value sum: Bpart[c][k] = sum_n p[n][c] x[n][k], x = (h0 + h1) / row scale.  A unit = the 8 k of one (k-step, hi)
        //      block; wave v owns units 8v .. 8v+7 in two passes of four; lane (rl = lane & 15, u = lane >> 4) sums rows rl,
        //      rl + 16, rl + 32, rl + 48 of unit 4 pass + u (v_fma_mix_f32 on the fp16 halves), then the 16 row lanes of a DPP
        //      row are reduced
        if constexpr ((ABL & 8) == 0) {                  // (ablation: no value sum)
            const int rl = lane & 15, u = lane >> 4;
            float* pb0 = a.part_B + (t.slot * a.C + c0) * (long long)Kv;
            float* pb1 = a.part_B + (t.slot * a.C + c1) * (long long)Kv;
            float w0[4], w1[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { w0[jj] = sPw[rl + 16 * jj]; w1[jj] = two ? sPw[F2_BM + rl + 16 * jj] : 0.f; }
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int U = 8 * wave + 4 * pass + u, su = U >> 1, hu = U & 1;
                const int sr = su < nks ? su : nks - 1;   // (K < 512: units past K re-read the last step and store nothing)
                const f32x4* p = sXp + (long long)(sr * 4 + hu) * F2_BM + rl;
                float acc0[8], acc1[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
                if (two) {                                // (block-uniform: one class — the C = 1 batch — does half the work)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        F2Frag f0, f1;
                        f0.f = p[16 * jj];
                        f1.f = p[2 * F2_BM + 16 * jj];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            acc0[e] = fmaf((float)f0.v[e], w0[jj], acc0[e]);
                            acc0[e] = fmaf((float)f1.v[e], w0[jj], acc0[e]);
                            acc1[e] = fmaf((float)f0.v[e], w1[jj], acc1[e]);
                            acc1[e] = fmaf((float)f1.v[e], w1[jj], acc1[e]);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) { acc0[e] = f2_row_sum(acc0[e]); acc1[e] = f2_row_sum(acc1[e]); }
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        F2Frag f0, f1;
                        f0.f = p[16 * jj];
                        f1.f = p[2 * F2_BM + 16 * jj];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            acc0[e] = fmaf((float)f0.v[e], w0[jj], acc0[e]);
                            acc0[e] = fmaf((float)f1.v[e], w0[jj], acc0[e]);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc0[e] = f2_row_sum(acc0[e]);
                }
                if (rl == 0 && su < nks) {
                    const int k = 16 * su + 8 * hu;
                    *reinterpret_cast<f32x4*>(pb0 + k) = f32x4{acc0[0], acc0[1], acc0[2], acc0[3]};
                    *reinterpret_cast<f32x4*>(pb0 + k + 4) = f32x4{acc0[4], acc0[5], acc0[6], acc0[7]};
                    if (two) {
                        *reinterpret_cast<f32x4*>(pb1 + k) = f32x4{acc1[0], acc1[1], acc1[2], acc1[3]};
                        *reinterpret_cast<f32x4*>(pb1 + k + 4) = f32x4{acc1[4], acc1[5], acc1[6], acc1[7]};
                    }
                }
            }
        }
        __syncthreads();                                  // T3: scratch and (after the last class pair) the planes are free
    }
}

// NK1 = K / 32 in {4, 8, 12, 16} (K % 128 == 0, K <= 512), feature rows 16-B aligned, vals == feats (v = Identity), a.wpk = the
// image of k_pack_agg_f2, rowmax[logical row] = max_k |x| (k_logits_stream).  PERSISTENT: workgroup g takes items g, g + grid,
// ... of the (bag, tile) list.  The weight ring of the compute waves and (NK1 % 8 == 0) the feature ring of the cutters run
// ACROSS tiles: while a tile is in GEMM 2 and its tail, the first half of the next tile is already on its way into the cutters'
// registers.  Everything inside a tile is straight-line code (NK1 a template parameter): hipcc counts the loads in flight
// exactly there, while at a loop header it waits for ALL of them (first form: one full HBM latency per eight chunks) — the one
// header left is the tile loop's, where the rings have had GEMM 2 and the tail to land.
// Small operands never come from memory on a tile's critical path: the two biases sit in LDS for the whole launch, the critical
// queries and row scales of the NEXT tile are requested at the start of the current one.
// ABL (experiment builds; timing only, wrong results): 1 no weight loads behind the first seven steps, 2 no feature loads
// behind the first ring fill, 4 no MFMAs, 8 no value sum, 16 no cut / plane writes.
template <int NK1, int ABL = 0>
__global__ __launch_bounds__(F2_THREADS, 1) void k_attend_f2(AttendArgs a, const float* __restrict__ rowmax, int tiles_per_bag,
                                                             int n_items) {
    static_assert(NK1 % 4 == 0 && NK1 >= 4 && 2 * NK1 <= F2_MAXSTEPS, "K a multiple of 128 up to 512");
    constexpr int NKS = 2 * NK1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* sXp = reinterpret_cast<f32x4*>(smem);
    f32x4* sHp = sXp + F2_XP_F4;
    float* scr = reinterpret_cast<float*>(sHp + F2_HP_F4);
    float* sS = scr;             // [4 waves][2 classes][64 rows] partial scores
    float* sMax = scr + 512;     // [4 waves][64 rows] hidden-layer row maxima
    float* sBias = scr + 768;    // [2][128]: q.0 / q.2 biases
    float* sPall = scr + 1024;   // [8 waves][2 classes][64 rows]: every wave's private value-sum weights p * (1 / row scale)
    // what the cutters hand to the compute waves a tile ahead, double-buffered by tile parity:
    float* sQall = scr + 2048;   // [2][2 classes][128]: critical queries of the tile's bag (first class pair)
    float* sInvAll = scr + 2560; // [2][64]: 1 / row scale of the tile's rows
    long long* sCtl = reinterpret_cast<long long*>(scr + 2688);   // [2][8]: {have, bag, off0, Nb, row0, slot}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int K = 32 * NK1;
    const int nst = NKS + (a.nonlinear ? 8 : 0);
    const f32x4* wimg = reinterpret_cast<const f32x4*>(a.wpk);
    const float* trailer = reinterpret_cast<const float*>(wimg + (long long)(NKS + 8) * F2_CHUNK_F4);
    const float* feats = reinterpret_cast<const float*>(a.feats);
    const int stride = (int)gridDim.x;
    int item = (int)blockIdx.x;
    F2Tile cur;
    bool have = f2_fetch(a, tiles_per_bag, n_items, stride, item, cur);
    if (!have) return;                                    // (block-uniform)
    // ABL & 32 (with DSMIL_EXPT=64: k_finish skipped): wave 0 and wave 4 keep s_memtime stamps of the phase boundaries and store
    // them into the tile's 64 rows of A (C = 1) at the end of the tile (tools/f2_stamps.py)
    unsigned long long stamps[16];
    auto STAMP = [&](int i) {
        if constexpr ((ABL & 32) != 0) stamps[i] = __builtin_readcyclecounter();
    };
    auto STAMP_OUT = [&](const F2Tile& t) {
        if constexpr ((ABL & 32) != 0) {
            if (lane == 0 && (wave == 0 || wave == 4) && a.C == 1 && t.row0 + F2_BM <= t.Nb) {
                unsigned long long* o = reinterpret_cast<unsigned long long*>(a.scores + (t.off0 + t.row0)) + (wave == 4 ? 16 : 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] = stamps[i];
            }
        }
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) stamps[i] = 0;
    // All workgroups start together and run tiles of the same length: their GEMM-1 phases — where half of a tile's feature loads
    // are issued — coincide, and the chip then asks for more than the HBM rate while the memory pipe idles through everybody's
    // exchange phase and tail.  Workgroup g therefore starts (g mod 8) eighths of a tile late: the phases of the eight groups
    // interleave and the memory pipe sees one even stream (F2_STAGGER x 64 cycles per eighth; 0 = off).
    {
        const int eighth = (int)(blockIdx.x & 7);
#ifdef DSMIL_EXPERIMENTS
        const int units = (a.expt >> 20) & 0xFF ? ((a.expt >> 20) & 0xFF) - 1 : F2_STAGGER;
#else
        const int units = F2_STAGGER;
#endif
        for (int i = 0; i < eighth * units; ++i) __builtin_amdgcn_s_sleep(1);
    }
    // once per launch: the biases -> LDS (the first tile's record, row scales and critical queries follow from the cutters)
    const int c1_first = a.C > 1 ? 1 : 0;
    if (tid < 256) sBias[tid] = tid < QD ? a.q0_b[tid] : (a.nonlinear ? a.q2_b[tid - QD] : 0.f);
    float* sPw = sPall + wave * 128;

    if (wave >= 4) {
        // ================= a CUTTER wave: rows 16j .. 16j+15 of every tile, all K =================
        const int j = wave - 4, rr = lane & 15, o = lane >> 4;   // this lane: row 16j + rr, k-octet o of every 32-k chunk
        const f32x16 noq[2] = {};
        f32x4 ring[NK1][2];                                // the WHOLE next tile in flight / in registers (8 NK1 registers)
        auto row_src = [&](const F2Tile& t, float& sc, float& sinv) -> const float* {
            long long gr = t.row0 + 16 * j + rr;
            if (gr >= t.Nb) gr = t.Nb - 1;                // rows past the bag end are cut like the last row, weight 0
            sc = f2_scale(rowmax[t.off0 + gr], sinv);
            return feats + phys_row(a.rowmap, t.off0 + gr) * (long long)K + 8 * o;
        };
        // the record, row scales and critical queries of tile `t` for the compute waves, buffer `par`
        auto hand_over = [&](int par, bool hv, const F2Tile& t, float sinv) {
            if (wave == 4 && lane == 0) {
                long long* c = sCtl + par * 8;
                c[0] = hv ? 1 : 0; c[1] = t.bag; c[2] = t.off0; c[3] = t.Nb; c[4] = t.row0; c[5] = t.slot;
            }
            if (o == 0) sInvAll[par * F2_BM + 16 * j + rr] = sinv;
        };
        auto load_q = [&](const F2Tile& t) {
            return a.qmax[((long long)t.bag * a.C + (tid < 384 ? 0 : c1_first)) * QD + (tid & 127)];
        };
        float sc = 1.f, sinv = 1.f, nsc = 1.f, nsinv = 1.f;
        const float* src = row_src(cur, sc, sinv);
        hand_over(0, true, cur, sinv);
        sQall[tid - 256] = load_q(cur);

#pragma unroll
        for (int c = 0; c < NK1; ++c) {
            ring[c][0] = *(const DSMIL_GLOBAL f32x4*)(src + 32 * c);
            ring[c][1] = *(const DSMIL_GLOBAL f32x4*)(src + 32 * c + 4);
        }
        int nitem = item + stride;
        F2Tile nxt = cur;
        bool have_next = f2_fetch(a, tiles_per_bag, n_items, stride, nitem, nxt);
        const float* nsrc = have_next ? row_src(nxt, nsc, nsinv) : src;
        __syncthreads();                                  // P0: biases and the first tile's hand-over are in LDS
        int par = 0;
        f32x4* dst0 = sXp + (long long)((o >> 1) * 4 + (o & 1)) * F2_BM + 16 * j + rr;   // + 8 F2_BM c: chunk c = steps 2c, 2c+1
        while (have) {
            STAMP(0);
            // the next tile's record and row scales for the compute waves; its bag's critical queries are requested now and
            // handed over behind the chunk loop (nobody reads buffer par ^ 1 during this tile)
            const float qn = load_q(nxt);
            hand_over(par ^ 1, have_next, nxt, nsinv);
            auto refill = [&](int k) {                    // (k a literal at every call site)
                if constexpr ((ABL & 2) == 0) {
                    if constexpr ((ABL & 64) != 0) {      // (variant: streaming loads)
                        ring[k][0] = __builtin_nontemporal_load((const DSMIL_GLOBAL f32x4*)(nsrc + 32 * k));
                        ring[k][1] = __builtin_nontemporal_load((const DSMIL_GLOBAL f32x4*)(nsrc + 32 * k + 4));
                    } else {
                        ring[k][0] = *(const DSMIL_GLOBAL f32x4*)(nsrc + 32 * k);
                        ring[k][1] = *(const DSMIL_GLOBAL f32x4*)(nsrc + 32 * k + 4);
                    }
                }
            };
#pragma unroll
            for (int c = 0; c < NK1; ++c) {
                if constexpr ((ABL & 16) == 0) {          // (ablation: no cut, no plane writes)
                    F2Frag f[2];
                    split2h_scaled(ring[c][0], ring[c][1], sc, f);
                    f32x4* dst = dst0 + (long long)c * (8 * F2_BM);
                    dst[0] = f[0].f;
                    dst[2 * F2_BM] = f[1].f;
                }
                // Refill: chunk c of the NEXT tile into slot c (no next tile: nsrc = src, a harmless re-read; no branch, so hipcc
                // counts the loads in flight exactly).  Only the FIRST half of the slots here, the second half goes out behind the
                // exchange barriers below: a load issued into a full memory queue blocks the issuing wave (stamps: ~800 cycles per
                // chunk with a refill behind every cut — the rate at which the chip drains 8 KB per CU from HBM — against ~450
                // without), so the requests are spread over the tile.
                if constexpr ((ABL & 2) == 0) {           // (ablation: no feature loads behind the first ring fill)
                    if constexpr ((ABL & 256) != 0) { if (c % 2 == 1) refill(c / 2); }   // (variant: one refill per two cuts, all through GEMM 1)
                    else if (c < NK1 / 2) refill(c);
                }
                // planes up to chunk c are visible to the compute waves: one barrier per chunk through the first half of the tile
                // (the cut is only a chunk ahead of the MFMAs there), then one per four chunks (a barrier costs ~300 cycles of eight
                // waves).  (Tried: per-cutter progress counters in LDS polled by the compute waves instead of barriers, cutters
                // running free — 0.60 - 0.63 ms against 0.53: a cutter is no faster without the barriers (it shares its SIMD's
                // issue port with a wave of MFMAs), and its refills then run into the compute waves' weight loads.)
                if (c < NK1 / 2 || c % 4 == 3) __syncthreads();
                if (c == 0) STAMP(1);
                if (c == 3) STAMP(2);
                if (c == 7) STAMP(3);
                if (c == 11) STAMP(4);
                if (c == NK1 - 1) STAMP(5);
            }
            sQall[(par ^ 1) * 256 + (tid - 256)] = qn;
            // The tile after next.  Its record (two dependent scalar loads) is looked up behind B2 and its rows' addresses and
            // scales (vector loads that depend on the record) behind B4: the cutters stall there while the compute waves run the
            // two halves of GEMM 2, instead of making them wait at a barrier.
            int nnitem = nitem + stride;
            F2Tile nn = nxt;
            bool have_nn = false;
            float nnsc = nsc, nnsinv = nsinv;
            const float* nnsrc = nsrc;
            auto look_record = [&]() { have_nn = have_next && f2_fetch(a, tiles_per_bag, n_items, stride, nnitem, nn); };
            auto look_rows = [&]() { if (have_nn) nnsrc = row_src(nn, nnsc, nnsinv); };
            // the second half of the next tile's chunks, a quarter behind each of the compute waves' exchange barriers B1 .. B4
            // (NK1 = 4, 12: no even quarters — all at once)
            auto refill_q = [&](int q) {
                if constexpr (NK1 % 8 == 0) {
#pragma unroll
                    for (int k = NK1 / 2 + q * (NK1 / 8); k < NK1 / 2 + (q + 1) * (NK1 / 8); ++k) refill(k);
                } else if (q == 0) {
#pragma unroll
                    for (int k = NK1 / 2; k < NK1; ++k) refill(k);
                }
            };
            if (a.nonlinear) {
                refill_q(0); __syncthreads();             // B1
                refill_q(1); __syncthreads();             // B2
                look_record();
                refill_q(2); __syncthreads();             // B3
                refill_q(3); __syncthreads();             // B4
                look_rows();
            } else {
                refill_q(0); refill_q(1); refill_q(2); refill_q(3);
                look_record(); look_rows();
            }
            // (the lookups are consumed HERE, before the tail's stores queue up behind them: at the loop head hipcc waits for
            // everything in flight that the next iteration reads)
            asm volatile("" : "+v"(nnsc), "+v"(nnsinv), "+v"(nnsrc));
            STAMP(6);
            f2_tail<ABL>(a, cur, noq, sXp, sS, sPw, sQall + par * 256, sInvAll + par * F2_BM, NKS);
            STAMP(7);
            STAMP_OUT(cur);
            item = nitem; cur = nxt; have = have_next; src = nsrc; sc = nsc;
            nitem = nnitem; nxt = nn; have_next = have_nn; nsrc = nnsrc; nsc = nnsc; nsinv = nnsinv;
            par ^= 1;
        }
        return;
    }

    // ================= a COMPUTE wave: hidden / query units 32 wave .. 32 wave + 31 of all 64 rows =================
    // weight fragments: a UNIFORM base (the wave's two pieces of a chunk) plus a 32-bit lane offset — scalar-base global loads;
    // (per-step vector pointers were hoisted out of the tile loop by hipcc and spilled: every reload waited for the whole ring)
    const char* wbase = reinterpret_cast<const char*>(wimg + (2 * wave) * 64);
    const unsigned loff = (unsigned)lane * 16u;
    F2Frag wr[F2_WRD][2];
    auto load_w = [&](auto slot_, int s) {                  // s may run past the last step: the ring wraps into the next tile
        constexpr int R = decltype(slot_)::value;
        if constexpr ((ABL & 1) != 0) return;
        int sw = s < nst ? s : s - nst;
        asm volatile("" : "+s"(sw));                        // (a scalar formed here, not a vector pointer hoisted out of the loop)
        const char* p = wbase + (size_t)sw * (F2_CHUNK_F4 * 16);
        wr[R][0].f = *(const DSMIL_GLOBAL f32x4*)(p + loff);
        wr[R][1].f = *(const DSMIL_GLOBAL f32x4*)(p + 1024 + loff);
    };
    // B fragments of GEMM-1 step s for both row groups, from the resident planes
    auto read_x = [&](int s, F2Frag (&fb)[2][2]) {
        const f32x4* p = sXp + (long long)(s * 4 + hi) * F2_BM + l31;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            fb[g][0].f = p[32 * g];
            fb[g][1].f = p[2 * F2_BM + 32 * g];
        }
    };
    // ... of GEMM-2 step st (its half of the exchange buffer)
    auto read_h = [&](int st, F2Frag (&fb)[2][2]) {
        const f32x4* p = sHp + (long long)((st & 3) * 2 + hi) * F2_BM + l31;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            fb[g][0].f = p[32 * g];
            fb[g][1].f = p[8 * F2_BM + 32 * g];
        }
    };
    // six MFMAs of a step, alternating between the row groups (independent accumulators), smallest products first
    auto mfma6 = [&](f32x16 (&acc)[2], const F2Frag (&w)[2], const F2Frag (&x)[2][2]) {
        if constexpr ((ABL & 4) != 0) {                      // (ablation: no MFMAs; the operands stay used)
            acc[0][0] += w[0].f[0] + w[1].f[0] + x[0][0].f[0] + x[0][1].f[0];
            acc[1][0] += x[1][0].f[0] + x[1][1].f[0];
            return;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1].v, x[g][0].v, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0].v, x[g][1].v, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0].v, x[g][0].v, acc[g], 0, 0, 0);
    };
    using F_ = std::false_type;
    using T_ = std::true_type;
    static_assert(F2_WRD == 8, "step groups below");
    constexpr int WLA = (ABL & 128) ? 7 : F2_WLA;          // (variant: one more step of weights in flight)
    // the weights of the first F2_WLA steps: the ring then runs across tiles (nst % 8 == 0)
    load_w(std::integral_constant<int, 0>{}, 0);
    load_w(std::integral_constant<int, 1>{}, 1);
    load_w(std::integral_constant<int, 2>{}, 2);
    load_w(std::integral_constant<int, 3>{}, 3);
    load_w(std::integral_constant<int, 4>{}, 4);
    load_w(std::integral_constant<int, 5>{}, 5);
    if constexpr (WLA == 7) load_w(std::integral_constant<int, 6>{}, 6);
    const float ia1 = trailer[0], ia2 = trailer[1];
    __syncthreads();                                      // P0: biases and the first tile's hand-over are in LDS
    int par = 0;
    while (true) {
        // this tile's record, from the cutters (LDS; behind the previous tile's last barrier)
        const int* ctl = reinterpret_cast<const int*>(sCtl + par * 8);
        auto ctl64 = [&](int i) {                         // (uniform values: into scalar registers)
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(ctl[2 * i]), hi_ = (unsigned)__builtin_amdgcn_readfirstlane(ctl[2 * i + 1]);
            return (long long)(((unsigned long long)hi_ << 32) | lo);
        };
        if (__builtin_amdgcn_readfirstlane(ctl[0]) == 0) break;   // (the cutters left their loop after the same tile)
        cur.bag = (int)ctl64(1); cur.off0 = ctl64(2); cur.Nb = ctl64(3); cur.row0 = ctl64(4); cur.slot = ctl64(5);
        const float* sInv = sInvAll + par * F2_BM;
        STAMP(0);
        f32x16 Hw[2], Qw[2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) Hw[g][r] = 0.f;
        __syncthreads();                                  // the cutters' first: planes of chunk 0
        STAMP(1);
        F2Frag xb[2][2];
        read_x(0, xb);
        // one 16-k step of GEMM 1: weights of step s + F2_WLA requested, fragments of step s + 1 read, six MFMAs
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const bool pre = s + 1 < NKS;
            if ((s & 1) && pre && f2_group_first<NK1>((s + 1) / 2)) __syncthreads();   // the cutters' barrier for the chunk group of step s + 1
            __builtin_amdgcn_sched_barrier(0);
            switch ((s + WLA) % 8) {                   // (a literal after unrolling: ring slots are registers)
                case 0: load_w(std::integral_constant<int, 0>{}, s + WLA); break;
                case 1: load_w(std::integral_constant<int, 1>{}, s + WLA); break;
                case 2: load_w(std::integral_constant<int, 2>{}, s + WLA); break;
                case 3: load_w(std::integral_constant<int, 3>{}, s + WLA); break;
                case 4: load_w(std::integral_constant<int, 4>{}, s + WLA); break;
                case 5: load_w(std::integral_constant<int, 5>{}, s + WLA); break;
                case 6: load_w(std::integral_constant<int, 6>{}, s + WLA); break;
                default: load_w(std::integral_constant<int, 7>{}, s + WLA); break;
            }
            __builtin_amdgcn_sched_barrier(0);
            F2Frag xn[2][2];
            if (pre) read_x(s + 1, xn);
            mfma6(Hw, wr[s % 8], xb);
            if (pre) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int p = 0; p < 2; ++p) xb[g][p] = xn[g][p];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s == 7) STAMP(2);
            if (s == 15) STAMP(3);
            if (s == 23) STAMP(4);
        }
        STAMP(5);
        const float inv_sc[2] = {sInv[l31], sInv[32 + l31]};   // 1 / row scale of rows 32g + l31
        // ---- un-scale, bias (+ReLU): reg 4q+e <-> unit 32 wave + 8q + 4hi + e, row 32g + l31
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + 32 * wave + 8 * q + 4 * hi);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float iv = ia1 * inv_sc[g];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaf(Hw[g][4 * q + e], iv, bq[e]);
                    Hw[g][4 * q + e] = a.nonlinear ? fmaxf(v, 0.f) : v;
                }
            }
        }
        if (!a.nonlinear) {
#pragma unroll
            for (int g = 0; g < 2; ++g) Qw[g] = Hw[g];
        } else {
            // row maxima of the hidden layer (>= 0 after the ReLU) -> per-row scale of its fp16 cut
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float m = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, Hw[g][r]);
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                if (hi == 0) sMax[wave * F2_BM + 32 * g + l31] = m;
            }
            __syncthreads();                              // B1: sMax complete
            STAMP(6);
            float hsc[2], hinv[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int r = 32 * g + l31;
                const float m = fmaxf(fmaxf(sMax[r], sMax[F2_BM + r]), fmaxf(sMax[2 * F2_BM + r], sMax[3 * F2_BM + r]));
                hsc[g] = f2_scale(m, hinv[g]);
            }
            // registers 8sx..8sx+7 of this wave's H tile are, for row l31, the 8 hidden units of GEMM-2 step 2 wave + sx
            // (the k permutation the packed W2 carries): scale, cut, publish in this wave's half of the exchange
            auto publish_h = [&]() {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int sx = 0; sx < 2; ++sx) {
                        const f32x4 h0 = {Hw[g][8 * sx], Hw[g][8 * sx + 1], Hw[g][8 * sx + 2], Hw[g][8 * sx + 3]};
                        const f32x4 h1 = {Hw[g][8 * sx + 4], Hw[g][8 * sx + 5], Hw[g][8 * sx + 6], Hw[g][8 * sx + 7]};
                        F2Frag f[2];
                        split2h_scaled(h0, h1, hsc[g], f);
                        f32x4* dst = sHp + (long long)((2 * (wave & 1) + sx) * 2 + hi) * F2_BM + 32 * g + l31;
                        dst[0] = f[0].f;
                        dst[8 * F2_BM] = f[1].f;
                    }
            };
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) Qw[g][r] = 0.f;
            auto step2 = [&](auto ri, auto pre_, int st) {
                constexpr int RI = decltype(ri)::value;
                constexpr bool PRE = decltype(pre_)::value;
                __builtin_amdgcn_sched_barrier(0);
                load_w(std::integral_constant<int, (RI + WLA) % F2_WRD>{}, NKS + st + WLA);
                __builtin_amdgcn_sched_barrier(0);
                F2Frag xn[2][2];
                if constexpr (PRE) read_h(st + 1, xn);
                mfma6(Qw, wr[RI], xb);
                if constexpr (PRE) {
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int p = 0; p < 2; ++p) xb[g][p] = xn[g][p];
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            if (wave < 2) publish_h();
            __syncthreads();                              // B2: half 0 (units of waves 0, 1) published
            STAMP(7);
            read_h(0, xb);
            step2(std::integral_constant<int, 0>{}, T_{}, 0);   // (NKS % 8 == 0: GEMM-2 step st sits in ring slot st)
            step2(std::integral_constant<int, 1>{}, T_{}, 1);
            step2(std::integral_constant<int, 2>{}, T_{}, 2);
            step2(std::integral_constant<int, 3>{}, F_{}, 3);
            STAMP(8);
            __syncthreads();                              // B3: half 0 consumed
            if (wave >= 2) publish_h();
            __syncthreads();                              // B4: half 1 published
            STAMP(9);
            read_h(4, xb);
            step2(std::integral_constant<int, 4>{}, T_{}, 4);
            step2(std::integral_constant<int, 5>{}, T_{}, 5);
            step2(std::integral_constant<int, 6>{}, T_{}, 6);
            step2(std::integral_constant<int, 7>{}, F_{}, 7);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + QD + 32 * wave + 8 * q + 4 * hi);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float iv = ia2 * hinv[g];
#pragma unroll
                    for (int e = 0; e < 4; ++e) Qw[g][4 * q + e] = fast_tanh(fmaf(Qw[g][4 * q + e], iv, bq[e]));
                }
            }
        }
        STAMP(10);
        f2_tail<ABL>(a, cur, Qw, sXp, sS, sPw, sQall + par * 256, sInv, NKS);
        STAMP(11);
        STAMP_OUT(cur);
        par ^= 1;
    }
}

// fp32 query weights -> two fp16 planes (round to nearest) of the power-of-two scaled values, MFMA-fragment order:
//   chunk s < nks (GEMM 1):  [t][p][lane (l31,hi)][e] = plane_p(a1 W1[32t + l31][16s + 8hi + e])   (0 past K)
//   chunk nks + 2t + sx:     [t2][p][lane][e] = plane_p(a2 W2[32t2 + l31][32t + 16sx + (e&3) + 8(e>>2) + 4hi])
//   trailer (behind chunk nks + 8): {1 / a1, 1 / a2, max_j ||W1[j]||_1}
// a1 / a2 from max |W1| / max |W2| (f2_scale), computed by every workgroup for itself (same values everywhere).
__global__ __launch_bounds__(256) void k_pack_agg_f2(const float* __restrict__ q0_w, const float* __restrict__ q2_w,
                                                     _Float16* __restrict__ out, int K, int nks) {
    __shared__ float s_m[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m1 = 0.f, m2 = 0.f;
    for (long long i = tid; i < (long long)QD * K; i += 256) m1 = fmaxf(m1, fabsf(q0_w[i]));
    if (q2_w)
        for (int i = tid; i < QD * QD; i += 256) m2 = fmaxf(m2, fabsf(q2_w[i]));
    m1 = wave_max(m1);
    m2 = wave_max(m2);
    if (lane == 0) { s_m[0][wave] = m1; s_m[1][wave] = m2; }
    __syncthreads();
    m1 = fmaxf(fmaxf(s_m[0][0], s_m[0][1]), fmaxf(s_m[0][2], s_m[0][3]));
    m2 = fmaxf(fmaxf(s_m[1][0], s_m[1][1]), fmaxf(s_m[1][2], s_m[1][3]));
    float i1, i2;
    const float a1 = f2_scale(m1, i1), a2 = f2_scale(m2, i2);
    const long long per = (long long)F2_CHUNK_F4 * 8;   // fp16 per chunk
    const long long total = (long long)(nks + (q2_w ? 8 : 0)) * per;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < total; i += (long long)gridDim.x * 256) {
        const int s = (int)(i / per);
        int r = (int)(i - s * per);
        const int e = r & 7; r >>= 3;
        const int ln = r & 63; r >>= 6;
        const int p = r & 1, t = r >> 1;
        const int l31 = ln & 31, hi = ln >> 5;
        float v;
        if (s < nks) {
            const int k = 16 * s + 8 * hi + e;
            v = k < K ? q0_w[(long long)(32 * t + l31) * K + k] * a1 : 0.f;
        } else {
            const int st = s - nks, tt = st >> 1, sx = st & 1;
            v = q2_w[(32 * t + l31) * QD + 32 * tt + 16 * sx + (e & 3) + 8 * (e >> 2) + 4 * hi] * a2;
        }
        const _Float16 h = (_Float16)v;
        out[i] = p == 0 ? h : (_Float16)(v - (float)h);
    }
    if (blockIdx.x == 0) {
        // trailer[2] = max_j sum_k |W1[j][k]|: with a row's max |x| it bounds the hidden layer of that row (k_attend_f3 scales its
        // hidden planes by that bound instead of exchanging the row's maximum between the waves)
        __shared__ float s_n1[4];
        float n1 = 0.f;
        for (int j = wave; j < QD; j += 4) {
            float sum = 0.f;
            for (int k = lane; k < K; k += 64) sum += fabsf(q0_w[(long long)j * K + k]);
            n1 = fmaxf(n1, wave_sum(sum));
        }
        if (lane == 0) s_n1[wave] = n1;
        __syncthreads();
        if (tid == 0) {
            float* tr = reinterpret_cast<float*>(out + (long long)(nks + 8) * per);
            tr[0] = i1;
            tr[1] = i2;
            tr[2] = fmaxf(fmaxf(s_n1[0], s_n1[1]), fmaxf(s_n1[2], s_n1[3]));
        }
    }
}

}  // namespace
