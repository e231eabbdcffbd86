// DSMIL dual-stream aggregator, forward — hand-written HIP for gfx950 (MI355X, CDNA4).
//
// What it computes (reference: dsmil.py FCLayer :6-12, BClassifier.forward :46-62,
// MILNet.forward :70-74), for a batch of independent variable-length bags:
//   c    = x W_i^T + b_i                          instance logits            [N,C]
//   idx  = argmax_n c[n,:]                         critical instances         [C]
//   Q    = tanh(relu(x W1^T + b1) W2^T + b2)      queries                    [N,128]
//   qmax = Q-MLP(x[idx])                                                      [C,128]
//   A    = softmax_n(Q qmax^T / sqrt(128))        attention over instances   [N,C]
//   B    = A^T V,  pred = Conv1d(C,C,K)(B)         bag embedding / bag logits
//
// Launch sequence on one stream (no host sync, hipGraph-capturable):
//   k_logits_stream   HBM stream over x: c, per-tile (max,idx) partials (8 lanes per row, full-line
//                     non-temporal loads; k_logits_argmax is the general form: bf16 rows, K % 4 != 0,
//                     caller-supplied logits)
//   k_qmax            per (bag,class): finish argmax, run the query MLP on the critical row
//   k_pack_agg_split  cut the query weights into three exact bf16 planes, MFMA-fragment order
//   k_query_attend_split  the dominant kernel: per 32-row wave tile the query MLP runs TRANSPOSED
//                     (H^T = W1 X^T keeps instances on the MFMA column axis, so the ReLU'd H^T
//                     accumulator registers are fed straight back as the B operand of Q^T = W2 H^T:
//                     no LDS round trip) on bf16 MFMA over exact three-plane cuts of the fp32
//                     operands (agg_split.h; DSMIL_MLP=f32 selects k_query_attend on
//                     v_mfma_f32_32x32x2_f32); scores, tile-local softmax statistics and the weighted
//                     value sum are fused behind it.  Q is never written to memory.
//   k_finish, k_pred  combine tile partials: A = exp(s-m)/l, B, pred
// dsmil_agg_shard_* (one bag sharded by instances over several GPUs) run the same kernels in two
// phases around the caller's exchanges.
//
// MFMA fragment maps used (cdna_hip_programming.md §3): 32x32x2 f32: A lane l = A[i=l&31][k=l>>5],
// B lane l = B[k=l>>5][j=l&31], D lane l reg r = D[(r&3)+8*(r>>2)+4*(l>>5)][l&31].

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "dsmil_hip.h"
#include "prof.h"

#include "agg_common.h"
#include "agg_split.h"
#include "agg_hs.h"
#include "agg_f2.h"
#include "agg_f3.h"
#include "agg_res.h"
#include "lds_attr.h"

namespace {

// --------------------------------------------------------------------------------------------
// k_logits_argmax: c = x W_i^T + b_i (dsmil.py:11) and per-tile arg-max partials (dsmil.py:52).
// One wave owns 32 rows, 4 rows in flight; lanes stride the feature axis with 16-B loads.
// GIVEN = true: the logits are taken from classes_in (BClassifier.forward(feats, c)).
// --------------------------------------------------------------------------------------------
// Work of ONE 128-row tile of one bag by a 256-thread workgroup; s_v / s_i: 8-entry LDS scratch.
template <int VEC, bool GIVEN, typename T = float>
__device__ __forceinline__ void logits_tile(
    const T* __restrict__ feats, const int64_t* __restrict__ offsets,
    const float* __restrict__ fc_w, const float* __restrict__ fc_b,
    const float* __restrict__ classes_in, float* __restrict__ classes_out,
    float* __restrict__ part_val, long long* __restrict__ part_idx, int K, int C,
    int bag, int tile, float* s_v, long long* s_i, const int64_t* __restrict__ rowmap) {
    const long long off0 = offsets[bag];
    const long long Nb = offsets[bag + 1] - off0;
    const long long row0 = (long long)tile * R0;
    if (row0 >= Nb) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long slot = off0 / R0 + bag + tile;

    for (int c0 = 0; c0 < C; c0 += 2) {
        const int c1 = (c0 + 1 < C) ? c0 + 1 : c0;
        float bv0 = -INFINITY, bv1 = -INFINITY;
        long long bi0 = 0x7fffffffffffffffLL, bi1 = 0x7fffffffffffffffLL;
        for (int rg = 0; rg < 8; ++rg) {
            const long long rbase = row0 + wave * 32 + rg * 4;
            if (rbase >= Nb) break;  // wave-uniform
            float va0, va1, vb0, vb1, vc0, vc1, vd0, vd1;  // rows a..d, classes c0/c1
            const long long ra = rbase, rb = (rbase + 1 < Nb) ? rbase + 1 : Nb - 1,
                            rc = (rbase + 2 < Nb) ? rbase + 2 : Nb - 1, rd = (rbase + 3 < Nb) ? rbase + 3 : Nb - 1;
            if constexpr (!GIVEN) {
                const T* xa = feats + phys_row(rowmap, off0 + ra) * (long long)K;
                const T* xb = feats + phys_row(rowmap, off0 + rb) * (long long)K;
                const T* xc = feats + phys_row(rowmap, off0 + rc) * (long long)K;
                const T* xd = feats + phys_row(rowmap, off0 + rd) * (long long)K;
                const float* w0p = fc_w + (long long)c0 * K;
                const float* w1p = fc_w + (long long)c1 * K;
                va0 = va1 = vb0 = vb1 = vc0 = vc1 = vd0 = vd1 = 0.f;
                for (int k0 = 0; k0 < K; k0 += 256) {
                    const int k = k0 + lane * 4;
                    const f32x4 w0 = load4<VEC>(w0p, k, K);
                    const f32x4 w1 = load4<VEC>(w1p, k, K);
                    const f32x4 a = load4<VEC, T>(xa, k, K);
                    const f32x4 b = load4<VEC, T>(xb, k, K);
                    const f32x4 c = load4<VEC, T>(xc, k, K);
                    const f32x4 d = load4<VEC, T>(xd, k, K);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        va0 = fmaf(a[e], w0[e], va0); va1 = fmaf(a[e], w1[e], va1);
                        vb0 = fmaf(b[e], w0[e], vb0); vb1 = fmaf(b[e], w1[e], vb1);
                        vc0 = fmaf(c[e], w0[e], vc0); vc1 = fmaf(c[e], w1[e], vc1);
                        vd0 = fmaf(d[e], w0[e], vd0); vd1 = fmaf(d[e], w1[e], vd1);
                    }
                }
                const float b0 = fc_b[c0], b1 = fc_b[c1];
                va0 = wave_sum(va0) + b0; va1 = wave_sum(va1) + b1;
                vb0 = wave_sum(vb0) + b0; vb1 = wave_sum(vb1) + b1;
                vc0 = wave_sum(vc0) + b0; vc1 = wave_sum(vc1) + b1;
                vd0 = wave_sum(vd0) + b0; vd1 = wave_sum(vd1) + b1;
                if (lane < 4 && rbase + lane < Nb) {
                    const float o0 = lane == 0 ? va0 : lane == 1 ? vb0 : lane == 2 ? vc0 : vd0;
                    const float o1 = lane == 0 ? va1 : lane == 1 ? vb1 : lane == 2 ? vc1 : vd1;
                    float* o = classes_out + (off0 + rbase + lane) * (long long)C;
                    o[c0] = o0;
                    if (c1 != c0) o[c1] = o1;
                }
            } else {
                const float* ci = classes_in + off0 * (long long)C;
                va0 = ci[ra * C + c0]; va1 = ci[ra * C + c1];
                vb0 = ci[rb * C + c0]; vb1 = ci[rb * C + c1];
                vc0 = ci[rc * C + c0]; vc1 = ci[rc * C + c1];
                vd0 = ci[rd * C + c0]; vd1 = ci[rd * C + c1];
            }
            // rows past the end were clamped to Nb-1: a duplicate can never beat itself (same index)
            if (better(va0, ra, bv0, bi0)) { bv0 = va0; bi0 = ra; }
            if (better(vb0, rb, bv0, bi0)) { bv0 = vb0; bi0 = rb; }
            if (better(vc0, rc, bv0, bi0)) { bv0 = vc0; bi0 = rc; }
            if (better(vd0, rd, bv0, bi0)) { bv0 = vd0; bi0 = rd; }
            if (better(va1, ra, bv1, bi1)) { bv1 = va1; bi1 = ra; }
            if (better(vb1, rb, bv1, bi1)) { bv1 = vb1; bi1 = rb; }
            if (better(vc1, rc, bv1, bi1)) { bv1 = vc1; bi1 = rc; }
            if (better(vd1, rd, bv1, bi1)) { bv1 = vd1; bi1 = rd; }
        }
        // combine the 4 waves (values are wave-uniform)
        __syncthreads();
        if (lane == 0) { s_v[wave] = bv0; s_i[wave] = bi0; s_v[4 + wave] = bv1; s_i[4 + wave] = bi1; }
        __syncthreads();
        if (threadIdx.x < 2 && (threadIdx.x == 0 || c1 != c0)) {
            const int h = threadIdx.x * 4;
            float bv = s_v[h];
            long long bi = s_i[h];
            for (int w = 1; w < 4; ++w)
                if (better(s_v[h + w], s_i[h + w], bv, bi)) { bv = s_v[h + w]; bi = s_i[h + w]; }
            const int c = threadIdx.x ? c1 : c0;
            part_val[slot * C + c] = bv;
            part_idx[slot * C + c] = bi;
        }
    }
}

template <int VEC, bool GIVEN, typename T = float>
__global__ __launch_bounds__(256) void k_logits_argmax(
    const T* __restrict__ feats, const int64_t* __restrict__ offsets,
    const float* __restrict__ fc_w, const float* __restrict__ fc_b,
    const float* __restrict__ classes_in, float* __restrict__ classes_out,
    float* __restrict__ part_val, long long* __restrict__ part_idx, int K, int C, int bag0,
    const int64_t* __restrict__ rowmap) {
    __shared__ float s_v[8];
    __shared__ long long s_i[8];
    logits_tile<VEC, GIVEN, T>(feats, offsets, fc_w, fc_b, classes_in, classes_out, part_val, part_idx, K, C,
                               bag0 + (int)blockIdx.y, (int)blockIdx.x, s_v, s_i, rowmap);
}

// --------------------------------------------------------------------------------------------
// k_logits_stream: the same work as k_logits_argmax for fp32 rows with K % 4 == 0 and computed
// logits, shaped for the HBM stream (this pass reads every feature byte once: 1.3 GB per 64 bags).
// 8 lanes share a row and walk it in 128-B segments, so one load instruction covers 8 rows x one
// full 128-B line each; 8 segments (1 KiB per lane-row) are in flight before the first FMA; the
// loads are unconditional (tail segments re-read a clamped address against zero-padded weights in
// LDS) and non-temporal (the stream is 5x the Infinity Cache and k_query_attend reads it again
// only after it has all gone by).  The lane reduction is 3 xor-shuffles per 8 rows.
// Same tile geometry and partial layout as k_logits_argmax (R0 rows per workgroup).
// --------------------------------------------------------------------------------------------
// 16 bytes of a feature row, non-temporal, as EPL = 16 / sizeof(T) floats (4 fp32 or 8 bf16)
template <typename T>
struct StreamVec;
template <>
struct StreamVec<float> {
    static constexpr int EPL = 4;
    f32x4 v;
    __device__ __forceinline__ void load(const float* p) { v = __builtin_nontemporal_load((const DSMIL_GLOBAL f32x4*)p); }
    __device__ __forceinline__ float at(int e) const { return v[e]; }
};
template <>
struct StreamVec<bf16_t> {
    static constexpr int EPL = 8;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = __builtin_nontemporal_load((const DSMIL_GLOBAL u32x4_t*)p); }
    __device__ __forceinline__ float at(int e) const {
        const unsigned w = v[e >> 1];
        return __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
    }
};

// Register budget of the one-class form: 80 (6 waves per SIMD).  The fp32 attend kernel of ANOTHER batch (another stream)
// holds 2 x 216 of a SIMD's 512 registers, so a logits wave fits beside it only below 80; measured same-box A/B at three
// streams: 77.7 k against 76.2 k bags/s (-0.6 % at one stream).  The two-class form does not fit the budget without
// halving its loads in flight, which costs more (bf16: 195 k against 215 k bags/s) than the co-residency buys: it keeps
// its ~130 registers.  Either way the streams overlap at kernel tails more than inside kernels.
template <int CP, typename T>  // classes per pass: 1 or 2; T = float or bf16 feature rows
__global__ __launch_bounds__(256, (CP == 1 ? 6 : 3)) void k_logits_stream(
    const T* __restrict__ feats, const int64_t* __restrict__ offsets,
    const float* __restrict__ fc_w, const float* __restrict__ fc_b, float* __restrict__ classes_out,
    float* __restrict__ part_val, long long* __restrict__ part_idx, int K, int C, int bag0,
    const int64_t* __restrict__ rowmap, int r0 = R0, int* __restrict__ qm_flag = nullptr,
    TrainPrologueJob job = TrainPrologueJob{}, float* __restrict__ rowmax = nullptr,
    const int* __restrict__ tile_pre = nullptr, int n_bags = 0) {
    // tile_pre (ragged batches, k_tile_prefix with BM = r0): a ONE-dimensional grid over the real tiles — tile_pre[b] of them in
    // front of bag b — instead of (tiles of the longest bag) x n_bags workgroups of which most find nothing to do
    // rowmax (fp32 rows, k_attend_f2 follows): max_k |x[row][k]| of every LOGICAL row, a by-product of this pass over the
    // bag — the attend kernel derives the row's power-of-two scale for its fp16 plane cut from it (agg_f2.h)
    // qm_flag: the hand-off flags of the attend launch that follows (AttendArgs::qm_flag), cleared here
    if (qm_flag && blockIdx.x == 0 && (int)threadIdx.x < C) qm_flag[(long long)(bag0 + (int)blockIdx.y) * C + threadIdx.x] = 0;
    if (job.blocks) {   // a training step: the last job.blocks workgroups cut the weight planes and write the offsets
        const int first = (int)gridDim.x - job.blocks;
        if ((int)blockIdx.x >= first) {
            const long long i0 = (long long)((int)blockIdx.x - first) * 256 + threadIdx.x, stride = (long long)job.blocks * 256;
            pack_agg_split_range(job.q0_w, job.q2_w, job.wsplit, job.K, job.nks, 0, i0, stride);
            if (job.q2_w) pack_agg_split_range(job.q2_w, nullptr, job.w2t, QD, 8, 1, i0, stride);
            if (i0 == 0) { job.off_a[0] = 0; job.off_a[1] = job.N; job.off_b[0] = 0; job.off_b[1] = job.N; }
            return;
        }
    }
    // r0 = rows per workgroup: R0 (128: a wave walks four 8-row groups) for batches, 32 (one group per wave) when there are
    // few rows — a lone 10 000-row bag is 79 workgroups at 128 rows, a third of the chip's CUs for an HBM-bound stream
    constexpr int EPL = StreamVec<T>::EPL;   // elements per lane per 128-B segment
    constexpr int SEG = 8 * EPL;             // elements per segment (32 fp32 / 64 bf16)
    constexpr int U = 8;                     // segments in flight per lane-row
    extern __shared__ __attribute__((aligned(16))) float s_w[];  // [CP][Kpad]: weights, zero past K, plus one zero segment
    __shared__ float s_v[8];
    __shared__ long long s_i[8];
    int bag = bag0 + (int)blockIdx.y, tile = (int)blockIdx.x;
    if (tile_pre) {
        if ((int)blockIdx.x >= tile_pre[n_bags]) return;
        const int b = tile_owner(tile_pre, n_bags, (int)blockIdx.x);
        bag = bag0 + b;
        tile = (int)blockIdx.x - tile_pre[b];
    }
    const long long off0 = job.blocks ? 0 : offsets[bag];       // (the job's offsets are being written by this very launch)
    const long long Nb = job.blocks ? job.N : offsets[bag + 1] - off0;
    const long long row0 = (long long)tile * r0;
    if (row0 >= Nb) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 7, rr = lane >> 3;
    const long long slot = off0 / r0 + bag + tile;
    const int nseg = (K + SEG - 1) / SEG, Kpad = (nseg + 1) * SEG;
    const int ngrp = r0 >> 5, rpw = r0 >> 2;   // 8-row groups per wave, rows per wave

    for (int c0 = 0; c0 < C; c0 += CP) {
        const int c1 = (CP == 2 && c0 + 1 < C) ? c0 + 1 : c0;
        __syncthreads();
        for (int i = tid; i < CP * Kpad; i += 256) {
            const int cc = i / Kpad, k = i - cc * Kpad;
            s_w[i] = k < K ? fc_w[(long long)(cc ? c1 : c0) * K + k] : 0.f;
        }
        __syncthreads();
        const float b0 = fc_b[c0], b1 = fc_b[c1];
        float bv0 = -INFINITY, bv1 = -INFINITY;
        long long bi0 = 0x7fffffffffffffffLL, bi1 = 0x7fffffffffffffffLL;
#pragma unroll 1
        for (int g = 0; g < ngrp; ++g) {
            const long long rbase = row0 + wave * rpw + g * 8;
            if (rbase >= Nb) break;  // wave-uniform
            const long long r = (rbase + rr < Nb) ? rbase + rr : Nb - 1;
            const T* x = feats + phys_row(rowmap, off0 + r) * (long long)K;
            float a0 = 0.f, a1 = 0.f, xm = 0.f;
            const bool want_max = std::is_same<T, float>::value && rowmax && c0 == 0;   // block-uniform
#pragma unroll 1
            for (int s0 = 0; s0 < nseg; s0 += U) {
                StreamVec<T> v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    int k = (s0 + u) * SEG + j * EPL;
                    k = k < K ? k : K - EPL;  // clamped re-read; its weight is zero
                    v[u].load(x + k);
                }
                if (want_max) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int e = 0; e < EPL; ++e) xm = fmaxf(xm, fabsf(v[u].at(e)));   // (a clamped re-read repeats row data)
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ks = (s0 + u < nseg ? s0 + u : nseg) * SEG + j * EPL;  // segment nseg is all zero
#pragma unroll
                    for (int q = 0; q < EPL / 4; ++q) {
                        const f32x4 w0 = *reinterpret_cast<const f32x4*>(s_w + ks + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) a0 = fmaf(v[u].at(4 * q + e), w0[e], a0);
                        if constexpr (CP == 2) {
                            const f32x4 w1 = *reinterpret_cast<const f32x4*>(s_w + Kpad + ks + 4 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) a1 = fmaf(v[u].at(4 * q + e), w1[e], a1);
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {
                a0 += __shfl_xor(a0, m, 64);
                if constexpr (CP == 2) a1 += __shfl_xor(a1, m, 64);
            }
            a0 += b0;
            a1 += b1;
            if (want_max) {
#pragma unroll
                for (int m = 1; m < 8; m <<= 1) xm = fmaxf(xm, __shfl_xor(xm, m, 64));
                if (j == 0 && rbase + rr < Nb) rowmax[off0 + r] = xm;
            }
            if (j == 0 && rbase + rr < Nb) {
                float* o = classes_out + (off0 + r) * (long long)C;
                o[c0] = a0;
                if (CP == 2 && c1 != c0) o[c1] = a1;
            }
            // rows past the end were clamped to Nb-1: a duplicate can never beat itself (same index)
            if (better(a0, r, bv0, bi0)) { bv0 = a0; bi0 = r; }
            if (CP == 2 && better(a1, r, bv1, bi1)) { bv1 = a1; bi1 = r; }
        }
        // best over the wave's 8 row lanes (lanes of one row agree), then over the 4 waves
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
            const float ov0 = __shfl_xor(bv0, m, 64);
            const long long oi0 = __shfl_xor(bi0, m, 64);
            if (better(ov0, oi0, bv0, bi0)) { bv0 = ov0; bi0 = oi0; }
            if constexpr (CP == 2) {
                const float ov1 = __shfl_xor(bv1, m, 64);
                const long long oi1 = __shfl_xor(bi1, m, 64);
                if (better(ov1, oi1, bv1, bi1)) { bv1 = ov1; bi1 = oi1; }
            }
        }
        __syncthreads();
        if (lane == 0) { s_v[wave] = bv0; s_i[wave] = bi0; s_v[4 + wave] = bv1; s_i[4 + wave] = bi1; }
        __syncthreads();
        if (tid < 2 && (tid == 0 || c1 != c0)) {
            const int h = tid * 4;
            float bv = s_v[h];
            long long bi = s_i[h];
            for (int w = 1; w < 4; ++w)
                if (better(s_v[h + w], s_i[h + w], bv, bi)) { bv = s_v[h + w]; bi = s_i[h + w]; }
            const int c = tid ? c1 : c0;
            part_val[slot * C + c] = bv;
            part_idx[slot * C + c] = bi;
        }
    }
}

// --------------------------------------------------------------------------------------------
// k_logits_pipe (round 6): k_logits_stream's work for the BATCH path of the bf16 aggregator (K = 512, C <= 2), shaped to run
// BESIDE a resident k_attend_bf16_res workgroup of another stream.  That kernel owns every CU for the whole launch (150 KiB of
// LDS, one wave per SIMD) — but since round 6 its waves hold 432 of a SIMD's 512 registers (agg_res.h), so ONE 80-register
// wave fits next to each of them, with the 4 KiB of LDS this kernel needs.  The second read of the features (dsmil.py:50-52
// needs the whole bag before any score) then runs UNDER the MFMA kernel of the batch in front instead of behind it.  What
// that asks of this kernel:
//   * <= 80 registers and no scratch (__launch_bounds__(256, 6); in-bag row indices in 32 bits, no class loop);
//   * its bytes in flight per CU are 4 waves x 4..8 KiB — they have to be in flight ALL the time: the stream is a continuous
//     pipeline of two 4-segment register sets running across the 8-row groups of a wave (the burst form has nothing in flight
//     while it reduces and stores a group).  Every load is unconditional — the last prefetch of a wave is peeled off, not
//     predicated (a conditional load makes hipcc's vmcnt model wait for the youngest load).
// Same tile geometry, partial layout and fmaf order as k_logits_stream: bit-identical logits and partials.
// --------------------------------------------------------------------------------------------
template <int CP>   // classes: 1 or 2 (= C).  (No row map: the bf16 entry point has none — its lookups would be loads in the same
                    // in-order queue as the stream.)
__global__ __launch_bounds__(256, 6) void k_logits_pipe(
    const bf16_t* __restrict__ feats, const int64_t* __restrict__ offsets,
    const float* __restrict__ fc_w, const float* __restrict__ fc_b, float* __restrict__ classes_out,
    float* __restrict__ part_val, long long* __restrict__ part_idx, int bag0,
    int r0, const int* __restrict__ tile_pre, int n_bags) {
    constexpr int K = 512, SEG = 64;         // 8 segments of 64 bf16 (128 B) per row; lane j of a row's 8 takes 16 B of each
    __shared__ __attribute__((aligned(16))) float s_w[CP * K];
    __shared__ float s_v[8];
    __shared__ int s_i[8];
    int bag = bag0 + (int)blockIdx.y, tile = (int)blockIdx.x;
    if (tile_pre) {
        if ((int)blockIdx.x >= tile_pre[n_bags]) return;
        const int b = tile_owner(tile_pre, n_bags, (int)blockIdx.x);
        bag = bag0 + b;
        tile = (int)blockIdx.x - tile_pre[b];
    }
    const long long off0 = offsets[bag];
    const long long Nbl = offsets[bag + 1] - off0;
    const long long row0l = (long long)tile * r0;
    if (row0l >= Nbl) return;
    const int Nb = (int)Nbl, row0 = (int)row0l;              // (a bag of the batch path has < 2^31 rows: n_items check of the host)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 7, rr = lane >> 3;
    const int ngrp = r0 >> 5, rpw = r0 >> 2;                 // 8-row groups per wave, rows per wave
    const int wbase = row0 + wave * rpw;
    int ng = 0;
    if (wbase < Nb) { const int left = (Nb - wbase + 7) >> 3; ng = left < ngrp ? left : ngrp; }
    for (int i = tid; i < CP * K; i += 256) s_w[i] = fc_w[i];
    __syncthreads();
    const float b0 = fc_b[0], b1 = fc_b[CP - 1];
    float bv0 = -INFINITY, bv1 = -INFINITY;
    int bi0 = 0x7fffffff, bi1 = 0x7fffffff;
    if (ng > 0) {
        typedef StreamVec<bf16_t> SV;
        const bf16_t* fbase = feats + j * 8;
        auto row_of = [&](int g) { const int r = wbase + g * 8 + rr; return r < Nb ? r : Nb - 1; };
        auto ptr_of = [&](int r) { return fbase + (off0 + r) * (long long)K; };
        SV va[4], vb[4];
        float a0 = 0.f, a1 = 0.f;
        // One 16-B piece (8 bf16 of one row) at a time: its 8 (16) weights come out of LDS inside an asm statement that also
        // takes the running sums as in/out operands — so hipcc can neither hoist the loop-invariant weight reads out of the
        // row loop (64-128 registers) nor gather the unpack work of all pieces in front of one long fma chain (both happened:
        // 50-120 scratch accesses inside the load pipeline, each of them a vmcnt(0) wait).  The LDS round trip per piece is
        // exposed on purpose: this wave is the filler beside an MFMA wave, its loads are in flight either way.
        const unsigned wad = (unsigned)(size_t)(__attribute__((address_space(3))) float*)s_w + j * 32;
        auto piece = [&](const SV& v, auto off_tag) {
            constexpr int OFF = decltype(off_tag)::value;     // byte offset of the segment's weights inside s_w
            f32x4 wa, wb, wc, wd;
            if constexpr (CP == 2) {
                asm volatile("ds_read_b128 %0, %6 offset:%7\n\tds_read_b128 %1, %6 offset:%8\n\t"
                             "ds_read_b128 %2, %6 offset:%9\n\tds_read_b128 %3, %6 offset:%10\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(wa), "=&v"(wb), "=&v"(wc), "=&v"(wd), "+v"(a0), "+v"(a1)
                             : "v"(wad), "n"(OFF), "n"(OFF + 16), "n"(OFF + K * 4), "n"(OFF + K * 4 + 16) : "memory");
            } else {
                asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(wa), "=&v"(wb), "+v"(a0) : "v"(wad), "n"(OFF), "n"(OFF + 16) : "memory");
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) a0 = fmaf(v.at(e), wa[e], a0);
#pragma unroll
            for (int e = 0; e < 4; ++e) a0 = fmaf(v.at(4 + e), wb[e], a0);
            if constexpr (CP == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) a1 = fmaf(v.at(e), wc[e], a1);
#pragma unroll
                for (int e = 0; e < 4; ++e) a1 = fmaf(v.at(4 + e), wd[e], a1);
            }
        };
        auto consume_lo = [&](const SV (&v)[4]) {
            piece(v[0], std::integral_constant<int, 0 * 256>{}); piece(v[1], std::integral_constant<int, 1 * 256>{});
            piece(v[2], std::integral_constant<int, 2 * 256>{}); piece(v[3], std::integral_constant<int, 3 * 256>{});
        };
        auto consume_hi = [&](const SV (&v)[4]) {
            piece(v[0], std::integral_constant<int, 4 * 256>{}); piece(v[1], std::integral_constant<int, 5 * 256>{});
            piece(v[2], std::integral_constant<int, 6 * 256>{}); piece(v[3], std::integral_constant<int, 7 * 256>{});
        };
        // finish one 8-row group: lane reduction (3 xor-shuffles), bias, stores, running arg-max
        auto finish_rows = [&](int g, int r) {
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {
                a0 += __shfl_xor(a0, m, 64);
                if constexpr (CP == 2) a1 += __shfl_xor(a1, m, 64);
            }
            a0 += b0;
            a1 += b1;
            if (j == 0 && wbase + g * 8 + rr < Nb) {
                float* o = classes_out + (off0 + r) * (long long)CP;
                o[0] = a0;
                if constexpr (CP == 2) o[1] = a1;
            }
            // rows past the end were clamped to Nb-1: a duplicate can never beat itself (same index)
            if (a0 > bv0 || (a0 == bv0 && r < bi0)) { bv0 = a0; bi0 = r; }
            if (CP == 2 && (a1 > bv1 || (a1 == bv1 && r < bi1))) { bv1 = a1; bi1 = r; }
            a0 = 0.f;
            a1 = 0.f;
        };
        int r = row_of(0);
        const bf16_t* x = ptr_of(r);
#pragma unroll
        for (int u = 0; u < 4; ++u) va[u].load(x + u * SEG);
#pragma unroll 1
        for (int g = 0; g + 1 < ng; ++g) {
            const int rn = row_of(g + 1);
            const bf16_t* xn = ptr_of(rn);
#pragma unroll
            for (int u = 0; u < 4; ++u) vb[u].load(x + (4 + u) * SEG);
            consume_lo(va);
#pragma unroll
            for (int u = 0; u < 4; ++u) va[u].load(xn + u * SEG);
            consume_hi(vb);
            finish_rows(g, r);
            r = rn;
            x = xn;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) vb[u].load(x + (4 + u) * SEG);
        consume_lo(va);
        consume_hi(vb);
        finish_rows(ng - 1, r);
    }
    // best over the wave's 8 row lanes (lanes of one row agree), then over the 4 waves
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) {
        const float ov0 = __shfl_xor(bv0, m, 64);
        const int oi0 = __shfl_xor(bi0, m, 64);
        if (ov0 > bv0 || (ov0 == bv0 && oi0 < bi0)) { bv0 = ov0; bi0 = oi0; }
        if constexpr (CP == 2) {
            const float ov1 = __shfl_xor(bv1, m, 64);
            const int oi1 = __shfl_xor(bi1, m, 64);
            if (ov1 > bv1 || (ov1 == bv1 && oi1 < bi1)) { bv1 = ov1; bi1 = oi1; }
        }
    }
    if (lane == 0) { s_v[wave] = bv0; s_i[wave] = bi0; s_v[4 + wave] = bv1; s_i[4 + wave] = bi1; }
    __syncthreads();
    if (tid < CP) {
        const int h = tid * 4;
        float bv = s_v[h];
        int bi = s_i[h];
        for (int w = 1; w < 4; ++w)
            if (s_v[h + w] > bv || (s_v[h + w] == bv && s_i[h + w] < bi)) { bv = s_v[h + w]; bi = s_i[h + w]; }
        const long long slot = off0 / r0 + bag + tile;
        part_val[slot * CP + tid] = bv;
        part_idx[slot * CP + tid] = bi == 0x7fffffff ? 0x7fffffffffffffffLL : (long long)bi;
    }
}

// --------------------------------------------------------------------------------------------
// k_qmax: one workgroup per (bag, class).  Finishes the arg-max over the bag's tile partials
// (dsmil.py:52), then q_max = q(feats[idx]) (dsmil.py:53-54) on the VALU, 8 hidden units in
// flight per wave so the dependent shuffle chains overlap.
// --------------------------------------------------------------------------------------------
// One (bag, class) by a 256-thread workgroup; s_v[4], s_i[4], s_h[128]: LDS scratch.  Ends with
// every thread past its last LDS read (callers may reuse the scratch after a __syncthreads()).
// PRE (the in-launch producer of k_attend_hs: 8 waves, a 256-register budget): with K = 512 and the two-layer query, the
// wave's whole share of both weight matrices (128 + 32 registers a lane) is requested BEFORE the arg-max, so the chain
// arg-max -> row -> layer 1 -> layer 2 has one dependent load (the row) instead of five.  Same fmaf order either way.
template <int VEC, typename T = float, bool PRE = false>
__device__ __forceinline__ void qmax_block(
    const T* __restrict__ feats, const int64_t* __restrict__ offsets,
    const float* __restrict__ part_val, const long long* __restrict__ part_idx,
    const float* __restrict__ q0_w, const float* __restrict__ q0_b,
    const float* __restrict__ q2_w, const float* __restrict__ q2_b,
    float* __restrict__ qmax, int64_t* __restrict__ idx_out, int K, int C, int nonlinear,
    int bag, int c, float* s_v, long long* s_i, float* s_h, int mode = 0, float* __restrict__ best_val_out = nullptr,
    const int64_t* __restrict__ rowmap = nullptr, int r0 = R0) {
    // mode 0: arg-max + query of the critical row.  Instance-sharded bags (dsmil_agg_shard_*):
    // mode 1 = arg-max only (index and value out), mode 2 = query of a GIVEN row (feats = [C,K] rows)
    const long long off0 = mode == 2 ? 0 : offsets[bag];
    const long long Nb = mode == 2 ? C : offsets[bag + 1] - off0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwv = (int)blockDim.x >> 6;          // 16 waves: every wave owns 8 hidden units -> one load round per layer
    const int upw = QD / nwv;                      // hidden units per wave (multiple of 8)
    const long long slot0 = off0 / r0 + bag;
    const long long ntile = mode == 2 ? 0 : (Nb + r0 - 1) / r0;   // r0 = rows per workgroup of the logits kernel that ran
    f32x4 w1r[PRE ? 32 : 1];   // [jb 2][k0 2][u 8]
    float w2r[PRE ? 32 : 1];   // [jb 2][u 8][half 2]
    bool pre = false;
    float bv = -INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    long long t_next = threadIdx.x;
    if constexpr (PRE) {
        pre = K == 512 && nwv == 8 && nonlinear && VEC == 4;
        if (pre) {
            if (t_next < ntile) {   // the first round of tile partials goes out ahead of the weights (loads return in order)
                bv = part_val[(slot0 + t_next) * C + c];
                bi = part_idx[(slot0 + t_next) * C + c];
                t_next += blockDim.x;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        w1r[(jb * 2 + kk) * 8 + u] = *reinterpret_cast<const f32x4*>(q0_w + (long long)(wave * 16 + jb * 8 + u) * 512 + kk * 256 + lane * 4);
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* wr = q2_w + (long long)(wave * 16 + jb * 8 + u) * QD;
                    w2r[(jb * 8 + u) * 2] = wr[lane];
                    w2r[(jb * 8 + u) * 2 + 1] = wr[lane + 64];
                }
            __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the arg-max
        }
    }
    for (long long t = t_next; t < ntile; t += blockDim.x) {
        const float v = part_val[(slot0 + t) * C + c];
        const long long i = part_idx[(slot0 + t) * C + c];
        if (better(v, i, bv, bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const long long oi = __shfl_xor(bi, o, 64);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
    __syncthreads();
    bv = s_v[0]; bi = s_i[0];
    for (int w = 1; w < nwv; ++w)
        if (better(s_v[w], s_i[w], bv, bi)) { bv = s_v[w]; bi = s_i[w]; }
    long long best = mode == 2 ? c : bi;
    if (best < 0 || best >= Nb) best = 0;  // all-NaN guard: stay in bounds
    if (threadIdx.x == 0 && mode != 2) {
        idx_out[(long long)bag * C + c] = best;
        if (best_val_out) best_val_out[(long long)bag * C + c] = bv;
    }
    if (mode == 1) return;
    const T* x = feats + (mode == 2 ? best : phys_row(rowmap, off0 + best)) * (long long)K;
    if (PRE && pre) {
        const f32x4 xv0 = load4<VEC, T>(x, lane * 4, K), xv1 = load4<VEC, T>(x, 256 + lane * 4, K);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const f32x4 xv = kk ? xv1 : xv0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const f32x4 wv = w1r[(jb * 2 + kk) * 8 + u];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[u] = fmaf(xv[e], wv[e], acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float a = fmaxf(wave_sum(acc[u]) + q0_b[wave * 16 + jb * 8 + u], 0.f);
                if (lane == 0) s_h[wave * 16 + jb * 8 + u] = a;
            }
        }
        __syncthreads();
        float* out = qmax + ((long long)bag * C + c) * QD;
        const float h0 = s_h[lane], h1 = s_h[lane + 64];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            float acc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = fmaf(h0, w2r[(jb * 8 + u) * 2], h1 * w2r[(jb * 8 + u) * 2 + 1]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float a = wave_sum(acc[u]) + q2_b[wave * 16 + jb * 8 + u];
                if (lane == 0) out[wave * 16 + jb * 8 + u] = tanhf(a);
            }
        }
        return;
    }
    // layer 1: wave w computes hidden units upw*w .. upw*w + upw - 1, 8 at a time; lanes stride k by 4
    for (int jb = 0; jb < upw; jb += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* wr = q0_w + (long long)(wave * upw + jb) * K;
        for (int k0 = 0; k0 < K; k0 += 256) {
            const int k = k0 + lane * 4;
            const f32x4 xv = load4<VEC, T>(x, k, K);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 wv = load4<VEC>(wr + (long long)u * K, k, K);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u] = fmaf(xv[e], wv[e], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float a = wave_sum(acc[u]) + q0_b[wave * upw + jb + u];
            if (nonlinear) a = fmaxf(a, 0.f);
            if (lane == 0) s_h[wave * upw + jb + u] = a;
        }
    }
    __syncthreads();
    float* out = qmax + ((long long)bag * C + c) * QD;
    if (!nonlinear) {
        if (threadIdx.x < QD) out[threadIdx.x] = s_h[threadIdx.x];
        return;
    }
    const float h0 = s_h[lane], h1 = s_h[lane + 64];
    for (int jb = 0; jb < upw; jb += 8) {
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* wr = q2_w + (long long)(wave * upw + jb + u) * QD;
            acc[u] = fmaf(h0, wr[lane], h1 * wr[lane + 64]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float a = wave_sum(acc[u]) + q2_b[wave * upw + jb + u];
            if (lane == 0) out[wave * upw + jb + u] = tanhf(a);
        }
    }
}

constexpr int QMAX_T = 1024;   // 16 waves per (bag, class): the block is a chain of dependent load rounds, so width pays
template <int VEC, typename T = float>
__global__ __launch_bounds__(QMAX_T) void k_qmax(
    const T* __restrict__ feats, const int64_t* __restrict__ offsets,
    const float* __restrict__ part_val, const long long* __restrict__ part_idx,
    const float* __restrict__ q0_w, const float* __restrict__ q0_b,
    const float* __restrict__ q2_w, const float* __restrict__ q2_b,
    float* __restrict__ qmax, int64_t* __restrict__ idx_out, int K, int C, int nonlinear, int bag0,
    int mode = 0, float* __restrict__ best_val_out = nullptr, const int64_t* __restrict__ rowmap = nullptr, int r0 = R0) {
    __shared__ float s_v[QMAX_T / 64];
    __shared__ long long s_i[QMAX_T / 64];
    __shared__ float s_h[QD];
    qmax_block<VEC, T>(feats, offsets, part_val, part_idx, q0_w, q0_b, q2_w, q2_b, qmax, idx_out, K, C, nonlinear,
                       bag0 + (int)blockIdx.x, (int)blockIdx.y, s_v, s_i, s_h, mode, best_val_out, rowmap, r0);
}

template <int NW, int VEC>
__device__ __forceinline__ void attend_tile(const AttendArgs& a, int bag, int tile, float* smem) {
    constexpr int BM = NW * 32;
    f32x16 H[4], Q[4];
    if (!mlp_tile<NW, VEC>(a, bag, tile, smem, H, Q)) return;
    const long long off0 = a.offsets[bag];
    const long long Nb = a.offsets[bag + 1] - off0;
    const long long row0 = (long long)tile * BM;
    const long long slot = off0 / BM + bag + tile;
    if (DSMIL_EXPT_ON(a, 4)) {  // experiment builds: stop after the MLP, keep the accumulators live
        float keep = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += Q[t][r];
        if (keep == 12345.678f) a.scores[0] = keep;
        return;
    }
    attend_tail<NW, VEC, float>(a, Q, smem, bag, off0, Nb, row0, slot);
}

template <int NW, int VEC>
__global__ __launch_bounds__(NW * 64, (NW == 1 ? 1 : 2)) void k_query_attend(AttendArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    attend_tile<NW, VEC>(a, a.bag0 + (int)blockIdx.y, (int)blockIdx.x, smem);
}

// fp32 in / fp32 out with the query MLP on bf16 MFMA over exact three-plane cuts (agg_split.h)
template <int NW, int VEC, int NP, bool XE = false, int TU = 8>
__global__ __launch_bounds__(NW * 64, (NW == 1 ? 1 : 2)) void k_query_attend_split(AttendArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = NW * 32;
    const int bag = a.bag0 + (int)blockIdx.y, tile = (int)blockIdx.x;
    f32x16 Q[4];
    if constexpr (VEC == 4) {
        if (!mlp_tile_split_dma<NW, NP, XE>(a, bag, tile, smem, Q)) return;
    } else {
        if (!mlp_tile_split<NW, VEC, NP>(a, bag, tile, smem, Q)) return;
    }
    const long long off0 = a.offsets[bag];
    const long long Nb = a.offsets[bag + 1] - off0;
    if (DSMIL_EXPT_ON(a, 4)) {  // experiment builds: stop after the MLP, keep the accumulators live
        float keep = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += Q[t][r];
        if (keep == 12345.678f) a.scores[0] = keep;
        return;
    }
    attend_tail<NW, VEC, float, TU>(a, Q, smem, bag, off0, Nb, (long long)tile * BM, off0 / BM + bag + tile);
}

// --------------------------------------------------------------------------------------------
// k_query_attend_bf16 — BASELINE config 2: bf16 storage (features + query weights), f32
// accumulate / softmax.  v_mfma_f32_32x32x16_bf16, same transposed chain as the fp32 kernel:
// the ReLU'd H^T accumulators are rounded to bf16 and fed back as the B operand; W2 is packed
// with its k axis permuted so that MFMA step (t, s) contracts exactly the hidden units that
// accumulator registers 8s..8s+7 of tile t hold:  W2p[j][32t+16s+8hi+e] = W2[j][32t+16s+(e&3)+8(e>>2)+4hi].
// 64 bf16 (128 B) per staged row, so LDS geometry and fragment addressing equal the fp32 kernel's.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) { return pack_bf16x2_hw(lo, hi); }

template <int NW>
__global__ __launch_bounds__(NW * 64, (NW == 1 ? 1 : 2)) void k_query_attend_bf16(AttendArgs a) {
    constexpr int T = NW * 64;
    constexpr int BM = NW * 32;
    constexpr int X_TILE = BM * LDK;
    constexpr int WPT = (QD * 8) / T;
    constexpr int XPT = (BM * 8) / T;
    constexpr int BKH = 64;  // bf16 elements per staged chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW = smem;
    float* sX = smem + 2 * W_TILE;
    const int bag = a.bag0 + blockIdx.y;
    const long long off0 = a.offsets[bag];
    const long long Nb = a.offsets[bag + 1] - off0;
    const long long row0 = (long long)blockIdx.x * BM;
    if (row0 >= Nb) return;
    const long long slot = off0 / BM + bag + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int K = a.K;
    const int K64 = (K + BKH - 1) / BKH * BKH;
    const int nk1 = K64 / BKH;
    const int nk = nk1 + (a.nonlinear ? QD / BKH : 0);
    const bf16_t* feats = reinterpret_cast<const bf16_t*>(a.feats);
    const bf16_t* w1p = a.wpk;
    const bf16_t* w2p = a.wpk + (long long)QD * K64;

    f32x4 wreg[WPT], xreg[XPT];
    auto stage_load = [&](int ci) {
        const bf16_t* wb;
        int ld, k0;
        if (ci < nk1) { wb = w1p; ld = K64; k0 = ci * BKH; }
        else { wb = w2p; ld = QD; k0 = (ci - nk1) * BKH; }
        const int k = k0 + (tid & 7) * 8;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int r = (tid + T * i) >> 3;
            wreg[i] = *reinterpret_cast<const f32x4*>(wb + (long long)r * ld + k);
        }
        if (ci < nk1) {
            const int kc = (k + 8 <= K) ? k : K - 8;  // beyond K the packed weights are zero
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int r = (tid + T * i) >> 3;
                long long gr = row0 + r;
                if (gr >= Nb) gr = Nb - 1;
                xreg[i] = *reinterpret_cast<const f32x4*>(feats + phys_row(a.rowmap, off0 + gr) * (long long)K + kc);
            }
        }
    };
    auto stage_write = [&](int ci) {
        float* w = sW + (ci & 1) * W_TILE;
        const int c4 = tid & 7;
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            *reinterpret_cast<f32x4*>(w + ((tid + T * i) >> 3) * LDK + c4 * 4) = wreg[i];
        if (ci < nk1) {
            float* x = sX + (ci & 1) * X_TILE;
#pragma unroll
            for (int i = 0; i < XPT; ++i)
                *reinterpret_cast<f32x4*>(x + ((tid + T * i) >> 3) * LDK + c4 * 4) = xreg[i];
        }
    };

    f32x16 H[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) H[t][r] = 0.f;
    stage_load(0);
    stage_write(0);
    __syncthreads();
    const int frag = l31 * LDK + 4 * hi;
    for (int ci = 0; ci < nk1; ++ci) {
        if (ci + 1 < nk) stage_load(ci + 1);
        const float* w = sW + (ci & 1) * W_TILE + frag;
        const float* x = sX + (ci & 1) * X_TILE + wave * 32 * LDK + frag;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 xb = *reinterpret_cast<const bf16x8*>(x + ks * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 wa = *reinterpret_cast<const bf16x8*>(w + t * 32 * LDK + ks * 8);
                H[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, H[t], 0, 0, 0);
            }
        }
        if (ci + 1 < nk) stage_write(ci + 1);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.q0_b + 32 * t + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = H[t][4 * g + e] + b[e];
                H[t][4 * g + e] = a.nonlinear ? fmaxf(v, 0.f) : v;
            }
        }
    f32x16 Q[4];
    if (a.nonlinear) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) Q[t][r] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int ci = nk1 + c2;
            if (c2 < 1) stage_load(ci + 1);
            const float* w = sW + (ci & 1) * W_TILE + frag;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * c2 + tt;
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
                    union { unsigned u[4]; bf16x8 v; } hb;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        hb.u[e] = pack_bf16x2(H[t][8 * sidx + 2 * e], H[t][8 * sidx + 2 * e + 1]);
#pragma unroll
                    for (int t2 = 0; t2 < 4; ++t2) {
                        const bf16x8 wa = *reinterpret_cast<const bf16x8*>(w + t2 * 32 * LDK + tt * 16 + sidx * 8);
                        Q[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, hb.v, Q[t2], 0, 0, 0);
                    }
                }
            }
            if (c2 < 1) stage_write(ci + 1);
            __syncthreads();
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(a.q2_b + 32 * t + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) Q[t][4 * g + e] = fast_tanh(Q[t][4 * g + e] + b[e]);
            }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) Q[t] = H[t];
    }
    attend_tail<NW, 4, bf16_t>(a, Q, smem, bag, off0, Nb, row0, slot);
}

// --------------------------------------------------------------------------------------------
// k_query_attend_bf16_dma — the bf16-storage query MLP on the LDS-DMA pipeline of the split kernel (128-row
// workgroups).  Both operands go global -> LDS by global_load_lds_dwordx4, no staging registers, no ds_write pass:
//   weights   one 16 KiB chunk per 64-k step in MFMA-fragment order ([ks][t][lane] x 16 B: a straight copy, every
//             fragment read a contiguous conflict-free ds_read_b128), 2 LDS buffers, issued one step ahead (L2-resident);
//   features  each wave stages its own 32 rows: 4 pieces of 8 rows x 128 B (64 bf16) per step, 3 LDS buffers, issued
//             TWO steps ahead (HBM-sourced); the 16-B slot of a row is permuted on the source side (slot c of row r
//             holds global slot c ^ f(r)), which makes the fragment reads conflict-free.
// A step is 16 MFMAs per wave; the kernel is bound by the feature stream, not by the matrix pipe.  Completion is
// counted by hand (vmcnt + raw s_barrier), see agg_split.h.  GEMM 2 as in k_query_attend_bf16.
// --------------------------------------------------------------------------------------------
constexpr int BD_WCHUNK_F4 = 4 * 4 * 64;      // float4 (16 B) per weight chunk = 16 KiB
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_query_attend_bf16_dma(AttendArgs a) {
    static_assert(NW == 4, "128-row workgroups");
    constexpr int BM = NW * 32;
    constexpr int X_BUF_F4 = BM * 8;          // float4 per feature buffer (128 B per row)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* sW = reinterpret_cast<f32x4*>(smem);                 // [2][BD_WCHUNK_F4]
    f32x4* sX = sW + 2 * BD_WCHUNK_F4;                          // [3][X_BUF_F4]
    const int bag = a.bag0 + blockIdx.y;
    const long long off0 = a.offsets[bag];
    const long long Nb = a.offsets[bag + 1] - off0;
    const long long row0 = (long long)blockIdx.x * BM;
    if (row0 >= Nb) return;
    const long long slot = off0 / BM + bag + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int K = a.K;
    const int nk1 = (K + 63) / 64;
    const int nst = nk1 + (a.nonlinear ? 2 : 0);
    const bf16_t* feats = reinterpret_cast<const bf16_t*>(a.feats);
    const f32x4* wpk = reinterpret_cast<const f32x4*>(a.wpk);  // chunk-major fragment layout (k_pack_agg_bf16)

    const bf16_t* xsrc[4];
    int xslot[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = p * 8 + (lane >> 3);
        long long gr = row0 + wave * 32 + r;
        if (gr >= Nb) gr = Nb - 1;                               // rows past the bag end are masked in attend_tail
        xsrc[p] = feats + phys_row(a.rowmap, off0 + gr) * (long long)K;
        xslot[p] = ((lane & 7) ^ ((r & 6) | ((r >> 4) & 1))) * 8;   // bf16 elements
    }
    auto issue_w = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * NW + wave;                         // 16 pieces of 1 KiB per chunk
            __builtin_amdgcn_global_load_lds((const DSMIL_GLOBAL void*)(wpk + (long long)s * BD_WCHUNK_F4 + q * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(sW + (s & 1) * BD_WCHUNK_F4 + q * 64), 16, 0, 0);
        }
    };
    auto issue_x = [&](int s) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int k = s * 64 + xslot[p];
            k = k + 8 <= K ? k : K - 8;                          // past K the packed weights are zero: any finite data will do
            __builtin_amdgcn_global_load_lds((const DSMIL_GLOBAL void*)(xsrc[p] + k),
                                             (__attribute__((address_space(3))) void*)(sX + (s % 3) * X_BUF_F4 + (wave * 32 + p * 8) * 8), 16, 0, 0);
        }
    };
    const int fr = (l31 & 6) | ((l31 >> 4) & 1);

    f32x16 H[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) H[t][r] = 0.f;

    issue_w(0);
    issue_x(0);
    if (nk1 > 1) { issue_x(1); S3_WAIT_VM(4); } else { S3_WAIT_VM(0); }
    __builtin_amdgcn_s_barrier();
    // ---- GEMM 1 (transposed): H^T[j][n] += W1[j][k] X[n][k], 64 k per step
    for (int s = 0; s < nk1; ++s) {
        const bool more_w = s + 1 < nst, more_x = s + 2 < nk1;    // block-uniform
        if (more_w) issue_w(s + 1);
        if (more_x) issue_x(s + 2);
        const f32x4* w = sW + (s & 1) * BD_WCHUNK_F4 + lane;
        const f32x4* x = sX + (s % 3) * X_BUF_F4 + (wave * 32 + l31) * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            union { f32x4 f; bf16x8 v; } xb, wa;
            xb.f = x[(ks * 2 + hi) ^ fr];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wa.f = w[(ks * 4 + t) * 64];
                H[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa.v, xb.v, H[t], 0, 0, 0);
            }
        }
        // W(s+1) and this wave's X(s+1) have landed; X(s+2) (the youngest pieces) may stay in flight
        if (more_x) S3_WAIT_VM(4); else S3_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.q0_b + 32 * t + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = H[t][4 * g + e] + b[e];
                H[t][4 * g + e] = a.nonlinear ? fmaxf(v, 0.f) : v;
            }
        }
    f32x16 Q[4];
    if (a.nonlinear) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) Q[t][r] = 0.f;
        // ---- GEMM 2 (transposed): two 64-k chunks; step (tt, sidx) of chunk c2 contracts the 16 hidden units that
        //      accumulator registers 8 sidx .. 8 sidx + 7 of H[2 c2 + tt] hold (packed W2 carries the k permutation)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int s = nk1 + c2;
            if (c2 == 0) issue_w(s + 1);
            const f32x4* w = sW + (s & 1) * BD_WCHUNK_F4 + lane;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * c2 + tt;
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
                    union { unsigned u[4]; bf16x8 v; } hb;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        hb.u[e] = pack_bf16x2(H[t][8 * sidx + 2 * e], H[t][8 * sidx + 2 * e + 1]);
#pragma unroll
                    for (int t2 = 0; t2 < 4; ++t2) {
                        union { f32x4 f; bf16x8 v; } wa;
                        wa.f = w[((tt * 2 + sidx) * 4 + t2) * 64];
                        Q[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa.v, hb.v, Q[t2], 0, 0, 0);
                    }
                }
            }
            S3_WAIT_VM(0);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(a.q2_b + 32 * t + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) Q[t][4 * g + e] = fast_tanh(Q[t][4 * g + e] + b[e]);
            }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) Q[t] = H[t];
    }
    attend_tail<NW, 4, bf16_t>(a, Q, smem, bag, off0, Nb, row0, slot);
}

// W1 [128,K] fp32 -> bf16 [128,K64] zero padded; W2 [128,128] fp32 -> bf16 with the k permutation
// described above.  RNE rounding (== torch .bfloat16()).
// Behind that row-major image (read by the register-staged kernel) the same weights follow in the chunk-major
// MFMA-fragment order of k_query_attend_bf16_dma: chunk s < K64/64: [ks][t][lane (l31,hi)][e] = W1[32t+l31][64s+16ks+8hi+e];
// chunk K64/64 + c2: [ks = 2tt + sidx][t2][lane][e] = W2[32 t2 + l31][32 (2 c2 + tt) + 16 sidx + (e&3) + 8(e>>2) + 4hi].
__global__ void k_pack_agg_bf16(const float* __restrict__ q0_w, const float* __restrict__ q2_w,
                                bf16_t* __restrict__ out, int K, int K64) {
    const int n1 = QD * K64;
    const int nrow = n1 + QD * QD;                       // row-major image (W2 part unused when !q2_w)
    const int nfrag = (K64 / 64 + 2) * BD_WCHUNK_F4 * 8; // fragment image
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nrow + nfrag; i += gridDim.x * blockDim.x) {
        if (i < n1) {
            const int j = i / K64, k = i - j * K64;
            out[i] = k < K ? f2bf(q0_w[(long long)j * K + k]) : (bf16_t)0;
        } else if (i < nrow) {
            const int q = i - n1, j = q / QD, kk = q - j * QD;
            const int base = kk & ~15, r = kk & 15, hi = r >> 3, e = r & 7;
            out[i] = q2_w ? f2bf(q2_w[j * QD + base + (e & 3) + 8 * (e >> 2) + 4 * hi]) : (bf16_t)0;
        } else {
            int r = i - nrow;
            const int e = r & 7; r >>= 3;
            const int lane = r & 63; r >>= 6;
            const int t = r & 3; r >>= 2;
            const int ks = r & 3; r >>= 2;
            const int s = r, l31 = lane & 31, hi = lane >> 5;
            float v = 0.f;
            if (s < K64 / 64) {
                const int k = 64 * s + 16 * ks + 8 * hi + e;
                if (k < K) v = q0_w[(long long)(32 * t + l31) * K + k];
            } else if (q2_w) {
                const int c2 = s - K64 / 64, tt = ks >> 1, sidx = ks & 1;
                v = q2_w[(32 * t + l31) * QD + 32 * (2 * c2 + tt) + 16 * sidx + (e & 3) + 8 * (e >> 2) + 4 * hi];
            }
            out[i] = f2bf(v);
        }
    }
}

// The producer half of the in-launch hand-off (MI355X_MICROARCH.md, "valid forms"): the workgroup's plain stores are out
// (the caller's __syncthreads()), ONE lane writes the XCD's L2 back with an agent-scope release, waits for the write-back
// with an s_waitcnt hipcc cannot drop (it elides its own when the wave's vmcnt scoreboard is provably empty, and the flag
// then overtakes the data), and only then sets the flag with a relaxed agent-scope store.
__device__ __forceinline__ void publish_flag(int* flag) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// few rows (a lone bag, a training step): the hidden units of a 32-row tile split over the four SIMDs of a CU (agg_hs.h)
template <int NP>
__global__ __launch_bounds__(HS_THREADS, 2) void k_attend_hs(AttendArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bag = a.bag0 + (int)blockIdx.y;
    int tile = (int)blockIdx.x;
    if (a.qm_flag) {   // the first C workgroups of the grid row: the critical row's query, beside the tiles' MLP
        if (tile < a.C) {
            if (a.offsets[bag + 1] <= a.offsets[bag]) {   // an empty bag has no tiles waiting and no row to read
                if (threadIdx.x == 0) {
                    a.qm_idx[(long long)bag * a.C + tile] = 0;
                    publish_flag(a.qm_flag + (long long)bag * a.C + tile);
                }
                return;
            }
            long long* s_i = reinterpret_cast<long long*>(smem);
            float* s_v = smem + 2 * (HS_THREADS / 64), *s_h = s_v + HS_THREADS / 64;
            qmax_block<4, float, true>(reinterpret_cast<const float*>(a.feats), a.offsets, a.qm_part_val, a.qm_part_idx, a.q0_w, a.q0_b,
                                 a.q2_w, a.q2_b, const_cast<float*>(a.qmax), a.qm_idx, a.K, a.C, a.nonlinear, bag, tile, s_v, s_i,
                                 s_h, 0, nullptr, a.rowmap, a.qm_r0);
            __syncthreads();   // every wave's stores are out (vmcnt 0) before the release below writes the L2 back
            if (threadIdx.x == 0) publish_flag(a.qm_flag + (long long)bag * a.C + tile);
            return;
        }
        tile -= a.C;
    }
    f32x16 Hw[HS_RG], Qw[HS_RG];
    if (!mlp_tile_hs<NP>(a, bag, tile, smem, Hw, Qw)) return;
    const long long off0 = a.offsets[bag];
    const long long Nb = a.offsets[bag + 1] - off0;
    attend_tail_hs<float>(a, Qw, smem + HS_STAGE + HS_PLANES, smem, bag, off0, Nb, (long long)tile * HS_BM, off0 / HS_BM + bag + tile);
#ifdef DSMIL_TRACE
    if (DSMIL_EXPT_ON(a, 64) && threadIdx.x == 0 && a.C == 1)
        reinterpret_cast<unsigned long long*>(a.scores + (off0 + (long long)tile * HS_BM) * (long long)a.C)[4] = __builtin_readcyclecounter();
#endif
}

// --------------------------------------------------------------------------------------------
// k_finish: per bag, combine tile partials (online-softmax merge), normalise A in place,
// produce B (dsmil.py:57-59) and pred = Conv1d(C,C,Kv)(B) (dsmil.py:60-61).
// grid = (nblk, n_bags), nblk = max(ceil(max_rows/FR), ceil(Kv/64)): block j normalises an even
// share of the bag's rows and owns the 64-wide k-run j of B (none when j >= ceil(Kv/64)).  The tile
// walk of a k-run is spread over 16 thread groups x float4 so that a lone 10k-row bag (313 tiles)
// is ~5 dependent load rounds instead of ~80.
// --------------------------------------------------------------------------------------------
// k_tile_prefix: pre[b] = sum over the bags in front of b of ceil(N / BM), pre[n_bags] = all tiles — the item list of the
// persistent batch kernels (AttendArgs::tile_pre, tiles of BMa rows) and of the logits pass (tiles of BMb rows) on a RAGGED
// batch.  One 1024-thread workgroup, rounds of 1024 bags.
__global__ __launch_bounds__(1024) void k_tile_prefix(const int64_t* __restrict__ offsets, int n_bags, int BMa, int BMb,
                                                      int* __restrict__ pre_a, int* __restrict__ pre_b) {
    __shared__ int s_w[2][16];
    __shared__ int s_base[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2) s_base[tid] = 0;
    __syncthreads();
    for (int b0 = 0; b0 < n_bags; b0 += 1024) {
        const int b = b0 + tid;
        int v[2] = {0, 0}, x[2];
        if (b < n_bags) {
            const long long n = offsets[b + 1] - offsets[b];
            v[0] = (int)((n + BMa - 1) / BMa);
            v[1] = (int)((n + BMb - 1) / BMb);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            x[q] = v[q];                                      // inclusive scan inside the wave
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(x[q], d);
                if (lane >= d) x[q] += y;
            }
            if (lane == 63) s_w[q][wave] = x[q];
        }
        __syncthreads();
        int before[2] = {s_base[0], s_base[1]};
        for (int w = 0; w < wave; ++w) { before[0] += s_w[0][w]; before[1] += s_w[1][w]; }
        if (b < n_bags) { pre_a[b] = before[0] + x[0] - v[0]; pre_b[b] = before[1] + x[1] - v[1]; }
        __syncthreads();
        if (tid == 1023) { s_base[0] = before[0] + x[0]; s_base[1] = before[1] + x[1]; }
        __syncthreads();
    }
    if (tid == 0) { pre_a[n_bags] = s_base[0]; pre_b[n_bags] = s_base[1]; }
}

constexpr int FR = 2048;
inline long long finish_blocks(long long max_rows, int Kv) {
    const long long a = (max_rows + FR - 1) / FR, b = (Kv + 63) / 64;
    return a > b ? (a > 1 ? a : 1) : b;
}

// UB: tiles in flight per thread group of the B walk.  UB = 2 is the LEAN form of the bf16 batch path (round 6): <= 80
// registers, so that it runs beside a resident k_attend_bf16_res workgroup of another stream like k_logits_pipe does (its bags
// have <= ~3 partials each; same summation order: a thread group walks its tiles in sequence whatever the unroll).
template <int VEC, int UB = 8>
__global__ __launch_bounds__(256, (UB == 8 ? 1 : 6)) void k_finish(
    const int64_t* __restrict__ offsets, const float* __restrict__ part_ml,
    const float* __restrict__ part_B, const float* __restrict__ fcc_w,
    float* __restrict__ A, float* __restrict__ B, float* __restrict__ pred_part, int Kv, int C, int BM,
    float* __restrict__ ml_out = nullptr, int seg_per = 0, int seg_T = 0, const int* __restrict__ tile_pre = nullptr) {
    // seg_per > 0 (k_attend_bf16_res): the partials are per (workgroup, bag) — workgroup g owns the BM-row tile items
    // [g seg_per, (g + 1) seg_per) of the (bag, tile) list with seg_T items per bag, and wrote slot g + bag
    // ml_out != null (instance-sharded bag): leave A and B relative to this shard's max, un-normalised
    // (A = exp(s - m), B = sum exp(s - m) V) and hand (m, l) per class to the caller's cross-shard merge
    const int bag = blockIdx.y, nblk = gridDim.x;
    const long long off0 = offsets[bag];
    const long long Nb = offsets[bag + 1] - off0;
    long long slot0 = off0 / BM + bag;
    long long ntile = (Nb + BM - 1) / BM;
    if (seg_per > 0 && ntile > 0) {   // (an EMPTY bag keeps ntile = 0: no workgroup wrote a partial for it — merging a slot
                                      // would read stale workspace; the bag then gets the same NaN / 0 outputs as on the tile path)
        // (ragged batches: the items are the real tiles, tile_pre[bag] of them in front of this bag — AttendArgs::tile_pre)
        const long long it0 = tile_pre ? (long long)tile_pre[bag] : (long long)bag * seg_T, g_lo = it0 / seg_per, g_hi = (it0 + ntile - 1) / seg_per;
        slot0 = g_lo + bag;
        ntile = g_hi - g_lo + 1;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float s_red[8];
    __shared__ __attribute__((aligned(16))) float s_acc[16][64];
    const int kb = (int)blockIdx.x * 64;           // this block's k-run
    const bool has_k = kb < Kv;
    long long rpb = (Nb + nblk - 1) / nblk;        // this block's share of the rows: an even split, but at least 256 (a short bag
    rpb = rpb < 256 ? 256 : rpb;                   // of a ragged batch keeps its rows in few blocks; the others leave at once)
    const long long rbeg = (long long)blockIdx.x * rpb;
    const long long rend = (rbeg + rpb < Nb) ? rbeg + rpb : Nb;
    const int kq = tid & 15, tg = tid >> 4;
    constexpr int UA = UB == 8 ? 8 : 4;            // rows per thread in flight of the A pass
    if (!has_k && rbeg >= rend && !ml_out) {   // a block of the grid (sized for the LONGEST bag) with neither rows nor a k-run of this bag
        if (tid < C * C) pred_part[(((long long)bag * nblk + blockIdx.x) * C) * C + tid] = 0.f;
        return;
    }
    for (int c = 0; c < C; ++c) {
        // global max / sum of the bag
        float m = -INFINITY;
        for (long long t = tid; t < ntile; t += 256) m = fmaxf(m, part_ml[((slot0 + t) * C + c) * 2]);
        m = wave_max(m);
        __syncthreads();
        if (lane == 0) s_red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        float l = 0.f;
        for (long long t = tid; t < ntile; t += 256) {
            const float* ml = part_ml + ((slot0 + t) * C + c) * 2;
            l += ml[1] * expf(ml[0] - m);
        }
        l = wave_sum(l);
        if (lane == 0) s_red[4 + wave] = l;
        __syncthreads();
        l = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
        const float il = ml_out ? 1.f : 1.f / l;
        if (ml_out && blockIdx.x == 0 && tid == 0) { ml_out[((long long)bag * C + c) * 2] = m; ml_out[((long long)bag * C + c) * 2 + 1] = l; }
        // A = exp(s - m) / l for this block's rows
        // (eight rows per thread in flight: as `*p = f(*p)` in a loop every load waits for the store before it — a block's
        // share is at most FR = 2048 rows, i.e. ONE round of loads instead of up to eight dependent round trips)
        for (long long r0 = rbeg + tid; r0 < rend; r0 += 256 * UA) {
            float sv[UA];
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const long long r = r0 + 256 * u;
                sv[u] = A[(off0 + (r < rend ? r : rend - 1)) * (long long)C + c];
            }
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const long long r = r0 + 256 * u;
                if (r < rend) A[(off0 + r) * (long long)C + c] = expf(sv[u] - m) * il;
            }
        }
        float* pp = pred_part + (((long long)bag * nblk + blockIdx.x) * C) * C + c;  // [o] stride C
        if (!has_k) {
            if (tid < C) pp[tid * C] = 0.f;
            continue;
        }
        // B[c][kb..kb+63]: 16 thread groups walk the tiles, 16 lanes x float4 cover the k-run
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* pb = part_B + (slot0 * C + c) * (long long)Kv;
        const float* pm = part_ml + (slot0 * C + c) * 2;
        // rounds of eight tiles, the last one padded with clamped re-reads that are not added (an unrolled run-time trip
        // count leaves a one-load-at-a-time remainder loop: 157 tiles / 16 groups = 2 rounds of 4 + 2 dependent round trips)
        for (long long t0 = tg; t0 < ntile; t0 += 16 * UB) {
            float wv[UB];
            f32x4 bv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                long long t = t0 + 16 * u;
                t = t < ntile ? t : ntile - 1;
                wv[u] = pm[t * C * 2];
                bv[u] = load4<VEC>(pb + t * C * (long long)Kv, kb + kq * 4, Kv);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (t0 + 16 * u < ntile) {
                    const float w = expf(wv[u] - m);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(w, bv[u][e], acc[e]);
                }
            }
        }
        *reinterpret_cast<f32x4*>(&s_acc[tg][kq * 4]) = acc;
        __syncthreads();
        if (wave == 0) {
            float b = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) b += s_acc[g][lane];
            b *= il;
            const int k = kb + lane;
            if (k < Kv) B[((long long)bag * C + c) * Kv + k] = b;
            // partial Conv1d dot products of this k-run: pred_part[bag][block][o][c]
            for (int o = 0; o < C; ++o) {
                const float d = wave_sum(k < Kv ? fcc_w[((long long)o * C + c) * Kv + k] * b : 0.f);
                if (lane == 0) pp[o * C] = d;
            }
        }
        __syncthreads();
    }
}

// pred[bag][o] = fcc_b[o] + sum_{block,c} pred_part  (fixed order => deterministic)
__global__ void k_pred(const float* __restrict__ pred_part, const float* __restrict__ fcc_b,
                       float* __restrict__ pred, int C, int nblk, int n_bags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bags * C) return;
    const int bag = i / C, o = i % C;
    float s = fcc_b[o];
    for (int j = 0; j < nblk; ++j)
        for (int c = 0; c < C; ++c) s += pred_part[(((long long)bag * nblk + j) * C + o) * C + c];
    pred[i] = s;
}

// FCLayer alone
template <int VEC>
__global__ __launch_bounds__(256) void k_fc(const float* __restrict__ feats,
                                            const float* __restrict__ fc_w,
                                            const float* __restrict__ fc_b,
                                            float* __restrict__ classes, long long N, int K, int C,
                                            const int64_t* __restrict__ rowmap) {
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    for (long long r = wid; r < N; r += nw) {
        const float* x = feats + phys_row(rowmap, r) * K;
        for (int c = 0; c < C; ++c) {
            float acc = 0.f;
            for (int k0 = 0; k0 < K; k0 += 256) {
                const int k = k0 + lane * 4;
                const f32x4 xv = load4<VEC>(x, k, K);
                const f32x4 wv = load4<VEC>(fc_w + (long long)c * K, k, K);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = fmaf(xv[e], wv[e], acc);
            }
            acc = wave_sum(acc) + fc_b[c];
            if (lane == 0) classes[r * C + c] = acc;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------
struct WsLayout {
    size_t part_val, part_idx, qmax, qflag, part_ml, part_B, pred_part, wsplit, wf2, rowmax, off2, tile_pre, total;
    long long slots0, slots, nchunk_max;
};
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// Experiment builds (-DDSMIL_EXPERIMENTS) read ablation / geometry knobs from the environment, once per
// process; the product build has none of them.
#ifdef DSMIL_EXPERIMENTS
int expt_env(const char* name) {
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
}
#endif

int pick_nw(int n_bags, long long total_rows) {
#ifdef DSMIL_EXPERIMENTS
    static const int force = expt_env("DSMIL_NW");
    if (force == 8 || force == 4 || force == 1) return force;
#endif
    // 128-row workgroups (4 waves) once they alone give >= 2 workgroups per CU; otherwise
    // 32-row single-wave workgroups so that a lone bag still spreads over the chip.
    const long long tiles128 = total_rows / 128 + n_bags;
    return tiles128 >= 512 ? 4 : 1;
}

inline size_t f2_image_bytes(int K) { return (size_t)(2 * ((K + 31) / 32) + 8) * F2_CHUNK_F4 * 16 + F2_TRAILER_BYTES; }

WsLayout ws_layout(int n_bags, long long total_rows, long long max_rows, int K, int Kv, int C, int BM) {
    WsLayout w;
    w.slots0 = total_rows / 32 + n_bags + 1;   // (32 = the smallest rows-per-workgroup of the logits kernels)
    const int bms = BM > F2_BM ? F2_BM : BM;   // (k_attend_f2 runs 64-row tiles in the 128-row regime)
    w.slots = total_rows / bms + n_bags + 1;
    if (w.slots < RS_MAX_WG + n_bags) w.slots = RS_MAX_WG + n_bags;   // k_attend_bf16_res: one slot per (workgroup, bag) pair, slot = workgroup + bag
    w.nchunk_max = finish_blocks(max_rows, Kv);
    size_t o = 0;
    w.part_val = o; o = al(o + (size_t)w.slots0 * C * sizeof(float));
    w.part_idx = o; o = al(o + (size_t)w.slots0 * C * sizeof(long long));
    w.qmax = o; o = al(o + (size_t)n_bags * C * QD * sizeof(float));
    w.qflag = o; o = al(o + (size_t)n_bags * C * sizeof(int));   // k_attend_hs: hand-off flags of the in-launch critical query
    w.part_ml = o; o = al(o + (size_t)w.slots * C * 2 * sizeof(float));
    w.part_B = o; o = al(o + (size_t)w.slots * C * Kv * sizeof(float));
    w.pred_part = o; o = al(o + (size_t)n_bags * w.nchunk_max * C * C * sizeof(float));
    w.wsplit = o; o = al(o + (size_t)(2 * ((K + 31) / 32) + 8) * S3_CHUNK_F4 * 16);  // cut query weights
    w.wf2 = o; o = al(o + f2_image_bytes(K));     // k_attend_f2: fp16 two-plane query weights (when the caller brought none)
    w.rowmax = o; o = al(o + (size_t)total_rows * sizeof(float));   // k_attend_f2: max |x| per row (k_logits_stream)
    w.off2 = o; o = al(o + 2 * sizeof(int64_t));  // {0, N} of a lone shard (dsmil_agg_shard_*)
    w.tile_pre = o; o = al(o + (size_t)2 * (n_bags + 1) * sizeof(int));   // ragged batches: tile prefixes of the persistent kernel and of the logits pass
    w.total = o;
    return w;
}

template <int NW, int VEC>
int launch_attend(const AttendArgs& a, long long max_rows, int n_bags, hipStream_t st) {
    constexpr int BM = NW * 32;
    const size_t lds = (size_t)(2 * W_TILE + 2 * BM * LDK) * sizeof(float);
    if (!dsmil_lds::allow((const void*)k_query_attend<NW, VEC>, (int)lds)) return DSMIL_E_LAUNCH;
    dim3 grid((unsigned)((max_rows + BM - 1) / BM), (unsigned)n_bags);
    const int slot = dsmil_prof::begin(dsmil_prof::CH_ATTEND, st);
    hipLaunchKernelGGL((k_query_attend<NW, VEC>), grid, dim3(NW * 64), lds, st, a);
    dsmil_prof::end(dsmil_prof::CH_ATTEND, slot, st);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

template <int NW, int VEC, int NP, bool XE = false, int TU = 8>
int launch_attend_split(const AttendArgs& a, long long max_rows, int n_bags, hipStream_t st) {
    constexpr int BM = NW * 32;
    size_t lds = VEC == 4 ? (size_t)(3 * S3_CHUNK_F4 * 4 + 2 * BM * 32) * sizeof(float)
                          : (size_t)(2 * S3_CHUNK_F4 * 4 + 2 * BM * LDK) * sizeof(float);
#ifdef DSMIL_EXPERIMENTS
    static const int lds_pad = expt_env("DSMIL_LDS_PAD");  // force 1 block/CU
    lds += (size_t)lds_pad;
#endif
    if (!dsmil_lds::allow((const void*)k_query_attend_split<NW, VEC, NP, XE, TU>, (int)lds)) return DSMIL_E_LAUNCH;
    dim3 grid((unsigned)((max_rows + BM - 1) / BM), (unsigned)n_bags);
    const int slot = dsmil_prof::begin(dsmil_prof::CH_ATTEND, st);
    hipLaunchKernelGGL((k_query_attend_split<NW, VEC, NP, XE, TU>), grid, dim3(NW * 64), lds, st, a);
    dsmil_prof::end(dsmil_prof::CH_ATTEND, slot, st);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

// Compute units of the current device (cached per device id; 256 on MI355X).
int device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// dsmil_agg_inline_query(): 1 = the critical row's query may run inside the k_attend_hs launch (default), 0 = always the
// separate k_qmax launch
std::atomic<int> g_inline_query{1};
// dsmil_agg_batch_form(): 2 (default) = batches of fp32 bags take k_attend_f3 (weights resident in registers, 32-row tiles;
// two-layer query and C <= 2 — k_attend_f2 otherwise), 1 = k_attend_f2 always, 0 = k_query_attend_split (rounds 2-4)
std::atomic<int> g_use_f2{2};
// dsmil_agg_persistent_grid(): workgroups of the persistent batch kernels (k_attend_f3 / k_attend_f2 / k_attend_bf16_res).
// A CONSTANT (256 = the CUs of an unpartitioned MI355X), not the visible CU count: the run length `per` fixes which tiles
// share a partial, i.e. the fp32 summation order, so the outputs must not depend on the box (partition mode, masked CUs).
std::atomic<int> g_persistent_grid{256};
// DSMIL_LOGITS_PIPE=0: the bf16 batch path keeps k_logits_stream (A/B of the co-resident logits pass, round 6)
// dsmil_agg_logits_form(): bf16 batches of K = 512 rows, C <= 2 have a second set of kernels around the persistent attend
// kernel — k_logits_pipe, a 4-wave k_qmax, the lean k_finish — that fit beside a resident k_attend_bf16_res workgroup of
// ANOTHER stream's batch.  2 = always, 0 = never (k_logits_stream, 16-wave k_qmax, k_finish: what every other shape takes),
// 1 (default) = when the library has recently been called on more than one stream: alone on the chip the co-resident
// kernels are ~6 % slower per pass (a narrower q_max launch, fewer bytes in flight per wave), beside another stream's attend
// kernel they make the pass 12-17 % faster.  Bit-identical outputs in every mode.
std::atomic<int> g_logits_form{1};
bool several_streams_recently(hipStream_t st) {   // the last four batch calls did not all come in on this stream
    static std::atomic<uintptr_t> ring[4];
    static std::atomic<unsigned> pos{0};
    const uintptr_t me = (uintptr_t)st + 1;       // (0 = empty slot; the null stream is a stream)
    bool several = false;
    for (auto& r : ring) {
        const uintptr_t v = r.load(std::memory_order_relaxed);
        several |= (v != 0 && v != me);
    }
    ring[pos.fetch_add(1, std::memory_order_relaxed) & 3].store(me, std::memory_order_relaxed);
    return several;
}
constexpr int PIPE_R0 = 512;   // rows per k_logits_pipe workgroup (workgroups enter a CU's one free slot one at a time)
constexpr int PIPE_QT = 256;   // threads of the k_qmax launch in front of a co-resident pass
bool coresident_wanted(hipStream_t st) {
    const int form = g_logits_form.load(std::memory_order_relaxed);
    const bool several = several_streams_recently(st);      // (always recorded, whatever the mode)
    return form == 2 || (form == 1 && several);
}
int persistent_grid(int cap) {
    int g = g_persistent_grid.load(std::memory_order_relaxed);
    if (g <= 0) g = 256;
    return g < cap ? g : cap;
}

// The in-launch hand-off needs its producers to RUN while tiles spin on their flags.  Producers are the first workgroups of
// every grid row and the hardware dispatches a grid in order, but HIP promises neither: the query is inlined only when
// every workgroup of the launch can be resident at once (k_attend_hs: 2 workgroups per CU by LDS and registers), so that
// progress does not depend on dispatch order at all.  Larger batches take the k_qmax launch.
bool hs_inline_fits(long long max_rows, int n_bags, int C) {
    const long long wgs = ((max_rows + HS_BM - 1) / HS_BM + C) * (long long)n_bags;
    return wgs <= 2LL * device_cus();
}

int launch_attend_hs(const AttendArgs& a, long long max_rows, int n_bags, hipStream_t st) {
    size_t lds = HS_LDS_BYTES;
#ifdef DSMIL_EXPERIMENTS
    static const int lds_pad = expt_env("DSMIL_LDS_PAD");  // force 1 block/CU
    lds += (size_t)lds_pad;
#endif
    if (!dsmil_lds::allow((const void*)k_attend_hs<6>, (int)lds)) return DSMIL_E_LAUNCH;
    dim3 grid((unsigned)((max_rows + HS_BM - 1) / HS_BM + (a.qm_flag ? a.C : 0)), (unsigned)n_bags);
    const int slot = dsmil_prof::begin(dsmil_prof::CH_ATTEND, st);
    hipLaunchKernelGGL(k_attend_hs<6>, grid, dim3(HS_THREADS), lds, st, a);
    dsmil_prof::end(dsmil_prof::CH_ATTEND, slot, st);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

// Which MFMA form the fp32 query MLP uses (see agg_split.h).  Default: 6 plane products — the three left
// out are together below 2^-20 of |x*w|, i.e. below the fp32 accumulation rounding the reference's own
// 512-long dot products carry (tests/accuracy_report.py: identical measured error for 0 / 9 / 6).
// Experiment builds (libdsmil_hip_expt.so): DSMIL_MLP = s9 | f32 selects the bit-exact-product forms, read once per
// process; the product library has the one form.
// batches of fp32 bags: resident 64-row tiles, fp16 two-plane MFMA (agg_f2.h)
int launch_attend_f2(const AttendArgs& a, const float* rowmax, long long max_rows, int n_bags, hipStream_t st) {
    void (*fn)(AttendArgs, const float*, int, int) = nullptr;
    switch (a.K / 32) {
        case 4: fn = k_attend_f2<4>; break;
        case 8: fn = k_attend_f2<8>; break;
        case 12: fn = k_attend_f2<12>; break;
        case 16: fn = k_attend_f2<16>; break;
        default: return DSMIL_E_UNSUPPORTED;
    }
#ifdef DSMIL_EXPERIMENTS   // timing-only ablations of the K = 512 form (tools/f2_ablate.py)
    static const int abl = expt_env("DSMIL_F2_ABL");
    switch (a.K == 512 ? abl : 0) {
        case 1: fn = k_attend_f2<16, 1>; break;
        case 2: fn = k_attend_f2<16, 2>; break;
        case 3: fn = k_attend_f2<16, 3>; break;
        case 4: fn = k_attend_f2<16, 4>; break;
        case 7: fn = k_attend_f2<16, 7>; break;
        case 8: fn = k_attend_f2<16, 8>; break;
        case 16: fn = k_attend_f2<16, 16>; break;
        case 31: fn = k_attend_f2<16, 31>; break;
        case 32: fn = k_attend_f2<16, 32>; break;
        case 64: fn = k_attend_f2<16, 64>; break;
        case 128: fn = k_attend_f2<16, 128>; break;
        case 192: fn = k_attend_f2<16, 192>; break;
        case 256: fn = k_attend_f2<16, 256>; break;
        default: break;
    }
#endif
    if (!dsmil_lds::allow((const void*)fn, F2_LDS_BYTES)) return DSMIL_E_LAUNCH;
    const int tiles_per_bag = (int)((max_rows + F2_BM - 1) / F2_BM);
    const long long n_items = (long long)tiles_per_bag * n_bags;
    if (n_items > 0x7fffffffLL) return DSMIL_E_UNSUPPORTED;
    int cus = persistent_grid(1 << 20);
#ifdef DSMIL_EXPERIMENTS
    static const int f2_grid = expt_env("DSMIL_F2_GRID");
    if (f2_grid > 0) cus = f2_grid;
#endif
    const long long grid = n_items < cus ? n_items : cus;   // persistent: one workgroup per CU (148 KiB of LDS each)
    const int slot = dsmil_prof::begin(dsmil_prof::CH_ATTEND, st);
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(F2_THREADS), F2_LDS_BYTES, st, a, rowmax, tiles_per_bag, (int)n_items);
    dsmil_prof::end(dsmil_prof::CH_ATTEND, slot, st);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

// batches of fp32 bags, two-layer query, C <= 2: 32-row tiles, the query weights resident in registers (agg_f3.h)
template <int NK1>
int launch_attend_f3_k(const AttendArgs& a, const float* rowmax, long long max_rows, long long total_rows, int n_bags, hipStream_t st, int* seg_per, int* seg_T) {
    void (*fn)(AttendArgs, const float*, int, int, int) = a.C == 2 ? k_attend_f3<NK1, true> : k_attend_f3<NK1, false>;
#ifdef DSMIL_EXPERIMENTS
    static const int f3_dbg = expt_env("DSMIL_F3_DBG");
    if (NK1 == 16 && a.C == 1 && f3_dbg == 1) fn = k_attend_f3<16, false, 1>;
    if (NK1 == 16 && a.C == 1 && f3_dbg == 2) fn = k_attend_f3<16, false, 2>;
    if (NK1 == 16 && a.C == 1 && f3_dbg == 3) fn = k_attend_f3<16, false, 3>;
    if (NK1 == 16 && a.C == 1 && f3_dbg == 4) fn = k_attend_f3<16, false, 4>;
    if (NK1 == 16 && a.C == 1 && f3_dbg == 5) fn = k_attend_f3<16, false, 5>;
#endif
    constexpr int lds = f3_lds_bytes(32 * NK1);
    if (!dsmil_lds::allow((const void*)fn, lds)) return DSMIL_E_LAUNCH;
    const int cus = persistent_grid(F3_MAX_WG);
    // ragged batch (a.tile_pre): the items are the real tiles — at most total_rows / 32 + n_bags of them (the kernel reads the
    // exact count from tile_pre[n_bags]); uniform: tiles_per_bag items per bag
    const long long tiles_per_bag = a.tile_pre ? 0 : (max_rows + F3_BM - 1) / F3_BM;
    const long long n_items = a.tile_pre ? total_rows / F3_BM + n_bags : tiles_per_bag * n_bags;
    if (n_items > 0x7fffffffLL) return DSMIL_E_UNSUPPORTED;
    const long long per = (n_items + cus - 1) / cus;   // contiguous runs of tile items per workgroup (k_attend_bf16_res's scheme)
    const unsigned grid = (unsigned)((n_items + per - 1) / per);
    *seg_per = (int)per;
    *seg_T = (int)tiles_per_bag;
    const int slot = dsmil_prof::begin(dsmil_prof::CH_ATTEND, st);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(F3_THREADS), lds, st, a, rowmax, (int)tiles_per_bag, (int)n_items, (int)per);
    dsmil_prof::end(dsmil_prof::CH_ATTEND, slot, st);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}
int launch_attend_f3(const AttendArgs& a, const float* rowmax, long long max_rows, long long total_rows, int n_bags, hipStream_t st, int* seg_per, int* seg_T) {
    switch (a.K / 32) {
        case 4: return launch_attend_f3_k<4>(a, rowmax, max_rows, total_rows, n_bags, st, seg_per, seg_T);
        case 8: return launch_attend_f3_k<8>(a, rowmax, max_rows, total_rows, n_bags, st, seg_per, seg_T);
        case 12: return launch_attend_f3_k<12>(a, rowmax, max_rows, total_rows, n_bags, st, seg_per, seg_T);
        case 16: return launch_attend_f3_k<16>(a, rowmax, max_rows, total_rows, n_bags, st, seg_per, seg_T);
        default: return DSMIL_E_UNSUPPORTED;
    }
}

int mlp_mode() {
#ifdef DSMIL_EXPERIMENTS
    static const int mode = [] {
        const char* e = getenv("DSMIL_MLP");
        if (!e) return 6;
        if (!strcmp(e, "f32")) return 0;
        if (!strcmp(e, "s9")) return 9;
        return 6;
    }();
    return mode;
#else
    return 6;
#endif
}

int launch_attend_bf16_dma(AttendArgs a, long long max_rows, int n_bags, hipStream_t st) {
    constexpr int NW = 4, BM = NW * 32;
    const size_t lds = (size_t)(2 * BD_WCHUNK_F4 + 3 * BM * 8) * 16;   // 32 KiB weights + 48 KiB features = 80 KiB: 2 per CU
    if (!dsmil_lds::allow((const void*)k_query_attend_bf16_dma<NW>, (int)lds)) return DSMIL_E_LAUNCH;
    const int K64 = (a.K + 63) / 64 * 64;
    a.wpk = a.wpk + (size_t)QD * K64 + QD * QD;   // the fragment image sits behind the row-major one
    dim3 grid((unsigned)((max_rows + BM - 1) / BM), (unsigned)n_bags);
    const int slot = dsmil_prof::begin(dsmil_prof::CH_ATTEND, st);
    hipLaunchKernelGGL((k_query_attend_bf16_dma<NW>), grid, dim3(NW * 64), lds, st, a);
    dsmil_prof::end(dsmil_prof::CH_ATTEND, slot, st);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

// The tile-resident kernel (agg_res.h): persistent, one 320-thread workgroup per CU.  Needs the whole K of a
// 128-row tile in LDS and this wave's weight slice in registers: instantiated for K = 512 and K = 256, values == features,
// C <= 2; everything else keeps the ring kernels.
bool bf16_res_ok(const AttendArgs& a) {
    return (a.K == 512 || a.K == 256) && a.Kv == a.K && a.vals == a.feats && a.C <= 2;
}
template <int NCH>
void (*bf16_res_fn(const AttendArgs& a))(AttendArgs, int, int, int) {
    return a.C == 2 ? (a.nonlinear ? k_attend_bf16_res<NCH, true, true> : k_attend_bf16_res<NCH, true, false>)
                    : (a.nonlinear ? k_attend_bf16_res<NCH, false, true> : k_attend_bf16_res<NCH, false, false>);
}
int launch_attend_bf16_res(AttendArgs a, long long max_rows, long long total_rows, int n_bags, hipStream_t st, int* seg_per, int* seg_T) {
    const int cus = persistent_grid(RS_MAX_WG);
    typedef void (*res_fn)(AttendArgs, int, int, int);
    res_fn fn = a.K == 512 ? bf16_res_fn<8>(a) : bf16_res_fn<4>(a);
#ifdef DSMIL_EXPERIMENTS
    if (a.K == 512 && a.C == 2 && a.nonlinear && (a.expt & 1024)) fn = k_attend_bf16_res<8, true, true, 1>;
    if (a.K == 512 && a.C == 2 && a.nonlinear && (a.expt & 2048)) fn = k_attend_bf16_res<8, true, true, 2>;
#endif
    if (!dsmil_lds::allow((const void*)fn, RS_LDS_BYTES)) return DSMIL_E_LAUNCH;
    const int K64 = (a.K + 63) / 64 * 64;
    a.wpk = a.wpk + (size_t)QD * K64 + QD * QD;   // the fragment image sits behind the row-major one
    const long long tiles_per_bag = a.tile_pre ? 1 : (max_rows + RS_BM - 1) / RS_BM;   // (ragged: unused by the kernel)
    const long long n_items = a.tile_pre ? total_rows / RS_BM + n_bags : tiles_per_bag * n_bags;
    if (n_items > 0x7fffffffLL) return DSMIL_E_UNSUPPORTED;
    // contiguous runs of tile items per workgroup: consecutive tiles belong to the same bag and share one partial
    const long long per = (n_items + cus - 1) / cus;
    const unsigned grid = (unsigned)((n_items + per - 1) / per);
    *seg_per = (int)per;
    *seg_T = (int)tiles_per_bag;
    const int slot = dsmil_prof::begin(dsmil_prof::CH_ATTEND, st);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(RS_THREADS), RS_LDS_BYTES, st, a, (int)tiles_per_bag, (int)n_items, (int)per);
    dsmil_prof::end(dsmil_prof::CH_ATTEND, slot, st);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

template <int NW>
int launch_attend_bf16(const AttendArgs& a, long long max_rows, int n_bags, hipStream_t st) {
    constexpr int BM = NW * 32;
    const size_t lds = (size_t)(2 * W_TILE + 2 * BM * LDK) * sizeof(float);
    if (!dsmil_lds::allow((const void*)k_query_attend_bf16<NW>, (int)lds)) return DSMIL_E_LAUNCH;
    dim3 grid((unsigned)((max_rows + BM - 1) / BM), (unsigned)n_bags);
    const int slot = dsmil_prof::begin(dsmil_prof::CH_ATTEND, st);
    hipLaunchKernelGGL((k_query_attend_bf16<NW>), grid, dim3(NW * 64), lds, st, a);
    dsmil_prof::end(dsmil_prof::CH_ATTEND, slot, st);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

}  // namespace

// library-internal: FCLayer over logical rows (row r of the output reads physical row rowmap[r]; nullptr = identity)
int dsmil_fc_forward_rows(const float* feats, int64_t total_rows, int32_t K, int32_t C, const float* fc_w,
                          const float* fc_b, float* classes, const int64_t* rowmap, void* stream) {
    if (!feats || !fc_w || !fc_b || !classes || total_rows <= 0 || K <= 0 || C <= 0) return DSMIL_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    long long blocks = (total_rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    const bool v4 = (K % 4 == 0) && (((uintptr_t)feats | (uintptr_t)fc_w) % 16 == 0);
    if (v4) hipLaunchKernelGGL(k_fc<4>, dim3((unsigned)blocks), dim3(256), 0, st, feats, fc_w, fc_b, classes, (long long)total_rows, K, C, rowmap);
    else hipLaunchKernelGGL(k_fc<1>, dim3((unsigned)blocks), dim3(256), 0, st, feats, fc_w, fc_b, classes, (long long)total_rows, K, C, rowmap);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

void dsmil_agg_forward_leftovers(void* ws, int32_t n_bags, int64_t total_rows, int32_t K, int32_t Kv, int32_t C,
                                 const void** packed_split, const float** qmax, const float** pred_part, int* pred_blocks) {
    const WsLayout L = ws_layout(n_bags, total_rows, total_rows, K, Kv, C, pick_nw(n_bags, total_rows) * 32);
    if (pred_part) *pred_part = (const float*)((char*)ws + L.pred_part);
    if (pred_blocks) *pred_blocks = (int)L.nchunk_max;
    if (packed_split) *packed_split = (pick_nw(n_bags, total_rows) != 8 && mlp_mode()) ? (const void*)((char*)ws + L.wsplit) : nullptr;
    if (qmax) *qmax = (const float*)((char*)ws + L.qmax);
}

extern "C" {

int dsmil_abi_version(void) { return DSMIL_ABI_VERSION; }

int dsmil_agg_mlp_form(void) { return mlp_mode(); }
int dsmil_agg_batch_form(int mode) {
    if (mode >= 0 && mode <= 2) return g_use_f2.exchange(mode);
    return g_use_f2.load();
}
int dsmil_agg_logits_form(int mode) {
    if (mode >= 0 && mode <= 2) return g_logits_form.exchange(mode);
    return g_logits_form.load();
}
int dsmil_agg_persistent_grid(int n) {
    if (n >= 1 && n <= 1024) return g_persistent_grid.exchange(n);
    return g_persistent_grid.load();
}
int dsmil_device_cus(void) { return device_cus(); }
int dsmil_agg_inline_query(int mode) {
    if (mode == 0 || mode == 1) return g_inline_query.exchange(mode);
    return g_inline_query.load();
}

const char* dsmil_strerror(int code) {
    switch (code) {
        case DSMIL_OK: return "ok";
        case DSMIL_E_INVALID: return "invalid argument";
        case DSMIL_E_UNSUPPORTED: return "unsupported shape or dtype";
        case DSMIL_E_WORKSPACE: return "workspace too small";
        case DSMIL_E_LAUNCH: return "kernel launch failed";
        case DSMIL_E_ALIGN: return "pointer not aligned";
        default: return "unknown error";
    }
}

int dsmil_agg_tile_rows(int32_t n_bags, int64_t total_rows) { return pick_nw(n_bags, total_rows) * 32; }

size_t dsmil_agg_workspace_bytes(int32_t n_bags, int64_t total_rows, int32_t K, int32_t Kv,
                                 int32_t C) {
    (void)K;
    if (n_bags <= 0 || total_rows <= 0 || Kv <= 0 || C <= 0) return 0;
    // max_rows <= total_rows bounds the chunk count; tile rows as the launcher will pick them
    return ws_layout(n_bags, total_rows, total_rows, K, Kv, C, pick_nw(n_bags, total_rows) * 32).total;
}

int dsmil_fc_forward(const float* feats, int64_t total_rows, int32_t K, int32_t C,
                     const float* fc_w, const float* fc_b, float* classes, void* stream) {
    return dsmil_fc_forward_rows(feats, total_rows, K, C, fc_w, fc_b, classes, nullptr, stream);
}

__global__ void k_set_offsets2(int64_t* off, long long N) { off[0] = 0; off[1] = N; }

// 16-B vector loads on every fp32 operand of the forward
static bool fwd_v4(const void* feats, const void* vals, const dsmil_agg_params* p) {
    return (p->K % 4 == 0) && (p->Kv % 4 == 0) &&
           (((uintptr_t)feats | (uintptr_t)vals | (uintptr_t)p->q0_w | (uintptr_t)p->fc_w |
             (uintptr_t)(p->nonlinear ? p->q2_w : p->q0_w)) % 16 == 0);
}

struct ShardCtl {
    int phase = 0;                    // 0 whole forward; 1 logits + arg-max only; 2 attend against given rows
    const float* crit_rows = nullptr; // phase 2: [C,K] feature rows of the bag-wide critical instances
    float* best_val = nullptr;        // phase 1 out: [C]
    float* ml_out = nullptr;          // phase 2 out: [C,2] (max, sum) of this shard
    bool skip_pred = false;           // phase 0: leave the last sum (k_pred) to the caller (dsmil_agg_forward_nopred)
    const TrainPrologueJob* job = nullptr;   // carried by the logits launch (dsmil_agg_forward_nopred)
    const void* packed_f2 = nullptr;         // dsmil_agg_opts::packed_f2 (k_attend_f2's weight image, prepared by the caller)
};

static int agg_forward_impl(const void* feats, const void* vals, const int64_t* offsets,
                            int32_t n_bags, int64_t total_rows, int64_t max_rows, const dsmil_agg_params* p,
                            const void* packed_bf16, bool bf16, const float* classes_in, float* classes_out,
                            float* A, float* B, float* pred, int64_t* idx, void* ws, size_t ws_bytes,
                            void* stream, const ShardCtl& sh = ShardCtl(), const void* packed_split = nullptr,
                            const int64_t* rowmap = nullptr) {
    if (!feats || !p || !ws) return DSMIL_E_INVALID;
    if (sh.phase == 0 && (!offsets || !A || !B || (!pred && !sh.skip_pred) || !idx)) return DSMIL_E_INVALID;
    if (sh.phase == 1 && (!classes_out || !idx || !sh.best_val)) return DSMIL_E_INVALID;
    if (sh.phase == 2 && (!A || !B || !sh.crit_rows || !sh.ml_out)) return DSMIL_E_INVALID;
    if (n_bags <= 0 || total_rows <= 0 || max_rows <= 0 || max_rows > total_rows) return DSMIL_E_INVALID;
    if (p->K <= 0 || p->Kv <= 0 || p->C <= 0) return DSMIL_E_INVALID;
    if (!p->q0_w || !p->q0_b || !p->fcc_w || !p->fcc_b) return DSMIL_E_INVALID;
    if (p->nonlinear && (!p->q2_w || !p->q2_b)) return DSMIL_E_INVALID;
    if (sh.phase != 2 && !classes_in && (!p->fc_w || !p->fc_b || !classes_out)) return DSMIL_E_INVALID;
    if (bf16 && !packed_bf16) return DSMIL_E_INVALID;
    if (n_bags > 65535) return DSMIL_E_UNSUPPORTED;
    if (!vals) vals = feats;
    if (vals == feats && p->Kv != p->K) return DSMIL_E_INVALID;
    if (((uintptr_t)ws % 256) || ((uintptr_t)p->q0_b % 16) || (p->nonlinear && ((uintptr_t)p->q2_b % 16)))
        return DSMIL_E_ALIGN;
    const int K = p->K, Kv = p->Kv, C = p->C;
    if (bf16 && ((K % 8) || (Kv % 4) || ((uintptr_t)feats % 16) || ((uintptr_t)vals % 8) ||
                 ((uintptr_t)packed_bf16 % 16)))
        return DSMIL_E_UNSUPPORTED;
    const int NW = pick_nw(n_bags, total_rows);
    const int BM = NW * 32;
    const WsLayout L = ws_layout(n_bags, total_rows, max_rows, K, Kv, C, BM);
    if (ws_bytes < L.total) return DSMIL_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* w8 = (char*)ws;
    if (sh.phase) {  // a lone shard: its {0, N} offsets live in the workspace
        int64_t* off2 = (int64_t*)(w8 + L.off2);
        hipLaunchKernelGGL(k_set_offsets2, dim3(1), dim3(1), 0, st, off2, (long long)total_rows);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
        offsets = off2;
    }
    float* part_val = (float*)(w8 + L.part_val);
    long long* part_idx = (long long*)(w8 + L.part_idx);
    float* qmax = (float*)(w8 + L.qmax);
    float* part_ml = (float*)(w8 + L.part_ml);
    float* part_B = (float*)(w8 + L.part_B);
    float* pred_part = (float*)(w8 + L.pred_part);
    const float* f32 = (const float*)feats;
    const bf16_t* b16 = (const bf16_t*)feats;

    const bool v4 = bf16 || fwd_v4(feats, vals, p);
    const bool w4 = (K % 4 == 0) && (((uintptr_t)p->q0_w | (uintptr_t)p->fc_w) % 16 == 0);
    AttendArgs a{feats, vals, (const bf16_t*)packed_bf16, offsets, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, A,
                 part_ml, part_B, K, Kv, C, p->nonlinear, 0, 0, rowmap};
#ifdef DSMIL_EXPERIMENTS
    static const int expt = expt_env("DSMIL_EXPT"), logits_old = expt_env("DSMIL_LOGITS_OLD"), no_hs = expt_env("DSMIL_NO_HS"),
                     no_qmi = expt_env("DSMIL_NO_QMI"), no_f2 = expt_env("DSMIL_NO_F2");
    a.expt = expt;
#else
    constexpr int logits_old = 0, no_hs = 0, no_qmi = 0, no_f2 = 0;
#endif
    int seg_per = 0, seg_T = 0;   // k_attend_bf16_res: partials per (workgroup, bag), see k_finish
    bool lean_tail = false;       // the bf16 batch path whose logits / q_max / combine kernels fit beside a resident attend workgroup
    int hs_bm = 0;                // k_attend_hs: rows per tile (the partial slots follow it)
    {
        // Tried and rejected here (round 1, numbers in DESIGN.md §3): (a) chunking the batch and running
        // chunk c+1's HBM-bound logits on a helper stream under chunk c's MFMA-bound attend, and (b) one
        // persistent launch pulling logits/attend work items from a device queue with in-launch
        // release/acquire hand-offs.  Both were slower than these plain back-to-back launches.
        const int b0 = 0, nb = n_bags;
        // 1. instance logits + arg-max partials
        // rows per workgroup of the logits pass: 32 instead of R0 when there are few rows (the 32-row-tile regime of pick_nw)
        // and the streaming kernel runs
        const bool stream_ok = !classes_in && !logits_old && ((bf16 && (K % 8 == 0)) || (!bf16 && v4));
        // bf16 batches of K = 512 rows: the continuous-pipeline logits kernel that fits beside a resident k_attend_bf16_res
        // workgroup of another stream (k_logits_pipe); its workgroups take more rows (they are dispatched one at a time into
        // the one free slot of a CU) and its k_qmax launch is 4 waves wide so that it fits the same slot
        const bool lpipe = bf16 && stream_ok && !logits_old && K == 512 && C <= 2 && NW == 4 && sh.phase == 0 &&
                           !rowmap && max_rows < 0x7fffffffLL && coresident_wanted(st);
        const int r0 = lpipe ? PIPE_R0 : (NW == 1 && stream_ok && sh.phase != 2) ? 32 : R0;
        const int qmax_t = lpipe ? PIPE_QT : QMAX_T;
        lean_tail = lpipe;
        // few rows, fp32: k_attend_hs (below); with the streaming logits kernel before it, the critical row's query runs
        // inside the attend launch (AttendArgs::qm_flag) instead of as k_qmax between the two
        bool use_hs = !bf16 && NW == 1 && v4 && !no_hs && mlp_mode() == 6;
#ifdef DSMIL_EXPERIMENTS
        if (a.expt & 8) use_hs = false;
#endif
        // batches (128-row regime), fp32, v = Identity, K a multiple of 128 up to 512: resident 64-row tiles on the fp16 two-plane
        // form (agg_f2.h); it needs the row maxima the streaming logits kernel leaves behind
        const bool use_f2 = !bf16 && NW == 4 && v4 && !no_f2 && mlp_mode() == 6 && stream_ok && sh.phase == 0 && vals == feats &&
                            (K % 128 == 0) && K <= 16 * F2_MAXSTEPS && g_use_f2.load(std::memory_order_relaxed);
        float* rowmax = use_f2 ? (float*)(w8 + L.rowmax) : nullptr;
        const bool qm_inline = use_hs && stream_ok && sh.phase == 0 && !no_qmi && g_inline_query.load(std::memory_order_relaxed) &&
                               hs_inline_fits(max_rows, nb, C);
        int* qflag = qm_inline ? (int*)(w8 + L.qflag) : nullptr;
        dim3 grid((unsigned)((max_rows + r0 - 1) / r0), (unsigned)nb);
        // RAGGED batch in the 128-row regime (the persistent attend kernels and the streaming logits pass): the work lists are the
        // real tiles (prefix per bag: k_tile_prefix), not n_bags x the tiles of the longest bag
        int* tile_pre_a = nullptr;
        int* tile_pre_l = nullptr;
        if (NW == 4 && stream_ok && sh.phase == 0 && (long long)max_rows * nb != (long long)total_rows &&
            total_rows / 32 + nb < 0x7fffffffLL) {
            tile_pre_a = (int*)(w8 + L.tile_pre);
            tile_pre_l = tile_pre_a + (nb + 1);
            hipLaunchKernelGGL(k_tile_prefix, dim3(1), dim3(1024), 0, st, offsets, nb, bf16 ? RS_BM : F3_BM, r0, tile_pre_a, tile_pre_l);
            if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
        }
        const dim3 grid_l = tile_pre_l ? dim3((unsigned)(total_rows / r0 + nb), 1u) : grid;   // (an upper bound of the real tiles)
        if (sh.phase == 2) {}  // the caller already knows the bag-wide critical rows
        else if (classes_in) hipLaunchKernelGGL((k_logits_argmax<1, true, float>), grid, dim3(256), 0, st, f32, offsets, p->fc_w, p->fc_b, classes_in, classes_out, part_val, part_idx, K, C, b0, rowmap);
        else if (bf16 && (K % 8 == 0) && !logits_old) {
            const size_t ldsw = (size_t)(C >= 2 ? 2 : 1) * (((K + 63) / 64 + 1) * 64) * sizeof(float);
            // batches of K = 512 rows: the continuous-pipeline form that fits beside a resident k_attend_bf16_res workgroup
            if (lpipe) {
                if (C == 2) hipLaunchKernelGGL(k_logits_pipe<2>, grid_l, dim3(256), 0, st, b16, offsets, p->fc_w, p->fc_b, classes_out, part_val, part_idx, b0, r0, tile_pre_l, nb);
                else hipLaunchKernelGGL(k_logits_pipe<1>, grid_l, dim3(256), 0, st, b16, offsets, p->fc_w, p->fc_b, classes_out, part_val, part_idx, b0, r0, tile_pre_l, nb);
            }
            else if (C >= 2) hipLaunchKernelGGL((k_logits_stream<2, bf16_t>), grid_l, dim3(256), ldsw, st, b16, offsets, p->fc_w, p->fc_b, classes_out, part_val, part_idx, K, C, b0, rowmap, r0, (int*)nullptr, TrainPrologueJob{}, (float*)nullptr, tile_pre_l, nb);
            else hipLaunchKernelGGL((k_logits_stream<1, bf16_t>), grid_l, dim3(256), ldsw, st, b16, offsets, p->fc_w, p->fc_b, classes_out, part_val, part_idx, K, C, b0, rowmap, r0, (int*)nullptr, TrainPrologueJob{}, (float*)nullptr, tile_pre_l, nb);
        }
        else if (bf16 && w4) hipLaunchKernelGGL((k_logits_argmax<4, false, bf16_t>), grid, dim3(256), 0, st, b16, offsets, p->fc_w, p->fc_b, classes_in, classes_out, part_val, part_idx, K, C, b0, rowmap);
        else if (bf16) hipLaunchKernelGGL((k_logits_argmax<1, false, bf16_t>), grid, dim3(256), 0, st, b16, offsets, p->fc_w, p->fc_b, classes_in, classes_out, part_val, part_idx, K, C, b0, rowmap);
        else if (v4 && !logits_old) {
            const size_t ldsw = (size_t)(C >= 2 ? 2 : 1) * (((K + 31) / 32 + 1) * 32) * sizeof(float);
            const TrainPrologueJob job = sh.job ? *sh.job : TrainPrologueJob{};
            dim3 gridj(grid_l.x + (unsigned)job.blocks, grid_l.y);   // (a training step is one bag: never the flat grid)
            if (C >= 2) hipLaunchKernelGGL((k_logits_stream<2, float>), gridj, dim3(256), ldsw, st, f32, offsets, p->fc_w, p->fc_b, classes_out, part_val, part_idx, K, C, b0, rowmap, r0, qflag, job, rowmax, tile_pre_l, nb);
            else hipLaunchKernelGGL((k_logits_stream<1, float>), gridj, dim3(256), ldsw, st, f32, offsets, p->fc_w, p->fc_b, classes_out, part_val, part_idx, K, C, b0, rowmap, r0, qflag, job, rowmax, tile_pre_l, nb);
        }
        else if (v4) hipLaunchKernelGGL((k_logits_argmax<4, false, float>), grid, dim3(256), 0, st, f32, offsets, p->fc_w, p->fc_b, classes_in, classes_out, part_val, part_idx, K, C, b0, rowmap);
        else hipLaunchKernelGGL((k_logits_argmax<1, false, float>), grid, dim3(256), 0, st, f32, offsets, p->fc_w, p->fc_b, classes_in, classes_out, part_val, part_idx, K, C, b0, rowmap);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
        // 2. critical instance + its query
        dim3 gq((unsigned)nb, (unsigned)C);
        if (sh.phase == 1) {
            if (v4) hipLaunchKernelGGL((k_qmax<4, float>), gq, dim3(QMAX_T), 0, st, f32, offsets, part_val, part_idx, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, idx, K, C, p->nonlinear, b0, 1, sh.best_val, rowmap, r0);
            else hipLaunchKernelGGL((k_qmax<1, float>), gq, dim3(QMAX_T), 0, st, f32, offsets, part_val, part_idx, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, idx, K, C, p->nonlinear, b0, 1, sh.best_val, rowmap, r0);
            return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
        }
        if (qm_inline) {   // produced by the first C workgroups of every grid row of k_attend_hs
            a.qm_flag = qflag; a.qm_part_val = part_val; a.qm_part_idx = part_idx; a.qm_idx = idx; a.qm_r0 = r0;
        }
        else if (sh.phase == 2) {
            const bool r4 = (K % 4 == 0) && ((uintptr_t)sh.crit_rows % 16 == 0) && (((uintptr_t)p->q0_w) % 16 == 0);
            if (r4) hipLaunchKernelGGL((k_qmax<4, float>), gq, dim3(QMAX_T), 0, st, sh.crit_rows, offsets, part_val, part_idx, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, idx, K, C, p->nonlinear, b0, 2, (float*)nullptr);
            else hipLaunchKernelGGL((k_qmax<1, float>), gq, dim3(QMAX_T), 0, st, sh.crit_rows, offsets, part_val, part_idx, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, idx, K, C, p->nonlinear, b0, 2, (float*)nullptr);
        }
        else if (bf16 && w4) hipLaunchKernelGGL((k_qmax<4, bf16_t>), gq, dim3(qmax_t), 0, st, b16, offsets, part_val, part_idx, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, idx, K, C, p->nonlinear, b0, 0, (float*)nullptr, rowmap, r0);
        else if (bf16) hipLaunchKernelGGL((k_qmax<1, bf16_t>), gq, dim3(qmax_t), 0, st, b16, offsets, part_val, part_idx, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, idx, K, C, p->nonlinear, b0, 0, (float*)nullptr, rowmap, r0);
        else if (v4) hipLaunchKernelGGL((k_qmax<4, float>), gq, dim3(QMAX_T), 0, st, f32, offsets, part_val, part_idx, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, idx, K, C, p->nonlinear, b0, 0, (float*)nullptr, rowmap, r0);
        else hipLaunchKernelGGL((k_qmax<1, float>), gq, dim3(QMAX_T), 0, st, f32, offsets, part_val, part_idx, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, idx, K, C, p->nonlinear, b0, 0, (float*)nullptr, rowmap, r0);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
        // 3. query MLP on MFMA + scores + tile softmax + weighted value sum
        int rc;
        seg_per = seg_T = 0;
        const int mode = (NW == 8) ? 0 : mlp_mode();
        if (use_f2 && sh.packed_f2) {
            a.wpk = (const bf16_t*)sh.packed_f2;  // the caller cut the weights once (dsmil_agg_pack_f2)
        } else if (use_f2) {
            _Float16* wf2 = (_Float16*)(w8 + L.wf2);
            hipLaunchKernelGGL(k_pack_agg_f2, dim3(64), dim3(256), 0, st, p->q0_w, p->nonlinear ? p->q2_w : nullptr, wf2, K, 2 * ((K + 31) / 32));
            if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
            a.wpk = (const bf16_t*)wf2;
        } else if (!bf16 && mode && packed_split) {
            a.wpk = (const bf16_t*)packed_split;  // the caller cut the weights once (dsmil_agg_pack_split)
        } else if (!bf16 && mode) {
            const int nks = 2 * ((K + 31) / 32);
            bf16_t* wsplit = (bf16_t*)(w8 + L.wsplit);
            hipLaunchKernelGGL(k_pack_agg_split, dim3(64), dim3(256), 0, st, p->q0_w, p->nonlinear ? p->q2_w : nullptr, wsplit, K, nks);
            if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
            a.wpk = wsplit;
        }
        bool bf16_dma = bf16 && NW == 4;   // feature rows are 16-B aligned on this path (K % 8 == 0, checked above)
#ifdef DSMIL_EXPERIMENTS
        if (a.expt & 256) bf16_dma = false;
#endif
        bool bf16_res = bf16_dma && bf16_res_ok(a);
#ifdef DSMIL_EXPERIMENTS
        if (a.expt & 512) bf16_res = false;
#endif
        const bool use_f3 = use_f2 && p->nonlinear && C <= 2 && g_use_f2.load(std::memory_order_relaxed) == 2;
        // ragged batch on a persistent kernel: the item list is the real tiles (prefix per bag), not max_rows-padded bags
        if (tile_pre_a && (use_f3 || bf16_res)) {
            a.tile_pre = tile_pre_a;
            a.n_bags = nb;
        }
        if (use_f3) { rc = launch_attend_f3(a, rowmax, max_rows, total_rows, nb, st, &seg_per, &seg_T); hs_bm = F3_BM; }
        else if (use_f2) { rc = launch_attend_f2(a, rowmax, max_rows, nb, st); hs_bm = F2_BM; }
        else if (bf16_res) rc = launch_attend_bf16_res(a, max_rows, total_rows, nb, st, &seg_per, &seg_T);
        else if (bf16_dma) rc = launch_attend_bf16_dma(a, max_rows, nb, st);
        else if (bf16) rc = (NW == 4) ? launch_attend_bf16<4>(a, max_rows, nb, st) : launch_attend_bf16<1>(a, max_rows, nb, st);
#ifdef DSMIL_EXPERIMENTS   // DSMIL_MLP=s9 and the ablation variants: not instantiated in the product library
        else if (mode == 9 && NW == 4) rc = v4 ? launch_attend_split<4, 4, 9>(a, max_rows, nb, st) : launch_attend_split<4, 1, 9>(a, max_rows, nb, st);
        else if (mode == 9) rc = v4 ? launch_attend_split<1, 4, 9>(a, max_rows, nb, st) : launch_attend_split<1, 1, 9>(a, max_rows, nb, st);
        else if (mode == 6 && NW == 4 && v4 && (a.expt & 16)) rc = launch_attend_split<4, 4, 6, false, 16>(a, max_rows, nb, st);
        else if (mode == 6 && NW == 4 && v4 && (a.expt & 32)) rc = launch_attend_split<4, 4, 6, false, 32>(a, max_rows, nb, st);
        else if (mode == 6 && NW == 4 && v4 && (a.expt & 8)) rc = launch_attend_split<4, 4, 6, true>(a, max_rows, nb, st);
        else if (mode == 6 && NW == 1 && v4 && (a.expt & 8)) rc = launch_attend_split<1, 4, 6, true>(a, max_rows, nb, st);
#endif
        else if (mode == 6 && NW == 4) rc = v4 ? launch_attend_split<4, 4, 6>(a, max_rows, nb, st) : launch_attend_split<4, 1, 6>(a, max_rows, nb, st);
        else if (use_hs) { rc = launch_attend_hs(a, max_rows, nb, st); hs_bm = HS_BM; }   // few rows: hidden units split over the SIMDs
        else if (mode == 6) rc = v4 ? launch_attend_split<1, 4, 6>(a, max_rows, nb, st) : launch_attend_split<1, 1, 6>(a, max_rows, nb, st);
        else if (NW == 8) rc = launch_attend<8, 4>(a, max_rows, nb, st);
        else if (NW == 4) rc = v4 ? launch_attend<4, 4>(a, max_rows, nb, st) : launch_attend<4, 1>(a, max_rows, nb, st);
        else rc = v4 ? launch_attend<1, 4>(a, max_rows, nb, st) : launch_attend<1, 1>(a, max_rows, nb, st);
        if (rc != DSMIL_OK) return rc;
    }
    // 4. combine (skipped under the stamp-trace knob, which leaves its stamps in A)
    if (!DSMIL_EXPT_ON(a, 64)) {
        dim3 grid((unsigned)L.nchunk_max, (unsigned)n_bags);
        if (Kv % 4 == 0 && lean_tail && seg_per && !hs_bm)
            hipLaunchKernelGGL((k_finish<4, 2>), grid, dim3(256), 0, st, offsets, part_ml, part_B, p->fcc_w, A, B, pred_part, Kv, C, RS_BM, sh.ml_out, seg_per, seg_T, a.tile_pre);
        else if (Kv % 4 == 0)
            hipLaunchKernelGGL(k_finish<4>, grid, dim3(256), 0, st, offsets, part_ml, part_B, p->fcc_w, A, B, pred_part, Kv, C, hs_bm ? hs_bm : (seg_per ? RS_BM : BM), sh.ml_out, seg_per, seg_T, a.tile_pre);
        else
            hipLaunchKernelGGL(k_finish<1>, grid, dim3(256), 0, st, offsets, part_ml, part_B, p->fcc_w, A, B, pred_part, Kv, C, BM, sh.ml_out, 0, 0);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
        if (sh.phase == 2 || sh.skip_pred) return DSMIL_OK;  // the bag head runs after the cross-shard merge / in the caller's next launch
        // (folding this sum into k_finish's last block per bag — a device-scope fence + ticket in every block — was tried:
        // 11 -> 52 us for 64 bags, the fences wait for the 5 MB of attention the blocks have just written)
        const int n = n_bags * C;
        hipLaunchKernelGGL(k_pred, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pred_part,
                           p->fcc_b, pred, C, (int)L.nchunk_max, n_bags);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    }
    return DSMIL_OK;
}

int dsmil_agg_forward(const float* feats, const float* vals, const int64_t* offsets,
                      int32_t n_bags, int64_t total_rows, int64_t max_rows,
                      const dsmil_agg_params* p, const float* classes_in, float* classes_out,
                      float* A, float* B, float* pred, int64_t* idx, void* ws, size_t ws_bytes,
                      void* stream) {
    return agg_forward_impl(feats, vals, offsets, n_bags, total_rows, max_rows, p, nullptr, false, classes_in,
                            classes_out, A, B, pred, idx, ws, ws_bytes, stream);
}

size_t dsmil_agg_packed_split_bytes(int32_t K, int32_t nonlinear) {
    if (K <= 0) return 0;
    return (size_t)(2 * ((K + 31) / 32) + (nonlinear ? 8 : 0)) * S3_CHUNK_F4 * 16;
}

int dsmil_agg_pack_split(const float* q0_w, const float* q2_w, int32_t K, void* packed, void* stream) {
    if (!q0_w || !packed || K <= 0) return DSMIL_E_INVALID;
    if ((uintptr_t)packed % 16) return DSMIL_E_ALIGN;
    hipLaunchKernelGGL(k_pack_agg_split, dim3(64), dim3(256), 0, (hipStream_t)stream, q0_w, q2_w, (bf16_t*)packed, K,
                       2 * ((K + 31) / 32));
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

size_t dsmil_agg_packed_f2_bytes(int32_t K) { return K <= 0 ? 0 : f2_image_bytes(K); }

int dsmil_agg_pack_f2(const float* q0_w, const float* q2_w, int32_t K, void* packed, void* stream) {
    if (!q0_w || !packed || K <= 0) return DSMIL_E_INVALID;
    if ((uintptr_t)packed % 16) return DSMIL_E_ALIGN;
    hipLaunchKernelGGL(k_pack_agg_f2, dim3(64), dim3(256), 0, (hipStream_t)stream, q0_w, q2_w, (_Float16*)packed, K,
                       2 * ((K + 31) / 32));
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

int dsmil_agg_forward_ex(const float* feats, const float* vals, const int64_t* offsets,
                         int32_t n_bags, int64_t total_rows, int64_t max_rows,
                         const dsmil_agg_params* p, const dsmil_agg_opts* opts, const float* classes_in,
                         float* classes_out, float* A, float* B, float* pred, int64_t* idx, void* ws,
                         size_t ws_bytes, void* stream) {
    const void* packed_split = opts ? opts->packed_split : nullptr;
    const int64_t* row_map = opts ? opts->row_map : nullptr;
    if (packed_split && ((uintptr_t)packed_split % 16)) return DSMIL_E_ALIGN;
    ShardCtl sh;
    sh.packed_f2 = opts ? opts->packed_f2 : nullptr;
    if (sh.packed_f2 && ((uintptr_t)sh.packed_f2 % 16)) return DSMIL_E_ALIGN;
    return agg_forward_impl(feats, vals, offsets, n_bags, total_rows, max_rows, p, nullptr, false, classes_in,
                            classes_out, A, B, pred, idx, ws, ws_bytes, stream, sh, packed_split, row_map);
}

}  // extern "C"
bool dsmil_agg_forward_carries_prologue(const float* feats, int64_t total_rows, const dsmil_agg_params* p) {
#ifdef DSMIL_EXPERIMENTS
    static const int off = expt_env("DSMIL_LOGITS_OLD") | expt_env("DSMIL_NO_PROLOGUE_ROLE");
    if (off) return false;
#endif
    (void)total_rows;
    return feats && p && p->fc_w && fwd_v4(feats, feats, p);   // = the fp32 k_logits_stream launch of agg_forward_impl
}
int dsmil_agg_forward_nopred(const float* feats, const int64_t* offsets, int64_t total_rows, const dsmil_agg_params* p,
                             const dsmil_agg_opts* opts, float* classes_out, float* A, float* B, int64_t* idx, void* ws,
                             size_t ws_bytes, void* stream, const TrainPrologueJob* job) {
    ShardCtl sh;
    sh.skip_pred = true;
    if (job && !dsmil_agg_forward_carries_prologue(feats, total_rows, p)) return DSMIL_E_INVALID;
    sh.job = job;
    return agg_forward_impl(feats, nullptr, offsets, 1, total_rows, total_rows, p, nullptr, false, nullptr, classes_out, A, B,
                            nullptr, idx, ws, ws_bytes, stream, sh, opts ? opts->packed_split : nullptr,
                            opts ? opts->row_map : nullptr);
}
extern "C" {

// ---- fused training objective of one bag (train_tcga.py:67-71) -----------------------------------------
// loss = 0.5 BCEWithLogits(pred, y) + 0.5 BCEWithLogits(max_n classes[n,:], y)   (mean over the C classes each),
// the max over instances IS classes[idx_c, c] (idx = the forward's critical-instance index), so the gradient of
// the instance stream is sparse: one row per class.
static __global__ void k_loss_head(const float* __restrict__ classes, const float* __restrict__ pred,
                            const int64_t* __restrict__ idx, const float* __restrict__ label, int C,
                            float* __restrict__ loss, float* __restrict__ max_pred, float* __restrict__ g_pred,
                            float* __restrict__ g_max) {
    const int c = threadIdx.x;
    float l = 0.f;
    if (c < C) {
        const float y = label[c];
        const float zb = pred[c], zm = classes[idx[c] * (long long)C + c];
        // BCEWithLogits(z, y) = max(z,0) - z y + log1p(exp(-|z|))  (torch's stable form)
        const float lb = fmaxf(zb, 0.f) - zb * y + log1pf(expf(-fabsf(zb)));
        const float lm = fmaxf(zm, 0.f) - zm * y + log1pf(expf(-fabsf(zm)));
        l = 0.5f * (lb + lm) / (float)C;
        const float sb = 1.f / (1.f + expf(-zb)), sm = 1.f / (1.f + expf(-zm));
        if (max_pred) max_pred[c] = zm;
        if (g_pred) g_pred[c] = 0.5f * (sb - y) / (float)C;
        if (g_max) g_max[c] = 0.5f * (sm - y) / (float)C;
    }
    l = wave_sum(l);   // C <= 64: one wave
    if (threadIdx.x == 0) *loss = l;
}

int dsmil_agg_loss_head(const float* classes, const float* pred, const int64_t* idx, const float* label,
                        int32_t C, float* loss, float* max_pred, float* g_pred, float* g_max, void* stream) {
    if (!classes || !pred || !idx || !label || !loss || C <= 0) return DSMIL_E_INVALID;
    if (C > 64) return DSMIL_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_loss_head, dim3(1), dim3(64), 0, (hipStream_t)stream, classes, pred, idx, label, C, loss,
                       max_pred, g_pred, g_max);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

int dsmil_agg_shard_argmax(const float* feats, int64_t rows, const dsmil_agg_params* p, float* classes_out,
                           float* best_val, int64_t* best_idx, void* ws, size_t ws_bytes, void* stream) {
    ShardCtl sh;
    sh.phase = 1;
    sh.best_val = best_val;
    return agg_forward_impl(feats, nullptr, nullptr, 1, rows, rows, p, nullptr, false, nullptr, classes_out, nullptr,
                            nullptr, nullptr, best_idx, ws, ws_bytes, stream, sh);
}

int dsmil_agg_shard_attend(const float* feats, const float* vals, int64_t rows, const dsmil_agg_params* p,
                           const float* crit_rows, float* A_unnorm, float* ml, float* B_unnorm, void* ws,
                           size_t ws_bytes, void* stream) {
    ShardCtl sh;
    sh.phase = 2;
    sh.crit_rows = crit_rows;
    sh.ml_out = ml;
    return agg_forward_impl(feats, vals, nullptr, 1, rows, rows, p, nullptr, false, nullptr, nullptr, A_unnorm,
                            B_unnorm, nullptr, nullptr, ws, ws_bytes, stream, sh);
}

size_t dsmil_agg_packed_bf16_bytes(int32_t K) {
    if (K <= 0) return 0;
    const size_t K64 = ((size_t)K + 63) / 64 * 64;
    return (QD * K64 + QD * QD) * sizeof(bf16_t) + (K64 / 64 + 2) * (size_t)BD_WCHUNK_F4 * 16;   // row-major + fragment image
}

int dsmil_agg_pack_bf16(const float* q0_w, const float* q2_w, int32_t K, void* packed, void* stream) {
    if (!q0_w || !packed || K <= 0) return DSMIL_E_INVALID;
    const int K64 = (K + 63) / 64 * 64;
    hipLaunchKernelGGL(k_pack_agg_bf16, dim3(512), dim3(256), 0, (hipStream_t)stream, q0_w, q2_w,
                       (bf16_t*)packed, K, K64);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

int dsmil_agg_forward_bf16(const void* feats_bf16, const void* vals_bf16, const int64_t* offsets,
                           int32_t n_bags, int64_t total_rows, int64_t max_rows,
                           const dsmil_agg_params* p, const void* packed, const float* classes_in,
                           float* classes_out, float* A, float* B, float* pred, int64_t* idx, void* ws,
                           size_t ws_bytes, void* stream) {
    return agg_forward_impl(feats_bf16, vals_bf16, offsets, n_bags, total_rows, max_rows, p, packed, true,
                            classes_in, classes_out, A, B, pred, idx, ws, ws_bytes, stream);
}

}  // extern "C"
