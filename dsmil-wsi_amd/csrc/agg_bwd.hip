// DSMIL dual-stream aggregator, backward + the fused training step — hand-written HIP for gfx950 (MI355X, CDNA4).
//
// Gradient of dsmil.py:46-62 (+ the fused FCLayer, dsmil.py:6-12) for ONE bag, as autograd would
// produce it for train_tcga.py:67-72 (the arg-max indices are constants; nonlinear or linear query;
// v = Identity or caller-supplied value rows).  Notation: x [N,K], c = x Wf^T + bf, H = relu(x W1^T+b1),
// Q = tanh(H W2^T + b2), q_c = Q[idx_c], s = Q q^T / sqrt(128), A = softmax_n(s), B = A^T V,
// pred = fcc(B).
//   gB  = g_B + sum_o g_pred[o] fcc_w[o]            g_fcc_w = g_pred (x) B,  g_fcc_b = g_pred
//   gA  = g_A + V gB^T                              D_c = <gB_c, B_c> (+ sum_n A g_A)   [= sum_n A gA]
//   gs  = A (gA - D) / sqrt(128)
//   gQ  = gs q  (+ at row idx_c:  sum_n gs[n,c] Q[n])        gz2 = gQ (1 - Q^2)
//   g_W2 = gz2^T H, g_b2 = colsum gz2, gH = (gz2 W2) [H > 0], g_W1 = gH^T x, g_b1 = colsum gH
//   g_Wf = g_c^T x, g_bf = colsum g_c
// Every matrix product runs on bf16 MFMA over EXACT three-plane cuts of both fp32 operands, six plane products
// (agg_split.h — the form the forward uses; round 3 had the backward on v_mfma_f32_32x32x2_f32, 1/16 of the rate).
// Launch sequence (one stream, no host sync; 9 launches, round 3: 17):
//   k_pack_agg_split x2  plane-cut W1 | W2 (skipped when the forward's packed image is handed in) and W2^T
//   k_bwd_prep           gB, D, g_fcc_*                                  (block 0: head; the others: g_fcc_w)
//   k_fc                 gA = V gB^T                                     (HBM stream, the forward's FCLayer kernel)
//   k_bwd_qrow           q_c = q(x[idx_c])                               (skipped when the forward's q_max is handed in)
//   k_bwd_rows           recompute H, Q per 32-row wave tile (the forward's MLP tile, H stored from the accumulators),
//                        gs, gz2 -> workspace; per-tile partials of g_q = gs^T Q (register butterfly over the rows)
//   k_bwd_critical       g_q = sum of the partials; gz2[idx_c] += g_q[c] (1 - Q[idx_c]^2)
//   k_bwd_gh             gH = (gz2 W2) [H>0]  — the forward GEMM-1 pipeline with X := gz2, W := W2^T
//   k_tn_split           g_W1 = gH^T x and g_W2 = gz2^T H in ONE launch: contraction over instances, 64-column slabs x
//                        row ranges, both operands cut into planes as they are staged (k = instance rows along the
//                        registers: the planes are written TRANSPOSED, one ds_write_b128 per 8 rows), column sums of
//                        gH / gz2 (bias gradients) from the staged values
//   k_bwd_reduce         fixed-order sums of the row-range partials -> g_W1, g_b1, g_W2, g_b2; the sparse max-stream
//                        gradient of the FCLayer (g_max) in the same launch
// dsmil_agg_train_step chains forward -> loss head -> this backward -> one Adam kernel over all eight tensors:
// one C call per train_tcga.py:60-75 step.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "agg_common.h"
#include "agg_split.h"
#include "agg_hs.h"
#include "lds_attr.h"

namespace {

constexpr int NP_BWD = 6;   // plane products per fp32 MAC (the forward's default form)

// instance rows per workgroup of k_tn_small (dense instance-logit gradient only): >= 128, multiple of 32, and at most
// 256 row ranges per bag
inline int tn_rows(long long N) {
    long long r = (N + 255) / 256;
    r = (r + 31) / 32 * 32;
    return (int)(r < 128 ? 128 : r);
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- k_bwd_prep ----------------------------------------------------------------------------
// block 0: gB, D, g_fcc_b, the zero bias of the gH pipeline; blocks 1..: g_fcc_w = g_pred (x) B, 1024 elements each.
// With `lh.label` set (dsmil_agg_train_step) the kernel FIRST forms the training objective of the bag and its two logit
// gradients — what dsmil_agg_loss_head does (train_tcga.py:67-71) — so the step needs no separate loss-head launch: every
// block derives g_pred for itself (C values), block 0 publishes loss / max_pred / g_pred / g_max.
// The last `fr.blocks` workgroups (dsmil_agg_train_step; 0 otherwise) are step 2 of the backward, gA = V gB^T, in the same
// launch: each forms gB for itself in LDS (block 0's arithmetic) and then runs agg_fwd.hip k_fc<4> over its rows (W := gB, b := 0).
constexpr int FC_ROLE_MAX = 2048;   // C * Kv floats of LDS
struct FcRole { const float* vals; const int64_t* rowmap; float* gA; int blocks; int first; };
struct LossHeadArgs {
    const float* label;     // [C] or null: g_pred comes from the caller
    const float* classes;   // [N,C]
    const float* pred;      // [C], or null: summed here from the forward's partials (k_pred's sum, same order)
    const int64_t* idx;     // [C]
    float* loss; float* max_pred; float* g_pred_out; float* g_max_out;
    const float* pred_part; const float* fcc_b; float* pred_out; int pred_blocks;   // pred == null: [blocks][C][C], [C], [C]
};
__global__ __launch_bounds__(256) void k_bwd_prep(
    const float* __restrict__ fcc_w, const float* __restrict__ Bm, const float* __restrict__ g_pred_in,
    const float* __restrict__ g_B, const float* __restrict__ A, const float* __restrict__ g_A,
    float* __restrict__ gB, float* __restrict__ Dv, float* __restrict__ g_fcc_w, float* __restrict__ g_fcc_b,
    float* __restrict__ zero128, long long N, int Kv, int C, LossHeadArgs lh, FcRole fr) {
    __shared__ float red[4];
    __shared__ float s_gp[64];
    __shared__ __attribute__((aligned(16))) float s_gB[FC_ROLE_MAX];
    const int tid = threadIdx.x;
    const float* g_pred = g_pred_in;
    if (lh.label) {   // C <= 64: one wave (the arithmetic of k_loss_head, agg_fwd.hip)
        if (tid < 64) {
            float l = 0.f;
            if (tid < C) {
                const float y = lh.label[tid];
                const long long ci = lh.idx[tid];          // (requested ahead of the partials: its row is the second hop)
                float zb;
                if (lh.pred) zb = lh.pred[tid];
                else {   // agg_fwd.hip k_pred: fcc_b + the partials in (block, class) order — for C <= 2 eight blocks at a time (every
                         // workgroup of this launch starts with this sum: one load at a time it was 8 dependent round trips)
                    zb = lh.fcc_b[tid];
                    if (C <= 2) {   // eight blocks' partials in flight
                        for (int j0 = 0; j0 < lh.pred_blocks; j0 += 8) {
                            float pv[8][2];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int j = j0 + u < lh.pred_blocks ? j0 + u : lh.pred_blocks - 1;
                                const float* q = lh.pred_part + ((long long)j * C + tid) * C;
                                pv[u][0] = q[0];
                                pv[u][1] = q[C - 1];
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (j0 + u < lh.pred_blocks) {
                                    zb += pv[u][0];
                                    if (C == 2) zb += pv[u][1];
                                }
                        }
                    } else {
                        for (int j = 0; j < lh.pred_blocks; ++j)
                            for (int c = 0; c < C; ++c) zb += lh.pred_part[((long long)j * C + tid) * C + c];
                    }
                    if (blockIdx.x == 0) lh.pred_out[tid] = zb;
                }
                const float zm = lh.classes[ci * (long long)C + tid];
                const float lb = fmaxf(zb, 0.f) - zb * y + log1pf(expf(-fabsf(zb)));
                const float lm = fmaxf(zm, 0.f) - zm * y + log1pf(expf(-fabsf(zm)));
                l = 0.5f * (lb + lm) / (float)C;
                const float sb = 1.f / (1.f + expf(-zb)), sm = 1.f / (1.f + expf(-zm));
                const float gp = 0.5f * (sb - y) / (float)C;
                s_gp[tid] = gp;
                if (blockIdx.x == 0) {
                    if (lh.max_pred) lh.max_pred[tid] = zm;
                    lh.g_pred_out[tid] = gp;
                    lh.g_max_out[tid] = 0.5f * (sm - y) / (float)C;
                }
            }
            l = wave_sum(l);
            if (blockIdx.x == 0 && tid == 0) *lh.loss = l;
        }
        __syncthreads();
        g_pred = s_gp;
    }
    if (fr.blocks && (int)blockIdx.x >= fr.first) {
        for (int i = tid; i < C * Kv; i += 256) {
            const int c = i / Kv, k = i - c * Kv;
            float g = g_B ? g_B[c * Kv + k] : 0.f;
            for (int o = 0; o < C; ++o) g = fmaf(g_pred[o], fcc_w[((long long)o * C + c) * Kv + k], g);
            s_gB[i] = g;
        }
        __syncthreads();
        const int lane = tid & 63;
        const long long nw = (long long)fr.blocks * 4;
        for (long long r = (long long)((int)blockIdx.x - fr.first) * 4 + (tid >> 6); r < N; r += nw) {
            const float* x = fr.vals + phys_row(fr.rowmap, r) * Kv;
            for (int c = 0; c < C; ++c) {
                float acc = 0.f;
                for (int k0 = 0; k0 < Kv; k0 += 256) {
                    const int k = k0 + lane * 4;
                    const f32x4 xv = load4<4>(x, k, Kv);
                    f32x4 wv = {0.f, 0.f, 0.f, 0.f};
                    if (k < Kv) wv = *reinterpret_cast<const f32x4*>(&s_gB[c * Kv + k]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = fmaf(xv[e], wv[e], acc);
                }
                acc = wave_sum(acc) + 0.f;
                if (lane == 0) fr.gA[r * C + c] = acc;
            }
        }
        return;
    }
    if (blockIdx.x > 0) {
        const long long n = (long long)C * C * Kv, i0 = (long long)(blockIdx.x - 1) * 1024;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = i0 + tid + 256 * u;
            if (i < n) {
                const int o = (int)(i / ((long long)C * Kv));
                g_fcc_w[i] = g_pred[o] * Bm[i - (long long)o * C * Kv];
            }
        }
        return;
    }
    for (int c = 0; c < C; ++c) {
        float dpart = 0.f;
        for (int k = tid; k < Kv; k += 256) {
            float g = g_B ? g_B[c * Kv + k] : 0.f;
            for (int o = 0; o < C; ++o) g = fmaf(g_pred[o], fcc_w[((long long)o * C + c) * Kv + k], g);
            gB[c * Kv + k] = g;
            dpart = fmaf(g, Bm[c * Kv + k], dpart);
        }
        if (g_A)
            for (long long n = tid; n < N; n += 256) dpart = fmaf(A[n * C + c], g_A[n * C + c], dpart);
        const float d = block_sum_256(dpart, red);
        if (tid == 0) Dv[c] = d;
    }
    if (tid < C) g_fcc_b[tid] = g_pred[tid];
    if (tid < QD) zero128[tid] = 0.f;
}

// ---- k_train_prologue: everything a training step needs before its first kernel, in one launch: the plane-cut query
//      weights W1 | W2 (forward + backward), W2^T (backward) and the {0, N} offsets of the lone bag --------------------
__global__ void k_train_prologue(const float* __restrict__ q0_w, const float* __restrict__ q2_w, bf16_t* __restrict__ wsplit,
                                 bf16_t* __restrict__ w2t, int K, int nks, int64_t* __restrict__ off_a,
                                 int64_t* __restrict__ off_b, long long N) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    pack_agg_split_range(q0_w, q2_w, wsplit, K, nks, 0, i0, stride);
    if (q2_w) pack_agg_split_range(q2_w, nullptr, w2t, QD, 8, 1, i0, stride);
    if (i0 == 0) { off_a[0] = 0; off_a[1] = N; off_b[0] = 0; off_b[1] = N; }
}

// ---- k_bwd_qrow: q_c = q(x[idx_c]) (dsmil.py:53-54), one workgroup per class ------------------
template <int VEC>
__global__ __launch_bounds__(256) void k_bwd_qrow(
    const float* __restrict__ feats, const int64_t* __restrict__ idx, const float* __restrict__ q0_w,
    const float* __restrict__ q0_b, const float* __restrict__ q2_w, const float* __restrict__ q2_b,
    float* __restrict__ qmax, int K, int nonlinear, const int64_t* __restrict__ rowmap) {
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float s_h[QD];
    const float* x = feats + phys_row(rowmap, idx[c]) * (long long)K;
    for (int jb = 0; jb < 32; jb += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* wr = q0_w + (long long)(wave * 32 + jb) * K;
        for (int k0 = 0; k0 < K; k0 += 256) {
            const int k = k0 + lane * 4;
            const f32x4 xv = load4<VEC, float>(x, k, K);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 wv = load4<VEC, float>(wr + (long long)u * K, k, K);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u] = fmaf(xv[e], wv[e], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float a = wave_sum(acc[u]) + q0_b[wave * 32 + jb + u];
            if (nonlinear) a = fmaxf(a, 0.f);
            if (lane == 0) s_h[wave * 32 + jb + u] = a;
        }
    }
    __syncthreads();
    float* out = qmax + (long long)c * QD;
    if (!nonlinear) {
        if (threadIdx.x < QD) out[threadIdx.x] = s_h[threadIdx.x];
        return;
    }
    const float h0 = s_h[lane], h1 = s_h[lane + 64];
    for (int jj = 0; jj < 32; ++jj) {
        const float* wr = q2_w + (long long)(wave * 32 + jj) * QD;
        const float a = wave_sum(fmaf(h0, wr[lane], h1 * wr[lane + 64])) + q2_b[wave * 32 + jj];
        if (lane == 0) out[wave * 32 + jj] = tanhf(a);
    }
}

// ---- k_bwd_rows ----------------------------------------------------------------------------
struct BwdRowsArgs {
    AttendArgs at;        // feats, offsets (2 entries: 0, N), q-weights (+ packed planes), qmax; bag 0
    const float* A;       // [N,C]
    const float* gA;      // [N,C]  = V gB^T
    const float* g_A;     // [N,C] or null
    const float* Dv;      // [C]
    float* gs;            // [N,C]
    float* gz2;           // [N,128]
    float* Hbuf;          // [N,128]
    float* Qbuf;          // [N,128]
    float* gqp;           // [ceil(N/32), C, 128]: per 32-row wave tile, sum_n gs[n,c] Q[n,:]
};

// stores the hidden layer from the accumulator registers: register 4g+e of H[t] = unit 32t + 8g + 4hi + e of row l31
struct StoreH {
    float* Hbuf;
    long long row;
    bool valid;
    int hi;
    __device__ __forceinline__ void operator()(const f32x16 (&H)[4]) const {
        if (!valid) return;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = H[t][4 * g + e];
                *reinterpret_cast<f32x4*>(Hbuf + row * QD + 32 * t + 8 * g + 4 * hi) = hv;
            }
    }
};

// Sum 64 per-lane values over the 32 lanes of a half-wave by recursive halving: after the exchange with mask 16 a lane
// keeps 32 of its 64 registers (each now a sum over 2 lanes), after mask 8 it keeps 16 ... after mask 1 it keeps 2,
// each the sum over all 32 lanes: 62 exchanges instead of 320.  Lane l31 ends with the sums of indices 2 l31, 2 l31 + 1.
__device__ __forceinline__ void halfwave_colsum64(const float (&v)[64], int l31, float& s0, float& s1) {
    float a32[32], a16[16], a8[8], a4[4];
    {
        const bool b = (l31 >> 4) & 1;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float send = b ? v[i] : v[i + 32], keep = b ? v[i + 32] : v[i];
            a32[i] = keep + __shfl_xor(send, 16, 64);
        }
    }
    {
        const bool b = (l31 >> 3) & 1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float send = b ? a32[i] : a32[i + 16], keep = b ? a32[i + 16] : a32[i];
            a16[i] = keep + __shfl_xor(send, 8, 64);
        }
    }
    {
        const bool b = (l31 >> 2) & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float send = b ? a16[i] : a16[i + 8], keep = b ? a16[i + 8] : a16[i];
            a8[i] = keep + __shfl_xor(send, 4, 64);
        }
    }
    {
        const bool b = (l31 >> 1) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float send = b ? a8[i] : a8[i + 4], keep = b ? a8[i + 4] : a8[i];
            a4[i] = keep + __shfl_xor(send, 2, 64);
        }
    }
    {
        const bool b = l31 & 1;
        const float send0 = b ? a4[0] : a4[2], keep0 = b ? a4[2] : a4[0];
        const float send1 = b ? a4[1] : a4[3], keep1 = b ? a4[3] : a4[1];
        s0 = keep0 + __shfl_xor(send0, 1, 64);
        s1 = keep1 + __shfl_xor(send1, 1, 64);
    }
}

template <int NW, int VEC>
__global__ __launch_bounds__(NW * 64, (NW == 1 ? 1 : 2)) void k_bwd_rows(BwdRowsArgs b) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AttendArgs& a = b.at;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const long long Nb = a.offsets[1] - a.offsets[0];
    const long long row = (long long)blockIdx.x * (NW * 32) + wave * 32 + l31;
    const bool valid = row < Nb;
    f32x16 Q[4];
    const StoreH hook{a.nonlinear ? b.Hbuf : nullptr, row, valid && a.nonlinear, hi};
    if constexpr (VEC == 4) {
        if (!mlp_tile_split_dma<NW, NP_BWD, false>(a, 0, (int)blockIdx.x, smem, Q, hook)) return;
    } else {
        if (!mlp_tile_split<NW, VEC, NP_BWD>(a, 0, (int)blockIdx.x, smem, Q, hook)) return;
    }
    const long long rc = valid ? row : Nb - 1;
    const int C = a.C;
    const float scale = 0.08838834764831845f;  // 1/sqrt(128)
    f32x16 G[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) G[t][r] = 0.f;
    const long long wtile = (long long)blockIdx.x * NW + wave;         // this wave's 32-row tile
    const bool wave_live = wtile * 32 < Nb;                            // (uniform per wave)
    for (int c = 0; c < C; ++c) {
        float ga = b.gA[rc * C + c];
        if (b.g_A) ga += b.g_A[rc * C + c];
        const float gsc = valid ? b.A[rc * C + c] * (ga - b.Dv[c]) * scale : 0.f;
        if (valid && hi == 0) b.gs[row * C + c] = gsc;
        const float* qm = a.qmax + (long long)c * QD;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(qm + 32 * t + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) G[t][4 * g + e] = fmaf(gsc, u[e], G[t][4 * g + e]);
            }
        // this tile's share of g_q[c] = sum_n gs[n,c] Q[n,:] (the gradient of the critical query)
        float v[64];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[16 * t + r] = gsc * Q[t][r];
        float s0, s1;
        halfwave_colsum64(v, l31, s0, s1);
        if (wave_live) {
            const int i = 2 * l31, t = i >> 4, r = i & 15;              // indices i, i+1 = registers r, r+1 of tile t
            float* o = b.gqp + (wtile * C + c) * QD + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
            o[0] = s0;
            o[1] = s1;
        }
    }
    if (!valid) return;
    // gz2 = gQ (1 - Q^2) for the tanh query; rows go to the workspace row-major (4 units per store)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 gz, qv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = Q[t][4 * g + e];
                gz[e] = a.nonlinear ? G[t][4 * g + e] * (1.f - q * q) : G[t][4 * g + e];
                qv[e] = q;
            }
            const long long o = row * QD + 32 * t + 8 * g + 4 * hi;
            *reinterpret_cast<f32x4*>(b.gz2 + o) = gz;
            *reinterpret_cast<f32x4*>(b.Qbuf + o) = qv;
        }
}

// sum_k p[k * stride], k = 0 .. S-1 in order (deterministic), sixteen loads in flight at a time: the additions are a chain, the
// loads are not.  The last round is padded with clamped re-reads that are not added, so that no round degenerates into the
// one-load-at-a-time remainder loop an unrolled run-time trip count leaves (k_bwd_reduce, S = 38: 4 rounds of 8 + SIX dependent
// round trips before: 11.8 -> 5.7 us).
__device__ __forceinline__ float sum_parts16(const float* __restrict__ p, long long stride, int S) {
    float s = 0.f;
    for (int k0 = 0; k0 < S; k0 += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = k0 + j < S ? k0 + j : S - 1;
            v[j] = p[(long long)k * stride];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) s = k0 + j < S ? s + v[j] : s;
    }
    return s;
}

// ---- the same two kernels on the hidden-split tile (agg_hs.h): few rows — a training step on one bag ----------------
// 16 per-lane values summed over the 32 lanes of a half-wave (recursive halving, 15 exchanges); every lane ends with the
// sum of index l31 >> 1
__device__ __forceinline__ float halfwave_colsum16(const float (&v)[16], int l31) {
    float a8[8], a4[4], a2[2];
    {
        const bool b = (l31 >> 4) & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float send = b ? v[i] : v[i + 8], keep = b ? v[i + 8] : v[i];
            a8[i] = keep + __shfl_xor(send, 16, 64);
        }
    }
    {
        const bool b = (l31 >> 3) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float send = b ? a8[i] : a8[i + 4], keep = b ? a8[i + 4] : a8[i];
            a4[i] = keep + __shfl_xor(send, 8, 64);
        }
    }
    {
        const bool b = (l31 >> 2) & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float send = b ? a4[i] : a4[i + 2], keep = b ? a4[i + 2] : a4[i];
            a2[i] = keep + __shfl_xor(send, 4, 64);
        }
    }
    const bool b = (l31 >> 1) & 1;
    const float send = b ? a2[0] : a2[1], keep = b ? a2[1] : a2[0];
    const float a1 = keep + __shfl_xor(send, 2, 64);
    return a1 + __shfl_xor(a1, 1, 64);
}

__global__ __launch_bounds__(HS_THREADS, 2) void k_bwd_rows_hs(BwdRowsArgs b) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AttendArgs& a = b.at;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const long long Nb = a.offsets[1] - a.offsets[0];
    f32x16 Hw[HS_RG], Qw[HS_RG];
    if (!mlp_tile_hs<NP_BWD>(a, 0, (int)blockIdx.x, smem, Hw, Qw)) return;
    const int C = a.C, u0 = 32 * wave + 4 * hi;     // reg 4q+e <-> unit u0 + 8q + e
    const float scale = 0.08838834764831845f;       // 1/sqrt(128)
#pragma unroll
    for (int g = 0; g < HS_RG; ++g) {
        const long long row = (long long)blockIdx.x * HS_BM + 32 * g + l31;
        const long long t32 = (long long)blockIdx.x * HS_RG + g;     // this group's 32-row tile
        if (t32 * 32 >= Nb) break;                                   // (block-uniform)
        const bool valid = row < Nb;
        const long long rc = valid ? row : Nb - 1;
        float G[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) G[r] = 0.f;
        for (int c = 0; c < C; ++c) {
            float ga = b.gA[rc * C + c];
            if (b.g_A) ga += b.g_A[rc * C + c];
            const float gsc = valid ? b.A[rc * C + c] * (ga - b.Dv[c]) * scale : 0.f;
            if (valid && hi == 0 && wave == 0) b.gs[row * C + c] = gsc;
            const float* qm = a.qmax + (long long)c * QD + u0;
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(qm + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    G[4 * q + e] = fmaf(gsc, u[e], G[4 * q + e]);
                    v[4 * q + e] = gsc * Qw[g][4 * q + e];
                }
            }
            // this 32-row tile's share of g_q[c] = sum_n gs[n,c] Q[n,:] for the wave's 32 units
            const float sum = halfwave_colsum16(v, l31);
            if (!(l31 & 1)) {
                const int i = l31 >> 1;   // register index 4q+e
                b.gqp[(t32 * C + c) * QD + u0 + 8 * (i >> 2) + (i & 3)] = sum;
            }
        }
        if (!valid) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 gz, qv, hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float qq = Qw[g][4 * q + e];
                gz[e] = a.nonlinear ? G[4 * q + e] * (1.f - qq * qq) : G[4 * q + e];
                qv[e] = qq;
                hv[e] = Hw[g][4 * q + e];
            }
            const long long o = row * QD + u0 + 8 * q;
            *reinterpret_cast<f32x4*>(b.gz2 + o) = gz;
            *reinterpret_cast<f32x4*>(b.Qbuf + o) = qv;
            if (a.nonlinear) *reinterpret_cast<f32x4*>(b.Hbuf + o) = hv;
        }
    }
}

// g_q[c] = sum over the 32-row tiles of their partials (fixed order); gz2[idx_c] += g_q[c] (1 - Q[idx_c]^2): q_c IS row
// idx_c of Q, so its gradient joins that row.  One workgroup of 1024 threads = 128 units x 8 strided tile groups.
__global__ __launch_bounds__(1024) void k_bwd_critical(const int64_t* __restrict__ idx, const float* __restrict__ gqp,
                                                       const float* __restrict__ Qbuf, float* __restrict__ gz2,
                                                       float* __restrict__ gq, long long ntile, int C, int nonlinear) {
    __shared__ float red[8][QD];
    const int j = threadIdx.x & (QD - 1), grp = threadIdx.x >> 7;
    for (int c = 0; c < C; ++c) {
        const int cnt = ntile > grp ? (int)((ntile - grp + 7) / 8) : 0;   // tiles grp, grp + 8, ...
        const float s = sum_parts16(gqp + ((long long)grp * C + c) * QD + j, 8LL * C * QD, cnt);
        __syncthreads();
        red[grp][j] = s;
        __syncthreads();
        if (grp == 0) {
            const float g = ((red[0][j] + red[1][j]) + (red[2][j] + red[3][j])) + ((red[4][j] + red[5][j]) + (red[6][j] + red[7][j]));
            gq[c * QD + j] = g;
            const long long o = idx[c] * QD + j;
            const float q = Qbuf[o];
            gz2[o] += g * (nonlinear ? (1.f - q * q) : 1.f);   // classes run in order: two classes may share a row
        }
    }
}

// ---- k_bwd_gh: gH = (gz2 W2) [H > 0] — forward GEMM-1 pipeline with X := gz2, W := W2^T ------------
struct GhArgs {
    AttendArgs at;  // feats = gz2 (K = 128), wpk = packed W2^T, q0_b = zeros, nonlinear = 0
    const float* Hbuf;
    float* gH;
};
template <int NW>
__global__ __launch_bounds__(NW * 64, (NW == 1 ? 1 : 2)) void k_bwd_gh(GhArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x16 H[4];
    if (!mlp_tile_split_dma<NW, NP_BWD, false>(g.at, 0, (int)blockIdx.x, smem, H)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
    const long long Nb = g.at.offsets[1] - g.at.offsets[0];
    const long long row = (long long)blockIdx.x * (NW * 32) + wave * 32 + l31;
    if (row >= Nb) return;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            const long long o = row * QD + 32 * t + 8 * gg + 4 * hi;
            const f32x4 hv = *reinterpret_cast<const f32x4*>(g.Hbuf + o);
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = hv[e] > 0.f ? H[t][4 * gg + e] : 0.f;
            *reinterpret_cast<f32x4*>(g.gH + o) = out;
        }
}

__global__ __launch_bounds__(HS_THREADS, 2) void k_bwd_gh_hs(GhArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x16 Hw[HS_RG], Qw[HS_RG];
    if (!mlp_tile_hs<NP_BWD>(g.at, 0, (int)blockIdx.x, smem, Hw, Qw)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
    const long long Nb = g.at.offsets[1] - g.at.offsets[0];
#pragma unroll
    for (int rg = 0; rg < HS_RG; ++rg) {
        const long long row = (long long)blockIdx.x * HS_BM + 32 * rg + l31;
        if (row >= Nb) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long o = row * QD + 32 * wave + 8 * q + 4 * hi;
            const f32x4 hv = *reinterpret_cast<const f32x4*>(g.Hbuf + o);
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = hv[e] > 0.f ? Hw[rg][4 * q + e] : 0.f;
            *reinterpret_cast<f32x4*>(g.gH + o) = out;
        }
    }
}

// ---- k_tn_split: contractions over the instance rows on bf16 MFMA (exact three-plane cuts, six products) -------
//   part0[s][128][K]   = A0[rows of split s]^T x[rows]        (A0 = gH; gz2 for the linear query)
//   part1[s][128][128] = A1[rows]^T Hbuf[rows]                (A1 = gz2; nonlinear query only)
// grid = (slabs of 64 output columns: ceil(K/64) of x, then 2 of H; row ranges).  The contraction index (instance rows)
// runs along the MFMA k axis, i.e. along a lane's registers, while memory holds rows along the slow axis: the staging
// threads therefore own one COLUMN each and 8 consecutive rows (8 coalesced dword loads: a wave covers 256 B of a row
// per instruction), cut the 8 values into three bf16 planes (split3: each element cut once) and write each plane's
// 16 bytes to sX[plane][column][8 rows] with one ds_write_b128; a fragment read is one ds_read_b128.  Row stride 80 B:
// both are conflict-free (16 consecutive columns hit 16 distinct 4-bank groups).  Global loads of step s+1 are issued
// before the MFMAs of step s.  Wave w = (column tile w & 1, unit-tile pair w >> 1): 2 accumulator tiles, 24 MFMAs per
// 32 rows.  Column sums of A0 / A1 (bias gradients) come from the staged values of the first slab of each kind.
constexpr int TN_LDW = 20;      // 32-bit words per (plane, column) row: 16 row pairs + 4 pad
constexpr int TN_WGS = 384;     // target workgroups per launch: sets the number of row ranges (fewer = fewer partials to reduce)
struct TnArgs {
    const float* A0;
    const float* A1;
    const float* X;
    const float* Hb;
    const int64_t* rowmap;
    float* part0;
    float* part1;
    float* pb0;   // [S][128]
    float* pb1;   // [S][128]
    long long N;
    int K, R, nx, nslab, S;
    unsigned long long* trace;   // trace builds (tools/stamp_tn.py): [workgroup][16] s_memtime stamps of wave 0; else null
};

// 1-D grid of nslab x (S rounded up to 8) workgroups.  Workgroup L runs on XCD L % 8 (round-robin dispatch); the slabs of
// ONE row range all read the same A rows, so they are given to consecutive workgroups of the SAME XCD — its L2 then serves
// the A tile to nine of the ten slabs (with (slab, split) = (L % nslab, L / nslab) the slabs of a range sat on eight
// different L2s: 75 MB of fabric reads per launch for 35 MB of operands).
// WIDE (rows 16-B aligned, K % 4 == 0): a staging thread loads 16 B — 4 columns x 8 rows = 8 loads instead of 32 — and the
// LDS holds the columns permuted (column 4g + c at position 32c + g, 16c + g for B) so that consecutive lanes still write
// consecutive positions; the epilogue undoes the permutation.  26.4 -> 23.5 us for the 10 000-row bag (staging-, not MFMA-bound: 3.3 us of MFMA time).
// MAP (a row map on X: dropout_patches): the physical rows of a prefetch are themselves loaded ONE prefetch ahead and IN FRONT of
// that call's data loads.  Vector memory returns in order: waiting for map entries requested after the previous step's data
// (as the first form did — and, the wait being emitted behind the `if (map)`, also when there was no map) drained the whole
// queue at every prefetch, so only one step's loads were ever in flight.
template <bool WIDE, bool MAP>
__global__ __launch_bounds__(256, 2) void k_tn_split(TnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned sA[3 * QD * TN_LDW];
    __shared__ __attribute__((aligned(16))) unsigned sB[3 * 64 * TN_LDW];
    __shared__ float s_cs[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int slab = q % a.nslab, split = xcd + 8 * (q / a.nslab);
    if (split >= a.S) return;
    const bool is_h = slab >= a.nx;
    const float* Am = is_h ? a.A1 : a.A0;
    const float* Bm = is_h ? a.Hb : a.X;
    const int ldb = is_h ? QD : a.K;
    const int col0 = (is_h ? slab - a.nx : slab) * 64;
    const int64_t* bmap = is_h ? nullptr : a.rowmap;
    const long long rbeg = (long long)split * a.R, rend = (rbeg + a.R < a.N) ? rbeg + a.R : a.N;
    const bool want_cs = (slab == 0) || (slab == a.nx);
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int ct = wave & 1, up = wave >> 1;
    constexpr int P0 = 9 - NP_BWD;
    // the MFMA phase of a 32-row step: the same for both staging forms (positions, not columns, index the LDS rows)
    auto mfma_phase = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int j = 2 * ks + hi;    // this lane's 8-row group: MFMA k = 8 hi + i  <->  row 16 ks + 8 hi + i
            S3Frag fb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[p].f = *reinterpret_cast<const f32x4*>(&sB[(p * 64 + 32 * ct + l31) * TN_LDW + 4 * j]);
            // (the two accumulators in turn: back to back on ONE accumulator a 32x32x16 MFMA issues every ~60 cycles instead of
            // 32 — measured on k_attend_f3; the products of each accumulator keep their order, the results their bits)
            S3Frag fa[2][3];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fa[tt][p].f = *reinterpret_cast<const f32x4*>(&sA[(p * QD + 32 * (2 * up + tt) + l31) * TN_LDW + 4 * j]);
#pragma unroll
            for (int qq = P0; qq < 9; ++qq)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tt][S3_PA(qq)].v, fb[S3_PB(qq)].v, acc[tt], 0, 0, 0);
        }
    };
    if constexpr (WIDE) {
        // staging roles: threads 0..127: A, columns 4g..4g+3 (g = tid & 31), 8-row group tid >> 5; threads 128..191: B, columns
        // col0 + 4g.. (g = tid & 15), 8-row group (tid >> 4) & 3; the fourth wave only multiplies
        const bool roleA = tid < 128, roleB = tid >= 128 && tid < 192;
        const int g = roleA ? (tid & 31) : (tid & 15), jg = roleA ? (tid >> 5) : ((tid >> 4) & 3);
        int bc = col0 + 4 * g;
        bc = bc < ldb ? bc : ldb - 4;                      // (a clamped column's products are never stored)
        const float* base = roleA ? Am + 4 * g : Bm + bc;
        const int ld = roleA ? QD : ldb;
        const int64_t* map = roleA ? nullptr : bmap;
        f32x4 v[2][8];
        long long nmap[8];                                 // MAP, B role: physical rows of the NEXT prefetch
        auto map_load = [&](long long r0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long long r = r0 + 8 * jg + e;
                nmap[e] = (long long)map[r < rend ? r : rend - 1];
            }
        };
        // The code below is shaped for hipcc's wait-count pass, which is exact only on straight-line code:
        // * the staging waves (0-2) and the multiply-only wave (3) take SEPARATE loops — with the staging behind an
        //   `if (role)` inside one loop, the pass assumed at the join that the refilled set's previous loads might still be in
        //   flight and waited vmcnt(7..0) before forming the new addresses: every prefetch drained the queue;
        // * steps always come in PAIRS (set 0, set 1; R is a multiple of 64, a range's tail is padded with a step whose A rows
        //   are zero) — behind `if (r0 + 32 < rend) step<1>` the wait for set 0 had to assume that no younger loads exist
        //   (vmcnt(0)) and drained set 1 as well;
        // * four pairs are unrolled straight-line: at a loop HEADER the pass merges the entry state with the back edge's and
        //   again waits vmcnt(0) for the older set (once per 256 rows now: once per workgroup for the 10 000-row bag).
        // Together: one step's loads in flight instead of two (24 us for the 10 000-row bag before).
        auto prefetch = [&](auto setc, long long r0) {     // branch-free: rows past the range re-read its last row
            constexpr int SET = decltype(setc)::value;
            long long pr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long long r = r0 + 8 * jg + e;
                pr[e] = r < rend ? r : rend - 1;
            }
            if constexpr (MAP) {
                if (roleB && map) {                        // (the H slabs have no map)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pr[e] = nmap[e];
                    map_load(r0 + 32);                     // in front of this call's data loads
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[SET][e] = *(const DSMIL_GLOBAL f32x4*)(base + pr[e] * (long long)ld);
        };
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
        auto step = [&](auto setc, long long r0) {
            constexpr int SET = decltype(setc)::value;
            unsigned* dstb = roleA ? sA : sB;
            const int npos = roleA ? QD : 64, cstride = roleA ? 32 : 16;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float xv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[e] = (roleA && r0 + 8 * jg + e >= rend) ? 0.f : v[SET][e][c];   // A rows past the range
                if (want_cs && roleA) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs[c] += xv[e];
                }
                S3Frag f[3];
                split3(xv, f);
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<f32x4*>(&dstb[(p * npos + c * cstride + g) * TN_LDW + 4 * jg]) = f[p].f;
            }
            __syncthreads();
            prefetch(setc, r0 + 64);
            mfma_phase();
            __syncthreads();
        };
#ifdef DSMIL_TRACE
        int tn_i = 0;
        auto TSTAMP = [&]() { if (a.trace && tid == 0 && tn_i < 16) a.trace[(long long)blockIdx.x * 16 + tn_i++] = __builtin_amdgcn_s_memtime(); };
#else
        auto TSTAMP = []() {};
#endif
        if (wave < 3) {
            TSTAMP();                                     // 0: entry
            if constexpr (MAP) {
                if (roleB && map) map_load(rbeg);
            }
            prefetch(std::integral_constant<int, 0>{}, rbeg);
            prefetch(std::integral_constant<int, 1>{}, rbeg + 32);
            TSTAMP();                                     // 1: first loads issued
            for (long long r0 = rbeg; r0 < rend; r0 += 256) {   // four pairs straight-line, forward exits only (see above)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    if (r0 + 64 * h >= rend) break;
                    step(std::integral_constant<int, 0>{}, r0 + 64 * h);
                    TSTAMP();                             // 2, 4, ...: even steps done
                    step(std::integral_constant<int, 1>{}, r0 + 64 * h + 32);
                    TSTAMP();                             // 3, 5, ...: odd steps done
                }
            }
        } else {
            for (long long r0 = rbeg; r0 < rend; r0 += 64) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    __syncthreads();
                    mfma_phase();
                    __syncthreads();
                }
            }
        }
        // D[m position][n position]: position -> unit 4 (m & 31) + (m >> 5), column col0 + 4 (n & 15) + (n >> 4)
        const int npos_ = 32 * ct + l31, col = col0 + 4 * (npos_ & 15) + (npos_ >> 4);
        if (col < ldb) {
            float* o = is_h ? a.part1 + (long long)split * QD * QD : a.part0 + (long long)split * QD * a.K;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mp = 32 * (2 * up + tt) + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    o[(long long)(4 * (mp & 31) + (mp >> 5)) * ldb + col] = acc[tt][r];
                }
        }
        if (want_cs) {
            if (roleA) {
#pragma unroll
                for (int c = 0; c < 4; ++c) s_cs[jg * QD + 4 * g + c] = cs[c];
            }
            __syncthreads();
            if (tid < QD) (is_h ? a.pb1 : a.pb0)[split * QD + tid] = (s_cs[tid] + s_cs[QD + tid]) + (s_cs[2 * QD + tid] + s_cs[3 * QD + tid]);
        }
#ifdef DSMIL_TRACE
        if (a.trace && tid == 0) a.trace[(long long)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memtime();   // 15: epilogue stores issued
#endif
    } else {
        // staging roles: A: column u, 8-row groups ja, ja + 2; B: column cb, 8-row group jb
        const int u = tid & 127, ja = tid >> 7;
        const int cb = tid & 63, jb = tid >> 6;
        const bool bcol_ok = col0 + cb < ldb;
        // two register sets: the loads of steps s+1 AND s+2 are in flight while step s runs
        float ra[2][2][8], rb[2][8];
        const int bcol = bcol_ok ? col0 + cb : ldb - 1;    // (a clamped column's products are never stored)
        // Branch-free: rows past the range read its last row (the A values are zeroed when they are cut), so the 24 loads of a
        // step issue back to back — behind a per-row bounds branch hipcc had waited for each one (and for the row-map entry
        // in front of it) before issuing the next: 33 us for 75 MB.
        auto prefetch = [&](auto setc, long long r0) {
            constexpr int SET = decltype(setc)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    long long r = r0 + 8 * (ja + 2 * i) + e;
                    r = r < rend ? r : rend - 1;
                    ra[SET][i][e] = Am[r * QD + u];
                }
            long long pr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long long r = r0 + 8 * jb + e;
                pr[e] = r < rend ? r : rend - 1;
            }
            if constexpr (MAP) {   // (the eight row-map entries load back to back; this form waits for them at once)
                if (bmap) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pr[e] = (long long)bmap[pr[e]];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) rb[SET][e] = Bm[pr[e] * (long long)ldb + bcol];
        };
        float colsum = 0.f;
        auto step = [&](auto setc, long long r0) {
            constexpr int SET = decltype(setc)::value;
            // cut the staged values into planes, transposed into LDS
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (r0 + 8 * (ja + 2 * i) + e >= rend) ra[SET][i][e] = 0.f;   // rows past the range (wave-uniform)
                S3Frag f[3];
                split3(ra[SET][i], f);
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<f32x4*>(&sA[(p * QD + u) * TN_LDW + 4 * (ja + 2 * i)]) = f[p].f;
                if (want_cs) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) colsum += ra[SET][i][e];
                }
            }
            {
                S3Frag f[3];
                split3(rb[SET], f);
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<f32x4*>(&sB[(p * 64 + cb) * TN_LDW + 4 * jb]) = f[p].f;
            }
            __syncthreads();
            prefetch(setc, r0 + 64);   // (rows past the range re-read its last row)
            mfma_phase();
            __syncthreads();
        };
        prefetch(std::integral_constant<int, 0>{}, rbeg);
        prefetch(std::integral_constant<int, 1>{}, rbeg + 32);
        for (long long r0 = rbeg; r0 < rend; r0 += 64) {
            step(std::integral_constant<int, 0>{}, r0);
            if (r0 + 32 < rend) step(std::integral_constant<int, 1>{}, r0 + 32);
        }
        // D[m = unit][n = column]: lane holds column 32 ct + l31, units 32 (2 up + tt) + (r & 3) + 8 (r >> 2) + 4 hi
        const int col = col0 + 32 * ct + l31;
        if (col < ldb) {
            float* o = is_h ? a.part1 + (long long)split * QD * QD : a.part0 + (long long)split * QD * a.K;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = 32 * (2 * up + tt) + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    o[(long long)m * ldb + col] = acc[tt][r];
                }
        }
        if (want_cs) {
            s_cs[tid] = colsum;
            __syncthreads();
            if (tid < QD) (is_h ? a.pb1 : a.pb0)[split * QD + tid] = s_cs[tid] + s_cs[tid + QD];
        }
    }
}

// ---- k_bwd_reduce: fixed-order sums over the row ranges (deterministic), every output of the query stream in one
//      launch, plus the sparse max-stream gradient of the FCLayer (train_tcga.py:68,70: only the critical rows carry
//      gradient: g_fc_w[c] (+)= g_max[c] x[idx_c], g_fc_b[c] (+)= g_max[c]) -------------------------------------
// One element of torch.optim.Adam (amsgrad = False, maximize = False), the operation order of torch's single-tensor /
// foreach implementation: g += wd p; m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g^2; p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
struct AdamScalars { float step_size, w1, beta2, w2, eps, wd, bc2_sqrt; };
__device__ __forceinline__ void adam_elem(float* __restrict__ p, float* __restrict__ mp, float* __restrict__ vp, float g,
                                          const AdamScalars& h) {
    // every fused multiply-add is spelled out: left to the compiler's contraction, the choice (e.g. which of the two products
    // of the second moment is rounded first) changes with the code around the call, and the update with it by an ulp
    const float pv = *p;
    if (h.wd != 0.f) g = fmaf(h.wd, pv, g);                   // grad.add(param, alpha = weight_decay)
    const float m0 = *mp;                                     // exp_avg.lerp_(grad, 1 - beta1): torch's two-sided formula
    const float m = h.w1 < 0.5f ? fmaf(h.w1, g - m0, m0) : fmaf(-(1.f - h.w1), g - m0, g);
    const float v = fmaf(h.w2, g * g, *vp * h.beta2);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    *mp = m;
    *vp = v;
    const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
    *p = fmaf(-h.step_size, m / denom, pv);                   // param.addcdiv_(exp_avg, denom, value = -step_size)
}
// dsmil_agg_train_step: optimizer.step() inside k_bwd_reduce — the thread that has just formed a gradient element applies
// it (every tensor's last reader is an earlier launch); tensors in parameter order fc_w, fc_b, q0_w, q0_b, q2_w, q2_b,
// fcc_w, fcc_b; the two fcc gradients (k_bwd_prep's) ride along as extra elements
struct AdamFuse {
    int on;
    float* p[8]; float* m[8]; float* v[8];
    AdamScalars h;
    const float* g_fcc_w; const float* g_fcc_b; long long n_fcc_w; int n_fcc_b;
};
struct ReduceArgs {
    const float* part0; const float* part1; const float* pb0; const float* pb1;
    float* g_w0; float* g_b0;    // outputs of (part0, pb0): g_q0_w [128,K], g_q0_b
    float* g_w1; float* g_b1;    // outputs of (part1, pb1): g_q2_w [128,128], g_q2_b (nonlinear only)
    int S, K, nonlinear;
    // sparse FCLayer gradient (g_max != null)
    const float* feats; const int64_t* idx; const float* g_max; const int64_t* rowmap;
    float* g_fc_w; float* g_fc_b; int C, accumulate;
    AdamFuse af;
};
__global__ __launch_bounds__(256) void k_bwd_reduce(ReduceArgs a) {
    const long long n0 = (long long)QD * a.K, n1 = a.nonlinear ? (long long)QD * QD : 0;
    const long long nb = QD * (a.nonlinear ? 2 : 1), nf = a.g_max ? (long long)a.C * a.K + a.C : 0;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n0) {
        const float s = sum_parts16(a.part0 + i, n0, a.S);
        a.g_w0[i] = s;
        if (a.af.on) adam_elem(a.af.p[2] + i, a.af.m[2] + i, a.af.v[2] + i, s, a.af.h);
        return;
    }
    i -= n0;
    if (i < n1) {
        const float s = sum_parts16(a.part1 + i, n1, a.S);
        a.g_w1[i] = s;
        if (a.af.on) adam_elem(a.af.p[4] + i, a.af.m[4] + i, a.af.v[4] + i, s, a.af.h);
        return;
    }
    i -= n1;
    if (i < nb) {
        const float* pb = i < QD ? a.pb0 : a.pb1;
        const int j = (int)(i & (QD - 1));
        const float s = sum_parts16(pb + j, QD, a.S);
        (i < QD ? a.g_b0 : a.g_b1)[j] = s;
        if (a.af.on) {
            const int t = i < QD ? 3 : 5;
            adam_elem(a.af.p[t] + j, a.af.m[t] + j, a.af.v[t] + j, s, a.af.h);
        }
        return;
    }
    i -= nb;
    if (i < nf) {
        if (i < (long long)a.C * a.K) {
            const int c = (int)(i / a.K), k = (int)(i - (long long)c * a.K);
            const float v = a.g_max[c] * a.feats[phys_row(a.rowmap, a.idx[c]) * (long long)a.K + k];
            const float g = a.accumulate ? a.g_fc_w[i] + v : v;
            a.g_fc_w[i] = g;
            if (a.af.on) adam_elem(a.af.p[0] + i, a.af.m[0] + i, a.af.v[0] + i, g, a.af.h);
        } else {
            const int c = (int)(i - (long long)a.C * a.K);
            const float g = a.accumulate ? a.g_fc_b[c] + a.g_max[c] : a.g_max[c];
            a.g_fc_b[c] = g;
            if (a.af.on) adam_elem(a.af.p[1] + c, a.af.m[1] + c, a.af.v[1] + c, g, a.af.h);
        }
        return;
    }
    i -= nf;
    if (a.af.on) {   // the bag stream's Conv1d: gradients formed by k_bwd_prep
        if (i < a.af.n_fcc_w) adam_elem(a.af.p[6] + i, a.af.m[6] + i, a.af.v[6] + i, a.af.g_fcc_w[i], a.af.h);
        else if (i - a.af.n_fcc_w < a.af.n_fcc_b) {
            const long long j = i - a.af.n_fcc_w;
            adam_elem(a.af.p[7] + j, a.af.m[7] + j, a.af.v[7] + j, a.af.g_fcc_b[j], a.af.h);
        }
    }
}

// out[i] = sum_s part[s][i]  (fixed order: deterministic) — the dense instance-logit gradient only
__global__ void k_reduce_parts(const float* __restrict__ part, float* __restrict__ out, int S, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += part[(long long)k * n + i];
        out[i] = s;
    }
}

// ---- k_tn_small: part[s][M][Kc] = a[rows][M]^T b[rows][Kc] for a handful of columns M (classes);
//      also part_a[s][M] = column sums of a.  grid = (row ranges, 256-wide k-segments); the 4 waves
//      stride the rows (b row read once for up to 4 classes), then merge through LDS.  Only the DENSE instance-logit
//      gradient (g_classes: never requested by the reference's training loop) takes it. -------------
template <int VEC>
__global__ __launch_bounds__(256) void k_tn_small(const float* __restrict__ a, const float* __restrict__ bm,
                                                  float* __restrict__ part, float* __restrict__ part_a,
                                                  long long N, int M, int Kc, int TNR, const int64_t* __restrict__ bmap) {
    __shared__ __attribute__((aligned(16))) float sacc[4][4][256];
    __shared__ float ssum[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int split = blockIdx.x;
    const int k = blockIdx.y * 256 + lane * 4;
    const long long rbeg = (long long)split * TNR, rend = (rbeg + TNR < N) ? rbeg + TNR : N;
    for (int m0 = 0; m0 < M; m0 += 4) {
        f32x4 acc[4];
        float asum[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; asum[j] = 0.f; }
#pragma unroll 4
        for (long long r = rbeg + wave; r < rend; r += 4) {
            const f32x4 bv = load4<VEC, float>(bm + phys_row(bmap, r) * (long long)Kc, k, Kc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float av = (m0 + j < M) ? a[r * M + m0 + j] : 0.f;
                acc[j] += av * bv;
                asum[j] += av;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&sacc[wave][j][lane * 4]) = acc[j];
        if (lane == 0)
#pragma unroll
            for (int j = 0; j < 4; ++j) ssum[wave][j] = asum[j];
        __syncthreads();
        const int m = m0 + wave;  // wave j merges class m0 + j
        if (m < M) {
            const f32x4 v = (*reinterpret_cast<const f32x4*>(&sacc[0][wave][lane * 4]) +
                             *reinterpret_cast<const f32x4*>(&sacc[1][wave][lane * 4])) +
                            (*reinterpret_cast<const f32x4*>(&sacc[2][wave][lane * 4]) +
                             *reinterpret_cast<const f32x4*>(&sacc[3][wave][lane * 4]));
            float* o = part + ((long long)split * M + m) * Kc;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + e < Kc) o[k + e] = v[e];
            if (part_a && blockIdx.y == 0 && lane == 0)
                part_a[split * M + m] = (ssum[0][wave] + ssum[1][wave]) + (ssum[2][wave] + ssum[3][wave]);
        }
    }
}

// g_vals[n][k] = sum_c A[n][c] gB[c][k]   (B = A^T V  =>  dV = A dB)
template <int VEC>
__global__ void k_bwd_gvals(const float* __restrict__ A, const float* __restrict__ gB, float* __restrict__ gv,
                            long long N, int Kv, int C) {
    const int k4n = (Kv + 3) / 4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * k4n) return;
    const long long n = i / k4n;
    const int k = (int)(i - n * k4n) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) acc += A[n * C + c] * load4<VEC, float>(gB + (long long)c * Kv, k, Kv);
    if constexpr (VEC == 4) {
        *reinterpret_cast<f32x4*>(gv + n * Kv + k) = acc;
    } else {
        for (int e = 0; e < 4; ++e)
            if (k + e < Kv) gv[n * Kv + k + e] = acc[e];
    }
}

// ---- Adam over all parameter tensors of the step in one launch (torch.optim.Adam, amsgrad = False, maximize = False:
//      g += wd p; m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps),
//      the operation order of torch's single-tensor / foreach implementation) --------------------------------------
struct AdamTensors {
    float* p[DSMIL_ADAM_MAX_TENSORS];
    float* m[DSMIL_ADAM_MAX_TENSORS];
    float* v[DSMIL_ADAM_MAX_TENSORS];
    const float* g[DSMIL_ADAM_MAX_TENSORS];
    long long end[DSMIL_ADAM_MAX_TENSORS];   // running element counts
    int n;
};
__global__ __launch_bounds__(256) void k_adam(AdamTensors t, float step_size, float w1, float beta2, float w2, float eps,
                                              float wd, float bc2_sqrt) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= t.end[t.n - 1]) return;
    int k = 0;
#pragma unroll
    for (int j = 0; j < DSMIL_ADAM_MAX_TENSORS - 1; ++j)
        if (j < t.n - 1 && i >= t.end[j]) k = j + 1;
    const long long o = i - (k ? t.end[k - 1] : 0);
    adam_elem(t.p[k] + o, t.m[k] + o, t.v[k] + o, t.g[k][o], AdamScalars{step_size, w1, beta2, w2, eps, wd, bc2_sqrt});
}

#ifdef DSMIL_TRACE
constexpr size_t TN_TRACE_WORDS = 1024 * 16;
inline unsigned long long* tn_trace_buffer() {
    static unsigned long long* buf = [] {
        unsigned long long* p = nullptr;
        (void)hipMalloc(&p, TN_TRACE_WORDS * 8);
        (void)hipMemset(p, 0, TN_TRACE_WORDS * 8);
        return p;
    }();
    return buf;
}
#endif
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
struct BwdWs {
    size_t gB, Dv, zero, qmax, gq, gA, gs, gz2, Hb, Qb, gH, gqp, wsplit, w2t, part0, part1, pb0, pb1, part, part_b, off, total;
    int splits, rows;   // k_tn_small (dense instance-logit gradient)
    int S, R, nx;       // k_tn_split
    long long T32;
};
BwdWs bwd_layout(long long N, int K, int Kv, int C, int nonlinear) {
    BwdWs w;
    w.rows = tn_rows(N);
    w.splits = (int)((N + w.rows - 1) / w.rows);
    w.nx = (K + 63) / 64;
    const int nslab = w.nx + (nonlinear ? 2 : 0);
    long long st = TN_WGS / nslab;
    if (st < 1) st = 1;
    if (st > 256) st = 256;
    long long R = ((N + st - 1) / st + 32) / 64 * 64;   // nearest multiple of 64: k_tn_split takes its 32-row steps in pairs
    if (R < 64) R = 64;
    w.R = (int)R;
    w.S = (int)((N + R - 1) / R);
    w.T32 = (N + 31) / 32;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t p = o; o = al(o + bytes); return p; };
    w.gB = take((size_t)C * Kv * 4);
    w.Dv = take((size_t)C * 4);
    w.zero = take(QD * 4);
    w.qmax = take((size_t)C * QD * 4);
    w.gq = take((size_t)C * QD * 4);
    w.gA = take((size_t)N * C * 4);
    w.gs = take((size_t)N * C * 4);
    w.gz2 = take((size_t)N * QD * 4);
    w.Hb = take((size_t)N * QD * 4);
    w.Qb = take((size_t)N * QD * 4);
    w.gH = take((size_t)N * QD * 4);
    w.gqp = take((size_t)w.T32 * C * QD * 4);
    w.wsplit = take((size_t)(2 * ((K + 31) / 32) + 8) * S3_CHUNK_F4 * 16);
    w.w2t = take((size_t)8 * S3_CHUNK_F4 * 16);
    w.part0 = take((size_t)w.S * QD * K * 4);
    w.part1 = take((size_t)w.S * QD * QD * 4);
    w.pb0 = take((size_t)w.S * QD * 4);
    w.pb1 = take((size_t)w.S * QD * 4);
    w.part = take((size_t)w.splits * (C > 4 ? C : 4) * (K > QD ? K : QD) * 4);   // k_tn_small: [splits][C][K]
    w.part_b = take((size_t)w.splits * QD * 4);
    w.off = take(2 * sizeof(int64_t));
    w.total = o;
    return w;
}

__global__ void k_set_offsets(int64_t* off, long long N) { off[0] = 0; off[1] = N; }

template <typename KernelT, typename ArgT>
int launch_tile_kernel(KernelT kern, const ArgT& arg, int nw, bool dma, long long N, hipStream_t st) {
    const int BM = nw * 32;
    const size_t lds = dma ? (size_t)(3 * S3_CHUNK_F4 * 4 + 2 * BM * 32) * sizeof(float)
                           : (size_t)(2 * S3_CHUNK_F4 * 4 + 2 * BM * LDK) * sizeof(float);
    if (!dsmil_lds::allow((const void*)kern, (int)lds)) return DSMIL_E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)((N + BM - 1) / BM)), dim3(nw * 64), lds, st, arg);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

int tn_small(const float* a, const float* bm, long long N, int M, int Kc, float* part, float* part_a, float* out,
             float* out_a, int splits, bool v4, hipStream_t st, const int64_t* bmap = nullptr) {
    const dim3 grid((unsigned)splits, (unsigned)((Kc + 255) / 256));
    if (v4) hipLaunchKernelGGL(k_tn_small<4>, grid, dim3(256), 0, st, a, bm, part, part_a, N, M, Kc, tn_rows(N), bmap);
    else hipLaunchKernelGGL(k_tn_small<1>, grid, dim3(256), 0, st, a, bm, part, part_a, N, M, Kc, tn_rows(N), bmap);
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    const long long n = (long long)M * Kc;
    hipLaunchKernelGGL(k_reduce_parts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, out, splits, n);
    if (out_a) hipLaunchKernelGGL(k_reduce_parts, dim3(1), dim3(64), 0, st, part_a, out_a, splits, (long long)M);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

// packed_split: the forward's plane-cut W1 | W2 image (dsmil_agg_pack_split layout) or null; qmax_in: the forward's
// q_max [C,128] or null — both are recomputed here when absent
int agg_backward_impl(const float* feats, const float* vals, int64_t N, const dsmil_agg_params* p,
                      const float* A, const float* Bm, const int64_t* idx, const float* g_classes,
                      const float* g_max, const float* g_pred, const float* g_A, const float* g_B,
                      const dsmil_agg_grads* g, float* g_vals, const int64_t* rowmap, void* ws, size_t ws_bytes,
                      void* stream, const void* packed_split, const float* qmax_in, bool prepared = false,
                      const LossHeadArgs* lhp = nullptr, const AdamFuse* adam = nullptr, bool own_fc_launch = false) {
    if (adam && (!g_max || g_classes)) return DSMIL_E_INVALID;   // the fused optimizer step is the training loop's
    if (!feats || !p || !A || !Bm || !idx || (!g_pred && !lhp) || !g || !ws) return DSMIL_E_INVALID;
    if (g_max && (!g->fc_w || !g->fc_b)) return DSMIL_E_INVALID;
    if (N <= 0 || p->K <= 0 || p->Kv <= 0 || p->C <= 0) return DSMIL_E_INVALID;
    if (!p->q0_w || !p->q0_b || !p->fcc_w || (p->nonlinear && (!p->q2_w || !p->q2_b))) return DSMIL_E_INVALID;
    if (!g->q0_w || !g->q0_b || !g->fcc_w || !g->fcc_b || (p->nonlinear && (!g->q2_w || !g->q2_b))) return DSMIL_E_INVALID;
    if (g_classes && (!g->fc_w || !g->fc_b)) return DSMIL_E_INVALID;
    if (!vals) vals = feats;
    if (vals == feats && p->Kv != p->K) return DSMIL_E_INVALID;
    if (((uintptr_t)ws % 256) || ((uintptr_t)p->q0_b % 16) || (p->nonlinear && ((uintptr_t)p->q2_b % 16))) return DSMIL_E_ALIGN;
    if (packed_split && ((uintptr_t)packed_split % 16)) return DSMIL_E_ALIGN;
    const int K = p->K, Kv = p->Kv, C = p->C;
    const BwdWs L = bwd_layout(N, K, Kv, C, p->nonlinear);
    if (ws_bytes < L.total) return DSMIL_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* w8 = (char*)ws;
    float* gB = (float*)(w8 + L.gB); float* Dv = (float*)(w8 + L.Dv); float* zero = (float*)(w8 + L.zero);
    float* qmax = (float*)(w8 + L.qmax); float* gq = (float*)(w8 + L.gq);
    float* gA = (float*)(w8 + L.gA); float* gs = (float*)(w8 + L.gs); float* gz2 = (float*)(w8 + L.gz2);
    float* Hb = (float*)(w8 + L.Hb); float* Qb = (float*)(w8 + L.Qb); float* gH = (float*)(w8 + L.gH);
    float* gqp = (float*)(w8 + L.gqp);
    bf16_t* wsplit = (bf16_t*)(w8 + L.wsplit); bf16_t* w2t = (bf16_t*)(w8 + L.w2t);
    float* part0 = (float*)(w8 + L.part0); float* part1 = (float*)(w8 + L.part1);
    float* pb0 = (float*)(w8 + L.pb0); float* pb1 = (float*)(w8 + L.pb1);
    float* part = (float*)(w8 + L.part); float* part_b = (float*)(w8 + L.part_b);
    int64_t* off = (int64_t*)(w8 + L.off);
    // the DMA tile needs 16-B aligned rows; K % 4 != 0 (MUSK: 166) takes the register-staged tile
    const bool v4 = (K % 4 == 0) && (((uintptr_t)feats) % 16 == 0);
    const bool w4 = (K % 4 == 0) && (((uintptr_t)feats | (uintptr_t)p->q0_w) % 16 == 0);
    const bool v4v = (Kv % 4 == 0) && ((uintptr_t)vals % 16 == 0);
    int rc;
    // 0. plane-cut weights: W1 | W2 unless the forward's image was handed in; W2^T for the gH pipeline
    // (`prepared`: dsmil_agg_train_step's prologue already filled wsplit, w2t and off of THIS workspace)
    const int nks = 2 * ((K + 31) / 32);
    if (prepared) packed_split = wsplit;
    else if (!packed_split || p->nonlinear) {   // (a nonlinear query needs W2^T from here anyway: the caller's image is then not used)
        hipLaunchKernelGGL(k_train_prologue, dim3(240), dim3(256), 0, st, p->q0_w, p->nonlinear ? p->q2_w : nullptr, wsplit, w2t, K, nks,
                           off, off, (long long)N);
        packed_split = wsplit;
    } else {
        hipLaunchKernelGGL(k_set_offsets, dim3(1), dim3(1), 0, st, off, (long long)N);
    }
    // 1. head: gB, D, g_fcc_*
    const long long nfcc = (long long)C * C * Kv;
    const int prep_blocks = (int)(1 + (nfcc + 1023) / 1024);
    FcRole fr{vals, rowmap, gA, 0, prep_blocks};
    if (lhp && !own_fc_launch && (Kv % 4 == 0) && ((uintptr_t)vals % 16 == 0) && (long long)C * Kv <= FC_ROLE_MAX) {   // the training step: one launch less
        long long fb = ((long long)N + 3) / 4;
        fr.blocks = (int)(fb > 4096 ? 4096 : fb);
    }
    hipLaunchKernelGGL(k_bwd_prep, dim3((unsigned)(prep_blocks + fr.blocks)), dim3(256), 0, st, p->fcc_w, Bm, g_pred, g_B, A, g_A,
                       gB, Dv, g->fcc_w, g->fcc_b, zero, (long long)N, Kv, C, lhp ? *lhp : LossHeadArgs{}, fr);
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    // 2. gA = V gB^T  (the forward's FCLayer kernel with W := gB, b := 0)
    if (!fr.blocks) {
        rc = dsmil_fc_forward_rows(vals, N, Kv, C, gB, zero, gA, rowmap, stream);
        if (rc) return rc;
    }
    // 3. critical queries
    if (qmax_in) qmax = const_cast<float*>(qmax_in);
    else {
        if (w4) hipLaunchKernelGGL(k_bwd_qrow<4>, dim3((unsigned)C), dim3(256), 0, st, feats, idx, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, K, p->nonlinear, rowmap);
        else hipLaunchKernelGGL(k_bwd_qrow<1>, dim3((unsigned)C), dim3(256), 0, st, feats, idx, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, K, p->nonlinear, rowmap);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    }
    // 4. per-row part on MFMA
    const int nw = (N / 128 >= 512) ? 4 : 1;
    BwdRowsArgs br{};
    br.at = AttendArgs{feats, feats, (const bf16_t*)packed_split, off, p->q0_w, p->q0_b, p->q2_w, p->q2_b, qmax, nullptr, nullptr, nullptr,
                       K, K, C, p->nonlinear, 0, 0, rowmap};
    br.A = A; br.gA = gA; br.g_A = g_A; br.Dv = Dv; br.gs = gs; br.gz2 = gz2; br.Hbuf = Hb; br.Qbuf = Qb; br.gqp = gqp;
    auto launch_hs = [&](auto kern, const auto& arg) {   // hidden-split tile: 256 threads per 64 rows
        if (!dsmil_lds::allow((const void*)kern, HS_LDS_BYTES)) return (int)DSMIL_E_LAUNCH;
        hipLaunchKernelGGL(kern, dim3((unsigned)((N + HS_BM - 1) / HS_BM)), dim3(HS_THREADS), HS_LDS_BYTES, st, arg);
        return hipGetLastError() == hipSuccess ? (int)DSMIL_OK : (int)DSMIL_E_LAUNCH;
    };
    if (nw == 4) rc = v4 ? launch_tile_kernel(k_bwd_rows<4, 4>, br, 4, true, N, st) : launch_tile_kernel(k_bwd_rows<4, 1>, br, 4, false, N, st);
    else if (v4) rc = launch_hs(k_bwd_rows_hs, br);
    else rc = launch_tile_kernel(k_bwd_rows<1, 1>, br, 1, false, N, st);
    if (rc) return rc;
    // 5. gradient of the critical queries joins their rows
    hipLaunchKernelGGL(k_bwd_critical, dim3(1), dim3(1024), 0, st, idx, gqp, Qb, gz2, gq, L.T32, C, p->nonlinear);
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    TnArgs tn{};
    tn.X = feats; tn.rowmap = rowmap; tn.N = N; tn.K = K; tn.R = L.R; tn.nx = L.nx;
    tn.nslab = L.nx + (p->nonlinear ? 2 : 0); tn.S = L.S;
    tn.part0 = part0; tn.part1 = part1; tn.pb0 = pb0; tn.pb1 = pb1;
    if (p->nonlinear) {
        // 6. gH = (gz2 W2) [H > 0]
        GhArgs gh{};
        gh.at = AttendArgs{gz2, gz2, w2t, off, nullptr, zero, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           QD, QD, C, 0, 0, 0, nullptr};
        gh.Hbuf = Hb; gh.gH = gH;
        rc = (nw == 4) ? launch_tile_kernel(k_bwd_gh<4>, gh, 4, true, N, st) : launch_hs(k_bwd_gh_hs, gh);
        if (rc) return rc;
        tn.A0 = gH; tn.A1 = gz2; tn.Hb = Hb;
    } else {
        tn.A0 = gz2; tn.A1 = nullptr; tn.Hb = nullptr;
    }
    // 7. weight gradients: contractions over instances, then the fixed-order reduction (+ the sparse FCLayer gradient)
#ifdef DSMIL_TRACE
    tn.trace = tn_trace_buffer();
#endif
    {
        const dim3 gtn((unsigned)(tn.nslab * ((L.S + 7) / 8 * 8)));
        if (v4 && rowmap) hipLaunchKernelGGL((k_tn_split<true, true>), gtn, dim3(256), 0, st, tn);
        else if (v4) hipLaunchKernelGGL((k_tn_split<true, false>), gtn, dim3(256), 0, st, tn);
        else if (rowmap) hipLaunchKernelGGL((k_tn_split<false, true>), gtn, dim3(256), 0, st, tn);
        else hipLaunchKernelGGL((k_tn_split<false, false>), gtn, dim3(256), 0, st, tn);
    }
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    // 8. dense instance stream (FCLayer): only when the caller has a dense upstream gradient on the instance logits
    if (g_classes) {
        rc = tn_small(g_classes, feats, N, C, K, part, part_b, g->fc_w, g->fc_b, L.splits, w4, st, rowmap);
        if (rc) return rc;
    }
    ReduceArgs ra{};
    ra.part0 = part0; ra.part1 = part1; ra.pb0 = pb0; ra.pb1 = pb1;
    ra.g_w0 = g->q0_w; ra.g_b0 = g->q0_b; ra.g_w1 = g->q2_w; ra.g_b1 = g->q2_b;
    ra.S = L.S; ra.K = K; ra.nonlinear = p->nonlinear;
    ra.feats = feats; ra.idx = idx; ra.g_max = g_max; ra.rowmap = rowmap; ra.g_fc_w = g->fc_w; ra.g_fc_b = g->fc_b;
    ra.C = C; ra.accumulate = g_classes ? 1 : 0;
    if (adam) {
        ra.af = *adam;
        ra.af.g_fcc_w = g->fcc_w; ra.af.g_fcc_b = g->fcc_b; ra.af.n_fcc_w = (long long)C * C * Kv; ra.af.n_fcc_b = C;
    }
    const long long nred = (long long)QD * K + (p->nonlinear ? QD * QD + 2 * QD : QD) + (g_max ? (long long)C * K + C : 0) +
                           (adam ? (long long)C * C * Kv + C : 0);
    hipLaunchKernelGGL(k_bwd_reduce, dim3((unsigned)((nred + 255) / 256)), dim3(256), 0, st, ra);
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    // 9. gradient of the value rows (only when v is a trainable layer of the caller)
    if (g_vals) {
        const long long n4 = (long long)N * ((Kv + 3) / 4);
        if (v4v && (uintptr_t)g_vals % 16 == 0)
            hipLaunchKernelGGL(k_bwd_gvals<4>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, A, gB, g_vals, (long long)N, Kv, C);
        else
            hipLaunchKernelGGL(k_bwd_gvals<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, A, gB, g_vals, (long long)N, Kv, C);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    }
    return DSMIL_OK;
}

// workspace of dsmil_agg_train_step: the forward's, the backward's, and the step's own tensors
struct StepWs {
    size_t fwd, bwd, classes, A, B, pred, idx, mx, gpred, gmax, off, grads, total;
    size_t fwd_bytes, bwd_bytes;
    long long gelems;
};
StepWs step_layout(long long N, int K, int C, int nonlinear) {
    StepWs s;
    s.fwd_bytes = dsmil_agg_workspace_bytes(1, N, K, K, C);
    s.bwd_bytes = bwd_layout(N, K, K, C, nonlinear).total;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t p = o; o = al(o + bytes); return p; };
    s.fwd = take(s.fwd_bytes);
    s.bwd = take(s.bwd_bytes);
    s.classes = take((size_t)N * C * 4);
    s.A = take((size_t)N * C * 4);
    s.B = take((size_t)C * K * 4);
    s.pred = take((size_t)C * 4);
    s.idx = take((size_t)C * 8);
    s.mx = take((size_t)C * 4);
    s.gpred = take((size_t)C * 4);
    s.gmax = take((size_t)C * 4);
    s.off = take(2 * sizeof(int64_t));
    // gradients in parameter order: fc_w, fc_b, q0_w, q0_b, q2_w, q2_b, fcc_w, fcc_b (each 256-B aligned)
    s.grads = o;
    const long long sizes[8] = {(long long)C * K, C, (long long)QD * K, QD, nonlinear ? QD * QD : 0, nonlinear ? QD : 0,
                                (long long)C * C * K, C};
    s.gelems = 0;
    for (int i = 0; i < 8; ++i) { take((size_t)sizes[i] * 4); s.gelems += sizes[i]; }
    s.total = o;
    return s;
}

}  // namespace

extern "C" {

size_t dsmil_agg_backward_workspace_bytes(int64_t N, int32_t K, int32_t Kv, int32_t C) {
    if (N <= 0 || K <= 0 || Kv <= 0 || C <= 0) return 0;
    return bwd_layout(N, K, Kv, C, 1).total;   // the nonlinear layout is the larger one
}

int dsmil_agg_backward(const float* feats, const float* vals, int64_t N, const dsmil_agg_params* p,
                       const float* A, const float* Bm, const int64_t* idx, const float* g_classes,
                       const float* g_pred, const float* g_A, const float* g_B, const dsmil_agg_grads* g,
                       float* g_vals, void* ws, size_t ws_bytes, void* stream) {
    return agg_backward_impl(feats, vals, N, p, A, Bm, idx, g_classes, nullptr, g_pred, g_A, g_B, g, g_vals, nullptr,
                             ws, ws_bytes, stream, nullptr, nullptr);
}

int dsmil_agg_backward_ex(const float* feats, const float* vals, int64_t N, const dsmil_agg_params* p,
                          const float* A, const float* Bm, const int64_t* idx, const float* g_classes,
                          const float* g_max, const float* g_pred, const float* g_A, const float* g_B,
                          const dsmil_agg_grads* g, float* g_vals, const int64_t* rowmap, void* ws, size_t ws_bytes,
                          void* stream) {
    return agg_backward_impl(feats, vals, N, p, A, Bm, idx, g_classes, g_max, g_pred, g_A, g_B, g, g_vals, rowmap,
                             ws, ws_bytes, stream, nullptr, nullptr);
}

int dsmil_adam_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, int64_t step, double lr, double beta1, double beta2,
                    double eps, double weight_decay, void* stream) {
    if (n_tensors <= 0 || n_tensors > DSMIL_ADAM_MAX_TENSORS || !params || !grads || !exp_avg || !exp_avg_sq || !numel || step <= 0)
        return DSMIL_E_INVALID;
    AdamTensors t{};
    long long tot = 0;
    int n = 0;
    for (int i = 0; i < n_tensors; ++i) {
        if (numel[i] < 0) return DSMIL_E_INVALID;
        if (numel[i] == 0) continue;
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) return DSMIL_E_INVALID;
        tot += numel[i];
        t.p[n] = params[i]; t.g[n] = grads[i]; t.m[n] = exp_avg[i]; t.v[n] = exp_avg_sq[i]; t.end[n] = tot;
        ++n;
    }
    if (n == 0) return DSMIL_OK;
    t.n = n;
    for (int i = n; i < DSMIL_ADAM_MAX_TENSORS; ++i) t.end[i] = tot;
    // scalars formed in double, as torch forms them from Python floats, then handed to the kernel as fp32
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t, (float)(lr / bc1),
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)sqrt(bc2));
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

size_t dsmil_agg_train_step_workspace_bytes(int64_t N, int32_t K, int32_t C, int32_t nonlinear) {
    if (N <= 0 || K <= 0 || C <= 0) return 0;
    return step_layout(N, K, C, nonlinear).total;
}

int dsmil_agg_train_step(const float* feats, int64_t N, const int64_t* row_map, const float* label,
                         const dsmil_agg_params* p, const dsmil_adam_state* opt, float* loss, void* ws,
                         size_t ws_bytes, void* stream) {
    if (!feats || !label || !p || !opt || !loss || !ws || N <= 0) return DSMIL_E_INVALID;
    if (p->K <= 0 || p->C <= 0 || p->Kv != p->K) return DSMIL_E_INVALID;
    if (p->C > 64) return DSMIL_E_UNSUPPORTED;
    if (!p->fc_w || !p->fc_b || !p->q0_w || !p->q0_b || !p->fcc_w || !p->fcc_b || (p->nonlinear && (!p->q2_w || !p->q2_b)))
        return DSMIL_E_INVALID;
    if ((uintptr_t)ws % 256) return DSMIL_E_ALIGN;
    const int K = p->K, C = p->C;
    const StepWs L = step_layout(N, K, C, p->nonlinear);
    if (ws_bytes < L.total) return DSMIL_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* w8 = (char*)ws;
    float* classes = (float*)(w8 + L.classes); float* A = (float*)(w8 + L.A); float* Bm = (float*)(w8 + L.B);
    float* pred = (float*)(w8 + L.pred); int64_t* idx = (int64_t*)(w8 + L.idx); float* mx = (float*)(w8 + L.mx);
    float* gpred = (float*)(w8 + L.gpred); float* gmax = (float*)(w8 + L.gmax); int64_t* off = (int64_t*)(w8 + L.off);
    // gradient tensors in parameter order
    const long long sizes[8] = {(long long)C * K, C, (long long)QD * K, QD, p->nonlinear ? QD * QD : 0, p->nonlinear ? QD : 0,
                                (long long)C * C * K, C};
    float* gr[8];
    {
        size_t o = L.grads;
        for (int i = 0; i < 8; ++i) { gr[i] = (float*)(w8 + o); o = al(o + (size_t)sizes[i] * 4); }
    }
#ifdef DSMIL_EXPERIMENTS   // DSMIL_TRAIN_UNFUSE: 1 = Adam as its own launch, 2 = gA as its own launch, 4 = k_pred and the prologue as their own launches
    static const int unfuse = getenv("DSMIL_TRAIN_UNFUSE") ? atoi(getenv("DSMIL_TRAIN_UNFUSE")) : 0;
#else
    constexpr int unfuse = 0;
#endif
    // prologue: plane-cut W1 | W2 and W2^T + the bag's offsets for the forward and the backward, one launch
    const BwdWs LB = bwd_layout(N, K, K, C, p->nonlinear);
    char* bw8 = w8 + L.bwd;
    const bool planes = dsmil_agg_mlp_form() == NP_BWD;   // (experiment builds may run the forward in another MFMA form)
    // (as 240 extra workgroups of the forward's first launch when that is k_logits_stream, else as its own launch)
    TrainPrologueJob job{p->q0_w, p->nonlinear ? p->q2_w : nullptr, (bf16_t*)(bw8 + LB.wsplit), (bf16_t*)(bw8 + LB.w2t), K,
                         2 * ((K + 31) / 32), off, (int64_t*)(bw8 + LB.off), (long long)N, 240};
    const bool carried = !(unfuse & 4) && dsmil_agg_forward_carries_prologue(feats, N, p);
    if (!carried) {
        hipLaunchKernelGGL(k_train_prologue, dim3(240), dim3(256), 0, st, job.q0_w, job.q2_w, job.wsplit, job.w2t, K, job.nks,
                           job.off_a, job.off_b, (long long)N);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    }
    // forward (train_tcga.py:67) — q_max stays in its workspace for the backward
    dsmil_agg_opts fo{};
    fo.row_map = row_map;
    fo.packed_split = planes ? (const void*)(bw8 + LB.wsplit) : nullptr;
    int rc = (unfuse & 4) ? dsmil_agg_forward_ex(feats, nullptr, off, 1, N, N, p, &fo, nullptr, classes, A, Bm, pred, idx, w8 + L.fwd,
                                                 L.fwd_bytes, stream)
                          : dsmil_agg_forward_nopred(feats, off, N, p, &fo, classes, A, Bm, idx, w8 + L.fwd, L.fwd_bytes, stream,
                                                     carried ? &job : nullptr);
    if (rc) return rc;
    // backward (train_tcga.py:72); its first kernel also forms loss = 0.5 BCE(bag) + 0.5 BCE(max instance) and both
    // upstream gradients (train_tcga.py:68-71)
    dsmil_agg_grads g{};
    g.fc_w = gr[0]; g.fc_b = gr[1]; g.q0_w = gr[2]; g.q0_b = gr[3]; g.q2_w = p->nonlinear ? gr[4] : nullptr;
    g.q2_b = p->nonlinear ? gr[5] : nullptr; g.fcc_w = gr[6]; g.fcc_b = gr[7];
    const float* qmax = nullptr;
    const float* pred_part = nullptr;
    int pred_blocks = 0;
    dsmil_agg_forward_leftovers(w8 + L.fwd, 1, N, K, K, C, nullptr, &qmax, &pred_part, &pred_blocks);
    const LossHeadArgs lh{label, classes, (unfuse & 4) ? pred : nullptr, idx, loss, mx, gpred, gmax, pred_part, p->fcc_b, pred, pred_blocks};
    // optimizer.step() (train_tcga.py:73): Adam over the eight tensors, applied by the backward's last launch (k_bwd_reduce)
    // to the gradient elements it has just formed; scalars formed in double as in dsmil_adam_step
    float* params[8] = {const_cast<float*>(p->fc_w), const_cast<float*>(p->fc_b), const_cast<float*>(p->q0_w),
                        const_cast<float*>(p->q0_b), const_cast<float*>(p->q2_w), const_cast<float*>(p->q2_b),
                        const_cast<float*>(p->fcc_w), const_cast<float*>(p->fcc_b)};
    if (opt->step <= 0) return DSMIL_E_INVALID;
    AdamFuse af{};
    af.on = 1;
    for (int i = 0; i < 8; ++i) {
        if (sizes[i] && (!opt->exp_avg[i] || !opt->exp_avg_sq[i])) return DSMIL_E_INVALID;
        af.p[i] = params[i]; af.m[i] = opt->exp_avg[i]; af.v[i] = opt->exp_avg_sq[i];
    }
    const double bc1 = 1.0 - pow(opt->beta1, (double)opt->step), bc2 = 1.0 - pow(opt->beta2, (double)opt->step);
    af.h = AdamScalars{(float)(opt->lr / bc1), (float)(1.0 - opt->beta1), (float)opt->beta2, (float)(1.0 - opt->beta2),
                       (float)opt->eps, (float)opt->weight_decay, (float)sqrt(bc2)};
    if (unfuse & 1) {
        rc = agg_backward_impl(feats, nullptr, N, p, A, Bm, idx, nullptr, gmax, gpred, nullptr, nullptr, &g, nullptr, row_map,
                               bw8, L.bwd_bytes, stream, nullptr, qmax, true, &lh, nullptr, (unfuse & 2) != 0);
        if (rc) return rc;
        int64_t numel[8];
        for (int i = 0; i < 8; ++i) numel[i] = sizes[i];
        return dsmil_adam_step(8, params, (const float* const*)gr, opt->exp_avg, opt->exp_avg_sq, numel, opt->step, opt->lr,
                               opt->beta1, opt->beta2, opt->eps, opt->weight_decay, stream);
    }
    return agg_backward_impl(feats, nullptr, N, p, A, Bm, idx, nullptr, gmax, gpred, nullptr, nullptr, &g, nullptr, row_map,
                             bw8, L.bwd_bytes, stream, nullptr, qmax, true, &lh, &af, (unfuse & 2) != 0);
}

#ifdef DSMIL_TRACE
int dsmil_debug_tn_trace(unsigned long long* out, int words) {
    if (!out || words < (int)TN_TRACE_WORDS) return DSMIL_E_INVALID;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out, tn_trace_buffer(), TN_TRACE_WORDS * 8, hipMemcpyDeviceToHost) == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}
#endif

}  // extern "C"
