// agg_hs.h — the query MLP tile for FEW rows (a lone bag, a training step): the HIDDEN units of a 32-row tile are split
// over the four SIMDs of a CU.
//
// The batched kernels (agg_split.h) give a wave 32 rows and all 128 hidden units: a 10 000-row bag is 313 waves on the
// chip's 1024 SIMDs, each walking 40 k-steps x 24 dependent MFMAs — 35 us for 1.6 GFLOP (dsmil.py:49 on one bag, the call
// train_tcga.py:67 and attention_map.py:85 make).  Here a 256-thread workgroup owns the 32 rows and wave w owns hidden
// units 32w..32w+31 (GEMM 1: H^T tile w) and query units 32w..32w+31 (GEMM 2: Q^T tile w): a quarter of the MFMA chain per
// wave, four times the waves.  Same arithmetic as the batched kernels — three exact bf16 planes per fp32 operand, the same
// six plane products in the same order, k ascending — so H and Q are bit-identical to theirs.
//   weights   the packed image of k_pack_agg_split: chunk s = [tile t][plane p][lane] x 16 B.  Wave w needs exactly pieces
//             3w..3w+2 of every chunk, so its weight stream is PRIVATE: LDS-DMA into its own slice of a 4-deep ring, its
//             own vmcnt wait, no barrier.
//   features  the 32 rows x 32 k chunk is shared by the four waves (each reads all of it as the MFMA B operand and cuts
//             its fragment itself); wave w issues rows 8w..8w+7; a 6-deep ring keeps five chunks (~4000 cycles of work)
//             in flight; one barrier per chunk = "chunk c landed for everybody, chunk c-1 released".
//   hidden    after GEMM 1 the four H tiles are exchanged through LDS (the feature ring is dead by then; rows padded to
//             132 floats: conflict-free 16-B accesses) and every wave reads the 8 hidden units of a GEMM-2 step for its
//             row as two 16-B loads.
// 48 KiB weights + 24 KiB features (+ 4 KiB scratch for the tail) = 76 KiB: two workgroups per CU.
#pragma once
#include "agg_split.h"

namespace {

constexpr int HS_WR = 4;                 // weight ring depth, in 16-k steps
constexpr int HS_XR = 6;                 // feature ring depth, in 32-k chunks
constexpr int HS_XT = 32 * 32;           // floats per feature chunk (32 rows x 128 B, slots permuted as in agg_split.h)
constexpr int HS_HLD = 132;              // row stride of the hidden-layer exchange buffer (floats)
constexpr int HS_W_FLOATS = HS_WR * S3_CHUNK_F4 * 4;
constexpr int HS_X_FLOATS = HS_XR * HS_XT;
constexpr int HS_SCRATCH = 1024;         // floats behind the rings, for the caller's tail
constexpr int HS_LDS_BYTES = (HS_W_FLOATS + HS_X_FLOATS + HS_SCRATCH) * 4;
static_assert(32 * HS_HLD <= HS_X_FLOATS, "the hidden-layer exchange buffer aliases the feature ring");

// On return wave w holds, in the MFMA D layout (lane (l31, hi), reg 4g+e <-> row l31, unit 32w + 8g + 4hi + e):
//   Hw = relu(x W1^T + b1) tile w (the plain linear query when !nonlinear), Qw = tanh(H W2^T + b2) tile w (= Hw when
// !nonlinear); every wave is past the last barrier and no DMA is in flight (the LDS is free).
// Returns false when the tile lies past the end of the bag (block-uniform).  Rows 16-B aligned, K % 4 == 0.
template <int NP>
__device__ __forceinline__ bool mlp_tile_hs(const AttendArgs& a, int bag, int tile, float* smem, f32x16& Hw, f32x16& Qw) {
    static_assert(NP == 6 || NP == 9, "plane products");
    constexpr int P0 = 9 - NP;
    f32x4* sW = reinterpret_cast<f32x4*>(smem);   // [HS_WR][S3_CHUNK_F4]
    float* sX = smem + HS_W_FLOATS;               // [HS_XR][HS_XT], later sH [32][HS_HLD]
    const long long off0 = a.offsets[bag];
    const long long Nb = a.offsets[bag + 1] - off0;
    const long long row0 = (long long)tile * 32;
    if (row0 >= Nb) return false;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int K = a.K;
    const int nk1 = (K + 31) / 32;
    const int nks = 2 * nk1;
    const int nst = nks + (a.nonlinear ? 8 : 0);
    const float* feats = reinterpret_cast<const float*>(a.feats);
    const f32x4* wpk = reinterpret_cast<const f32x4*>(a.wpk);

    // this lane's share of the wave's feature piece (rows 8 wave .. 8 wave + 7), and its permuted 16-B slot
    const int xr = wave * 8 + (lane >> 3);
    long long gr = row0 + xr;
    if (gr >= Nb) gr = Nb - 1;   // rows past the bag end are masked by the caller
    const float* xsrc = feats + phys_row(a.rowmap, off0 + gr) * (long long)K;
    const int xslot = ((lane & 7) ^ ((xr & 6) | ((xr >> 4) & 1))) * 4;
    auto issue_w = [&](int s) {   // past the end the last chunk is re-read into a dead ring slot: uniform counts
        const int sc = s < nst ? s : nst - 1;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int q = 3 * wave + p;
            __builtin_amdgcn_global_load_lds((const DSMIL_GLOBAL void*)(wpk + (long long)sc * S3_CHUNK_F4 + q * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(sW + (s % HS_WR) * S3_CHUNK_F4 + q * 64), 16, 0, 0);
        }
    };
    auto issue_x = [&](int c) {   // c < nk1
        int k = c * 32 + xslot;
        k = k < K ? k : K - 4;    // past K the packed weights are zero: any finite data will do
        __builtin_amdgcn_global_load_lds((const DSMIL_GLOBAL void*)(xsrc + k),
                                         (__attribute__((address_space(3))) void*)(sX + (c % HS_XR) * HS_XT + (wave * 8) * 32), 16, 0, 0);
    };
    // feature pieces that are YOUNGER than the weight pieces of step s when step s begins: those issued at the even steps
    // among s-3, s-2, s-1 (step e issues chunk (e >> 1) + HS_XR - 1 after its own weights)
    auto x_younger = [&](int s) {
        int n = 0;
#pragma unroll
        for (int d = 1; d <= 3; ++d) {
            const int e = s - d;
            if (e >= 0 && !(e & 1) && (e >> 1) + HS_XR - 1 < nk1) ++n;
        }
        return n;
    };
    const int fr = (l31 & 6) | ((l31 >> 4) & 1);   // this lane's row permutation as a reader
    auto read_cut = [&](int s, S3Frag (&xb)[3]) {
        const float* x = sX + ((s >> 1) % HS_XR) * HS_XT + l31 * 32;
        const int j0 = ((s & 1) * 4 + hi * 2) ^ fr;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + j0 * 4);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(x + (j0 ^ 1) * 4);
        const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        split3(xv, xb);
    };

#pragma unroll
    for (int r = 0; r < 16; ++r) Hw[r] = 0.f;
    // prologue: weights of steps 0..2, the first HS_XR - 1 feature chunks
    issue_w(0);
    issue_w(1);
    issue_w(2);
    for (int c = 0; c < HS_XR - 1 && c < nk1; ++c) issue_x(c);
    // ---- GEMM 1 (transposed): H^T[32 wave + j][n] += W1[32 wave + j][k] X[n][k], 16 k per step
    for (int s = 0; s < nks; ++s) {
        if (s == 0) S3_WAIT_VM(0);
        else s3_wait_vm_dyn(6 + x_younger(s));          // weights of step s (and everything older) have landed
        if (!(s & 1)) {
            __builtin_amdgcn_s_barrier();                // chunk s/2 landed for every wave; chunk s/2 - 1 is released
            if ((s >> 1) + HS_XR - 1 < nk1) { issue_w(s + HS_WR - 1); issue_x((s >> 1) + HS_XR - 1); }
            else issue_w(s + HS_WR - 1);
        } else {
            issue_w(s + HS_WR - 1);
        }
        S3Frag xb[3], wa[3];
        read_cut(s, xb);
        const f32x4* w = sW + (s % HS_WR) * S3_CHUNK_F4 + (3 * wave) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) wa[p].f = w[p * 64];
#pragma unroll
        for (int q = P0; q < 9; ++q)
            Hw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[S3_PA(q)].v, xb[S3_PB(q)].v, Hw, 0, 0, 0);
    }
    // ---- bias (+ReLU): reg 4g+e <-> unit 32 wave + 8g + 4hi + e
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.q0_b + 32 * wave + 8 * g + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = Hw[4 * g + e] + b[e];
            Hw[4 * g + e] = a.nonlinear ? fmaxf(v, 0.f) : v;
        }
    }
    if (!a.nonlinear) {
        Qw = Hw;
        S3_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
        return true;
    }
    // ---- exchange the hidden layer: sH[row][unit], rows padded to HS_HLD floats
    __syncthreads();                                     // every wave is done reading the feature ring
    float* sH = sX;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = Hw[4 * g + e];
        *reinterpret_cast<f32x4*>(sH + l31 * HS_HLD + 32 * wave + 8 * g + 4 * hi) = hv;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) Qw[r] = 0.f;
    // ---- GEMM 2 (transposed): Q^T[32 wave + j][n] += W2[32 wave + j][k] H^T[k][n]; step (t, sx) contracts hidden units
    //      32t + 16sx + {0..3, 8..11} + 4hi (the k permutation the packed W2 carries)
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int t = st >> 1, sx = st & 1, s = nks + st;
        s3_wait_vm_dyn(6 + x_younger(s));
        issue_w(s + HS_WR - 1);
        const float* h = sH + l31 * HS_HLD + 32 * t + 16 * sx + 4 * hi;
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(h);
        const f32x4 h1 = *reinterpret_cast<const f32x4*>(h + 8);
        const float hv[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        S3Frag hb[3], wa[3];
        split3(hv, hb);
        const f32x4* w = sW + (s % HS_WR) * S3_CHUNK_F4 + (3 * wave) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) wa[p].f = w[p * 64];
#pragma unroll
        for (int q = P0; q < 9; ++q)
            Qw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[S3_PA(q)].v, hb[S3_PB(q)].v, Qw, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.q2_b + 32 * wave + 8 * g + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) Qw[4 * g + e] = fast_tanh(Qw[4 * g + e] + b[e]);
    }
    S3_WAIT_VM(0);                                       // the clamped weight re-reads of the last steps
    __syncthreads();
    return true;
}

// Everything behind the query MLP for the hidden-split tile: scores (dsmil.py:55-56) from the four waves' partial dot
// products, the tile's softmax statistics, the weighted value sum (dsmil.py:57) with the ROWS split over the waves (8 each,
// all features: the loop of attend_tail, row index wave-uniform) and a four-way merge through LDS.  `scr` = HS_SCRATCH
// floats of LDS, `merge` = 4096 floats (the dead feature ring).  Partials go to slot `slot` as attend_tail writes them.
template <typename T>
__device__ __forceinline__ void attend_tail_hs(const AttendArgs& a, const f32x16& Qw, float* scr, float* merge, int bag,
                                               long long off0, long long Nb, long long row0, long long slot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long long myrow = row0 + l31;
    const bool valid = myrow < Nb;
    const float scale = 0.08838834764831845f;            // 1/sqrt(128), dsmil.py:56
    const int Kv = a.Kv;
    const T* vbase = reinterpret_cast<const T*>(a.vals);
    float* sS = scr;                                      // [4 waves][2 classes][32 rows]
    // physical value row of this lane's instance (rows past the bag end: the last row, weight 0)
    const long long myphys = phys_row(a.rowmap, off0 + (valid ? myrow : Nb - 1));
    for (int c0 = 0; c0 < a.C; c0 += 2) {
        const int c1 = (c0 + 1 < a.C) ? c0 + 1 : c0;
        const float* qm0 = a.qmax + ((long long)bag * a.C + c0) * QD + 32 * wave;
        const float* qm1 = a.qmax + ((long long)bag * a.C + c1) * QD + 32 * wave;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(qm0 + 8 * g + 4 * hi);
            const f32x4 u1 = *reinterpret_cast<const f32x4*>(qm1 + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s0 = fmaf(Qw[4 * g + e], u0[e], s0);
                s1 = fmaf(Qw[4 * g + e], u1[e], s1);
            }
        }
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (hi == 0) { sS[(wave * 2 + 0) * 32 + l31] = s0; sS[(wave * 2 + 1) * 32 + l31] = s1; }
        __syncthreads();
        // every wave forms the same scores in the same (fixed) order
        s0 = ((sS[0 * 32 + l31] + sS[2 * 32 + l31]) + (sS[4 * 32 + l31] + sS[6 * 32 + l31])) * scale;
        s1 = ((sS[1 * 32 + l31] + sS[3 * 32 + l31]) + (sS[5 * 32 + l31] + sS[7 * 32 + l31])) * scale;
        const float m0 = wave_max(valid ? s0 : -INFINITY), m1 = wave_max(valid ? s1 : -INFINITY);
        const float p0 = valid ? expf(s0 - m0) : 0.f, p1 = valid ? expf(s1 - m1) : 0.f;
        const float l0 = wave_sum(hi == 0 ? p0 : 0.f), l1 = wave_sum(hi == 0 ? p1 : 0.f);
        if (wave == 0) {
            if (valid && hi == 0) {
                float* o = a.scores + (off0 + myrow) * (long long)a.C;
                o[c0] = s0;
                if (c1 != c0) o[c1] = s1;
            }
            if (lane == 0) {
                float* ml = a.part_ml + (slot * a.C + c0) * 2;
                ml[0] = m0; ml[1] = l0;
                if (c1 != c0) { ml[2] = m1; ml[3] = l1; }
            }
        }
        // ---- weighted value sum: Bpart[c][k] = sum_n p[n][c] V[n][k], 512 k per sweep; wave w sums rows 8w..8w+7
        float* pb0 = a.part_B + (slot * a.C + c0) * (long long)Kv;
        float* pb1 = a.part_B + (slot * a.C + c1) * (long long)Kv;
        for (int k0 = 0; k0 < Kv; k0 += 512) {
            f32x4 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
            const int ka = k0 + lane * 4, kb = ka + 256;
            // unconditional loads from a clamped column (a lane past Kv accumulates junk it never stores)
            const int kac = ka < Kv ? ka : Kv - 4, kbc = kb < Kv ? kb : Kv - 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = 8 * wave + j;
                const long long r = __shfl(myphys, n, 64);   // rows past the bag end were clamped (weight 0)
                const float w0 = __shfl(p0, n, 64), w1 = __shfl(p1, n, 64);
                const T* vr = vbase + r * (long long)Kv;
                const f32x4 va = load4_nocheck<T>(vr, kac), vb = load4_nocheck<T>(vr, kbc);
                acc00 += w0 * va; acc01 += w0 * vb;
                acc10 += w1 * va; acc11 += w1 * vb;
            }
            __syncthreads();                                  // (also: every wave is done with sS / the previous sweep)
            float* my = merge + wave * 1024;
            *reinterpret_cast<f32x4*>(my + lane * 4) = acc00;
            *reinterpret_cast<f32x4*>(my + 256 + lane * 4) = acc01;
            *reinterpret_cast<f32x4*>(my + 512 + lane * 4) = acc10;
            *reinterpret_cast<f32x4*>(my + 768 + lane * 4) = acc11;
            __syncthreads();
            for (int e = tid; e < 1024; e += 256) {
                const float sum = (merge[e] + merge[1024 + e]) + (merge[2048 + e] + merge[3072 + e]);
                const int cc = e >> 9, k = k0 + (e & 511);
                if (k < Kv && (cc == 0 || c1 != c0)) (cc ? pb1 : pb0)[k] = sum;
            }
        }
        __syncthreads();
    }
}

}  // namespace
