// agg_split.h — the query MLP of the aggregator (dsmil.py:31-33,49) on bf16 MFMA with EXACT fp32
// operands: every fp32 value v is cut into three bf16 planes v = h + m + l (8 significand bits
// each, truncation, no rounding: the cut is exact), and a product x*w is formed as the sum of the
// plane products (each exact in fp32: 8 x 8 significand bits) accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  NP = 9 adds all nine plane products (the fp32 product exactly);
// NP = 6 leaves out (m,l), (l,m), (l,l): with truncating cuts |m| < 2^-7 |v| and |l| < 2^-14 |v|, so the
// omission is below 2^-20 of |x*w| (typically ~2^-23, the size of one fp32 rounding).  The cut is exact
// while no residual is subnormal (|v| >= 2^-100); smaller magnitudes lose their low plane.  One bf16 MFMA moves 16x the MACs of v_mfma_f32_32x32x2_f32 per
// cycle, so the 9-product form costs 9/16 of the exact-f32 MFMA time (6/16 for NP = 6); the number
// of fp32 accumulator roundings per 16 k is 9 (6) against 8 for the f32 MFMA chain.
//
// Weights are pre-cut and laid out in MFMA-fragment order by k_pack_agg_split (one 12 KiB chunk per
// 16-k step: [tile t][plane p][lane] x 16 B), so staging a chunk is a straight copy and every
// fragment read is a contiguous, conflict-free ds_read_b128.  Feature rows stay fp32 in LDS (full
// 128-B lines per row and 32-k chunk) and are cut in registers by the lane that feeds them to the
// MFMA as the B operand — each element is cut exactly once.
#pragma once
#include <type_traits>
#include "agg_common.h"

namespace {

constexpr int S3_CHUNK_F4 = 4 * 3 * 64;  // float4 (16 B) per packed weight chunk = 12 KiB

// (weight plane, feature plane) per product, smallest terms first; NP = 6 uses the last six
__host__ __device__ constexpr int S3_PA(int q) { return (int)((0x212201100ull >> (4 * (8 - q))) & 15); }
__host__ __device__ constexpr int S3_PB(int q) { return (int)((0x221021010ull >> (4 * (8 - q))) & 15); }
static_assert(S3_PA(0) == 2 && S3_PB(0) == 2 && S3_PA(8) == 0 && S3_PB(8) == 0 && S3_PA(3) == 2 && S3_PB(3) == 0 &&
              S3_PA(4) == 0 && S3_PB(4) == 2 && S3_PA(5) == 1 && S3_PB(5) == 1, "product table");

union S3Frag {
    unsigned u[4];
    bf16x8 v;
    f32x4 f;
};

// cut 8 fp32 values into three bf16 planes, packed in k order (element i -> bits 16(i&1) of word i/2)
__device__ __forceinline__ void split3(const float (&x)[8], S3Frag (&o)[3]) {
    unsigned xu[8], r1u[8], r2u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xu[i] = __float_as_uint(x[i]);
        const float r1 = x[i] - __uint_as_float(xu[i] & 0xFFFF0000u);   // exact: <= 16 significant bits
        r1u[i] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(r1u[i] & 0xFFFF0000u);    // exact: <= 8 significant bits
        r2u[i] = __float_as_uint(r2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[0].u[i] = __builtin_amdgcn_perm(xu[2 * i + 1], xu[2 * i], 0x07060302u);
        o[1].u[i] = __builtin_amdgcn_perm(r1u[2 * i + 1], r1u[2 * i], 0x07060302u);
        o[2].u[i] = __builtin_amdgcn_perm(r2u[2 * i + 1], r2u[2 * i], 0x07060302u);
    }
}

// Same contract as mlp_tile (agg_common.h) except that only Q is returned and the query weights
// come from the packed planes in a.wpk: [nks + 8 (nonlinear)] chunks of S3_CHUNK_F4 float4.
// `on_h(H)` is called with the hidden layer (after bias + ReLU) while it is still in the accumulator registers — the
// backward stores it there (agg_bwd.hip); the default does nothing.
struct S3NoHook {
    __device__ __forceinline__ void operator()(const f32x16 (&)[4]) const {}
};
template <int NW, int VEC, int NP, typename OnH = S3NoHook>
__device__ __forceinline__ bool mlp_tile_split(const AttendArgs& a, int bag, int tile, float* smem, f32x16 (&Q)[4],
                                               OnH on_h = OnH()) {
    static_assert(NP == 6 || NP == 9, "plane products");
    constexpr int T = NW * 64;
    constexpr int BM = NW * 32;
    constexpr int X_TILE = BM * LDK;
    constexpr int WPT = S3_CHUNK_F4 / T;  // 3 or 12 float4 per thread per weight chunk
    constexpr int XPT = (BM * 8) / T;     // 4
    constexpr int P0 = 9 - NP;
    f32x4* sW = reinterpret_cast<f32x4*>(smem);  // [2][S3_CHUNK_F4]
    float* sX = smem + 2 * S3_CHUNK_F4 * 4;      // [2][X_TILE] fp32 rows, 32 k per chunk

    const long long off0 = a.offsets[bag];
    const long long Nb = a.offsets[bag + 1] - off0;
    const long long row0 = (long long)tile * BM;
    if (row0 >= Nb) return false;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int K = a.K;
    const int nk1 = (K + 31) / 32;  // feature chunks
    const int nks = 2 * nk1;        // 16-k steps of GEMM 1
    const int nst = nks + (a.nonlinear ? 8 : 0);
    const float* feats = reinterpret_cast<const float*>(a.feats);
    const f32x4* wpk = reinterpret_cast<const f32x4*>(a.wpk);
    const int c4 = tid & 7;

    f32x4 wreg[WPT], xreg[XPT];
    const float* xrow[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        long long gr = row0 + ((tid + T * i) >> 3);
        if (gr >= Nb) gr = Nb - 1;  // rows past the bag end are masked in attend_tail
        xrow[i] = feats + phys_row(a.rowmap, off0 + gr) * (long long)K;
    }
    // branch-free loads: past the end the last chunk is re-read (and written to a dead buffer)
    auto load_w = [&](int s) {
        const f32x4* src = wpk + (long long)(s < nst ? s : nst - 1) * S3_CHUNK_F4 + tid;
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[i] = *(const DSMIL_GLOBAL f32x4*)(src + T * i);
    };
    auto write_w = [&](int s) {
        f32x4* dst = sW + (s & 1) * S3_CHUNK_F4 + tid;
#pragma unroll
        for (int i = 0; i < WPT; ++i) dst[T * i] = wreg[i];
    };
    auto load_x = [&](int c) {
        const int cx = c < nk1 ? c : nk1 - 1;
#pragma unroll
        for (int i = 0; i < XPT; ++i) xreg[i] = load4_clamped<VEC>(xrow[i], cx * 32 + c4 * 4, K);
    };
    auto write_x = [&](int c) {
        float* dst = sX + (c & 1) * X_TILE + c4 * 4;
#pragma unroll
        for (int i = 0; i < XPT; ++i) *reinterpret_cast<f32x4*>(dst + ((tid + T * i) >> 3) * LDK) = xreg[i];
    };

    f32x16 H[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) H[t][r] = 0.f;

    load_w(0);
    load_x(0);
    write_w(0);
    write_x(0);
    load_w(1);
    load_x(1);
    __syncthreads();
    // ---- GEMM 1 (transposed): H^T[j][n] += W1[j][k] X[n][k], 16 k per step
    for (int s = 0; s < nks; ++s) {
        const f32x4* w = sW + (s & 1) * S3_CHUNK_F4 + lane;
        const float* x = sX + ((s >> 1) & 1) * X_TILE + (wave * 32 + l31) * LDK + (s & 1) * 16 + 8 * hi;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(x);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(x + 4);
        const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        S3Frag xb[3];
        split3(xv, xb);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            S3Frag wa[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) wa[p].f = w[(t * 3 + p) * 64];
#pragma unroll
            for (int q = P0; q < 9; ++q)
                H[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[S3_PA(q)].v, xb[S3_PB(q)].v, H[t], 0, 0, 0);
        }
        write_w(s + 1);
        load_w(s + 2);
        if (s & 1) {
            write_x((s + 1) >> 1);
            load_x(((s + 1) >> 1) + 1);
        }
        __syncthreads();
    }
    // ---- bias (+ReLU): H^T row j = 32t + 8g + 4hi + e  for reg r = 4g + e
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
    on_h(H);
    if (!a.nonlinear) {
#pragma unroll
        for (int t = 0; t < 4; ++t) Q[t] = H[t];
        return true;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) Q[t][r] = 0.f;
    // ---- GEMM 2 (transposed): Q^T[j2][n] += W2[j2][k] H^T[k][n].  Step (t, sx) contracts the 16
    // hidden units that accumulator registers 8sx..8sx+7 of H[t] hold (the packed W2 carries the
    // matching k permutation), so H goes from the accumulators straight into the B operand.
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int t = st >> 1, sx = st & 1, s = nks + st;
        const f32x4* w = sW + (s & 1) * S3_CHUNK_F4 + lane;
        float hv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) hv[i] = H[t][8 * sx + i];
        S3Frag hb[3];
        split3(hv, hb);
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            S3Frag wa[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) wa[p].f = w[(t2 * 3 + p) * 64];
#pragma unroll
            for (int q = P0; q < 9; ++q)
                Q[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[S3_PA(q)].v, hb[S3_PB(q)].v, Q[t2], 0, 0, 0);
        }
        write_w(s + 1);
        load_w(s + 2);
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
    return true;
}

// ---- LDS-DMA variant (16-B aligned rows, K % 4 == 0) ---------------------------------------------
// Both operands go global -> LDS by global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass: the
// register-staged form above spends ~45 % of the CU's LDS cycles on ds_write_b128 alone).
//   weights:  a chunk is 12 lane-linear 1 KiB pieces — a straight copy; 3 LDS buffers, issued two
//             steps ahead.
//   features: each wave stages its own 32 rows, 4 pieces of 8 rows x 128 B per 32-k chunk, 2 buffers.
//             LDS-DMA writes lane-linear (no row padding), so the 16-B slot of a row is permuted on
//             the SOURCE side: LDS slot c of row r holds global slot c ^ f(r), f(r) = (r & 6) | (r>>4 & 1);
//             with it the two ds_read_b128 of a fragment are conflict-free in every b128 lane group.
// Completion is counted by hand (s_waitcnt vmcnt(N) + raw s_barrier): a __syncthreads() would drain
// the DMA queue at every step.
#define S3_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

__device__ __forceinline__ void s3_wait_vm_dyn(int n) {  // n is wave-uniform; a count without a case waits for everything
    switch (n) {
#define S3_CASE(k) case k: S3_WAIT_VM(k); break;
        S3_CASE(0) S3_CASE(2) S3_CASE(3) S3_CASE(4) S3_CASE(6) S3_CASE(7) S3_CASE(8) S3_CASE(10) S3_CASE(12) S3_CASE(14)
        S3_CASE(16) S3_CASE(18) S3_CASE(20) S3_CASE(22) S3_CASE(24) S3_CASE(26) S3_CASE(28) S3_CASE(30) S3_CASE(32) S3_CASE(48)
#undef S3_CASE
        default: S3_WAIT_VM(0); break;
    }
}

// XE ("x early"): feature chunk c+2 is issued at the ODD step 2c+1 (its buffer, chunk c's, was last read in the
// middle of step 2c by this very wave) instead of chunk c+1 at the even step 2c: the HBM-sourced pieces get a full
// extra step (~1000+ cycles) of flight before their first read.
template <int NW, int NP, bool XE = false, typename OnH = S3NoHook>
__device__ __forceinline__ bool mlp_tile_split_dma(const AttendArgs& a, int bag, int tile, float* smem, f32x16 (&Q)[4],
                                                   OnH on_h = OnH()) {
    static_assert(NP == 6 || NP == 9, "plane products");
    constexpr int BM = NW * 32;
    constexpr int X_TILE = BM * 32;       // floats per feature buffer (128 B per row, no padding)
    constexpr int WPW = 12 / NW;          // weight pieces per wave per step
    constexpr int P0 = 9 - NP;
    f32x4* sW = reinterpret_cast<f32x4*>(smem);  // [3][S3_CHUNK_F4]
    float* sX = smem + 3 * S3_CHUNK_F4 * 4;      // [2][X_TILE]

    const long long off0 = a.offsets[bag];
    const long long Nb = a.offsets[bag + 1] - off0;
    const long long row0 = (long long)tile * BM;
    if (row0 >= Nb) return false;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int K = a.K;
    const int nk1 = (K + 31) / 32;
    const int nks = 2 * nk1;
    const int nst = nks + (a.nonlinear ? 8 : 0);
    const float* feats = reinterpret_cast<const float*>(a.feats);
    const f32x4* wpk = reinterpret_cast<const f32x4*>(a.wpk);

    // source rows of this lane's share of the wave's 4 feature pieces, and its permuted 16-B slot
    const float* xsrc[4];
    int xslot[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = p * 8 + (lane >> 3);
        long long gr = row0 + wave * 32 + r;
        if (gr >= Nb) gr = Nb - 1;  // rows past the bag end are masked in attend_tail
        xsrc[p] = feats + phys_row(a.rowmap, off0 + gr) * (long long)K;
        xslot[p] = ((lane & 7) ^ ((r & 6) | ((r >> 4) & 1))) * 4;
    }
    // one 1 KiB piece per call: the pieces of a step are spread behind its MFMA groups (issued in a
    // burst they queue behind each other at 100+ cycles apiece and stall the wave's MFMA stream)
    auto issue_w_piece = [&](int s, int i) {
        const int q = i * NW + wave;
        __builtin_amdgcn_global_load_lds((const DSMIL_GLOBAL void*)(wpk + (long long)s * S3_CHUNK_F4 + q * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(sW + (s % 3) * S3_CHUNK_F4 + q * 64), 16, 0, 0);
    };
    auto issue_x_piece = [&](int c, int p) {
        int k = c * 32 + xslot[p];
        k = k < K ? k : K - 4;  // past K the packed weights are zero: any finite data will do
        __builtin_amdgcn_global_load_lds((const DSMIL_GLOBAL void*)(xsrc[p] + k),
                                         (__attribute__((address_space(3))) void*)(sX + (c & 1) * X_TILE + (wave * 32 + p * 8) * 32), 16, 0, 0);
    };
    auto issue_w = [&](int s) {
#pragma unroll
        for (int i = 0; i < WPW; ++i) issue_w_piece(s, i);
    };
    auto issue_x = [&](int c) {
#pragma unroll
        for (int p = 0; p < 4; ++p) issue_x_piece(c, p);
    };
    const int fr = (l31 & 6) | ((l31 >> 4) & 1);  // this lane's row permutation as a reader

    f32x16 H[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) H[t][r] = 0.f;

#ifdef DSMIL_TRACE
    // build with -DDSMIL_TRACE, run with DSMIL_EXPT=68: wave 0 of every tile stamps s_memtime at the end
    // of every step into its rows of A (decoded by tools_stamp.py).  Not compiled into the product.
    unsigned* stamp = reinterpret_cast<unsigned*>(a.scores + (off0 + row0) * (long long)a.C);
    int nstamp = 0;
    auto STAMP = [&]() {
        if (DSMIL_EXPT_ON(a, 64) && tid == 0 && nstamp < 62) {
            const unsigned long long tt = __builtin_readcyclecounter();
            stamp[2 * nstamp] = (unsigned)tt;
            stamp[2 * nstamp + 1] = (unsigned)(tt >> 32);
        }
        ++nstamp;
    };
#else
    auto STAMP = [] {};
#endif
    STAMP();
    issue_w(0);
    issue_x(0);
    issue_w(1);  // nst >= 2 always
    if (XE && a.nonlinear && nk1 > 1) {
        issue_x(1);
        S3_WAIT_VM(WPW + 4);
    } else {
        S3_WAIT_VM(WPW);
    }
    __builtin_amdgcn_s_barrier();
    STAMP();
    // ---- GEMM 1 (transposed): H^T[j][n] += W1[j][k] X[n][k], 16 k per step.
    // The B operand of step s+1 is read from LDS and cut into planes in the MIDDLE of step s, between
    // its MFMA groups, so the ~45 VALU ops of the cut issue in MFMA shadows instead of ahead of the
    // step's first MFMA.  A feature chunk is wave-private (each wave stages and reads its own rows),
    // so the odd step only has to wait for its OWN DMA pieces of chunk c+1 — no barrier involved.
    auto read_cut = [&](int s, S3Frag (&xb)[3]) {
        const float* x = sX + ((s >> 1) & 1) * X_TILE + (wave * 32 + l31) * 32;
        const int j0 = ((s & 1) * 4 + hi * 2) ^ fr;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + j0 * 4);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(x + (j0 ^ 1) * 4);
        const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        split3(xv, xb);
    };
    S3Frag xb[3];
    read_cut(0, xb);
    // one 16-k step; PARITY / NEXT_X / CUT_NEXT are literals at every call site so that the whole
    // step is ONE basic block (the compiler then interleaves the cut with the MFMAs) and every
    // wait count is an immediate
    // XE: NEXT_X on an ODD step issues chunk (s>>1)+2; HAS_X1 says whether chunk (s>>1)+1 exists (and is then in
    // flight or landed): it sets how many younger pieces each counted wait may leave outstanding
    auto step = [&](int s, auto parity, auto next_x, auto cut_next, auto has_x1) {
        constexpr bool ODD = decltype(parity)::value, NEXT_X = decltype(next_x)::value, CUT = decltype(cut_next)::value;
        constexpr bool HAS_X1 = decltype(has_x1)::value;
        issue_w(s + 2);
        if constexpr (NEXT_X) issue_x((s >> 1) + (XE ? 2 : 1));
        const f32x4* w = sW + (s % 3) * S3_CHUNK_F4 + lane;
        // hand-placed issue order (sched_barrier after every MFMA slot): the compiler otherwise
        // clumps the ~45 VALU ops of the cut, and the MFMA pipe idles behind the clump
        const float* xnp = sX + (((s + 1) >> 1) & 1) * X_TILE + (wave * 32 + l31) * 32;
        const int jn = (((s + 1) & 1) * 4 + hi * 2) ^ fr;
        f32x4 xr0, xr1;
        unsigned xu[8], r1u[8], r2u[8];
        S3Frag xn[3];
        S3Frag wa[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) wa[0][p].f = w[p * 64];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int q = P0; q < 9; ++q) {
                H[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[t & 1][S3_PA(q)].v, xb[S3_PB(q)].v, H[t], 0, 0, 0);
                const int k = t * NP + (q - P0);  // filler slot behind this MFMA
                if (q == P0 && t < 3) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) wa[(t + 1) & 1][p].f = w[((t + 1) * 3 + p) * 64];
                }
                if constexpr (CUT) {
                    if (k == 1) {
                        // own pieces of chunk c+1 landed.  Plain: issued a step ago, only this step's weight pieces are
                        // younger.  XE: issued two steps ago; younger = w(s+1), w(s+2) and, if issued, x(c+2)
                        if constexpr (ODD && !XE) s3_wait_vm_dyn(WPW);
                        if constexpr (ODD && XE) s3_wait_vm_dyn(2 * WPW + (NEXT_X ? 4 : 0));
                        xr0 = *reinterpret_cast<const f32x4*>(xnp + jn * 4);
                        xr1 = *reinterpret_cast<const f32x4*>(xnp + (jn ^ 1) * 4);
                    }
                    if (k >= 6 && k < 14) {  // cut element e: exact residuals
                        const int e = k - 6;
                        const float xv = e < 4 ? xr0[e & 3] : xr1[e & 3];
                        xu[e] = __float_as_uint(xv);
                        const float r1 = xv - __uint_as_float(xu[e] & 0xFFFF0000u);
                        r1u[e] = __float_as_uint(r1);
                        r2u[e] = __float_as_uint(r1 - __uint_as_float(r1u[e] & 0xFFFF0000u));
                        asm volatile("" : "+v"(r1u[e]), "+v"(r2u[e]));  // pin the piece into this slot
                    }
                    if (k >= 14 && k < 18) {  // pack word i of the three planes
                        const int i = k - 14;
                        xn[0].u[i] = __builtin_amdgcn_perm(xu[2 * i + 1], xu[2 * i], 0x07060302u);
                        xn[1].u[i] = __builtin_amdgcn_perm(r1u[2 * i + 1], r1u[2 * i], 0x07060302u);
                        xn[2].u[i] = __builtin_amdgcn_perm(r2u[2 * i + 1], r2u[2 * i], 0x07060302u);
                        asm volatile("" : "+v"(xn[0].u[i]), "+v"(xn[1].u[i]), "+v"(xn[2].u[i]));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (CUT) {
#pragma unroll
            for (int p = 0; p < 3; ++p) xb[p] = xn[p];
        }
        // w(s+1) has landed.  Plain: everything issued BEFORE this step has landed.  XE: the feature chunk issued
        // at the previous odd step (even steps) or at this one (odd steps) may stay in flight
        if constexpr (!XE) s3_wait_vm_dyn(WPW + (NEXT_X ? 4 : 0));
        else s3_wait_vm_dyn(WPW + ((ODD ? NEXT_X : HAS_X1) ? 4 : 0));
        __builtin_amdgcn_s_barrier();
        STAMP();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (a.nonlinear && XE) {
        for (int c = 0; c + 2 < nk1; ++c) {
            step(2 * c, F_{}, F_{}, T_{}, T_{});
            step(2 * c + 1, T_{}, T_{}, T_{}, T_{});
        }
        if (nk1 > 1) {
            step(nks - 4, F_{}, F_{}, T_{}, T_{});
            step(nks - 3, T_{}, F_{}, T_{}, T_{});
        }
        step(nks - 2, F_{}, F_{}, T_{}, F_{});
        step(nks - 1, T_{}, F_{}, F_{}, F_{});
    } else if (a.nonlinear) {  // a weight chunk is due two steps ahead throughout GEMM 1
        for (int c = 0; c + 1 < nk1; ++c) {
            step(2 * c, F_{}, T_{}, T_{}, F_{});
            step(2 * c + 1, T_{}, F_{}, T_{}, F_{});
        }
        step(nks - 2, F_{}, F_{}, T_{}, F_{});
        step(nks - 1, T_{}, F_{}, F_{}, F_{});
    } else
    for (int s = 0; s < nks; ++s) {
        // feature chunk c+1 goes out on the even step 2c; its first use is the mid-step read of step 2c+1
        const bool do_w = s + 2 < nst, do_x = !(s & 1) && (s >> 1) + 1 < nk1;
        const int issued = (do_w ? WPW : 0) + (do_x ? 4 : 0);
        if (do_w) issue_w(s + 2);
        if (do_x) issue_x((s >> 1) + 1);
        const f32x4* w = sW + (s % 3) * S3_CHUNK_F4 + lane;
        S3Frag xn[3];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            S3Frag wa[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) wa[p].f = w[(t * 3 + p) * 64];
#pragma unroll
            for (int q = P0; q < 9; ++q)
                H[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[S3_PA(q)].v, xb[S3_PB(q)].v, H[t], 0, 0, 0);
            if (t == 0 && s + 1 < nks) {
                if (s & 1) s3_wait_vm_dyn(do_w ? WPW : 0);  // own pieces of chunk c+1 (issued a step ago) landed
                read_cut(s + 1, xn);
            }
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) xb[p] = xn[p];
        s3_wait_vm_dyn(issued);  // everything issued BEFORE this step has landed
        __builtin_amdgcn_s_barrier();
        STAMP();
    }
    // ---- bias (+ReLU): H^T row j = 32t + 8g + 4hi + e  for reg r = 4g + e
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
    on_h(H);
    if (!a.nonlinear) {
#pragma unroll
        for (int t = 0; t < 4; ++t) Q[t] = H[t];
        return true;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) Q[t][r] = 0.f;
    // ---- GEMM 2 (transposed), see mlp_tile_split
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int t = st >> 1, sx = st & 1, s = nks + st;
        if (st < 6) issue_w(s + 2);
        const f32x4* w = sW + (s % 3) * S3_CHUNK_F4 + lane;
        float hv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) hv[i] = H[t][8 * sx + i];
        S3Frag hb[3];
        split3(hv, hb);
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            S3Frag wa[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) wa[p].f = w[(t2 * 3 + p) * 64];
#pragma unroll
            for (int q = P0; q < 9; ++q)
                Q[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[S3_PA(q)].v, hb[S3_PB(q)].v, Q[t2], 0, 0, 0);
        }
        if (st < 6) S3_WAIT_VM(WPW); else S3_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
        STAMP();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.q2_b + 32 * t + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) Q[t][4 * g + e] = fast_tanh(Q[t][4 * g + e] + b[e]);
        }
    return true;
}

// fp32 query weights -> three truncated bf16 planes in MFMA-fragment order.
//   chunk s < nks (GEMM 1):  [t][p][lane (l31,hi)][e] = plane_p(W1[32t + l31][16s + 8hi + e])   (0 past K)
//   chunk nks + 2t + sx:     [t2][p][lane][e] = plane_p(W2[32t2 + l31][32t + 16sx + (e&3) + 8(e>>2) + 4hi])
// tr != 0: the first matrix is read TRANSPOSED — element (row j, column k) = q0_w[k * QD + j] (K = 128): the backward's
// gH = gz2 W2 runs the GEMM-1 pipeline with W := W2^T without materialising the transpose.
__device__ __forceinline__ void pack_agg_split_range(const float* __restrict__ q0_w, const float* __restrict__ q2_w,
                                                     bf16_t* __restrict__ out, int K, int nks, int tr, long long i0,
                                                     long long stride) {
    const long long per = (long long)S3_CHUNK_F4 * 8;  // bf16 per chunk
    const long long total = (long long)(nks + (q2_w ? 8 : 0)) * per;
    for (long long i = i0; i < total; i += stride) {
        const int s = (int)(i / per);
        int r = (int)(i - s * per);
        const int e = r & 7; r >>= 3;
        const int lane = r & 63; r >>= 6;
        const int p = r % 3, t = r / 3;
        const int l31 = lane & 31, hi = lane >> 5;
        float v;
        if (s < nks) {
            const int k = 16 * s + 8 * hi + e;
            v = k < K ? (tr ? q0_w[(long long)k * QD + (32 * t + l31)] : q0_w[(long long)(32 * t + l31) * K + k]) : 0.f;
        } else {
            const int st = s - nks, tt = st >> 1, sx = st & 1;
            v = q2_w[(32 * t + l31) * QD + 32 * tt + 16 * sx + (e & 3) + 8 * (e >> 2) + 4 * hi];
        }
        const unsigned h = __float_as_uint(v) & 0xFFFF0000u;
        const float r1 = v - __uint_as_float(h);
        const unsigned m = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(m);
        const unsigned bits = p == 0 ? h : (p == 1 ? m : __float_as_uint(r2));
        out[i] = (bf16_t)(bits >> 16);
    }
}
__global__ void k_pack_agg_split(const float* __restrict__ q0_w, const float* __restrict__ q2_w,
                                 bf16_t* __restrict__ out, int K, int nks, int tr = 0) {
    pack_agg_split_range(q0_w, q2_w, out, K, nks, tr, (long long)blockIdx.x * blockDim.x + threadIdx.x,
                         (long long)gridDim.x * blockDim.x);
}

}  // namespace
