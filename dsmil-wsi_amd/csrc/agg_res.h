// agg_res.h — k_attend_bf16_res: the bf16-storage query / attend kernel with its 128-row tile RESIDENT in LDS.
//
// What it replaces: k_query_attend_bf16_dma streamed each 64-k chunk of the tile through a 3-slot LDS ring for the
// query MLP and then read the whole tile a SECOND time from L2 / HBM for the value sum  B = sum_n p[n] x[n,:]
// (dsmil.py:57) once the attention of every row was known — 1.34 GB of fabric traffic per 64 bags against 0.68 GB
// algorithmic, and every tile paid a cold start.  Here:
//   * the tile (128 rows x K <= 512 bf16 = up to 128 KiB) stays in LDS from its first use by GEMM 1 until the value
//     sum has consumed it: ONE read of every feature byte;
//   * the workgroup is PERSISTENT (one per CU, 160 KiB of LDS) and walks (bag, tile) work items; a fifth wave does
//     nothing but issue the feature stream (global_load_lds, 1 KiB pieces) and wait for it, so the HBM-sourced
//     pieces never sit in front of the L2-sourced weight pieces in a compute wave's in-order vmcnt queue, and the
//     next tile's chunks are requested as soon as the value sum has released their slots;
//   * the value sum is split by FEATURE CHUNK over the four compute waves (wave w owns chunks w and 4 + w, all 128
//     rows), so no cross-wave reduction of B is needed and slots are released four at a time.
// LDS map: sX [8 chunks][128 rows][128 B] (slot c of row r holds global 16-B slot c ^ f(r), as in the DMA kernel:
// conflict-free ds_read_b128 for the MFMA fragments AND for the row-major reads of the value sum), then the weight
// ring sW [2][16 KiB]; the 2 KiB of softmax scratch (tile max / sum exchange, p[128][2]) alias the END of sW[1],
// which is dead between the last step of GEMM 2 and step 1 of the next tile.
// Barrier protocol (all 5 waves execute the same sequence per tile):
//   B_s (s = 0 .. nst-1)  before step s: W(s) landed (each compute wave waited for its own pieces), X(s) landed
//                         (loader waited), everyone is past step s-1 (so W buffer (s+1)&1 may be refilled)
//   E0                    GEMM 2 done: the weight ring is free (scratch may be written, W(0) of the next tile issued)
//   E1                    wave maxima published          E2   p[row][class] and wave sums published
//   E3                    value-sum step 0 done: chunks 0..3 released (loader issues chunks 0, 1 of the next tile)
//   E4                    value-sum step 1 done: chunks 4..7 released (loader issues chunks 2, 3; chunk s + 4 follows B_s)
#pragma once
#include "agg_common.h"
#include "agg_split.h"

namespace {

constexpr int RS_BM = 128;                  // rows per tile
constexpr int RS_CH_F4 = 1024;              // float4 (16 B) per 16 KiB chunk (features: 128 rows x 128 B; weights: 128 units x 64 k)
constexpr int RS_MAXCH = 8;                 // K <= 512
constexpr int RS_THREADS = 320;             // 4 compute waves + 1 feature-stream wave
constexpr int RS_LDS_BYTES = (RS_MAXCH + 2) * RS_CH_F4 * 16;   // 163 840 = all of a CU's LDS
constexpr int RS_SCRATCH_F4 = 2 * RS_CH_F4 - 128;              // last 2 KiB of sW[1], in float4 units from sW

// tanh x = 1 - 2 / (1 + e^{2x}) on v_exp_f32 / v_rcp_f32 (5 VALU ops against ~31 + branches for tanhf): abs error
// ~1e-7, exact saturation (e -> inf gives 1, e -> 0 gives -1).  64 of these per lane and tile: with tanhf they cost
// more issue slots than the tile's 160 MFMAs.
__device__ __forceinline__ float rs_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
}

struct RsWork {
    int bag;
    long long off0, Nb, row0, slot;
};

// first work item >= item (stepping by the grid) whose tile lies inside its bag; false when the list is exhausted.
// Evaluated identically by every wave of the workgroup.
__device__ __forceinline__ bool rs_fetch(const AttendArgs& a, int tiles_per_bag, int n_items, int& item, RsWork& w) {
    while (item < n_items) {
        const int b = item / tiles_per_bag, tile = item - b * tiles_per_bag;
        const int bag = a.bag0 + b;
        const long long off0 = a.offsets[bag];
        const long long Nb = a.offsets[bag + 1] - off0;
        const long long row0 = (long long)tile * RS_BM;
        if (row0 < Nb) {
            w.bag = bag; w.off0 = off0; w.Nb = Nb; w.row0 = row0; w.slot = off0 / RS_BM + bag + tile;
            return true;
        }
        item += (int)gridDim.x;
    }
    return false;
}

template <bool TWO, bool NL>   // TWO: C == 2 (else C == 1); NL: the two-layer (nonlinear) query of dsmil.py:31-32
__global__ __launch_bounds__(RS_THREADS) void k_attend_bf16_res(AttendArgs a, int tiles_per_bag, int n_items) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* sX = reinterpret_cast<f32x4*>(smem);              // [8][RS_CH_F4]
    f32x4* sW = sX + RS_MAXCH * RS_CH_F4;                    // [2][RS_CH_F4]
    float* sRed = reinterpret_cast<float*>(sW + RS_SCRATCH_F4);   // [4 waves][2 classes] max, then [8..15] sums
    float* sP = sRed + 16;                                   // [128 rows][2 classes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K;
    const int nk1 = K >> 6;                                  // feature chunks (K % 64 == 0, K <= 512: checked by the launcher)
    const int nst = nk1 + (NL ? 2 : 0);
    const bf16_t* feats = reinterpret_cast<const bf16_t*>(a.feats);
    const f32x4* wpk = reinterpret_cast<const f32x4*>(a.wpk);   // chunk-major fragment image (k_pack_agg_bf16)

    int item = (int)blockIdx.x;
    RsWork cur, nxt;
    if (!rs_fetch(a, tiles_per_bag, n_items, item, cur)) return;
#ifdef DSMIL_TRACE
    // trace builds, DSMIL_EXPT & 64: lane 0 of compute wave 0 (slots 0..31) and of the feature-stream wave (slots 32..63)
    // stamp s_memtime after every barrier into the tile's rows of A (tools/stamp_res.py); k_finish is skipped.
    int nstamp = 0;
    auto STAMP = [&](const RsWork& w) {
        if (DSMIL_EXPT_ON(a, 64) && lane == 0 && (wave == 0 || wave == 4) && nstamp < 32 && w.row0 + RS_BM <= w.Nb) {
            const unsigned long long tt = __builtin_readcyclecounter();
            unsigned long long* o = reinterpret_cast<unsigned long long*>(a.scores + (w.off0 + w.row0) * (long long)a.C);
            o[(wave == 4 ? 32 : 0) + nstamp] = tt;
        }
        ++nstamp;
    };
#define RS_STAMP(w) STAMP(w)
#define RS_STAMP_RESET() nstamp = 0
#else
#define RS_STAMP(w)
#define RS_STAMP_RESET()
#endif

    if (wave == 4) {
        // ------------------------------------------------------------------ feature-stream wave
        const bf16_t* src[16];
        auto set_rows = [&](const RsWork& w) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r32 = (q & 3) * 8 + (lane >> 3);                  // row inside its compute wave's 32
                long long gr = w.row0 + (q >> 2) * 32 + r32;
                if (gr >= w.Nb) gr = w.Nb - 1;                              // rows past the bag end get weight 0 later
                const int gslot = (lane & 7) ^ ((r32 & 6) | ((r32 >> 4) & 1));
                src[q] = feats + phys_row(a.rowmap, w.off0 + gr) * (long long)K + gslot * 8;
            }
        };
        auto issue_chunk = [&](int c) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                __builtin_amdgcn_global_load_lds((const DSMIL_GLOBAL void*)(src[q] + c * 64),
                                                 (__attribute__((address_space(3))) void*)(sX + c * RS_CH_F4 + q * 64), 16, 0, 0);
        };
        set_rows(cur);
        for (int c = 0; c < 4 && c < nk1; ++c) issue_chunk(c);
        for (;;) {
            int in = item + (int)gridDim.x;
            const bool has_next = rs_fetch(a, tiles_per_bag, n_items, in, nxt);
            RS_STAMP_RESET();
            RS_STAMP(cur);                                                  // 0: tile start
            for (int s = 0; s < nk1; ++s) {
                // chunk s has landed; chunks issued after it may stay in flight
                const int issued = (s + 4 < nk1 ? s + 4 : nk1);
                s3_wait_vm_dyn(16 * (issued - (s + 1)));
                RS_STAMP(cur);                                              // 1 + 2s: chunk s landed
                __builtin_amdgcn_s_barrier();                               // B_s
                RS_STAMP(cur);                                              // 2 + 2s: past B_s
                if (s + 4 < nk1) issue_chunk(s + 4);
            }
            for (int s = nk1; s < nst; ++s) __builtin_amdgcn_s_barrier();   // B_s of GEMM 2
            __builtin_amdgcn_s_barrier();                                   // E0
            RS_STAMP(cur);
            __builtin_amdgcn_s_barrier();                                   // E1
            __builtin_amdgcn_s_barrier();                                   // E2
            RS_STAMP(cur);
            if (has_next) set_rows(nxt);
            __builtin_amdgcn_s_barrier();                                   // E3: chunks 0..3 released
            RS_STAMP(cur);
            if (has_next) {
                issue_chunk(0);
                if (nk1 > 1) issue_chunk(1);
            }
            RS_STAMP(cur);
            __builtin_amdgcn_s_barrier();                                   // E4: chunks 4..7 released
            RS_STAMP(cur);
            if (!has_next) break;
            if (nk1 > 2) issue_chunk(2);
            if (nk1 > 3) issue_chunk(3);
            cur = nxt;
            item = in;
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const int l31 = lane & 31, hi = lane >> 5;
    const int fr = (l31 & 6) | ((l31 >> 4) & 1);
    auto issue_w = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * 4 + wave;                                     // 16 pieces of 1 KiB per chunk
            __builtin_amdgcn_global_load_lds((const DSMIL_GLOBAL void*)(wpk + (long long)s * RS_CH_F4 + q * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(sW + (s & 1) * RS_CH_F4 + q * 64), 16, 0, 0);
        }
    };
    const float scale = 0.08838834764831845f;                               // 1/sqrt(128), dsmil.py:56
    const int C = a.C;
    issue_w(0);
    for (;;) {
        int in = item + (int)gridDim.x;
        const bool has_next = rs_fetch(a, tiles_per_bag, n_items, in, nxt);
        RS_STAMP_RESET();
        RS_STAMP(cur);                                                      // 0: tile start

        // the accumulators start from the bias (unit 32t + 8g + 4hi + e lives in register 4g + e): no bias pass later
        f32x16 H[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(a.q0_b + 32 * t + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) H[t][4 * g + e] = b[e];
            }
        // ---- GEMM 1 (transposed): H^T[j][n] += W1[j][k] X[n][k], 64 k per step
        for (int s = 0; s < nk1; ++s) {
            S3_WAIT_VM(0);
            RS_STAMP(cur);                                                  // 1 + 2s: at B_s
            __builtin_amdgcn_s_barrier();                                   // B_s
            RS_STAMP(cur);                                                  // 2 + 2s: past B_s
            if (s + 1 < nst) issue_w(s + 1);
            const f32x4* w = sW + (s & 1) * RS_CH_F4 + lane;
            const f32x4* x = sX + s * RS_CH_F4 + (wave * 32 + l31) * 8;
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
        }
        f32x16 Q[4];
        if constexpr (NL) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(a.q2_b + 32 * t + 8 * g + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        H[t][4 * g + e] = fmaxf(H[t][4 * g + e], 0.f);   // ReLU
                        Q[t][4 * g + e] = b[e];
                    }
                }
            // ---- GEMM 2 (transposed): step (tt, sidx) of chunk c2 contracts the 16 hidden units that accumulator
            //      registers 8 sidx .. 8 sidx + 7 of H[2 c2 + tt] hold (the packed W2 carries the k permutation)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int s = nk1 + c2;
                S3_WAIT_VM(0);
                __builtin_amdgcn_s_barrier();                               // B_s
                RS_STAMP(cur);                                              // 17, 18: past B_8, B_9
                if (c2 == 0) issue_w(s + 1);
                const f32x4* w = sW + (s & 1) * RS_CH_F4 + lane;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = 2 * c2 + tt;
#pragma unroll
                    for (int sidx = 0; sidx < 2; ++sidx) {
                        union { unsigned u[4]; bf16x8 v; } hb;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            hb.u[e] = pack_bf16x2_hw(H[t][8 * sidx + 2 * e], H[t][8 * sidx + 2 * e + 1]);
#pragma unroll
                        for (int t2 = 0; t2 < 4; ++t2) {
                            union { f32x4 f; bf16x8 v; } wa;
                            wa.f = w[((tt * 2 + sidx) * 4 + t2) * 64];
                            Q[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa.v, hb.v, Q[t2], 0, 0, 0);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) Q[t] = H[t];
        }
        // the reads of the last weight chunk are complete once the MFMAs that consume them have issued; make it so
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RS_STAMP(cur);                                                      // 19: at E0
        __builtin_amdgcn_s_barrier();                                       // E0: weight ring free
        RS_STAMP(cur);                                                      // 20
        if (has_next) issue_w(0);                                           // W(0) is the same for every tile
        // ---- tanh, scores (dsmil.py:55-56)
        const float* qm0 = a.qmax + ((long long)cur.bag * C) * QD;
        const float* qm1 = qm0 + (TWO ? QD : 0);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(qm0 + 32 * t + 8 * g + 4 * hi);
                f32x4 u1 = u0;
                if constexpr (TWO) u1 = *reinterpret_cast<const f32x4*>(qm1 + 32 * t + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float q = NL ? rs_tanh(Q[t][4 * g + e]) : Q[t][4 * g + e];
                    s0 = fmaf(q, u0[e], s0);
                    if constexpr (TWO) s1 = fmaf(q, u1[e], s1);
                }
            }
        s0 = (s0 + __shfl_xor(s0, 32, 64)) * scale;
        if constexpr (TWO) s1 = (s1 + __shfl_xor(s1, 32, 64)) * scale;
        const long long myrow = cur.row0 + wave * 32 + l31;
        const bool valid = myrow < cur.Nb;
        if (valid && hi == 0 && !DSMIL_EXPT_ON(a, 64)) {
            float* o = a.scores + (cur.off0 + myrow) * (long long)C;
            o[0] = s0;
            if constexpr (TWO) o[1] = s1;
        }
        const float mw0 = wave_max(valid ? s0 : -INFINITY);
        const float mw1 = TWO ? wave_max(valid ? s1 : -INFINITY) : 0.f;
        if (lane == 0) { sRed[wave * 2] = mw0; sRed[wave * 2 + 1] = mw1; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RS_STAMP(cur);                                                      // 21: at E1 (tanh + scores done)
        __builtin_amdgcn_s_barrier();                                       // E1
        const float mb0 = fmaxf(fmaxf(sRed[0], sRed[2]), fmaxf(sRed[4], sRed[6]));
        const float mb1 = TWO ? fmaxf(fmaxf(sRed[1], sRed[3]), fmaxf(sRed[5], sRed[7])) : 0.f;
        const float p0 = valid ? expf(s0 - mb0) : 0.f;                      // weights relative to the TILE max
        const float p1 = (TWO && valid) ? expf(s1 - mb1) : 0.f;
        if (hi == 0) *reinterpret_cast<float2*>(sP + (wave * 32 + l31) * 2) = make_float2(p0, p1);
        const float lw0 = wave_sum(hi == 0 ? p0 : 0.f);
        const float lw1 = TWO ? wave_sum(hi == 0 ? p1 : 0.f) : 0.f;
        if (lane == 0) { sRed[8 + wave * 2] = lw0; sRed[9 + wave * 2] = lw1; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                       // E2
        RS_STAMP(cur);                                                      // 22: past E2
        if (tid == 0) {
            float* ml = a.part_ml + cur.slot * C * 2;
            ml[0] = mb0; ml[1] = (sRed[8] + sRed[10]) + (sRed[12] + sRed[14]);
            if constexpr (TWO) { ml[2] = mb1; ml[3] = (sRed[9] + sRed[11]) + (sRed[13] + sRed[15]); }
        }
        // ---- weighted value sum (dsmil.py:57), split by feature chunk: wave w owns chunks w and 4 + w, all 128 rows.
        // lane (rr = lane >> 3, g = lane & 7) walks rows 8 i + rr and owns the 8 features of global 16-B slot g.
        const int rr = lane >> 3, g8 = lane & 7;
        float2 pw[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pw[i] = *reinterpret_cast<const float2*>(sP + (8 * i + rr) * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = 4 * j + wave;
            if (c < nk1) {
                float a0[8], a1[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
                const f32x4* xc = sX + c * RS_CH_F4 + rr * 8;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int f = (rr & 6) | ((i & 3) >> 1);
                    union { f32x4 f4; unsigned u[4]; } v;
                    v.f4 = xc[i * 64 + (g8 ^ f)];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const float lo = __uint_as_float(v.u[d] << 16), hi_ = __uint_as_float(v.u[d] & 0xffff0000u);
                        a0[2 * d] = fmaf(pw[i].x, lo, a0[2 * d]);
                        a0[2 * d + 1] = fmaf(pw[i].x, hi_, a0[2 * d + 1]);
                        if constexpr (TWO) {
                            a1[2 * d] = fmaf(pw[i].y, lo, a1[2 * d]);
                            a1[2 * d + 1] = fmaf(pw[i].y, hi_, a1[2 * d + 1]);
                        }
                    }
                }
                // reduce over the 8 row groups (lane bits 3..5), halving the payload at every stage:
                // bit 5 picks the class, bit 4 the upper / lower 4 features, bit 3 the upper / lower 2 of those
                float b4[8];
                {
                    const bool up = lane & 32;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float send = up ? a0[e] : a1[e], keep = up ? a1[e] : a0[e];
                        b4[e] = keep + __shfl_xor(send, 32, 64);
                    }
                }
                float b2[4];
                {
                    const bool up = lane & 16;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float send = up ? b4[e] : b4[4 + e], keep = up ? b4[4 + e] : b4[e];
                        b2[e] = keep + __shfl_xor(send, 16, 64);
                    }
                }
                float b1[2];
                {
                    const bool up = lane & 8;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float send = up ? b2[e] : b2[2 + e], keep = up ? b2[2 + e] : b2[e];
                        b1[e] = keep + __shfl_xor(send, 8, 64);
                    }
                }
                const int cls = (lane >> 5) & 1;
                if (TWO || cls == 0) {
                    const int k = c * 64 + g8 * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2;
                    *reinterpret_cast<float2*>(a.part_B + (cur.slot * C + cls) * (long long)a.Kv + k) = make_float2(b1[0], b1[1]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            RS_STAMP(cur);                                                  // 23, 25: value-sum step done
            __builtin_amdgcn_s_barrier();                                   // E3 / E4: this step's chunks are released
            RS_STAMP(cur);                                                  // 24, 26
        }
        if (!has_next) break;
        cur = nxt;
        item = in;
    }
}

}  // namespace
