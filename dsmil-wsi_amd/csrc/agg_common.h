// Shared device code of the aggregator kernels (forward: agg_fwd.hip, backward: agg_bwd.hip):
// typedefs, element loaders, wave reductions and the query-MLP tile on exact-f32 MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "dsmil_hip.h"

// library-internal (defined in agg_fwd.hip): FCLayer over logical rows, physical row = rowmap[r] (nullptr = identity)
int dsmil_fc_forward_rows(const float* feats, int64_t total_rows, int32_t K, int32_t C, const float* fc_w,
                          const float* fc_b, float* classes, const int64_t* rowmap, void* stream);

// library-internal (defined in agg_fwd.hip): what a finished fp32 dsmil_agg_forward(_ex) call leaves in its workspace for
// the backward of the same bag — the plane-cut query weights (null if the forward did not cut them: caller-supplied
// packed_split, or an MFMA form without planes) and q_max [n_bags, C, 128]
void dsmil_agg_forward_leftovers(void* ws, int32_t n_bags, int64_t total_rows, int32_t K, int32_t Kv, int32_t C,
                                 const void** packed_split, const float** qmax, const float** pred_part = nullptr,
                                 int* pred_blocks = nullptr);
// library-internal: dsmil_agg_forward_ex WITHOUT its last launch (k_pred): pred[o] = fcc_b[o] + the sum of
// pred_part[block][o][c] in (block, c) order is left to the caller (dsmil_agg_train_step: the loss head of k_bwd_prep)
// `job` (dsmil_agg_train_step): what k_train_prologue does — plane-cut W1 | W2 (-> wsplit), W2^T (-> w2t) and the lone bag's
// {0, N} offsets (-> off_a, off_b) — done by extra workgroups of the forward's FIRST launch (k_logits_stream, which then takes
// the bag's extent from job->N instead of the offsets in memory).  *job_taken tells whether the forward's launch path could
// carry it (dsmil_agg_forward_carries_prologue); if not, the caller runs the prologue as its own launch BEFORE this call.
struct TrainPrologueJob {
    const float* q0_w; const float* q2_w; unsigned short* wsplit; unsigned short* w2t; int K, nks;
    int64_t* off_a; int64_t* off_b; long long N; int blocks;
};
bool dsmil_agg_forward_carries_prologue(const float* feats, int64_t total_rows, const dsmil_agg_params* p);
int dsmil_agg_forward_nopred(const float* feats, const int64_t* offsets, int64_t total_rows, const dsmil_agg_params* p,
                             const dsmil_agg_opts* opts, float* classes_out, float* A, float* B, int64_t* idx, void* ws,
                             size_t ws_bytes, void* stream, const TrainPrologueJob* job = nullptr);

namespace {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int QD = DSMIL_Q_DIM;  // 128
constexpr int BK = 32;           // k-chunk staged per pipeline step
constexpr int LDK = BK + 4;      // LDS row stride in floats (144 B): conflict-free ds_read_b128
constexpr int W_TILE = QD * LDK; // floats per staged weight chunk
constexpr int R0 = 128;          // rows per workgroup of k_logits_argmax

// Wave-wide sum / max, every lane gets the result.  On DPP: row_ror 8, 4, 2, 1 inside the four 16-lane rows (every lane of a
// row then holds the row's value), then the four rows through v_readlane and three scalar-operand VALU ops — ~11 instructions,
// no LDS.  (Rounds 1-4 wrote these as six __shfl_xor stages, which hipcc lowers to ds_bpermute_b32: six dependent LDS round
// trips of ~100 cycles each; k_qmax is a chain of 32 such sums, the tile softmax of k_attend_hs of four.)  Fixed order:
// ((r0 + r1) + (r2 + r3)) over the rows, rotation order inside a row — deterministic, NOT the xor tree's rounding.
// PRECONDITION: all 64 lanes active (EXEC == ~0): the DPP rotations and v_readlane read the VGPRs of inactive lanes as they
// are (stale), where ds_bpermute returned 0.  Every caller reduces in wave-uniform control flow and feeds the neutral element
// from lanes without data; experiment builds (-DDSMIL_EXPERIMENTS) trap on a partial wave.
#ifdef DSMIL_EXPERIMENTS
#define DSMIL_FULL_WAVE() do { if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap(); } while (0)
#else
#define DSMIL_FULL_WAVE() do { } while (0)
#endif
__device__ __forceinline__ float dpp_row_sum(float v) {
#define DSMIL_ROR(n) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xf, 0xf, false))
    DSMIL_ROR(8); DSMIL_ROR(4); DSMIL_ROR(2); DSMIL_ROR(1);
#undef DSMIL_ROR
    return v;
}
__device__ __forceinline__ float dpp_row_max(float v) {
#define DSMIL_ROR(n) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x120 + (n), 0xf, 0xf, false)))
    DSMIL_ROR(8); DSMIL_ROR(4); DSMIL_ROR(2); DSMIL_ROR(1);
#undef DSMIL_ROR
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    DSMIL_FULL_WAVE();
    v = dpp_row_sum(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
    DSMIL_FULL_WAVE();
    v = dpp_row_max(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Every feature / weight / workspace pointer handed to this library is device GLOBAL memory.  Inside
// out-of-line (noinline) device functions hipcc cannot infer that and would emit FLAT loads, whose
// lgkmcnt accounting serialises them with the LDS reads; the hot loads therefore say so explicitly.
#define DSMIL_GLOBAL __attribute__((address_space(1)))

typedef unsigned short bf16_t;  // raw bfloat16 bits (storage type of the bf16 path)
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // round to nearest even, like torch .bfloat16()
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 (lo in bits 0..15), round to nearest even: ONE v_cvt_pk_bf16_f32 on gfx950 (the integer
// form above costs ~8 VALU ops per value); identical results for every finite input
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2_hw(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2_hw r = __builtin_convertvector(v, bf16x2_hw);
    return __builtin_bit_cast(unsigned, r);
}

// tanh x = 1 - 2 / (1 + e^{2x}) on v_exp_f32 / v_rcp_f32: 5 VALU ops where tanhf is ~31 plus branches.  A tile applies
// it to 64 values per lane — with tanhf that is ~2000 issue slots per wave and tile, a sixth of the fp32 kernel's tile
// time and more than all MFMAs of the bf16 kernel's.  Abs error ~1e-7 (one ulp of the "1 -"), exact saturation
// (e -> inf gives 1, e -> 0 gives -1), NaN stays NaN.  Switched by DSMIL_PRECISE_TANH in experiment builds for A/B.
__device__ __forceinline__ float fast_tanh(float x) {
#ifdef DSMIL_PRECISE_TANH
    return tanhf(x);
#else
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
#endif
}

// 4 consecutive elements at p[k..k+3] as floats, zero beyond klim.  VEC=4 needs rows aligned to
// 4 elements (16 B for fp32, 8 B for bf16).
template <int VEC, typename T = float>
__device__ __forceinline__ f32x4 load4(const T* __restrict__ p, int k, int klim) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) {
        if constexpr (VEC == 4) {
            if (k < klim) {
                const u32x2 t = *(const DSMIL_GLOBAL u32x2*)(p + k);
                v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
                v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + e < klim) v[e] = bf2f(p[k + e]);
        }
    } else if constexpr (VEC == 4) {
        if (k < klim) v = *(const DSMIL_GLOBAL f32x4*)(p + k);
    } else {
        if (k + 0 < klim) v[0] = p[k + 0];
        if (k + 1 < klim) v[1] = p[k + 1];
        if (k + 2 < klim) v[2] = p[k + 2];
        if (k + 3 < klim) v[3] = p[k + 3];
    }
    return v;
}

// 4 consecutive elements at p[k..k+3], 16-B (fp32) / 8-B (bf16) aligned, no range check
template <typename T>
__device__ __forceinline__ f32x4 load4_nocheck(const T* __restrict__ p, int k) {
    if constexpr (sizeof(T) == 2) {
        const u32x2 t = *(const DSMIL_GLOBAL u32x2*)(p + k);
        f32x4 v;
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
        return v;
    } else {
        return *(const DSMIL_GLOBAL f32x4*)(p + k);
    }
}

// Staging variant: never branches for VEC=4 — the address is clamped into the row (klim % 4 == 0,
// klim >= 4) and the caller zeroes out-of-range k later (at LDS-write time), so a run of these
// loads issues back to back and stays in flight under the MFMAs.
template <int VEC>
__device__ __forceinline__ f32x4 load4_clamped(const float* __restrict__ p, int k, int klim) {
    if constexpr (VEC == 4) {
        const int kc = k < klim ? k : klim - 4;
        return *(const DSMIL_GLOBAL f32x4*)(p + kc);
    } else {
        return load4<1>(p, k, klim);
    }
}

// better (value, index): larger value wins, lowest index wins on exact ties
__device__ __forceinline__ bool better(float v, long long i, float bv, long long bi) {
    return (v > bv) || (v == bv && i < bi);
}

// --------------------------------------------------------------------------------------------
// k_query_attend — the dominant kernel.  NW waves per workgroup, 32 instance rows per wave.
// --------------------------------------------------------------------------------------------
struct AttendArgs {
    const void* feats;  // fp32 or bf16 [total_rows, K]
    const void* vals;   // fp32 or bf16 [total_rows, Kv]
    const bf16_t* wpk;  // bf16 path: packed W1 [128][K64] then W2 permuted [128][128]
    const int64_t* offsets;
    const float* q0_w;
    const float* q0_b;
    const float* q2_w;
    const float* q2_b;
    const float* qmax;  // [n_bags, C, 128]
    float* scores;      // [total_rows, C]  (the A buffer; normalised in place by k_finish)
    float* part_ml;     // [slots, C, 2]
    float* part_B;      // [slots, C, Kv]
    int K, Kv, C, nonlinear;
    int expt;  // ablation switches of -DDSMIL_EXPERIMENTS builds (DSMIL_EXPT); always 0 in the product build
    int bag0;  // first bag of this launch (chunked pipelining over bags)
    // Row indirection (train_tcga.py:78-83 dropout_patches as an index list instead of a gathered copy): logical row i
    // of the batch lives at physical row rowmap[i] of feats / vals; nullptr = identity.  Outputs stay logical.
    const int64_t* rowmap;
    // k_attend_hs only (qm_flag != nullptr): the critical row's query (k_qmax) runs INSIDE the attend launch — workgroups
    // [0, C) of a bag's grid row produce qmax / the arg-max index while the tiles run their MLP, and publish through
    // qm_flag[bag * C + c] (cleared by the logits launch before); the tiles read qmax behind the flag with agent-scope loads.
    int* qm_flag;
    const float* qm_part_val;
    const long long* qm_part_idx;
    int64_t* qm_idx;
    int qm_r0;
    // Persistent batch kernels (k_attend_f3, k_attend_bf16_res) on RAGGED batches: tile_pre[b] = number of tiles of the bags in
    // front of bag b, tile_pre[n_bags] = all tiles (k_tile_prefix); the tile items of the launch are then the REAL tiles only,
    // dealt in contiguous runs.  nullptr (uniform batches: every bag max_rows long) = item b * tiles_per_bag + tile.
    const int* tile_pre;
    int n_bags;
};

// Largest b in [0, n_bags) with pre[b] <= item (pre non-decreasing, pre[0] = 0, item < pre[n_bags]): the bag that owns tile
// item `item` — behind runs of equal entries (empty bags).  Uniform address: scalar loads; once per workgroup.
__device__ __forceinline__ int tile_owner(const int* pre_, int n_bags, int item) {
    const __attribute__((address_space(4))) int* pre = (const __attribute__((address_space(4))) int*)(uintptr_t)pre_;
    int lo = 0, hi = n_bags;           // invariant: pre[lo] <= item < pre[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pre[mid] <= item) lo = mid; else hi = mid;
    }
    return lo;
}
// the same with a guess (the bag of the item in front): two independent loads when the item stays in that bag or moves to the next
__device__ __forceinline__ int tile_owner_near(const int* pre_, int n_bags, int item, int guess) {
    const __attribute__((address_space(4))) int* pre = (const __attribute__((address_space(4))) int*)(uintptr_t)pre_;
    if (guess >= 0 && guess < n_bags) {
        const int g1 = pre[guess + 1], g2 = pre[guess + 2 <= n_bags ? guess + 2 : n_bags];
        if (pre[guess] <= item) {
            if (item < g1) return guess;
            if (item < g2) return guess + 1;
        }
    }
    return tile_owner(pre_, n_bags, item);
}

__device__ __forceinline__ long long phys_row(const int64_t* __restrict__ rowmap, long long logical) {
    return rowmap ? (long long)rowmap[logical] : logical;
}

// Ablation / tracing branches exist only in experiment builds (DSMIL_CFLAGS=-DDSMIL_EXPERIMENTS): the
// product kernels carry none of them.
#ifdef DSMIL_EXPERIMENTS
#define DSMIL_EXPT_ON(a, bit) (((a).expt & (bit)) != 0)
#else
#define DSMIL_EXPT_ON(a, bit) false
#endif

// --------------------------------------------------------------------------------------------
// attend_tail: everything behind the query MLP, shared by the fp32 and bf16 kernels.  Q holds
// Q^T in the MFMA D layout: lane (l31, hi), tile t, reg 4g+e  <->  Q[row l31][32t + 8g + 4hi + e].
// Scores (dsmil.py:55-56), tile softmax statistics, weighted value sum (dsmil.py:57).
// --------------------------------------------------------------------------------------------
template <int NW, int VEC, typename T, int TU = 8>   // TU: value rows in flight per lane-pair of loads
__device__ __forceinline__ void attend_tail(const AttendArgs& a, const f32x16 (&Q)[4], float* smem, int bag,
                                            long long off0, long long Nb, long long row0, long long slot) {
    constexpr int T_ = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // From here on the staging LDS is free (every wave is past the last barrier above).
    // ---- scores, tile softmax statistics, weighted value sum — two classes per sweep
    const long long wrow0 = row0 + wave * 32;           // first row of this wave
    const long long myrow = wrow0 + l31;                // this lane's instance row (bag-local)
    const bool valid = myrow < Nb;
    const float scale = 0.08838834764831845f;           // 1/sqrt(128), dsmil.py:56
    const int Kv = a.Kv;
    const T* vbase = reinterpret_cast<const T*>(a.vals);
    // physical value row of this lane's instance (rows past the bag end: the last row, weight 0)
    const long long myphys = phys_row(a.rowmap, off0 + (valid ? myrow : Nb - 1));
    float* sRed = smem;                                  // [NW][4]: m0,l0,m1,l1 per wave
    float* sB = smem + 64;                               // [NW][2][512]
    for (int c0 = 0; c0 < a.C; c0 += 2) {
        const int c1 = (c0 + 1 < a.C) ? c0 + 1 : c0;
        const float* qm0 = a.qmax + ((long long)bag * a.C + c0) * QD;
        const float* qm1 = a.qmax + ((long long)bag * a.C + c1) * QD;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(qm0 + 32 * t + 8 * g + 4 * hi);
                const f32x4 u1 = *reinterpret_cast<const f32x4*>(qm1 + 32 * t + 8 * g + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0 = fmaf(Q[t][4 * g + e], u0[e], s0);
                    s1 = fmaf(Q[t][4 * g + e], u1[e], s1);
                }
            }
        s0 = (s0 + __shfl_xor(s0, 32, 64)) * scale;
        s1 = (s1 + __shfl_xor(s1, 32, 64)) * scale;
        if (valid && hi == 0) {
            float* o = a.scores + (off0 + myrow) * (long long)a.C;
            o[c0] = s0;
            if (c1 != c0) o[c1] = s1;
        }
        const float mw0 = wave_max(valid ? s0 : -INFINITY);
        const float mw1 = wave_max(valid ? s1 : -INFINITY);
        const float p0 = valid ? expf(s0 - mw0) : 0.f;
        const float p1 = valid ? expf(s1 - mw1) : 0.f;
        const float lw0 = wave_sum(hi == 0 ? p0 : 0.f);
        const float lw1 = wave_sum(hi == 0 ? p1 : 0.f);
        // block-level max / sum
        float f0 = 1.f, f1 = 1.f;
        if constexpr (NW > 1) {
            __syncthreads();
            if (lane == 0) {
                sRed[wave * 4 + 0] = mw0; sRed[wave * 4 + 1] = lw0;
                sRed[wave * 4 + 2] = mw1; sRed[wave * 4 + 3] = lw1;
            }
            __syncthreads();
            float mb0 = -INFINITY, mb1 = -INFINITY;
#pragma unroll
            for (int w = 0; w < NW; ++w) { mb0 = fmaxf(mb0, sRed[w * 4 + 0]); mb1 = fmaxf(mb1, sRed[w * 4 + 2]); }
            f0 = expf(mw0 - mb0);  // 0 for a wave with no valid row (mw = -inf, mb finite)
            f1 = expf(mw1 - mb1);
            if (tid == 0) {
                float lb0 = 0.f, lb1 = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    lb0 += sRed[w * 4 + 1] * expf(sRed[w * 4 + 0] - mb0);
                    lb1 += sRed[w * 4 + 3] * expf(sRed[w * 4 + 2] - mb1);
                }
                float* ml = a.part_ml + (slot * a.C + c0) * 2;
                ml[0] = mb0; ml[1] = lb0;
                if (c1 != c0) { ml[2] = mb1; ml[3] = lb1; }
            }
        } else {
            if (tid == 0) {
                float* ml = a.part_ml + (slot * a.C + c0) * 2;
                ml[0] = mw0; ml[1] = lw0;
                if (c1 != c0) { ml[2] = mw1; ml[3] = lw1; }
            }
        }
        const float pp0 = p0 * f0, pp1 = p1 * f1;  // weights relative to the BLOCK max
        // ---- weighted value sum: Bpart[c][k] = sum_n p[n][c] * V[n][k], 512 k per sweep
        for (int k0 = 0; k0 < (DSMIL_EXPT_ON(a, 1) ? 0 : Kv); k0 += 512) {
            f32x4 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
            const int ka = k0 + lane * 4, kb = ka + 256;
            // VEC = 4: unconditional loads from a clamped column (a lane past Kv accumulates junk it
            // never stores) — a load behind a per-lane branch is issued alone and waited for at once
            const int kac = ka < Kv ? ka : Kv - 4, kbc = kb < Kv ? kb : Kv - 4;
#pragma unroll TU
            for (int n = 0; n < 32; ++n) {
                const long long r = __shfl(myphys, n, 64);   // rows past the bag end were clamped (weight 0)
                const float w0 = __shfl(pp0, n, 64), w1 = __shfl(pp1, n, 64);
                const T* vr = vbase + r * (long long)Kv;
                f32x4 va, vb;
                if constexpr (VEC == 4) {
                    va = load4_nocheck<T>(vr, kac);
                    vb = load4_nocheck<T>(vr, kbc);
                } else {
                    va = load4<VEC, T>(vr, ka, Kv);
                    vb = load4<VEC, T>(vr, kb, Kv);
                }
                acc00 += w0 * va; acc01 += w0 * vb;
                acc10 += w1 * va; acc11 += w1 * vb;
            }
            float* pb0 = a.part_B + (slot * a.C + c0) * (long long)Kv;
            float* pb1 = a.part_B + (slot * a.C + c1) * (long long)Kv;
            if constexpr (NW > 1) {
                __syncthreads();
                float* my = sB + wave * 1024;
                *reinterpret_cast<f32x4*>(my + lane * 4) = acc00;
                *reinterpret_cast<f32x4*>(my + 256 + lane * 4) = acc01;
                *reinterpret_cast<f32x4*>(my + 512 + lane * 4) = acc10;
                *reinterpret_cast<f32x4*>(my + 768 + lane * 4) = acc11;
                __syncthreads();
                for (int e = tid; e < 1024; e += T_) {
                    float s = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) s += sB[w * 1024 + e];
                    const int cc = e >> 9, k = k0 + (e & 511);
                    if (k < Kv && (cc == 0 || c1 != c0)) (cc ? pb1 : pb0)[k] = s;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (ka + e < Kv) { pb0[ka + e] = acc00[e]; if (c1 != c0) pb1[ka + e] = acc10[e]; }
                    if (kb + e < Kv) { pb0[kb + e] = acc01[e]; if (c1 != c0) pb1[kb + e] = acc11[e]; }
                }
            }
        }
        if constexpr (NW > 1) __syncthreads();
    }
}

// Query MLP of ONE tile (NW waves x 32 rows) on exact-f32 MFMA.  On return every wave holds, in the
// MFMA D layout (lane (l31, hi), tile t, reg 4g+e <-> row l31, unit 32t+8g+4hi+e):
//   H = relu(x W1^T + b1)  (or the plain linear query when !nonlinear)   and   Q = tanh(H W2^T + b2)
// (Q == H when !nonlinear), and is past the last barrier (the staging LDS is free).
// Returns false when the tile lies past the end of the bag (block-uniform).
template <int NW, int VEC>
__device__ __forceinline__ bool mlp_tile(const AttendArgs& a, int bag, int tile, float* smem, f32x16 (&H)[4], f32x16 (&Q)[4]) {
    static_assert(NW == 1 || NW == 4 || NW == 8, "tile geometries: 32, 128 or 256 rows");
    constexpr int T = NW * 64;
    constexpr int BM = NW * 32;
    constexpr int X_TILE = BM * LDK;
    constexpr int WPT = (QD * (BK / 4)) / T;  // float4 per thread per weight chunk (4 or 16)
    constexpr int XPT = (BM * (BK / 4)) / T;  // == 4
    constexpr int WPS = WPT >= 4 ? WPT / 4 : 1;  // weight float4 per pipeline slot (slots past WPT idle)
    float* sW = smem;               // [2][W_TILE]
    float* sX = smem + 2 * W_TILE;  // [2][X_TILE]

    const long long off0 = a.offsets[bag];
    const long long Nb = a.offsets[bag + 1] - off0;
    const long long row0 = (long long)tile * BM;
    if (row0 >= Nb) return false;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int K = a.K;
    const int nk1 = (K + BK - 1) / BK;
    const int nk = nk1 + (a.nonlinear ? QD / BK : 0);
    const float* feats = reinterpret_cast<const float*>(a.feats);
    const int c4 = tid & 7;

    // Staging pipeline, distance 2, ONE register set (write-then-reissue):
    //   iteration ci:  MFMAs on LDS buffer ci&1  ||  registers (chunk ci+1, loaded during
    //   iteration ci-1) -> LDS buffer (ci+1)&1  ||  global loads of chunk ci+2 -> same registers.
    // Each of the 4 k-groups of a chunk carries one slot (1/4 of the chunk's registers), placed
    // behind that k-group's MFMAs so address arithmetic, ds_write and load issue hide under the
    // 64-cycle MFMAs instead of forming a bubble at the chunk boundary.  One barrier per chunk.
    f32x4 wreg[WPT], xreg[XPT];
    bool kok = true;                 // weight k-slice of the chunk held in registers is in range
    const float* xrow[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        long long gr = row0 + ((tid + T * i) >> 3);
        if (gr >= Nb) gr = Nb - 1;   // clamp: rows past the bag end are masked later
        xrow[i] = feats + phys_row(a.rowmap, off0 + gr) * (long long)K;
    }
    auto chunk_src = [&](int ci, const float*& wb, int& ld, int& k, int& klim) {
        int k0;
        if (ci < nk1) { wb = a.q0_w; ld = K; k0 = ci * BK; klim = K; }
        else { wb = a.q2_w; ld = QD; k0 = (ci - nk1) * BK; klim = QD; }
        k = k0 + c4 * 4;
    };
    // issue the loads of pipeline slot q of chunk ci.  Branch-free on purpose (a branch around a
    // load makes hipcc fall back to vmcnt(0) waits): past the last chunk the last one is simply
    // re-loaded, and `with_x` is a literal at every call site.
    auto load_slot = [&](int ci, int q, const bool with_x) {
        const int cw = ci < nk ? ci : nk - 1;
        const float* wb; int ld, k, klim;
        chunk_src(cw, wb, ld, k, klim);
#pragma unroll
        for (int j = 0; j < WPS; ++j) {
            const int i = q * WPS + j;
            if (i < WPT) wreg[i] = load4_clamped<VEC>(wb + (long long)((tid + T * i) >> 3) * ld, k, klim);
        }
        if (with_x) {
            const int cx = ci < nk1 ? ci : nk1 - 1;
            xreg[q] = load4_clamped<VEC>(xrow[q], cx * BK + c4 * 4, K);
        }
    };
    // move pipeline slot q of the chunk held in registers (chunk ci) to its LDS buffer; past the
    // last chunk this writes into a buffer nobody reads any more
    auto write_slot = [&](int ci, int q, bool ok, const bool with_x) {
        float* w = sW + (ci & 1) * W_TILE;
#pragma unroll
        for (int j = 0; j < WPS; ++j) {
            const int i = q * WPS + j;
            if (i < WPT) {
                f32x4 v = wreg[i];
                if constexpr (VEC == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;   // zero the weight k-tail
                }
                *reinterpret_cast<f32x4*>(w + ((tid + T * i) >> 3) * LDK + c4 * 4) = v;
            }
        }
        if (with_x)
            *reinterpret_cast<f32x4*>(sX + (ci & 1) * X_TILE + ((tid + T * q) >> 3) * LDK + c4 * 4) = xreg[q];
    };
    auto k_in_range = [&](int ci) {
        const float* wb; int ld, k, klim;
        chunk_src(ci < nk ? ci : nk - 1, wb, ld, k, klim);
        return k < klim;
    };

#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) H[t][r] = 0.f;

    // prologue: chunk 0 -> LDS, chunk 1 -> registers
#pragma unroll
    for (int q = 0; q < 4; ++q) load_slot(0, q, true);
    kok = k_in_range(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) write_slot(0, q, kok, true);
#pragma unroll
    for (int q = 0; q < 4; ++q) load_slot(1, q, true);
    kok = k_in_range(1);
    __syncthreads();
    const int frag_off = l31 * LDK + 4 * hi;  // this lane's row / k-half inside a chunk
    // ---- GEMM 1 (transposed): H^T[j][n] += W1[j][k] * X[n][k]
    for (int ci = 0; ci < nk1; ++ci) {
        const float* w = sW + (ci & 1) * W_TILE + frag_off;
        const float* x = sX + (ci & 1) * X_TILE + wave * 32 * LDK + frag_off;
        const bool kok_next = k_in_range(ci + 2);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const f32x4 xb = *reinterpret_cast<const f32x4*>(x + kg * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 wa = *reinterpret_cast<const f32x4*>(w + t * 32 * LDK + kg * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) H[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], xb[j], H[t], 0, 0, 0);
            }
            write_slot(ci + 1, kg, kok, true);
            load_slot(ci + 2, kg, true);
            __builtin_amdgcn_sched_barrier(0);  // keep each slot inside its own k-group
        }
        kok = kok_next;
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
                float v = H[t][4 * g + e] + b[e];
                H[t][4 * g + e] = a.nonlinear ? fmaxf(v, 0.f) : v;
            }
        }
    if (a.nonlinear) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) Q[t][r] = 0.f;
        // ---- GEMM 2 (transposed): Q^T[j2][n] += W2[j2][k] * H^T[k][n]; chunk t feeds k=32t..32t+31
        // straight from the accumulator registers of H[t]: reg 4g+e holds k = 8g + 4hi + e.
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ci = nk1 + t;
            const float* w = sW + (ci & 1) * W_TILE + frag_off;
            const bool kok_next = k_in_range(ci + 2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    const f32x4 wa = *reinterpret_cast<const f32x4*>(w + t2 * 32 * LDK + g * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        Q[t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[e], H[t][4 * g + e], Q[t2], 0, 0, 0);
                }
                write_slot(ci + 1, g, kok, false);
                load_slot(ci + 2, g, false);
                __builtin_amdgcn_sched_barrier(0);
            }
            kok = kok_next;
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
    return true;
}

}  // namespace
