// ResNet-18 (InstanceNorm2d, fc = Identity) patch embedder, forward — hand-written HIP for gfx950.
//
// Replaces the torchvision backbone behind dsmil.IClassifier (reference call sites:
// compute_feats.py:146-170,211; dsmil.py:21-25).  fp32 end to end on exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32), activations NHWC in HBM.
//
// InstanceNorm cannot be folded into the conv weights (per-image, per-channel statistics over
// H x W, eps = 1e-5, biased variance, no affine).  Each convolution therefore writes its RAW output
// plus per-32-pixel-tile (mean, M2) partials from its epilogue registers; a tiny finalize kernel
// merges them (Chan) into mean/rstd per (image, channel); and the CONSUMER applies
// (x - mean) * rstd (+ReLU) while it stages its input tile into LDS.  Residual add + ReLU is the
// one remaining elementwise pass per BasicBlock.
//
//   k_stem            7x7 s2 conv on the NCHW input (Cin = 3, K = 147): input window + weights in LDS
//   k_norm_relu_maxpool  IN + ReLU + 3x3 s2 max-pool of the stem output (max commutes with the
//                     increasing map x -> relu((x-mean)*rstd), so the window max is taken on raw values)
//   k_conv<NT,NORM>   implicit-GEMM 3x3 / 1x1 conv, M = flattened (image, y, x) output pixels,
//                     128 pixels x NT*32 channels per workgroup, K looped as (cin-chunk, tap)
//   k_in_finalize_*   partials -> mean, rstd
//   k_norm_add_relu   out = relu(IN(y2) + identity | IN(y_downsample)); the last one also
//                     average-pools (adaptive_avg_pool2d(1)) into the 512-d feature row
//
// MFMA maps as in agg_fwd.hip; here A = pixels (rows), B = output channels (cols), so that for one
// accumulator register the 32 lanes of a half-wave hold 32 consecutive channels of one pixel:
// the NHWC store is 128-B coalesced.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "dsmil_hip.h"
#include "prof.h"
#include "lds_attr.h"

namespace {

// Experiment builds (-DDSMIL_EXPERIMENTS) read ablation knobs from the environment, once per process.
#ifdef DSMIL_EXPERIMENTS
inline int expt_env(const char* name) {
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
}
#define DSMIL_WEXPT_ON(a, bit) (((a).expt & (bit)) != 0)
#else
#define DSMIL_WEXPT_ON(a, bit) false
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDK = BK + 4;
constexpr float IN_EPS = 1e-5f;

__device__ __forceinline__ int drow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---- exact three-plane bf16 cuts (see agg_split.h) ---------------------------------------------
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));

// ---- operand forms of the conv kernels (template parameter NP = plane products per fp32 MAC) -------------------------------
//   9 / 6  bf16 x 3 planes by truncation (exact cut), nine / the six largest products (rounds 2-4; experiment builds:
//          DSMIL_WINO / DSMIL_CONV = s9 | s6)
//   3      fp16 x 2 planes by round-to-nearest, products h0 w0 + h0 w1 + h1 w0 (round 5, the product form).  Two RNE fp16
//          planes carry a value to 2^-24 relative while the second plane is normal (absolute 2^-25 below: gfx950's f16 MFMA
//          keeps subnormals), the dropped h1 w1 is <= 2^-24 |x w|: the accuracy class of the six-product bf16 form
//          (tools/form_error_study.py: 5.7e-7 against 3.6e-7 max feature error through the 20 convs) for HALF the MFMAs and
//          2/3 of the plane bytes.  fp16 has five exponent bits: WEIGHTS are multiplied by 2^EMB_WSHIFT at pack time (a conv
//          weight of ~0.03 would otherwise keep ~8 bits in its second plane) and the accumulators by 2^-EMB_WSHIFT before
//          anything reads them; ACTIVATIONS are what an InstanceNorm / frozen BatchNorm + ReLU (or the [0,1] image) produces:
//          far inside +-65504, also after the Winograd input transform (sums of four).
constexpr int NPD = 3;                        // the product library's form
constexpr int EMB_WSHIFT = 8;
constexpr float EMB_WSCALE = 256.f, EMB_OSCALE = 1.f / 256.f;
//   1      fp16 x 1 plane (RNE), one product: the OPT-IN reduced-precision path (dsmil_resnet_forward_ex, precision = 1):
//          operands rounded to 11 significand bits, f32 accumulation, fp32 activations / InstanceNorm between the layers
template <int NP> __host__ __device__ constexpr int emb_planes() { return NP == 1 ? 1 : (NP == 3 ? 2 : 3); }
template <int NP> __host__ __device__ constexpr bool emb_f16() { return NP == 3 || NP == 1; }
// (operand plane of the activations, of the weights) per product, smallest terms first
template <int NP> struct PlaneProducts;
template <> struct PlaneProducts<9> { static constexpr int N = 9; static constexpr int X[9] = {2, 1, 2, 2, 0, 1, 1, 0, 0}, W[9] = {2, 2, 1, 0, 2, 1, 0, 1, 0}; };
template <> struct PlaneProducts<6> { static constexpr int N = 6; static constexpr int X[6] = {2, 0, 1, 1, 0, 0}, W[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct PlaneProducts<3> { static constexpr int N = 3; static constexpr int X[3] = {1, 0, 0}, W[3] = {0, 1, 0}; };
template <> struct PlaneProducts<1> { static constexpr int N = 1; static constexpr int X[1] = {0}, W[1] = {0}; };

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// one 32x32x16 plane product on the matrix pipe: fp16 operands for NP = 3, bf16 otherwise (same rate, same layout)
template <int NP>
__device__ __forceinline__ f32x16 plane_mfma(const u32x4_t& a, const u32x4_t& b, const f32x16& c) {
    if constexpr (emb_f16<NP>()) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// two fp32 values -> their packed fp16 planes: h = (rne16(a), rne16(b)), l = (rne16(a - h.lo), rne16(b - h.hi)); v_fma_mix
// forms each half in one instruction (hipcc's own sequence: a v_cvt_pk, two conversions back and two subtractions per pair)
__device__ __forceinline__ void cut2h(float a, float b, unsigned& h, unsigned& l) {
    const float one = 1.f;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "v"(one));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "v"(one));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(one), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(one), "v"(h));
}
// host/pack side of the same cut (one value): plane 0 / plane 1 of v (already scaled), as fp16 bits
__device__ __forceinline__ unsigned short pack_h0(float v) { const _Float16 h = (_Float16)v; return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ unsigned short pack_h1(float v) { const _Float16 h = (_Float16)v; const _Float16 l = (_Float16)(v - (float)h); return __builtin_bit_cast(unsigned short, l); }

// cut 4 fp32 values into operand planes, 2 packed dwords per plane: NP = 3 -> (h0, h1, -) fp16; else three bf16 planes
// (truncation: exact)
__device__ __forceinline__ unsigned cut2h1(float a, float b) {   // one plane: (rne16(a), rne16(b))
    unsigned h;
    const float one = 1.f;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "v"(one));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "v"(one));
    return h;
}
template <int NP = 6>
__device__ __forceinline__ void cut4(const f32x4& x, u32x2_t& ph, u32x2_t& pm, u32x2_t& pl) {
    if constexpr (NP == 1) {
        ph = u32x2_t{cut2h1(x[0], x[1]), cut2h1(x[2], x[3])};
        pm = u32x2_t{0u, 0u};
        pl = u32x2_t{0u, 0u};
        return;
    }
    if constexpr (NP == 3) {
        unsigned h0, l0, h1, l1;
        cut2h(x[0], x[1], h0, l0);
        cut2h(x[2], x[3], h1, l1);
        ph = u32x2_t{h0, h1};
        pm = u32x2_t{l0, l1};
        pl = u32x2_t{0u, 0u};
        return;
    }
    unsigned xu[4], r1u[4], r2u[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        xu[e] = __float_as_uint(x[e]);
        const float r1 = x[e] - __uint_as_float(xu[e] & 0xFFFF0000u);
        r1u[e] = __float_as_uint(r1);
        r2u[e] = __float_as_uint(r1 - __uint_as_float(r1u[e] & 0xFFFF0000u));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ph[i] = __builtin_amdgcn_perm(xu[2 * i + 1], xu[2 * i], 0x07060302u);
        pm[i] = __builtin_amdgcn_perm(r1u[2 * i + 1], r1u[2 * i], 0x07060302u);
        pl[i] = __builtin_amdgcn_perm(r2u[2 * i + 1], r2u[2 * i], 0x07060302u);
    }
}

union Frag16 { u32x4_t u; bf16x8_t v; };
// cut 8 consecutive-k fp32 values into three bf16x8 MFMA operands (plane h, m, l)
__device__ __forceinline__ void cut8(const f32x4& a0, const f32x4& a1, Frag16 (&o)[3]) {
    u32x2_t h0, m0, l0, h1, m1, l1;
    cut4(a0, h0, m0, l0);
    cut4(a1, h1, m1, l1);
    o[0].u = u32x4_t{h0[0], h0[1], h1[0], h1[1]};
    o[1].u = u32x4_t{m0[0], m0[1], m1[0], m1[1]};
    o[2].u = u32x4_t{l0[0], l0[1], l1[0], l1[1]};
}

// ---------------------------------------------------------------------------------------------
// statistics epilogue shared by the conv kernels: one wave holds a 32(pixel) x 32(channel)
// accumulator tile; rows [lo, hi_) of it belong to one image.  Returns (mean, M2) of those rows
// for this lane's channel (valid in every lane).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_stats(const f32x16& acc, int hi, int lo, int hi_, float& mean, float& m2) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = drow(r, hi);
        s += (row >= lo && row < hi_) ? acc[r] : 0.f;
    }
    s += __shfl_xor(s, 32, 64);
    const float cnt = (float)(hi_ - lo);
    mean = s / cnt;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = drow(r, hi);
        const float d = acc[r] - mean;
        q += (row >= lo && row < hi_) ? d * d : 0.f;
    }
    q += __shfl_xor(q, 32, 64);
    m2 = q;
}

// ---------------------------------------------------------------------------------------------
// k_conv: implicit GEMM.  x NHWC [B,H,W,Cin] (Cin % 32 == 0), w packed [taps][Cout][Cin],
// y raw NHWC [B,Ho,Wo,Cout].  NORM: x is a RAW conv output; apply relu((x-mean)*rstd) on load.
// ---------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* x;
    const float* w;
    const float* in_mean;  // [B,Cin]
    const float* in_rstd;
    float* y;
    float* part;           // [tiles32][nslots][Cout][2]
    int B, H, W, Cin, Ho, Wo, Cout, ks, stride, pad, nslots;
    long long Mtot;
};

// Epilogue shared by the direct-conv kernels: raw NHWC store (128-B coalesced per half-wave) of a wave's
// 32(pixel) x NT*32(channel) accumulators + the (mean, M2) statistics partials per image slot.
template <int NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const f32x16 (&acc)[NT], long long tbase, int cbase,
                                              int HW, int l31, int hi) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = cbase + t * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long p = tbase + drow(r, hi);
            if (p < a.Mtot) a.y[p * a.Cout + co] = acc[t][r];
        }
    }
    if (tbase < a.Mtot) {
        const long long tile32 = tbase >> 5;
        const int nfirst = (int)(tbase / HW);
        for (int s = 0; s < a.nslots; ++s) {
            const long long ibeg = (long long)(nfirst + s) * HW, iend = ibeg + HW;
            long long lo = ibeg - tbase, hi_ = ((iend < a.Mtot) ? iend : a.Mtot) - tbase;
            if (lo < 0) lo = 0;
            if (hi_ > 32) hi_ = 32;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float mean = 0.f, m2 = 0.f;
                if (hi_ > lo) tile_stats(acc[t], hi, (int)lo, (int)hi_, mean, m2);
                if (hi == 0) {
                    float* o = a.part + ((tile32 * a.nslots + s) * a.Cout + cbase + t * 32 + l31) * 2;
                    o[0] = mean;
                    o[1] = m2;
                }
            }
        }
    }
}

// MW = m-tiles (of 32 pixels) per workgroup (4 -> 128 px, 2 -> 64 px); the 4 waves form an
// MW x (4/MW) grid, each wave owning 32 pixels x NT*32 channels.  <4,4>: 128x128, <4,2>: 128x64,
// <2,1>: 64x64 (finer work units for the late layers, whose 128x128 tiling yields only 392
// workgroups for 256 CUs).  fp32 operands on v_mfma_f32_32x32x2_f32 (DSMIL_CONV=f32).
template <int MW, int NT, bool NORM, int NBUF = 2>
__global__ __launch_bounds__(256, 2) void k_conv(ConvArgs a) {
    constexpr int NWN = 4 / MW;
    constexpr int BM = MW * 32;
    constexpr int TN = NWN * NT * 32;
    constexpr int XPT = BM / 32;  // float4 per thread per activation chunk
    constexpr int WPT = TN / 32;  // float4 per thread per weight chunk
    constexpr int X_TILE = BM * LDK, W_TILE = TN * LDK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem;                   // [NBUF][X_TILE]
    float* sW = smem + NBUF * X_TILE;   // [NBUF][W_TILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % MW, wn = wave / MW;
    const int l31 = lane & 31, hi = lane >> 5;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * TN;
    const int HW = a.Ho * a.Wo;
    const int taps = a.ks * a.ks;
    const int nsteps = taps * (a.Cin / BK);
    const int c4 = tid & 7;

    // per-thread pixel metadata for the rows it stages
    int iy0[XPT], ix0[XPT], nb[XPT], nimg[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const long long p = m0 + (tid >> 3) + 32 * i;
        if (p < a.Mtot) {
            const int n = (int)(p / HW), rem = (int)(p - (long long)n * HW);
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            iy0[i] = oy * a.stride - a.pad;
            ix0[i] = ox * a.stride - a.pad;
            nb[i] = n * a.H * a.W;
            nimg[i] = n;
        } else {
            iy0[i] = -100000; ix0[i] = -100000; nb[i] = 0; nimg[i] = 0;
        }
    }
    f32x4 xreg[XPT], wreg[WPT];
    f32x4 mu[XPT], rs[XPT];
    unsigned okmask = 0;  // bit i: tap of row i is inside the image (else zero padding)
    // stage_load only ISSUES the global loads (branch-free, clamped addresses) so that they fly
    // under this step's MFMAs; normalisation, padding select and the LDS write happen in
    // stage_write, after the MFMAs, where the data is first needed.
    auto stage_load = [&](int st) {
        const int cc = st / taps, tap = st - cc * taps;
        const int kh = tap / a.ks, kw = tap - kh * a.ks;
        const int c0 = cc * BK + c4 * 4;
        if constexpr (NORM) {
            if (tap == 0) {
#pragma unroll
                for (int i = 0; i < XPT; ++i) {
                    mu[i] = *reinterpret_cast<const f32x4*>(a.in_mean + (long long)nimg[i] * a.Cin + c0);
                    rs[i] = *reinterpret_cast<const f32x4*>(a.in_rstd + (long long)nimg[i] * a.Cin + c0);
                }
            }
        }
        okmask = 0;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int iy = iy0[i] + kh, ix = ix0[i] + kw;
            const bool ok = (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);
            okmask |= ok ? (1u << i) : 0u;
            const int iyc = min(max(iy, 0), a.H - 1), ixc = min(max(ix, 0), a.W - 1);
            xreg[i] = *reinterpret_cast<const f32x4*>(a.x + (long long)(nb[i] + iyc * a.W + ixc) * a.Cin + c0);
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int r = (tid >> 3) + 32 * i;
            wreg[i] = *reinterpret_cast<const f32x4*>(a.w + ((long long)tap * a.Cout + n0 + r) * a.Cin + c0);
        }
    };
    auto stage_write = [&](int st) {
        float* x = sX + (st & (NBUF - 1)) * X_TILE;
        float* w = sW + (st & (NBUF - 1)) * W_TILE;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            f32x4 v = xreg[i];
            const bool ok = (okmask >> i) & 1u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (NORM) v[e] = fmaxf((v[e] - mu[i][e]) * rs[i][e], 0.f);
                v[e] = ok ? v[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(x + ((tid >> 3) + 32 * i) * LDK + c4 * 4) = v;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            *reinterpret_cast<f32x4*>(w + ((tid >> 3) + 32 * i) * LDK + c4 * 4) = wreg[i];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    stage_load(0);
    stage_write(0);
    __syncthreads();
    const int frag = l31 * LDK + 4 * hi;
    for (int st = 0; st < nsteps; ++st) {
        if (st + 1 < nsteps) stage_load(st + 1);
        const float* x = sX + (st & (NBUF - 1)) * X_TILE + wm * 32 * LDK + frag;
        const float* w = sW + (st & (NBUF - 1)) * W_TILE + wn * NT * 32 * LDK + frag;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const f32x4 xa = *reinterpret_cast<const f32x4*>(x + kg * 8);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 wb = *reinterpret_cast<const f32x4*>(w + t * 32 * LDK + kg * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j], wb[j], acc[t], 0, 0, 0);
            }
        }
        if constexpr (NBUF == 1) __syncthreads();  // single LDS buffer: everyone done reading first
        if (st + 1 < nsteps) stage_write(st + 1);
        __syncthreads();
    }
    conv_epilogue<NT>(a, acc, m0 + wm * 32, n0 + wn * NT * 32, HW, l31, hi);
}

// --------------------------------------------------------------------------------------------
// k_conv_s6 — the same implicit GEMM on bf16 MFMA over exact three-plane cuts (6 largest plane products per fp32
// product, see agg_split.h): 6/16 of the f32-MFMA time for the same fp32-class result.  Every operand element is
// cut ONCE: weights at pack time (k_pack_conv_s6: [tap][Cin/16][3 planes][Cout][16] bf16, the Winograd kernel's U
// layout), activations by the staging thread as it writes them to LDS (behind the producer's IN + ReLU and the zero
// padding).  16 k per step; an LDS row holds the three planes of its 16 k (96 B) + 16 B pad = 112 B, which makes every
// ds_read_b128 of a fragment conflict-free (row stride 7 slots of 16 B, coprime to 16); two buffers, one barrier per step.
// --------------------------------------------------------------------------------------------
constexpr int S6K = 16;
constexpr int S6LD = 28;   // dwords per LDS row (112 B)

template <int MW, int NT, bool NORM, int NP = NPD>
__global__ __launch_bounds__(256, 2) void k_conv_s6(ConvArgs a) {
    constexpr int PLN = emb_planes<NP>();
    constexpr int NWN = 4 / MW;
    constexpr int BM = MW * 32;
    constexpr int TN = NWN * NT * 32;
    constexpr int XPT = BM / 64;                    // float4 (4 k) per thread per activation step
    constexpr int WIT = 2 * PLN * TN;               // 16-B weight items per step: planes x TN rows x 2 halves
    constexpr int WPT = (WIT + 255) / 256;
    constexpr int X_TILE = BM * S6LD, W_TILE = TN * S6LD;   // dwords
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sX = reinterpret_cast<unsigned*>(smem);       // [2][X_TILE]
    unsigned* sW = sX + 2 * X_TILE;                          // [2][W_TILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % MW, wn = wave / MW;
    const int l31 = lane & 31, hi = lane >> 5;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * TN;
    const int HW = a.Ho * a.Wo;
    const int taps = a.ks * a.ks;
    const int nchunks = a.Cin / S6K;
    const int nsteps = taps * nchunks;
    // staging thread -> (row, 4-channel group): 16 consecutive lanes cover 8 rows x 2 groups, so their ds_write_b64 hit 32
    // distinct banks (row stride 28 dwords = -4 mod 32; the plain (tid >> 2, tid & 3) map had 4 rows x 4 groups per 16 lanes,
    // 2-way conflicts: 30 % of the LDS cycles in profiles/r02_emb were bank conflicts)
    const int c4 = (tid & 1) | (((tid >> 4) & 1) << 1);
    const int xrow = ((tid >> 1) & 7) | ((tid >> 5) << 3);
    const unsigned short* wpk = reinterpret_cast<const unsigned short*>(a.w);

    int iy0[XPT], ix0[XPT], nb[XPT], nimg[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const long long p = m0 + xrow + 64 * i;
        if (p < a.Mtot) {
            const int n = (int)(p / HW), rem = (int)(p - (long long)n * HW);
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            iy0[i] = oy * a.stride - a.pad;
            ix0[i] = ox * a.stride - a.pad;
            nb[i] = n * a.H * a.W;
            nimg[i] = n;
        } else {
            iy0[i] = -100000; ix0[i] = -100000; nb[i] = 0; nimg[i] = 0;
        }
    }
    // this thread's weight items: item e -> plane e / (2 TN); inside a plane 8 consecutive lanes take 8 consecutive cout
    // rows of the same 16-B half (their ds_write_b128 cover all 32 banks once)
    int wsrc[WPT], wdst[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int e = tid + 256 * i;
        const int pl = e / (2 * TN), rem = e - pl * 2 * TN, r = (rem & 7) | ((rem >> 4) << 3), h = (rem >> 3) & 1;
        wsrc[i] = e < WIT ? (pl * a.Cout + n0 + r) * S6K + h * 8 : -1;      // bf16 elements inside a (tap, chunk) slab
        wdst[i] = r * S6LD + pl * 8 + h * 4;                                // dwords
    }
    // Two register sets: the global loads of step st+2 are issued while step st computes (the 24 bf16 MFMAs of a step
    // are ~770 cycles per wave — shorter than an L2 / HBM round trip, so one step of distance leaves the wave waiting
    // for its loads at every stage_write: measured 305 us for the layer-2 strided conv against 71 us of MFMA time).
    struct Regs {
        f32x4 x[XPT], mu[XPT], rs[XPT];
        u32x4_t w[WPT];
        unsigned okmask;
    };
    Regs ra, rb;
    // tap-major inside a channel chunk: the producer's statistics are read once per chunk
    auto stage_load = [&](int st, Regs& R) {
        const int cc = st / taps, tap = st - cc * taps;
        const int kh = tap / a.ks, kw = tap - kh * a.ks;
        const int c0 = cc * S6K + c4 * 4;
        if constexpr (NORM) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) {   // L1/L2-resident [B, Cin] arrays; re-read per step keeps the sets independent
                R.mu[i] = *reinterpret_cast<const f32x4*>(a.in_mean + (long long)nimg[i] * a.Cin + c0);
                R.rs[i] = *reinterpret_cast<const f32x4*>(a.in_rstd + (long long)nimg[i] * a.Cin + c0);
            }
        }
        R.okmask = 0;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int iy = iy0[i] + kh, ix = ix0[i] + kw;
            const bool ok = (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);
            R.okmask |= ok ? (1u << i) : 0u;
            const int iyc = min(max(iy, 0), a.H - 1), ixc = min(max(ix, 0), a.W - 1);
            R.x[i] = *reinterpret_cast<const f32x4*>(a.x + (long long)(nb[i] + iyc * a.W + ixc) * a.Cin + c0);
        }
        const unsigned short* slab = wpk + ((long long)tap * nchunks + cc) * 3 * a.Cout * S6K;
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            R.w[i] = *reinterpret_cast<const u32x4_t*>(slab + (wsrc[i] < 0 ? 0 : wsrc[i]));
    };
    auto stage_write = [&](int st, const Regs& R) {
        unsigned* x = sX + (st & 1) * X_TILE;
        unsigned* w = sW + (st & 1) * W_TILE;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            f32x4 v = R.x[i];
            const bool ok = (R.okmask >> i) & 1u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (NORM) v[e] = fmaxf((v[e] - R.mu[i][e]) * R.rs[i][e], 0.f);
                v[e] = ok ? v[e] : 0.f;
            }
            u32x2_t ph, pm, pl;
            cut4<NP>(v, ph, pm, pl);
            unsigned* d = x + (xrow + 64 * i) * S6LD + c4 * 2;
            *reinterpret_cast<u32x2_t*>(d) = ph;
            if constexpr (PLN >= 2) *reinterpret_cast<u32x2_t*>(d + 8) = pm;
            if constexpr (PLN == 3) *reinterpret_cast<u32x2_t*>(d + 16) = pl;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (wsrc[i] >= 0) *reinterpret_cast<u32x4_t*>(w + wdst[i]) = R.w[i];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int frag = l31 * S6LD + 4 * hi;   // dwords: this lane's row, k-half
    auto mfmas = [&](int st) {
        const unsigned* x = sX + (st & 1) * X_TILE + wm * 32 * S6LD + frag;
        const unsigned* w = sW + (st & 1) * W_TILE + wn * NT * 32 * S6LD + frag;
        Frag16 xa[PLN];
#pragma unroll
        for (int p = 0; p < PLN; ++p) xa[p].u = *reinterpret_cast<const u32x4_t*>(x + p * 8);
        Frag16 wb[NT][PLN];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < PLN; ++p) wb[t][p].u = *reinterpret_cast<const u32x4_t*>(w + t * 32 * S6LD + p * 8);
        // smallest products first (PlaneProducts).  The N-tiles ALTERNATE inside each product, so no MFMA accumulates into
        // the tile the previous one is still writing (a dependent 32x32x16 chain issues every ~48 cycles instead of 32:
        // MI355X_MICROARCH.md, measured constants)
        using PP = PlaneProducts<NP>;
#pragma unroll
        for (int k = 0; k < PP::N; ++k)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = plane_mfma<NP>(xa[PP::X[k]].u, wb[t][PP::W[k]].u, acc[t]);
    };
    // step st: [issue loads of st+2 -> the set st's data just left] [MFMAs on LDS buffer st&1] [set of st+1 -> buffer (st+1)&1]
    stage_load(0, ra);
    stage_write(0, ra);
    if (nsteps > 1) stage_load(1, rb);
    __syncthreads();
    for (int st = 0; st < nsteps; st += 2) {
        if (st + 2 < nsteps) stage_load(st + 2, ra);
        mfmas(st);
        if (st + 1 < nsteps) stage_write(st + 1, rb);   // the other buffer: its readers passed the previous barrier
        __syncthreads();
        if (st + 1 < nsteps) {
            if (st + 3 < nsteps) stage_load(st + 3, rb);
            mfmas(st + 1);
            if (st + 2 < nsteps) stage_write(st + 2, ra);
            __syncthreads();
        }
    }
    if constexpr (emb_f16<NP>()) {                  // the weights carry 2^EMB_WSHIFT
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] *= EMB_OSCALE;
    }
    conv_epilogue<NT>(a, acc, m0 + wm * 32, n0 + wn * NT * 32, HW, l31, hi);
}

// conv weight [O][I][k][k] -> operand planes [tap][I/16][3][O][16] (the third plane slot stays in the layout for every form):
// np = 3: two fp16 planes of 2^EMB_WSHIFT w; else three truncated bf16 planes
__global__ void k_pack_conv_s6(const float* __restrict__ w, unsigned short* __restrict__ out, int O, int I, int taps, int np) {
    const long long total = (long long)O * I * taps;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % taps);
        const long long r = i / taps;
        const int ci = (int)(r % I), o = (int)(r / I);
        const float v = w[i];                                   // OIHW: ((o*I + ci)*taps + t)
        const long long base = ((((long long)t * (I / 16) + ci / 16) * 3) * O + o) * 16 + (ci & 15);
        if (np == 3 || np == 1) {
            out[base] = pack_h0(v * EMB_WSCALE);
            out[base + (long long)O * 16] = np == 3 ? pack_h1(v * EMB_WSCALE) : (unsigned short)0;
            out[base + 2LL * O * 16] = 0;
            continue;
        }
        const unsigned hb = __float_as_uint(v) & 0xFFFF0000u;
        const float r1 = v - __uint_as_float(hb);
        const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
        const unsigned lb = __float_as_uint(r1 - __uint_as_float(mb));
        out[base] = (unsigned short)(hb >> 16);
        out[base + (long long)O * 16] = (unsigned short)(mb >> 16);
        out[base + 2LL * O * 16] = (unsigned short)(lb >> 16);
    }
}

constexpr int FIN_U = 8;   // partial loads in flight per thread in the k_in_finalize_stem / _cnt walks
// partials of flattened 32-pixel tiles -> mean / rstd per (image, channel).  One workgroup per
// image; thread (c mod 64, g) sums the tiles t = g mod 4 of its channel: first the weighted means,
// then M2 about the image mean (M2 = sum M2_t + cnt_t (mean_t - mean)^2) — two flat reductions,
// no serial Chan chain.
__global__ __launch_bounds__(256) void k_in_finalize_flat(const float* __restrict__ part, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int B, int HW, int C,
                                                         int nslots, long long Mtot) {
    const int n = blockIdx.x;
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long long ibeg = (long long)n * HW, iend = ibeg + HW;
    const long long t0 = ibeg >> 5, t1 = (iend - 1) >> 5;
    __shared__ float red[4][64];
    // grid.y = C / 64: one 64-channel group per workgroup (a deep layer's 8 groups used to be 8 serial rounds of two
    // block-wide reductions in one workgroup per image — pure latency)
    for (int c = (int)blockIdx.y * 64 + cl; c < C && c < ((int)blockIdx.y + 1) * 64; c += 64) {
        float s = 0.f;
        for (long long t = t0 + g; t <= t1; t += 4) {
            const long long tb = t << 5;
            const int sl = n - (int)(tb / HW);
            long long lo = ibeg - tb, hi = ((iend < Mtot) ? iend : Mtot) - tb;
            if (lo < 0) lo = 0;
            if (hi > 32) hi = 32;
            if (hi <= lo || sl < 0 || sl >= nslots) continue;
            s += (float)(hi - lo) * part[((t * nslots + sl) * C + c) * 2];
        }
        __syncthreads();
        red[g][cl] = s;
        __syncthreads();
        const float mu = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)HW;
        float q = 0.f;
        for (long long t = t0 + g; t <= t1; t += 4) {
            const long long tb = t << 5;
            const int sl = n - (int)(tb / HW);
            long long lo = ibeg - tb, hi = ((iend < Mtot) ? iend : Mtot) - tb;
            if (lo < 0) lo = 0;
            if (hi > 32) hi = 32;
            if (hi <= lo || sl < 0 || sl >= nslots) continue;
            const float* o = part + ((t * nslots + sl) * C + c) * 2;
            const float d = o[0] - mu;
            q += o[1] + (float)(hi - lo) * d * d;
        }
        __syncthreads();
        red[g][cl] = q;
        __syncthreads();
        if (g == 0) {
            const float m2 = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
            mean[(long long)n * C + c] = mu;
            rstd[(long long)n * C + c] = 1.0f / sqrtf(m2 / (float)HW + IN_EPS);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_conv_wino: 3x3 stride-1 pad-1 convolution by Winograd F(2x2, 3x3), fused end to end
// (13 of the 20 convolutions, ~80 % of the direct-form FLOPs; 2.25x fewer MFMA FLOPs).
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A,   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],
//   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
// Work unit: up to 32 output tiles (2x2 pixels each) arranged as IB images x TYB x TXB tiles (the
// host picks the shape that fills the 32 MFMA rows best per layer) x 64 output channels.
// 4 waves = 2 channel groups (wn) x 2 position halves (wp: transform rows xi in {2wp, 2wp+1});
// a wave owns 32 tiles x 32 channels x 8 positions = 8 accumulator tiles (128 VGPRs), so TWO
// independent workgroups share a CU (2 waves per SIMD) and hide each other's prologue, epilogue
// and barrier stalls — single-wave-per-SIMD versions with all 16 positions per wave were 3x off.
// Data flow per 8-channel chunk, one barrier interval:
//   * the unit's input region ((2TYB+2) x (2TXB+2) pixels per image) is loaded ONCE from global
//     into a raw LDS buffer (no per-tile re-fetch of overlapping 4x4 patches);
//   * each thread transforms one (tile, 4-channel group, column half h, row half v): 9 ds_reads of
//     raw pixels, IN + ReLU of the producer and zero padding applied on the fly, B^T d B for
//     4 of the 16 positions, 4 ds_writes into V[16][32][8] of the NEXT chunk (double buffered);
//   * the pre-transformed weights U[16][C/8][Cout][8] are read by each lane straight from L2 in
//     MFMA B-operand order (fully coalesced, no LDS), re-issued position by position;
//   * 8 positions x 4 MFMAs (K = 8) on the CURRENT chunk.
// The inverse transform is linear: each position half reduces its 8 accumulator tiles to four
// partial outputs, the wp = 1 wave hands them to its wp = 0 partner through LDS, which stores the
// raw NHWC pixels and the (cnt, mean, M2) statistics partials per (image, unit).
// ---------------------------------------------------------------------------------------------
constexpr int WK = 8;            // channels per chunk
constexpr int WLD = WK + 4;      // LDS row stride (48 B): conflict-free ds_read_b128
constexpr int WTT = 32;          // tile slots per unit
constexpr int WTILE = 16 * WTT * WLD;
constexpr int WRAW_MAX = 256;    // raw-region pixels per unit the host may choose (x WLD floats)
constexpr int WRPT = (WRAW_MAX * 2 + 255) / 256;  // raw float4 loads per thread per chunk

struct WinoArgs {
    const float* x;        // NHWC [B,H,W,C]
    const float* u;        // [16][C/8][Cout][8]
    const float* in_mean;  // [B,C] or null
    const float* in_rstd;
    float* y;              // raw NHWC [B,H,W,Cout]
    float* part;           // [B][PB][2][Cout][3]   (cnt, mean, M2) per output-row half
    int B, H, W, C, Cout, TY, TX;
    int IB, TYB, TXB;      // unit shape: images x tile rows x tile cols (IB*TYB*TXB <= 32)
    int nby, nbx, PB;      // units per image along y / x, PB = nby*nbx
    int expt;              // DSMIL_WINO_EXPT ablation knob (0 in production)
#ifdef DSMIL_TRACE
    unsigned long long* trace;   // trace builds: s_memtime stamps of workgroups 0-3 (waves 0 and 4), [wg][role][step][8]
#endif
};

// Epilogue shared by the Winograd kernels: inverse transform of this wave's 8 positions, exchange of
// the partner wave's output row through LDS (smem must be free: call behind a barrier), raw NHWC
// store and the (cnt, mean, M2) statistics partials.  acc[p][r] = M[position 8wp+p][slot drow(r,hi)][cout].
__device__ __forceinline__ void wino_epilogue(const WinoArgs& a, const f32x16 (&acc)[8], float* smem, int lane,
                                              int wn, int wp, int n0, int img0, int ty0, int tx0, int tpi, int pb,
                                              float oscale = 1.f) {   // oscale: 2^-EMB_WSHIFT for the fp16 form (exact)
    const int l31 = lane & 31, hi = lane >> 5;
    // ---- inverse transform: partial outputs of this wave's two xi rows
    //   ra[xi] = m[xi][0]+m[xi][1]+m[xi][2], rb[xi] = m[xi][1]-m[xi][2]-m[xi][3]
    //   Y[0][.] = r.[0]+r.[1]+r.[2],  Y[1][.] = r.[1]-r.[2]-r.[3]
    float y00[16], y01[16], y10[16], y11[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float raA = acc[0][r] + acc[1][r] + acc[2][r], rbA = acc[1][r] - acc[2][r] - acc[3][r];  // xi = 2wp
        const float raB = acc[4][r] + acc[5][r] + acc[6][r], rbB = acc[5][r] - acc[6][r] - acc[7][r];  // xi = 2wp+1
        if (wp == 0) { y00[r] = raA + raB; y01[r] = rbA + rbB; y10[r] = raB; y11[r] = rbB; }
        else { y00[r] = raA; y01[r] = rbA; y10[r] = -raA - raB; y11[r] = -rbA - rbB; }
    }
    // exchange through LDS (the V buffers are free now): the wp = 0 wave finishes output row 0
    // (y00, y01), the wp = 1 wave output row 1 (y10, y11); each sends the other row's partials
    float* xch = smem + ((wn * 2 + wp) * 64 + lane) * 33;        // [2 wn][2 wp][64 lanes][32 (+1 pad)]
    float* xpr = smem + ((wn * 2 + (wp ^ 1)) * 64 + lane) * 33;  // the partner's slot
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        xch[r] = wp == 0 ? y10[r] : y00[r];
        xch[16 + r] = wp == 0 ? y11[r] : y01[r];
    }
    __syncthreads();
    float ya[16], yb[16];  // this wave's output row: pixels (oy+wp, ox) and (oy+wp, ox+1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        ya[r] = ((wp == 0 ? y00[r] : y10[r]) + xpr[r]) * oscale;
        yb[r] = ((wp == 0 ? y01[r] : y11[r]) + xpr[16 + r]) * oscale;
    }
    // ---- raw store + statistics partials
    const int co = n0 + wn * 32 + l31;
    unsigned vmask[16];  // 2 validity bits per accumulator row (tile slot)
    int rimg[16];        // local image of the row's tile slot (or -1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int slot = drow(r, hi);
        const int il = slot / tpi, rem = slot - il * tpi, tyl = rem / a.TXB, txl = rem - tyl * a.TXB;
        const int n = img0 + il, ty = ty0 + tyl, tx = tx0 + txl;
        unsigned vm = 0;
        rimg[r] = -1;
        const int oy = 2 * ty + wp, ox = 2 * tx;
        if (slot < a.IB * tpi && n < a.B && ty < a.TY && tx < a.TX && oy < a.H) {
            float* o = a.y + ((long long)(n * a.H + oy) * a.W + ox) * a.Cout + co;
            o[0] = ya[r]; vm = 1u;
            if (ox + 1 < a.W) { o[a.Cout] = yb[r]; vm |= 2u; }
            rimg[r] = il;
        }
        vmask[r] = vm;
    }
    for (int il = 0; il < a.IB; ++il) {
        const int n = img0 + il;
        if (n >= a.B) break;
        float sum = 0.f, cnt = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned vm = (rimg[r] == il) ? vmask[r] : 0u;
            sum += ((vm & 1u) ? ya[r] : 0.f) + ((vm & 2u) ? yb[r] : 0.f);
            cnt += (float)__popc(vm);
        }
        sum += __shfl_xor(sum, 32, 64);
        cnt += __shfl_xor(cnt, 32, 64);
        const float mean = cnt > 0.f ? sum / cnt : 0.f;
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned vm = (rimg[r] == il) ? vmask[r] : 0u;
            const float d0 = ya[r] - mean, d1 = yb[r] - mean;
            q += ((vm & 1u) ? d0 * d0 : 0.f) + ((vm & 2u) ? d1 * d1 : 0.f);
        }
        q += __shfl_xor(q, 32, 64);
        if (hi == 0) {
            float* o = a.part + ((((long long)n * a.PB + pb) * 2 + wp) * a.Cout + co) * 3;
            o[0] = cnt; o[1] = mean; o[2] = q;
        }
    }
}

template <bool NORM>
__global__ __launch_bounds__(256, 2) void k_conv_wino(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sV = smem;                       // [2][WTILE]
    float* sR = smem + 2 * WTILE;           // [2][WRAW_MAX * WLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wp = wave >> 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.y * 64;
    const int nchunks = a.C / WK;
    // ---- which unit
    int bid = blockIdx.x;
    const int bx = bid % a.nbx; bid /= a.nbx;
    const int by = bid % a.nby; bid /= a.nby;
    const int img0 = bid * a.IB;
    const int ty0 = by * a.TYB, tx0 = bx * a.TXB;
    const int pb = by * a.nbx + bx;
    const int RH = 2 * a.TYB + 2, RW = 2 * a.TXB + 2, RP = RH * RW;   // raw region per image
    const int tpi = a.TYB * a.TXB;                                   // tile slots per image
    const int iy_org = 2 * ty0 - 1, ix_org = 2 * tx0 - 1;            // image coords of raw (0,0)

    // ---- raw staging role: element e = tid + 256 q -> (pixel e>>1, channel group e&1).  The producer's
    // IN + ReLU and the zero padding are applied HERE, once per staged pixel, so the transform below
    // is pure adds (a first version normalised and masked per tile: ~2x the VALU work, and the
    // instruction stream, not the MFMA pipe, was the limit).
    int roff[WRPT];   // global element offset of the pixel (x C); -1 = zero padding, -2 = unused slot
    int rlds[WRPT];   // LDS float offset inside a raw buffer
    int rsto[WRPT];   // element offset of the pixel's image in in_mean / in_rstd
#pragma unroll
    for (int q = 0; q < WRPT; ++q) {
        const int e = tid + 256 * q, px = e >> 1, gg = e & 1;
        roff[q] = -2; rlds[q] = 0; rsto[q] = gg * 4;
        if (px < a.IB * RP) {
            const int il = px / RP, rem = px - il * RP, ry = rem / RW, rx = rem - ry * RW;
            const int n = img0 + il, iy = iy_org + ry, ix = ix_org + rx;
            roff[q] = -1;
            if (n < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                roff[q] = ((n * a.H + iy) * a.W + ix) * a.C + gg * 4;
                rsto[q] = n * a.C + gg * 4;
            }
            rlds[q] = px * WLD + gg * 4;
        }
    }
    // ---- transform role: channel group g, tile slot ts, column half h (nu in {2h,2h+1}),
    //      row half v (xi in {2v,2v+1}): patch rows v..v+2, columns h..h+2
    const int g = tid & 1, ts = (tid >> 1) & 31, h = (tid >> 6) & 1, v = tid >> 7;
    const int sil = ts / tpi, srem = ts - sil * tpi, styl = srem / a.TXB, stxl = srem - styl * a.TXB;
    // LDS float offset of this role's first patch pixel (row 2*styl+v, col 2*stxl+h) in a raw buffer;
    // slots past the unit's tiles read pixel 0 (their accumulator rows are never stored)
    const int praw = (ts < a.IB * tpi) ? ((sil * RH + 2 * styl + v) * RW + 2 * stxl + h) * WLD + g * 4 : g * 4;

    f32x4 rreg[WRPT], rmu[WRPT], rrs[WRPT];
    auto raw_load = [&](int cc) {
#pragma unroll
        for (int q = 0; q < WRPT; ++q) {
            const int off = roff[q] < 0 ? 0 : roff[q];
            rreg[q] = *reinterpret_cast<const f32x4*>(a.x + (long long)off + cc * WK);
            if constexpr (NORM) {
                rmu[q] = *reinterpret_cast<const f32x4*>(a.in_mean + rsto[q] + cc * WK);
                rrs[q] = *reinterpret_cast<const f32x4*>(a.in_rstd + rsto[q] + cc * WK);
            }
        }
    };
    auto raw_write = [&](int cc) {
        float* r = sR + (cc & 1) * (WRAW_MAX * WLD);
#pragma unroll
        for (int q = 0; q < WRPT; ++q) {
            if (roff[q] == -2) continue;
            f32x4 x = rreg[q];
            const bool ok = roff[q] >= 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float xv = x[e];
                if constexpr (NORM) xv = fmaxf((xv - rmu[q][e]) * rrs[q][e], 0.f);
                x[e] = ok ? xv : 0.f;
            }
            *reinterpret_cast<f32x4*>(r + rlds[q]) = x;
        }
    };
    // transform this role's share of chunk cc: raw[cc&1] -> V[cc&1][positions (2v+{0,1})*4 + 2h+{0,1}]
    auto transform = [&](int cc) {
        const float* r = sR + (cc & 1) * (WRAW_MAX * WLD) + praw;
        float* vv = sV + (cc & 1) * WTILE;
        f32x4 d[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) d[i * 3 + jj] = *reinterpret_cast<const f32x4*>(r + (i * RW + jj) * WLD);
        // rows (B^T d): v = 0 -> xi 0,1 from patch rows 0,1,2; v = 1 -> xi 2,3 from patch rows 1,2,3
        f32x4 t[2][3];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            if (v == 0) { t[0][jj] = d[jj] - d[6 + jj]; t[1][jj] = d[3 + jj] + d[6 + jj]; }
            else { t[0][jj] = d[3 + jj] - d[jj]; t[1][jj] = d[jj] - d[6 + jj]; }
        }
        // columns (.. B): h = 0 -> nu 0,1 from patch columns 0,1,2; h = 1 -> nu 2,3 from columns 1,2,3
#pragma unroll
        for (int x2 = 0; x2 < 2; ++x2) {
            f32x4 va, vb;
            if (h == 0) { va = t[x2][0] - t[x2][2]; vb = t[x2][1] + t[x2][2]; }
            else { va = t[x2][1] - t[x2][0]; vb = t[x2][0] - t[x2][2]; }
            const int pos = (2 * v + x2) * 4 + 2 * h;
            *reinterpret_cast<f32x4*>(vv + (pos * WTT + ts) * WLD + g * 4) = va;
            *reinterpret_cast<f32x4*>(vv + ((pos + 1) * WTT + ts) * WLD + g * 4) = vb;
        }
    };
    // B operand (weights) of position p (of this wave's half) for this lane — straight from
    // global/L2; one running pointer per position, advanced by a chunk (Cout*8 floats) per step
    const float* up[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
        up[p] = a.u + ((long long)(8 * wp + p) * nchunks * a.Cout + n0 + wn * 32 + l31) * WK + 4 * hi;
    const int ustep = a.Cout * WK;

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    f32x4 ub[8];
    // ---- prologue: raw(0) -> LDS, V(0), raw(1) -> LDS, raw(2) -> registers, weights of chunk 0
    raw_load(0);
#pragma unroll
    for (int p = 0; p < 8; ++p) { ub[p] = *reinterpret_cast<const f32x4*>(up[p]); up[p] += ustep; }
    raw_write(0);
    __syncthreads();
    transform(0);
    if (nchunks > 1) { raw_load(1); raw_write(1); }
    if (nchunks > 2) raw_load(2);
    __syncthreads();
    const int vfo = ((8 * wp) * WTT + l31) * WLD + 4 * hi;
    // iteration cc: transform(cc+1) [raw(cc+1) was written an iteration ago], MFMAs(cc) with the weight
    // loads of chunk cc+1 re-issued behind each position, then raw(cc+2) registers -> LDS (loaded a
    // whole iteration ago) and the global loads of raw(cc+3) into the same registers.
    for (int cc = 0; cc < nchunks; ++cc) {
        const bool more = cc + 1 < nchunks, more2 = cc + 2 < nchunks, more3 = cc + 3 < nchunks;   // block-uniform
        if (more && !DSMIL_WEXPT_ON(a, 1)) transform(cc + 1);
        const float* vb = sV + (cc & 1) * WTILE + vfo;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const f32x4 va = *reinterpret_cast<const f32x4*>(vb + p * WTT * WLD);
            const f32x4 w = ub[p];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[j], w[j], acc[p], 0, 0, 0);
            if (more && !DSMIL_WEXPT_ON(a, 4)) { ub[p] = *reinterpret_cast<const f32x4*>(up[p]); up[p] += ustep; }
        }
        if (!DSMIL_WEXPT_ON(a, 2)) {
            if (more2) raw_write(cc + 2);   // raw[cc&1] was consumed by transform(cc) an iteration ago
            if (more3) raw_load(cc + 3);
        }
        __syncthreads();
    }
    if (DSMIL_WEXPT_ON(a, 8)) {  // ablation: no epilogue
        float keep = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[p][r];
        if (keep == 123.456f) a.y[0] = keep;
        return;
    }
    wino_epilogue(a, acc, smem, lane, wn, wp, n0, img0, ty0, tx0, tpi, pb);
}

// --------------------------------------------------------------------------------------------
// k_conv_wino_s3: the same Winograd F(2x2,3x3) unit on bf16 MFMA over EXACT three-plane cuts of the
// fp32 operands (see agg_split.h): U is cut at pack time, V by the transform threads when they write
// it; every fp32 product U*V is the sum of nine exact plane products accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 (9/16 of the f32 MFMA time).  On gfx950 the f32 "MFMA" shares the VALU's
// fp32 FMA rate, so the transform / staging VALU work of k_conv_wino costs MFMA throughput; the bf16
// MFMA runs beside the VALU.  16-channel chunks (one MFMA k step); LDS is single-buffered so that two
// workgroups still share a CU (the partner covers the serial phases):
//   raw[256 px][16 ch (+4)] fp32  20 KB      V[16 pos][32 slots][3 planes x 16 bf16 (+16 B)]  56 KB
//   per chunk:  MFMAs(cc) | raw(cc+1) regs -> LDS, global loads of raw(cc+2) | barrier |
//               transform(cc+1): raw -> V planes | barrier
// Weights U[16 pos][C/16][3 planes][Cout][16] bf16 are read by each lane straight from L2 in MFMA
// B-operand order (1 KiB coalesced per wave instruction), one position ahead.
// --------------------------------------------------------------------------------------------
constexpr int SK = 16;                               // channels per chunk
constexpr int SRLD = SK + 4;                         // raw LDS row stride (floats, 80 B)
constexpr int SVLD = 28;                             // V LDS slot stride (dwords, 112 B)
constexpr int SV_DW = 16 * WTT * SVLD;               // dwords of V
constexpr int SRPT = (WRAW_MAX * 4 + 255) / 256;     // raw float4 per thread per chunk

// UD: how many transform positions ahead the weight fragments are requested (1 or 2); LS: the producer's
// (mean, rstd) of the next chunk are staged through LDS by a few threads instead of being loaded by every staging
// thread right before use.
// NP: plane products per fp32 product (9 = every product formed exactly; 6 = the three smallest, together < 2^-20 of
// |u*v|, left out — see agg_split.h).
#ifdef DSMIL_TRACE
#define PP_STAMP(slot)                                                                                              \
    do {                                                                                                            \
        if (a.trace && blockIdx.x < 4 && lane == 0 && (wave8 & 3) == 0 && s < 256)                                  \
            a.trace[(((long long)blockIdx.x * 2 + (wave8 >> 2)) * 256 + s) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PP_STAMP(slot) do { } while (0)
#endif
#ifdef DSMIL_TRACE
#define UNIT_STAMP(slot)                                                                                            \
    do {                                                                                                            \
        if (a.trace && blockIdx.x < 4 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == NT / 64 - 1) && cc < 256) \
            a.trace[(((long long)blockIdx.x * 2 + (wave != 0)) * 256 + cc) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define UNIT_STAMP(slot) do { } while (0)
#endif
// WNN: cout tiles of 32 per workgroup (2: 64 couts, 256 threads, two workgroups per CU; 4: 128 couts, 512 threads, one
// workgroup per CU): the staged + transformed input of a unit serves WNN x 32 couts, so the wide form does HALF the
// staging / transform / cut work per MFMA (that work, not the MFMAs, bounds the narrow form: stamps in profiles/README.md).
// UC: the weight-fragment ring runs on ACROSS chunks (needs a ring that divides the 8 positions: UD = 3): the first
// positions of a chunk find their fragments already there instead of paying an L2 round trip behind the barrier.
// Measured on the 128-cout form: 39.98k patches/s against 40.78k without (8-15 registers spill into the staging phase);
// compiled only with -DWIDE_UC=true.
template <bool NORM, int UD = 1, bool LS = false, int NP = 9, int WNN = 2, bool UC = false>
__global__ __launch_bounds__(WNN * 128, 2) void k_conv_wino_s3(WinoArgs a) {
    constexpr int NT = WNN * 128;                       // threads
    static_assert(!UC || UD == 3, "carrying the ring needs (UD + 1) | 8");
    constexpr int RPT = (WRAW_MAX * 4 + NT - 1) / NT;   // raw float4 per thread per chunk
    constexpr int XH = NT / 256;                        // threads sharing one (g, ts, h) transform role: 4 / XH xi rows each
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sV = reinterpret_cast<unsigned*>(smem);   // [16][WTT][SVLD] dwords
    float* sR = smem + SV_DW;                           // [WRAW_MAX][SRLD]
    float* sS = sR + WRAW_MAX * SRLD;                   // LS: [2 buffers][16 images][2 (mean, rstd)][16 ch]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WNN, wp = wave / WNN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.y * (WNN * 32);
    const int nchunks = a.C / SK;
    int bid = blockIdx.x;
    const int bx = bid % a.nbx; bid /= a.nbx;
    const int by = bid % a.nby; bid /= a.nby;
    const int img0 = bid * a.IB;
    const int ty0 = by * a.TYB, tx0 = bx * a.TXB;
    const int pb = by * a.nbx + bx;
    const int RH = 2 * a.TYB + 2, RW = 2 * a.TXB + 2, RP = RH * RW;
    const int tpi = a.TYB * a.TXB;
    const int iy_org = 2 * ty0 - 1, ix_org = 2 * tx0 - 1;

    // ---- raw staging role: element e = tid + 256 q -> (pixel, channel group); 8 consecutive lanes take 8 consecutive
    //      pixels of one group, so that a ds_write_b128 lane group (8 lanes, 80-B rows) covers the 32 banks once
    int roff[RPT], rlds[RPT], rsto[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int e = tid + NT * q, px = (e & 7) | ((e >> 5) << 3), gg = (e >> 3) & 3;
        roff[q] = -2; rlds[q] = 0; rsto[q] = gg * 4;
        if (px < a.IB * RP) {
            const int il = px / RP, rem = px - il * RP, ry = rem / RW, rx = rem - ry * RW;
            const int n = img0 + il, iy = iy_org + ry, ix = ix_org + rx;
            roff[q] = -1;
            if (n < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                roff[q] = ((n * a.H + iy) * a.W + ix) * a.C + gg * 4;
                rsto[q] = LS ? il * 32 + gg * 4 : n * a.C + gg * 4;
            }
            rlds[q] = px * SRLD + gg * 4;
        }
    }
    // LS: thread t < 128 fetches 4 channels of (mean | rstd) of local image t>>3 for the chunk two ahead
    f32x4 sreg = {0.f, 0.f, 0.f, 0.f};
    auto stat_load = [&](int cc) {
        if constexpr (NORM && LS) {
            if (tid < 128) {
                const int il = tid >> 3, which = (tid >> 2) & 1, c4 = tid & 3;
                const int n = img0 + il < a.B ? img0 + il : a.B - 1;
                sreg = *reinterpret_cast<const f32x4*>((which ? a.in_rstd : a.in_mean) + (long long)n * a.C + cc * SK + c4 * 4);
            }
        }
    };
    auto stat_write = [&](int cc) {
        if constexpr (NORM && LS) {
            if (tid < 128) {
                const int il = tid >> 3, which = (tid >> 2) & 1, c4 = tid & 3;
                *reinterpret_cast<f32x4*>(sS + (cc & 1) * 512 + il * 32 + which * 16 + c4 * 4) = sreg;
            }
        }
    };
    // ---- transform role: channel group g (4 channels), tile slot ts, column half h (nu in {2h,2h+1});
    //      all four xi rows of the patch columns h..h+2
    //      Lane -> (g, ts): 16 consecutive lanes hold 4 groups x the EVEN (or odd) slots of 8, which makes the 8-byte plane
    //      writes below conflict-free (slot stride 28 dwords) and the window reads 1.5-way instead of 2-way (LDS simulation
    //      over all lane-bit assignments; profiles/r02_emb had 38 % of this kernel's LDS cycles in bank conflicts)
    const int g = tid & 3, ts = ((lane >> 5) & 1) | (((lane >> 2) & 7) << 1) | ((wave & 1) << 4);
    const int h = (wave >> 1) & 1;   // wave-uniform
    const int xh = wave >> 2;        // XH == 2: which pair of xi rows (0 when XH == 1)
    const int sil = ts / tpi, srem = ts - sil * tpi, styl = srem / a.TXB, stxl = srem - styl * a.TXB;
    const int praw = (ts < a.IB * tpi) ? ((sil * RH + 2 * styl) * RW + 2 * stxl + h) * SRLD + g * 4 : g * 4;

    f32x4 rreg[RPT];
    auto raw_load = [&](int cc) {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int off = roff[q] < 0 ? 0 : roff[q];
            rreg[q] = *reinterpret_cast<const f32x4*>(a.x + (long long)off + cc * SK);
        }
    };
    auto raw_write = [&](int cc) {   // producer's IN + ReLU and the zero padding applied once per staged pixel
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            if (roff[q] == -2) continue;
            f32x4 x = rreg[q];
            const bool ok = roff[q] >= 0;
            if constexpr (NORM) {
                // (mean, rstd) are read here, not prefetched with the pixels: 32 more live registers
                // across the MFMA phase spill (measured: 132 B of scratch)
                f32x4 mu, rs;
                if constexpr (LS) {
                    mu = *reinterpret_cast<const f32x4*>(sS + (cc & 1) * 512 + rsto[q]);
                    rs = *reinterpret_cast<const f32x4*>(sS + (cc & 1) * 512 + rsto[q] + 16);
                } else {
                    mu = *reinterpret_cast<const f32x4*>(a.in_mean + rsto[q] + cc * SK);
                    rs = *reinterpret_cast<const f32x4*>(a.in_rstd + rsto[q] + cc * SK);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = fmaxf((x[e] - mu[e]) * rs[e], 0.f);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = ok ? x[e] : 0.f;
            *reinterpret_cast<f32x4*>(sR + rlds[q]) = x;
        }
    };
    // raw -> V planes: B^T d B of this role's 4x3 patch window, 8 outputs (4 xi x 2 nu), each cut and
    // written as three 8-byte pieces (plane, k = 4g..4g+3) of slot ts
    auto transform = [&]() {
        const float* r = sR + praw;
        f32x4 T[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4 R0 = *reinterpret_cast<const f32x4*>(r + (0 * RW + c) * SRLD);
            const f32x4 R1 = *reinterpret_cast<const f32x4*>(r + (1 * RW + c) * SRLD);
            const f32x4 R2 = *reinterpret_cast<const f32x4*>(r + (2 * RW + c) * SRLD);
            const f32x4 R3 = *reinterpret_cast<const f32x4*>(r + (3 * RW + c) * SRLD);
            T[0][c] = R0 - R2; T[1][c] = R1 + R2; T[2][c] = R2 - R1; T[3][c] = R1 - R3;
        }
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            if (XH == 2 && (xi >> 1) != xh) continue;   // the other thread of this role does these rows
            f32x4 o0, o1;
            if (h == 0) { o0 = T[xi][0] - T[xi][2]; o1 = T[xi][1] + T[xi][2]; }
            else { o0 = T[xi][1] - T[xi][0]; o1 = T[xi][0] - T[xi][2]; }
            const int pos = xi * 4 + 2 * h;
            u32x2_t ph, pm, pl;
            unsigned* d0 = sV + (pos * WTT + ts) * SVLD + g * 2;
            cut4<NP>(o0, ph, pm, pl);
            *reinterpret_cast<u32x2_t*>(d0) = ph;
            if constexpr (emb_planes<NP>() >= 2) *reinterpret_cast<u32x2_t*>(d0 + 8) = pm;
            if constexpr (emb_planes<NP>() == 3) *reinterpret_cast<u32x2_t*>(d0 + 16) = pl;
            unsigned* d1 = d0 + WTT * SVLD;
            cut4<NP>(o1, ph, pm, pl);
            *reinterpret_cast<u32x2_t*>(d1) = ph;
            if constexpr (emb_planes<NP>() >= 2) *reinterpret_cast<u32x2_t*>(d1 + 8) = pm;
            if constexpr (emb_planes<NP>() == 3) *reinterpret_cast<u32x2_t*>(d1 + 16) = pl;
        }
    };
    // weights, TILED (k_pack_wino_s3, tiled = 1: [Cout/32][C/16][16 pos][3 planes][32 couts][16] bf16 — the 24 fragments a wave
    // needs for one chunk lie inside one 48 KiB slab) and fetched by buffer loads: one resource for the tensor, the lane's 16
    // bytes as the only address VGPR, a wave-uniform soffset and the 12-bit immediate — no per-load 64-bit address arithmetic
    // on the VALU (see wino_w1.h, which introduced the layout)
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u), 0, (int)((long long)a.C * a.Cout * 96), 0x00020000);
    const int ulane_b = l31 * 32 + hi * 16;
    const int utile_b = (((n0 >> 5) + wn) * nchunks) * (48 * 1024) + (8 * wp) * 3072;   // wave-uniform: cout tile, position half
    constexpr int PLN = emb_planes<NP>();
    auto uload = [&](int p, int cc, u32x4_t (&w)[3]) {
        const int sb = utile_b + cc * (48 * 1024);
#pragma unroll
        for (int pl = 0; pl < PLN; ++pl) {
            const int f = p * 3 + pl;                        // fragment inside the wave's 24 KiB of the slab
            w[pl] = __builtin_amdgcn_raw_buffer_load_b128(urs, ulane_b + (f & 3) * 1024, sb + (f >> 2) * 4096, 0);
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // ---- prologue: raw(0) -> LDS -> V(0); raw(1) in registers
    raw_load(0);
    stat_load(0);
    stat_write(0);
    if constexpr (NORM && LS) __syncthreads();
    raw_write(0);
    if (nchunks > 1) { raw_load(1); stat_load(1); }
    __syncthreads();
    transform();
    if (nchunks > 1) stat_write(1);
    __syncthreads();
    const int vfo = ((8 * wp) * WTT + l31) * SVLD + 4 * hi;   // dwords
    union Frag { u32x4_t u; bf16x8_t v; };
    u32x4_t w[UD + 1][3];
    if constexpr (UC) {
#pragma unroll
        for (int p = 0; p < UD; ++p) uload(p, 0, w[p]);
    }
    for (int cc = 0; cc < nchunks; ++cc) {
        const bool more = cc + 1 < nchunks, more2 = cc + 2 < nchunks;   // block-uniform
        UNIT_STAMP(0);
        // ---- 8 positions x NP plane products
        if constexpr (!UC) {
            uload(0, cc, w[0]);
            if constexpr (UD >= 2) uload(1, cc, w[1]);
            if constexpr (UD >= 3) uload(2, cc, w[2]);
        }
        if (more2) stat_load(cc + 2);
        // V fragments: one position ahead where the registers allow it (the 512-thread form), else read at use
        constexpr bool VA2 = WNN == 4;
        Frag vq[2][3];
        if constexpr (VA2) {
#pragma unroll
            for (int pl = 0; pl < PLN; ++pl) vq[0][pl].u = *reinterpret_cast<const u32x4_t*>(sV + vfo + pl * 8);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (p + UD < 8) { if (!DSMIL_WEXPT_ON(a, 4)) uload(p + UD, cc, w[(p + UD) % (UD + 1)]); }
            else if (UC) uload(p + UD - 8, more ? cc + 1 : cc, w[(p + UD) % (UD + 1)]);   // next chunk's first positions
            Frag va[3], wb[3];
#pragma unroll
            for (int pl = 0; pl < PLN; ++pl) {
                if constexpr (VA2) {
                    if (p < 7) vq[(p + 1) & 1][pl].u = *reinterpret_cast<const u32x4_t*>(sV + vfo + (p + 1) * WTT * SVLD + pl * 8);
                    va[pl] = vq[p & 1][pl];
                } else {
                    va[pl].u = *reinterpret_cast<const u32x4_t*>(sV + vfo + p * WTT * SVLD + pl * 8);
                }
                wb[pl].u = w[DSMIL_WEXPT_ON(a, 4) ? 0 : (p % (UD + 1))][pl];
            }
            // the requests stay where they are written: left alone, the scheduler of the 512-thread form sinks every
            // weight load to just before its use (an s_waitcnt vmcnt(0) in front of each MFMA group: 413 us against 257)
            __builtin_amdgcn_sched_barrier(0);
            if (DSMIL_WEXPT_ON(a, 16)) {   // ablation: no MFMAs (operands kept live)
                asm volatile("" ::"v"(va[0].u), "v"(wb[0].u));
                continue;
            }
            // smallest products first (PlaneProducts<NP>)
            {
                using PP = PlaneProducts<NP>;
#pragma unroll
                for (int k = 0; k < PP::N; ++k) acc[p] = plane_mfma<NP>(va[PP::X[k]].u, wb[PP::W[k]].u, acc[p]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- raw(cc+1): registers -> LDS (the raw buffer was consumed before the last barrier), then the
        //      global loads of raw(cc+2)
        UNIT_STAMP(1);
        if (!DSMIL_WEXPT_ON(a, 2)) {
            if (more) raw_write(cc + 1);
            if (more2) raw_load(cc + 2);
        }
        UNIT_STAMP(2);
        __syncthreads();                 // V(cc) is free, raw(cc+1) is in LDS
        UNIT_STAMP(3);
        if (more && !DSMIL_WEXPT_ON(a, 1)) transform();
        if (more2) stat_write(cc + 2);   // read by raw_write(cc+2) in the next iteration, behind the barrier below
        UNIT_STAMP(4);
        __syncthreads();                 // V(cc+1) is ready
        UNIT_STAMP(5);
    }
    if (DSMIL_WEXPT_ON(a, 8)) {  // ablation: no epilogue
        float keep = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[p][r];
        if (keep == 123.456f) a.y[0] = keep;
        return;
    }
    wino_epilogue(a, acc, smem, lane, wn, wp, n0, img0, ty0, tx0, tpi, pb, emb_f16<NP>() ? EMB_OSCALE : 1.f);
}

#include "wino_w1.h"        // k_conv_wino_w1: the same unit for one wave per SIMD

#ifdef DSMIL_EXPERIMENTS   // measured-and-lost restructurings of the Winograd unit: experiment builds only
#include "experiments/wino_variants.h"
#endif  // DSMIL_EXPERIMENTS

// conv weight [O][I][3][3] -> U = G g G^T cut into three bf16 planes: [16 pos][I/16][3][O][16], or (tiled, for
// k_conv_wino_w1) [O/32][I/16][16 pos][3][32][16]
__global__ void k_pack_wino_s3(const float* __restrict__ w, unsigned short* __restrict__ out, int O, int I, int tiled, int np) {
    const long long total = (long long)O * I;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % I), o = (int)(i / I);
        const float* gw = w + i * 9;
        float gg[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) gg[r][c] = gw[r * 3 + c];
        float tmp[4][3];  // G g
        for (int c = 0; c < 3; ++c) {
            tmp[0][c] = gg[0][c];
            tmp[1][c] = 0.5f * (gg[0][c] + gg[1][c] + gg[2][c]);
            tmp[2][c] = 0.5f * (gg[0][c] - gg[1][c] + gg[2][c]);
            tmp[3][c] = gg[2][c];
        }
        for (int xi = 0; xi < 4; ++xi) {
            float u4[4];
            u4[0] = tmp[xi][0];
            u4[1] = 0.5f * (tmp[xi][0] + tmp[xi][1] + tmp[xi][2]);
            u4[2] = 0.5f * (tmp[xi][0] - tmp[xi][1] + tmp[xi][2]);
            u4[3] = tmp[xi][2];
            for (int nu = 0; nu < 4; ++nu) {
                const float v = u4[nu];
                const long long base = tiled
                    ? (((((long long)(o >> 5) * (I / 16) + ci / 16) * 16 + (xi * 4 + nu)) * 3) * 32 + (o & 31)) * 16 + (ci & 15)
                    : ((((long long)(xi * 4 + nu) * (I / 16) + ci / 16) * 3) * O + o) * 16 + (ci & 15);
                const long long pstride = tiled ? 32 * 16 : (long long)O * 16;
                if (np == 3 || np == 1) {   // two (one) fp16 planes of 2^EMB_WSHIFT U (see PlaneProducts)
                    out[base] = pack_h0(v * EMB_WSCALE);
                    out[base + pstride] = np == 3 ? pack_h1(v * EMB_WSCALE) : (unsigned short)0;
                    out[base + 2 * pstride] = 0;
                    continue;
                }
                const unsigned hb = __float_as_uint(v) & 0xFFFF0000u;
                const float r1 = v - __uint_as_float(hb);
                const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
                const unsigned lb = __float_as_uint(r1 - __uint_as_float(mb));
                out[base] = (unsigned short)(hb >> 16);
                out[base + pstride] = (unsigned short)(mb >> 16);
                out[base + 2 * pstride] = (unsigned short)(lb >> 16);
            }
        }
    }
}

// (cnt, mean, M2) partials [B][nparts][C][3] -> mean / rstd per (image, channel)
__global__ __launch_bounds__(256) void k_in_finalize_cnt(const float* __restrict__ part, float* __restrict__ mean,
                                                        float* __restrict__ rstd, int nparts, int C) {
    const int n = blockIdx.x;
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    __shared__ float red[2][4][64];
    for (int c = (int)blockIdx.y * 64 + cl; c < C && c < ((int)blockIdx.y + 1) * 64; c += 64) {   // grid.y = C / 64 (see k_in_finalize_flat)
        float s = 0.f, cnt = 0.f;
        for (int t0 = g; t0 < nparts; t0 += 4 * FIN_U) {   // rounds of FIN_U loads (see k_in_finalize_stem)
            float o0[FIN_U], o1[FIN_U];
#pragma unroll
            for (int u = 0; u < FIN_U; ++u) {
                const int t = t0 + 4 * u < nparts ? t0 + 4 * u : nparts - 1;
                const float* o = part + (((long long)n * nparts + t) * C + c) * 3;
                o0[u] = o[0]; o1[u] = o[1];
            }
#pragma unroll
            for (int u = 0; u < FIN_U; ++u)
                if (t0 + 4 * u < nparts) { s += o0[u] * o1[u]; cnt += o0[u]; }
        }
        __syncthreads();
        red[0][g][cl] = s;
        red[1][g][cl] = cnt;
        __syncthreads();
        const float tot = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
        const float mu = ((red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl])) / tot;
        float q = 0.f;
        for (int t0 = g; t0 < nparts; t0 += 4 * FIN_U) {
            float o0[FIN_U], o1[FIN_U], o2[FIN_U];
#pragma unroll
            for (int u = 0; u < FIN_U; ++u) {
                const int t = t0 + 4 * u < nparts ? t0 + 4 * u : nparts - 1;
                const float* o = part + (((long long)n * nparts + t) * C + c) * 3;
                o0[u] = o[0]; o1[u] = o[1]; o2[u] = o[2];
            }
#pragma unroll
            for (int u = 0; u < FIN_U; ++u)
                if (t0 + 4 * u < nparts) {
                    const float dlt = o1[u] - mu;
                    q += o2[u] + o0[u] * dlt * dlt;
                }
        }
        __syncthreads();
        red[0][g][cl] = q;
        __syncthreads();
        if (g == 0) {
            const float m2 = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
            mean[(long long)n * C + c] = mu;
            rstd[(long long)n * C + c] = 1.0f / sqrtf(m2 / tot + IN_EPS);
        }
    }
}

// OIHW 3x3 weights -> Winograd domain U = G g G^T, laid out [16][I/8][O][8]
__global__ void k_pack_wino(const float* __restrict__ w, float* __restrict__ out, int O, int I) {
    const long long total = (long long)O * I;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % I), o = (int)(i / I);
        const float* gw = w + i * 9;
        float gg[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) gg[r][c] = gw[r * 3 + c];
        float tmp[4][3];  // G g
        for (int c = 0; c < 3; ++c) {
            tmp[0][c] = gg[0][c];
            tmp[1][c] = 0.5f * (gg[0][c] + gg[1][c] + gg[2][c]);
            tmp[2][c] = 0.5f * (gg[0][c] - gg[1][c] + gg[2][c]);
            tmp[3][c] = gg[2][c];
        }
        for (int xi = 0; xi < 4; ++xi) {
            float u4[4];
            u4[0] = tmp[xi][0];
            u4[1] = 0.5f * (tmp[xi][0] + tmp[xi][1] + tmp[xi][2]);
            u4[2] = 0.5f * (tmp[xi][0] - tmp[xi][1] + tmp[xi][2]);
            u4[3] = tmp[xi][2];
            for (int nu = 0; nu < 4; ++nu)
                out[(((long long)(xi * 4 + nu) * (I / 8) + ci / 8) * O + o) * 8 + (ci & 7)] = u4[nu];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_stem: conv 7x7 stride 2 pad 3, Cin = 3 -> 64, input NCHW fp32 (what VF.to_tensor yields,
// compute_feats.py:35-39), weight [64][3][7][7] read as [64][147] (k = c*49 + kh*7 + kw).
// Workgroup = 8 x 16 output pixels (4 waves x 32 px) x 64 channels.
// ---------------------------------------------------------------------------------------------
constexpr int ST_ROWS = 21, ST_LW = 40, ST_PLANE = ST_ROWS * ST_LW;  // input window per channel
constexpr int ST_K = 147, ST_KP = 152, ST_LDW = 156;

__host__ __device__ constexpr int stem_off(int k) {
    return k >= ST_K ? 0 : (k / 49) * ST_PLANE + ((k % 49) / 7) * ST_LW + (k % 7);
}

template <int KG>
__device__ __forceinline__ void stem_kgroup(const float* __restrict__ sIn, const float* __restrict__ sW,
                                            int pixbase, int frag, int hi, f32x16 (&acc)[2]) {
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(sW + frag + KG * 8);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(sW + 32 * ST_LDW + frag + KG * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int off = hi ? stem_off(KG * 8 + 4 + j) : stem_off(KG * 8 + j);
        const float av = sIn[pixbase + off];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w0[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w1[j], acc[1], 0, 0, 0);
    }
}

template <int KG>
struct StemLoop {
    static __device__ __forceinline__ void run(const float* sIn, const float* sW, int pixbase, int frag, int hi,
                                               f32x16 (&acc)[2]) {
        StemLoop<KG - 1>::run(sIn, sW, pixbase, frag, hi, acc);
        stem_kgroup<KG>(sIn, sW, pixbase, frag, hi, acc);
    }
};
template <>
struct StemLoop<-1> {
    static __device__ __forceinline__ void run(const float*, const float*, int, int, int, f32x16 (&)[2]) {}
};

// One workgroup walks a whole ROW of 8x16-pixel tiles of one image: the 64x147 weights are staged
// in LDS once per row (not once per tile), and the next tile's input window is prefetched into
// registers under the current tile's MFMAs and written to the other LDS window buffer.
constexpr int ST_WPT = (3 * ST_PLANE + 255) / 256;  // window floats per thread

// U8 = true: the input is uint8 NHWC [B,H,W,3] (a decoded image as PIL/numpy hold it) and the
// to_tensor scaling x/255 (compute_feats.py:35-39 -> VF.to_tensor: IEEE division) is applied here.
template <bool U8>
__global__ __launch_bounds__(256) void k_stem(const void* __restrict__ xin, const float* __restrict__ w,
                                              float* __restrict__ y, float* __restrict__ part, int B,
                                              int H, int W, int Ho, int Wo, int tiles_x, int tiles_y) {
    const float* x = reinterpret_cast<const float*>(xin);
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(xin);
    __shared__ __attribute__((aligned(16))) float sIn[2][3 * ST_PLANE];
    __shared__ __attribute__((aligned(16))) float sW[64 * ST_LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int ty = blockIdx.x, n = blockIdx.y;
    const int oy0 = ty * 8;
    const int iy00 = oy0 * 2 - 3;
    // per-thread window elements: row base offset in x (or -1) and column inside the window
    int wrow[ST_WPT], wcol[ST_WPT];
#pragma unroll
    for (int q = 0; q < ST_WPT; ++q) {
        const int e = tid + 256 * q;
        wrow[q] = -1; wcol[q] = 0;
        if (e < 3 * ST_PLANE) {
            const int c = e / ST_PLANE, r = (e - c * ST_PLANE) / ST_LW, col = e - c * ST_PLANE - r * ST_LW;
            const int iy = iy00 + r;
            wcol[q] = col;
            if (col < 37 && iy >= 0 && iy < H) wrow[q] = U8 ? ((n * H + iy) * W) * 3 + c : ((n * 3 + c) * H + iy) * W;
        }
    }
    float wreg[ST_WPT];
    auto win_load = [&](int tx) {
        const int ix00 = tx * 32 - 3;
#pragma unroll
        for (int q = 0; q < ST_WPT; ++q) {
            const int ix = ix00 + wcol[q];
            const bool ok = wrow[q] >= 0 && ix >= 0 && ix < W;
            float v;
            if constexpr (U8) v = __fdiv_rn((float)xb[ok ? (long long)wrow[q] + 3 * ix : 0], 255.f);
            else v = x[ok ? (long long)wrow[q] + ix : 0];
            wreg[q] = ok ? v : 0.f;
        }
    };
    auto win_write = [&](int buf) {
#pragma unroll
        for (int q = 0; q < ST_WPT; ++q) {
            const int e = tid + 256 * q;
            if (e < 3 * ST_PLANE) sIn[buf][e] = wreg[q];
        }
    };
    win_load(0);
    for (int e = tid; e < 64 * ST_KP; e += 256) {
        const int co = e / ST_KP, k = e - co * ST_KP;
        sW[co * ST_LDW + k] = (k < ST_K) ? w[co * ST_K + k] : 0.f;
    }
    win_write(0);
    __syncthreads();
    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;
    const int pixbase = (2 * py) * ST_LW + 2 * px;
    const int frag = l31 * ST_LDW + 4 * hi;
    for (int tx = 0; tx < tiles_x; ++tx) {
        const int ox0 = tx * 16;
        if (tx + 1 < tiles_x) win_load(tx + 1);
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        StemLoop<ST_KP / 8 - 1>::run(sIn[tx & 1], sW, pixbase, frag, hi, acc);
        if (tx + 1 < tiles_x) win_write((tx + 1) & 1);
        // store raw NHWC + statistics partial (cnt, mean, M2) per (image, tile, wave, channel)
        int cnt = 0;
        float s0 = 0.f, s1 = 0.f;
        bool okr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = drow(r, hi);
            const int oy = oy0 + 2 * wave + (q >> 4), ox = ox0 + (q & 15);
            okr[r] = (oy < Ho) && (ox < Wo);
            if (okr[r]) {
                float* o = y + (((long long)n * Ho + oy) * Wo + ox) * 64;
                o[l31] = acc[0][r];
                o[32 + l31] = acc[1][r];
                s0 += acc[0][r];
                s1 += acc[1][r];
                ++cnt;
            }
        }
        cnt += __shfl_xor(cnt, 32, 64);
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        const float fc = (float)cnt;
        const float m0 = cnt ? s0 / fc : 0.f, m1 = cnt ? s1 / fc : 0.f;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (okr[r]) {
                const float d0 = acc[0][r] - m0, d1 = acc[1][r] - m1;
                q0 += d0 * d0;
                q1 += d1 * d1;
            }
        q0 += __shfl_xor(q0, 32, 64);
        q1 += __shfl_xor(q1, 32, 64);
        if (hi == 0) {
            float* o = part + ((((long long)n * tiles_x * tiles_y + ty * tiles_x + tx) * 4 + wave) * 64) * 3;
            o[l31 * 3 + 0] = fc; o[l31 * 3 + 1] = m0; o[l31 * 3 + 2] = q0;
            o[(32 + l31) * 3 + 0] = fc; o[(32 + l31) * 3 + 1] = m1; o[(32 + l31) * 3 + 2] = q1;
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------
// k_stem_s6: the same 7x7 stride-2 conv on bf16 MFMA over exact three-plane cuts (6 plane products, see agg_split.h).
// K is laid out (c, kh, kw padded to 8): 21 rows of 8 = 168, padded to 176 = 11 steps of 16, so the 8 k of a lane are
// 8 CONSECUTIVE columns of one window row — the A fragment is read straight from the input window in LDS (4 dwords per
// plane), no im2col.  Every input element is cut ONCE when its window is staged (the 49/4 windows that overlap it
// share the planes); the weights are cut at pack time (k_pack_stem_s6: [3 planes][64][184] bf16, 368-B rows =
// conflict-free ds_read_b128).  Workgroup = 16 x 16 output pixels x 64 channels, 8 waves (two rows each), walking a
// row of tiles with the next window prefetched into registers; the two cout halves alternate inside each plane
// product, so no MFMA accumulates into the tile the previous one is still writing.
//   LDS: window 2 buffers x 3 planes x 3 x 37 x 40 bf16 = 52 KB, weights 3 x 64 x 184 bf16 = 69 KB.  (48-column window
//   rows would put the two pixel rows of a wave's A read 16 banks apart — conflict-free — but cost 20 % more staging:
//   470 us against 433; the 31 % of LDS cycles spent in conflicts are not what bounds this kernel.)
// ---------------------------------------------------------------------------------------------
constexpr int SS_TR = 16;                                   // output rows per tile
constexpr int SS_ROWS = 2 * SS_TR + 5, SS_LW = 40, SS_PLANE = SS_ROWS * SS_LW, SS_WIN = 3 * SS_PLANE;
constexpr int SS_KP = 176, SS_LDW = 184;                    // padded K, weight row stride (bf16)
constexpr int SS_WIMG = 3 * 64 * SS_LDW;                    // bf16 elements of the packed stem image
constexpr int SS_PPT = (SS_WIN / 2 + 511) / 512;            // window element PAIRS per thread
constexpr size_t SS_LDS = (size_t)(2 * 3 * SS_WIN + SS_WIMG) * 2;

// conv1 weight [64][3][7][7] -> three truncated bf16 planes [3][64][184], k' = (c*7 + kh)*8 + kw
__global__ void k_pack_stem_s6(const float* __restrict__ w, unsigned short* __restrict__ out, int np) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 64 * SS_LDW; i += gridDim.x * blockDim.x) {
        const int co = i / SS_LDW, k = i - co * SS_LDW;
        const int row = k >> 3, kw = k & 7;
        float v = 0.f;
        if (row < 21 && kw < 7) v = w[co * 147 + (row / 7) * 49 + (row % 7) * 7 + kw];
        if (np == 3 || np == 1) {   // two (one) fp16 planes of 2^EMB_WSHIFT w (see PlaneProducts)
            out[i] = pack_h0(v * EMB_WSCALE);
            out[64 * SS_LDW + i] = np == 3 ? pack_h1(v * EMB_WSCALE) : (unsigned short)0;
            out[2 * 64 * SS_LDW + i] = 0;
            continue;
        }
        const unsigned hb = __float_as_uint(v) & 0xFFFF0000u;
        const float r1 = v - __uint_as_float(hb);
        const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
        const unsigned lb = __float_as_uint(r1 - __uint_as_float(mb));
        out[i] = (unsigned short)(hb >> 16);
        out[64 * SS_LDW + i] = (unsigned short)(mb >> 16);
        out[2 * 64 * SS_LDW + i] = (unsigned short)(lb >> 16);
    }
}

__host__ __device__ constexpr int stem6_off(int row) {      // window element offset of K row (c, kh); row 21 is padding
    return row >= 21 ? 0 : (row / 7) * SS_PLANE + (row % 7) * SS_LW;
}

// POOL: the 3x3 stride-2 max-pool is taken HERE, on the raw conv values while they are in registers (max commutes with the
// increasing map IN + ReLU, as in k_norm_relu_maxpool): the kernel writes the pooled raw map [B,Hp,Wp,64] (4x smaller than
// the 822 MB raw stem output it replaces) plus the last conv row of every tile row (`halo`, [B][tiles_y][Wo][64]) — the one
// row a tile's first pool row needs from the tile above; k_pool_fix_norm folds it in and normalises.  A wave holds two conv
// rows of the tile = pool row `wave`'s centre and lower row; its upper row is the second row of the wave above (through
// LDS, 32 KB); columns: a lane holds 8 of the 16 (two groups of 4), the 3-wide stride-2 windows need one value from the
// other half-wave (two shuffles) and, at the left edge, the last column of the previous tile of the walk (a carried register).
constexpr size_t SS_POOL_LDS = (size_t)8 * 16 * 64 * 4;
template <bool U8, bool POOL = false, int NP = NPD>
__global__ __launch_bounds__(512, 2) void k_stem_s6(const void* __restrict__ xin, const unsigned short* __restrict__ wimg,
                                                    float* __restrict__ y, float* __restrict__ part, int B, int H, int W,
                                                    int Ho, int Wo, int tiles_x, int tiles_y,
                                                    float* __restrict__ halo = nullptr, int Hp = 0, int Wp = 0) {
    const float* x = reinterpret_cast<const float*>(xin);
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(xin);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* sIn = reinterpret_cast<unsigned short*>(smem);         // [2][3 planes][SS_WIN]
    unsigned short* sW = sIn + 2 * 3 * SS_WIN;                              // [3][64][SS_LDW]
    float* sE = reinterpret_cast<float*>(sW + SS_WIMG);                     // POOL: [8 waves][16 cols][64 ch] second conv rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int ty = blockIdx.x, n = blockIdx.y;
    float carry[2] = {-INFINITY, -INFINITY};                                // POOL: column-max of the previous tile's last column
    const int oy0 = ty * SS_TR;
    const int iy00 = oy0 * 2 - 3;
    // per-thread window pairs (two adjacent columns of one row): row base offset in x (or -1) and the first column
    int wrow[SS_PPT], wcol[SS_PPT];
#pragma unroll
    for (int q = 0; q < SS_PPT; ++q) {
        const int e = 2 * (tid + 512 * q);
        wrow[q] = -1; wcol[q] = 0;
        if (e < SS_WIN) {
            const int c = e / SS_PLANE, r = (e - c * SS_PLANE) / SS_LW, col = e - c * SS_PLANE - r * SS_LW;
            const int iy = iy00 + r;
            wcol[q] = col;
            if (iy >= 0 && iy < H) wrow[q] = U8 ? ((n * H + iy) * W) * 3 + c : ((n * 3 + c) * H + iy) * W;
        }
    }
    float wreg[SS_PPT][2];
    auto win_load = [&](int tx) {
        const int ix00 = tx * 32 - 3;
#pragma unroll
        for (int q = 0; q < SS_PPT; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ix = ix00 + wcol[q] + j;
                const bool ok = wrow[q] >= 0 && ix >= 0 && ix < W && wcol[q] + j < 39;
                float v;
                if constexpr (U8) v = __fdiv_rn((float)xb[ok ? (long long)wrow[q] + 3 * ix : 0], 255.f);
                else v = x[ok ? (long long)wrow[q] + ix : 0];
                wreg[q][j] = ok ? v : 0.f;
            }
    };
    auto win_write = [&](int buf) {                         // cut once, three 4-byte plane writes per pair
        unsigned* dst = reinterpret_cast<unsigned*>(sIn + buf * 3 * SS_WIN);
#pragma unroll
        for (int q = 0; q < SS_PPT; ++q) {
            const int e2 = tid + 512 * q;                   // pair index
            if (2 * e2 < SS_WIN) {
                if constexpr (NP == 1) {
                    dst[e2] = cut2h1(wreg[q][0], wreg[q][1]);
                    continue;
                }
                if constexpr (NP == 3) {
                    unsigned h, l;
                    cut2h(wreg[q][0], wreg[q][1], h, l);
                    dst[e2] = h;
                    dst[SS_WIN / 2 + e2] = l;
                    continue;
                }
                unsigned xu[2], r1u[2], r2u[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    xu[j] = __float_as_uint(wreg[q][j]);
                    const float r1 = wreg[q][j] - __uint_as_float(xu[j] & 0xFFFF0000u);
                    r1u[j] = __float_as_uint(r1);
                    r2u[j] = __float_as_uint(r1 - __uint_as_float(r1u[j] & 0xFFFF0000u));
                }
                dst[e2] = __builtin_amdgcn_perm(xu[1], xu[0], 0x07060302u);
                dst[SS_WIN / 2 + e2] = __builtin_amdgcn_perm(r1u[1], r1u[0], 0x07060302u);
                dst[SS_WIN + e2] = __builtin_amdgcn_perm(r2u[1], r2u[0], 0x07060302u);
            }
        }
    };
    win_load(0);
    {   // the packed weight image -> LDS (16-B pieces)
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(wimg);
        u32x4_t* dstw = reinterpret_cast<u32x4_t*>(sW);
        for (int e = tid; e < SS_WIMG / 8; e += 512) dstw[e] = src[e];
    }
    win_write(0);
    __syncthreads();
    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;
    const int pixbase = (2 * py) * SS_LW + 2 * px;          // window element of this lane's pixel (even: 4-byte aligned)
    const int wfrag = l31 * SS_LDW + 8 * hi;                // bf16 elements inside a plane's [64][184]
    for (int tx = 0; tx < tiles_x; ++tx) {
        const int ox0 = tx * 16;
        if (tx + 1 < tiles_x) win_load(tx + 1);
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        const unsigned short* win = sIn + (tx & 1) * 3 * SS_WIN + pixbase;
#pragma unroll
        for (int st = 0; st < SS_KP / 16; ++st) {
            const int off = hi ? stem6_off(2 * st + 1) : stem6_off(2 * st);
            constexpr int PLN = emb_planes<NP>();
            Frag16 xa[PLN], wb[2][PLN];
#pragma unroll
            for (int pl = 0; pl < PLN; ++pl) {
                const unsigned* a4 = reinterpret_cast<const unsigned*>(win + pl * SS_WIN + off);
                xa[pl].u = u32x4_t{a4[0], a4[1], a4[2], a4[3]};
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    wb[t][pl].u = *reinterpret_cast<const u32x4_t*>(sW + (pl * 64 + t * 32) * SS_LDW + wfrag + st * 16);
            }
            using PP = PlaneProducts<NP>;   // smallest products first
#pragma unroll
            for (int k = 0; k < PP::N; ++k)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[t] = plane_mfma<NP>(xa[PP::X[k]].u, wb[t][PP::W[k]].u, acc[t]);
        }
        if constexpr (emb_f16<NP>()) {                      // the weights carry 2^EMB_WSHIFT
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][r] *= EMB_OSCALE; acc[1][r] *= EMB_OSCALE; }
        }
        if (tx + 1 < tiles_x) win_write((tx + 1) & 1);
        // store raw NHWC + statistics partial (cnt, mean, M2) per (image, tile, wave, channel)
        int cnt = 0;
        float s0 = 0.f, s1 = 0.f;
        bool okr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = drow(r, hi);
            const int oy = oy0 + 2 * wave + (q >> 4), ox = ox0 + (q & 15);
            okr[r] = (oy < Ho) && (ox < Wo);
            if (okr[r]) {
                if constexpr (!POOL) {
                    float* o = y + (((long long)n * Ho + oy) * Wo + ox) * 64;
                    o[l31] = acc[0][r];
                    o[32 + l31] = acc[1][r];
                }
                s0 += acc[0][r];
                s1 += acc[1][r];
                ++cnt;
            }
        }
        cnt += __shfl_xor(cnt, 32, 64);
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        const float fc = (float)cnt;
        const float m0 = cnt ? s0 / fc : 0.f, m1 = cnt ? s1 / fc : 0.f;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (okr[r]) {
                const float d0 = acc[0][r] - m0, d1 = acc[1][r] - m1;
                q0 += d0 * d0;
                q1 += d1 * d1;
            }
        q0 += __shfl_xor(q0, 32, 64);
        q1 += __shfl_xor(q1, 32, 64);
        if (hi == 0) {
            float* o = part + ((((long long)n * tiles_x * tiles_y + ty * tiles_x + tx) * 8 + wave) * 64) * 3;
            o[l31 * 3 + 0] = fc; o[l31 * 3 + 1] = m0; o[l31 * 3 + 2] = q0;
            o[(32 + l31) * 3 + 0] = fc; o[(32 + l31) * 3 + 1] = m1; o[(32 + l31) * 3 + 2] = q1;
        }
        if constexpr (POOL) {
            // register k (0..7) of a lane = column 4 hi + k (k < 4) or 8 + 4 hi + (k - 4); registers k and k + 8 are the two rows
            float v[2][8], low[2][8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float r0 = okr[k] ? acc[t][k] : -INFINITY;
                    low[t][k] = okr[k + 8] ? acc[t][k + 8] : -INFINITY;
                    v[t][k] = fmaxf(r0, low[t][k]);
                    const int col = (k < 4 ? 4 * hi + k : 8 + 4 * hi + (k - 4));
                    sE[(wave * 16 + col) * 64 + t * 32 + l31] = low[t][k];
                }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (wave > 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int col = (k < 4 ? 4 * hi + k : 8 + 4 * hi + (k - 4));
                        v[t][k] = fmaxf(v[t][k], sE[((wave - 1) * 16 + col) * 64 + t * 32 + l31]);
                    }
                }
                // pool columns: hi = 0 lanes form 0, 1, 4, 5; hi = 1 lanes 2, 3, 6, 7 (windows 2i-1 .. 2i+1)
                const float t3 = __shfl_xor(v[t][3], 32, 64), t7 = __shfl_xor(v[t][7], 32, 64);
                const float X = hi ? t3 : carry[t], Y = hi ? t7 : t3;
                float o4[4];
                o4[0] = fmaxf(fmaxf(X, v[t][0]), v[t][1]);
                o4[1] = fmaxf(fmaxf(v[t][1], v[t][2]), v[t][3]);
                o4[2] = fmaxf(fmaxf(Y, v[t][4]), v[t][5]);
                o4[3] = fmaxf(fmaxf(v[t][5], v[t][6]), v[t][7]);
                carry[t] = t7;                                            // (hi = 0 lanes: column 15 of this tile)
                const int py = (oy0 >> 1) + wave;
                if (py < Hp) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int px = (ox0 >> 1) + 2 * hi + (j & 1) + 4 * (j >> 1);
                        if (px < Wp) y[(((long long)n * Hp + py) * Wp + px) * 64 + t * 32 + l31] = o4[j];
                    }
                }
                if (wave == 7 && ty + 1 < tiles_y) {                      // the row the tile below needs above its first pool row
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int ox = ox0 + (k < 4 ? 4 * hi + k : 8 + 4 * hi + (k - 4));
                        if (ox < Wo) halo[(((long long)n * tiles_y + ty) * Wo + ox) * 64 + t * 32 + l31] = low[t][k];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// POOL path, second half: fold the row from the tile above into the first pool row of every tile row, then IN + ReLU
// (in place on the pooled raw map; the same arithmetic as k_norm_relu_maxpool for r > 0, so the result is bit-identical)
// b16 != nullptr (the opt-in bf16-activation trunk, resnet_b16.h): the normalised map goes out as bf16 into that trunk's
// shared-border layout ([rows][Wp + 1][64], row 0 of an image and column Wp are zero: written by k_b16_borders) instead of
// fp32 in place — the trunk's first conv reads it as it is
__global__ __launch_bounds__(256) void k_pool_fix_norm(float* __restrict__ p, const float* __restrict__ halo,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       int B, int Hp, int Wp, int Wo, int tiles_y, unsigned short* __restrict__ b16 = nullptr,
                                                       int b16_is_f16 = 0) {
    const int c4 = threadIdx.x & 15;
    const unsigned np = (unsigned)B * Hp * Wp;
    for (unsigned pp = blockIdx.x * 16 + threadIdx.x / 16; pp < np; pp += gridDim.x * 16) {
        unsigned q = pp;
        const int px = (int)(q % (unsigned)Wp); q /= (unsigned)Wp;
        const int py = (int)(q % (unsigned)Hp);
        const int n = (int)(q / (unsigned)Hp);
        float* o = p + (size_t)pp * 64 + c4 * 4;
        f32x4 m = *reinterpret_cast<const f32x4*>(o);
        if ((py & 7) == 0 && py > 0) {
            const float* hrow = halo + ((size_t)n * tiles_y + (py >> 3) - 1) * Wo * 64 + c4 * 4;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int ox = 2 * px + dx;
                if (ox < 0 || ox >= Wo) continue;
                const f32x4 h = *reinterpret_cast<const f32x4*>(hrow + (size_t)ox * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], h[e]);
            }
        }
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (long long)n * 64 + c4 * 4);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + (long long)n * 64 + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf((m[e] - mu[e]) * rs[e], 0.f);
        if (b16) {
            auto bf = [&](float f) -> unsigned {   // RNE to bf16, or to fp16 (the fp16-activation trunk)
                if (b16_is_f16) { const _Float16 h = (_Float16)f; return (unsigned)__builtin_bit_cast(unsigned short, h); }
                const unsigned u = __float_as_uint(f);
                return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
            };
            const size_t q = ((size_t)n * (Hp + 1) + py + 1) * (Wp + 1) + px;
            unsigned* d = reinterpret_cast<unsigned*>(b16 + q * 64 + c4 * 4);
            d[0] = bf(m[0]) | (bf(m[1]) << 16);
            d[1] = bf(m[2]) | (bf(m[3]) << 16);
        } else {
            *reinterpret_cast<f32x4*>(o) = m;
        }
    }
}

// stem partials (cnt, mean, M2) -> mean / rstd; one workgroup per image, 64 channels x 16 tile groups (392 partials
// per channel at 224x224: a chain of dependent loads per thread, so width is what makes it short)
constexpr int FS_G = 16;
__global__ __launch_bounds__(64 * FS_G) void k_in_finalize_stem(const float* __restrict__ part, float* __restrict__ mean,
                                                              float* __restrict__ rstd, int B, int nparts) {
    const int n = blockIdx.x, c = threadIdx.x & 63, g = threadIdx.x >> 6;
    __shared__ float red[2][FS_G][64];
    // The partial walks run in rounds of FIN_U loads, the last round padded with clamped re-reads that are not added: as
    // plain loops hipcc issued ONE load, waited for it (vmcnt(0)) and added.  Same order of additions.  (Measured: 25.7 ->
    // 22.9 us here — the walk is 2 x 77 MB of partials per 256 images, i.e. bandwidth — and 5.8 -> 4.4 us for
    // k_in_finalize_cnt; the same form of k_in_finalize_flat, whose tiles need a division each, was slower and is not kept.)
    float s = 0.f, cnt = 0.f;
    for (int t0 = g; t0 < nparts; t0 += FS_G * FIN_U) {
        float o0[FIN_U], o1[FIN_U];
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
            const int t = t0 + FS_G * u < nparts ? t0 + FS_G * u : nparts - 1;
            const float* o = part + (((long long)n * nparts + t) * 64 + c) * 3;
            o0[u] = o[0]; o1[u] = o[1];
        }
#pragma unroll
        for (int u = 0; u < FIN_U; ++u)
            if (t0 + FS_G * u < nparts) { s += o0[u] * o1[u]; cnt += o0[u]; }
    }
    red[0][g][c] = s;
    red[1][g][c] = cnt;
    __syncthreads();
    float tot = 0.f, sm = 0.f;
#pragma unroll
    for (int k = 0; k < FS_G; ++k) { tot += red[1][k][c]; sm += red[0][k][c]; }
    const float mu = sm / tot;
    float q = 0.f;
    for (int t0 = g; t0 < nparts; t0 += FS_G * FIN_U) {
        float o0[FIN_U], o1[FIN_U], o2[FIN_U];
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
            const int t = t0 + FS_G * u < nparts ? t0 + FS_G * u : nparts - 1;
            const float* o = part + (((long long)n * nparts + t) * 64 + c) * 3;
            o0[u] = o[0]; o1[u] = o[1]; o2[u] = o[2];
        }
#pragma unroll
        for (int u = 0; u < FIN_U; ++u)
            if (t0 + FS_G * u < nparts) {
                const float d = o1[u] - mu;
                q += o2[u] + o0[u] * d * d;
            }
    }
    __syncthreads();
    red[0][g][c] = q;
    __syncthreads();
    if (g == 0) {
        float m2 = 0.f;
#pragma unroll
        for (int k = 0; k < FS_G; ++k) m2 += red[0][k][c];
        mean[n * 64 + c] = mu;
        rstd[n * 64 + c] = 1.0f / sqrtf(m2 / tot + IN_EPS);
    }
}

// IN + ReLU + MaxPool2d(3, stride 2, pad 1) of the raw stem output, NHWC, C = 64
__global__ __launch_bounds__(256) void k_norm_relu_maxpool(const float* __restrict__ y,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           float* __restrict__ out, int B, int Hi, int Wi,
                                                           int Ho, int Wo, int C) {
    const int c4n = C / 4;                                  // 16: a thread keeps its channel group (256 % 16 == 0)
    const int c4 = threadIdx.x % c4n;
    const unsigned tpb = 256 / c4n, np = (unsigned)B * Ho * Wo;
    for (unsigned pp = blockIdx.x * tpb + threadIdx.x / c4n; pp < np; pp += gridDim.x * tpb) {
        unsigned p = pp;
        const int ox = (int)(p % (unsigned)Wo); p /= (unsigned)Wo;
        const int oy = (int)(p % (unsigned)Ho);
        const int n = (int)(p / (unsigned)Ho);
        // max of the normalised window = normalised max (r >= 0: InstanceNorm always) or normalised
        // min (r < 0: a frozen BatchNorm with negative weight), so both extremes of the raw window are kept
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        f32x4 mn = {INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if (iy < 0 || iy >= Hi) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if (ix < 0 || ix >= Wi) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(y + (((long long)n * Hi + iy) * Wi + ix) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { m[e] = fmaxf(m[e], v[e]); mn[e] = fminf(mn[e], v[e]); }
            }
        }
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (long long)n * C + c4 * 4);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + (long long)n * C + c4 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(((rs[e] >= 0.f ? m[e] : mn[e]) - mu[e]) * rs[e], 0.f);
        *reinterpret_cast<f32x4*>(out + (((long long)n * Ho + oy) * Wo + ox) * C + c4 * 4) = o;
    }
}

// out = relu(IN(y2) + (DOWN ? IN(yd) : idn))   NHWC.  HBM-bound streaming: a thread keeps ONE 4-channel group (256 is a
// multiple of C/4 for every ResNet width, so the grid stride preserves it) and walks pixels with 32-bit arithmetic —
// the first version spent three 64-bit divisions per 48 bytes of traffic (4.9 TB/s).
template <bool DOWN>
__global__ __launch_bounds__(256) void k_norm_add_relu(const float* __restrict__ y2, const float* __restrict__ m2,
                                                       const float* __restrict__ r2, const float* __restrict__ idn,
                                                       const float* __restrict__ md, const float* __restrict__ rd,
                                                       float* __restrict__ out, long long npix, int HW, int C) {
    const int c4n = C / 4;
    const int lanes_c = c4n < 256 ? c4n : 256;              // threads across the channel groups of one pixel
    const int cpt = (c4n + 255) / 256;                      // channel groups per thread (2 only for the 2048-wide Bottleneck tail)
    const int tpb = 256 / lanes_c;                          // pixels per workgroup per iteration
    const int c40 = threadIdx.x % lanes_c, pl = threadIdx.x / lanes_c;
    const unsigned np = (unsigned)npix, pstep = gridDim.x * (unsigned)tpb;
    int ncur = -1;
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = mu, mud = mu, rsd = mu;
    for (unsigned p = blockIdx.x * (unsigned)tpb + pl; p < np; p += pstep) {
        const int n = (int)(p / (unsigned)HW);
        for (int j = 0; j < cpt; ++j) {
            const int c4 = c40 + 256 * j;
            const size_t o = (size_t)p * C + c4 * 4;
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y2 + o));
            f32x4 id = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(idn + o));
            if (n != ncur || cpt > 1) {                     // the image changes every HW pixels: reload its statistics
                mu = *reinterpret_cast<const f32x4*>(m2 + (size_t)n * C + c4 * 4);
                rs = *reinterpret_cast<const f32x4*>(r2 + (size_t)n * C + c4 * 4);
                if constexpr (DOWN) {
                    mud = *reinterpret_cast<const f32x4*>(md + (size_t)n * C + c4 * 4);
                    rsd = *reinterpret_cast<const f32x4*>(rd + (size_t)n * C + c4 * 4);
                }
            }
            if constexpr (DOWN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) id[e] = (id[e] - mud[e]) * rsd[e];
            }
            f32x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = fmaxf((v[e] - mu[e]) * rs[e] + id[e], 0.f);
            *reinterpret_cast<f32x4*>(out + o) = o4;
        }
        ncur = n;
    }
}

// last block: feats[n][c] = mean_p relu(IN(y2) + idn)   (adaptive_avg_pool2d(1) + flatten)
__global__ void k_norm_add_relu_pool(const float* __restrict__ y2, const float* __restrict__ m2,
                                     const float* __restrict__ r2, const float* __restrict__ idn,
                                     float* __restrict__ feats, int B, int HW, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int n = i / C, c = i - n * C;
    const float mu = m2[i], rs = r2[i];
    float s = 0.f;
    for (int p = 0; p < HW; ++p) {
        const long long o = ((long long)n * HW + p) * C + c;
        s += fmaxf((y2[o] - mu) * rs + idn[o], 0.f);
    }
    feats[i] = s / (float)HW;
}

// OIHW -> [tap][O][I]
__global__ void k_pack_conv(const float* __restrict__ w, float* __restrict__ out, int O, int I, int taps) {
    const long long total = (long long)O * I * taps;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % I);
        const long long r = i / I;
        const int o = (int)(r % O), t = (int)(r / O);
        out[i] = w[((long long)o * I + ci) * taps + t];
    }
}

// ---- host ----------------------------------------------------------------------------------
struct ConvSpec { int cout, cin, ks, stride, pad; };
// torchvision state_dict order of the bias-free convolutions of a BasicBlock ResNet:
// stem, then per block conv1, conv2 and — first block of layers 2..4 — downsample.0.
// depth 18 = blocks [2,2,2,2] (20 convs), depth 34 = [3,4,6,3] (36 convs).
// depth 50 = Bottleneck blocks [3,4,6,3] (53 convs), depth 101 = [3,4,23,3] (104 convs): per block conv1 1x1,
// conv2 3x3 (stride on conv2: torchvision's ResNet v1.5), conv3 1x1 to 4x the width, and — first block of every
// layer — downsample.0 1x1; feature width 2048 (compute_feats.py:161-167).
struct Arch { int depth, nblk[4], nconv, bottleneck, feat; ConvSpec specs[112]; };
Arch make_arch(int depth) {
    Arch a;
    a.depth = depth;
    const int n18[4] = {2, 2, 2, 2}, n34[4] = {3, 4, 6, 3}, n101[4] = {3, 4, 23, 3};
    a.bottleneck = depth >= 50;
    for (int l = 0; l < 4; ++l) a.nblk[l] = depth == 18 ? n18[l] : depth == 101 ? n101[l] : n34[l];
    int n = 0;
    a.specs[n++] = ConvSpec{64, 3, 7, 2, 3};
    int cin = 64;
    for (int l = 0; l < 4; ++l) {
        const int c = 64 << l;
        for (int b = 0; b < a.nblk[l]; ++b) {
            if (a.bottleneck) {
                const int stride = (l > 0 && b == 0) ? 2 : 1;
                a.specs[n++] = ConvSpec{c, cin, 1, 1, 0};
                a.specs[n++] = ConvSpec{c, c, 3, stride, 1};
                a.specs[n++] = ConvSpec{4 * c, c, 1, 1, 0};
                if (b == 0) a.specs[n++] = ConvSpec{4 * c, cin, 1, stride, 0};
                cin = 4 * c;
            } else {
                const bool down = l > 0 && b == 0;
                a.specs[n++] = ConvSpec{c, cin, 3, down ? 2 : 1, 1};
                a.specs[n++] = ConvSpec{c, c, 3, 1, 1};
                if (down) a.specs[n++] = ConvSpec{c, cin, 1, 2, 0};
                cin = c;
            }
        }
    }
    a.nconv = n;
    a.feat = cin;
    return a;
}
const Arch* arch_of(int depth) {
    static const Arch a18 = make_arch(18), a34 = make_arch(34), a50 = make_arch(50), a101 = make_arch(101);
    return depth == 18 ? &a18 : depth == 34 ? &a34 : depth == 50 ? &a50 : depth == 101 ? &a101 : nullptr;
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// hipFuncAttributeMaxDynamicSharedMemorySize once per (device, kernel), not once per launch (lds_attr.h: the attribute
// applies to the current device only, so a process that drives several GPUs must set it on each)
inline void allow_lds(const void* kern, size_t bytes) { (void)dsmil_lds::allow(kern, (int)bytes); }
}  // namespace
// Round 6: the opt-in bf16-ACTIVATION trunk behind the stem (dsmil_resnet_pack_ex / dsmil_resnet_forward_ex, precision = 2).
// Needs Arch / ConvSpec / allow_lds, lives in namespace b16.
#include "resnet_b16.h"
namespace {
thread_local int g_b16_trunk = 0;    // precision 2 / 3 on THIS host thread: one-plane stem, then b16::trunk on bf16 (1) / fp16 (2) activations
inline bool stem_fuse() {   // expt builds: DSMIL_STEM_FUSE=0 keeps the stem + k_norm_relu_maxpool pair (A/B, bit-identity test)
#ifdef DSMIL_EXPERIMENTS
    static const int off = [] { const char* e = getenv("DSMIL_STEM_FUSE"); return (e && !strcmp(e, "0")) ? 1 : 0; }();
    return !off;
#else
    return true;
#endif
}
inline bool use_wino(const ConvSpec& s) {  // 3x3 stride-1 convs run as Winograd F(2x2,3x3)
#ifdef DSMIL_EXPERIMENTS
    static const int off = expt_env("DSMIL_NO_WINO");
#else
    constexpr int off = 0;
#endif
    return !off && s.ks == 3 && s.stride == 1 && s.pad == 1 && s.cin % WK == 0 && s.cout % 64 == 0;
}
// Experiment builds: DSMIL_WINO = h3 (default) | s6 | s9 | f32: which MFMA form the Winograd convs use (read once per process;
// the packed weights and the kernels must agree): h3 = fp16 MFMA over two-plane cuts, three products (PlaneProducts, round 5);
// s6 / s9 = bf16 MFMA over exact three-plane cuts with the 6 largest / all 9 plane products; f32 = v_mfma_f32_32x32x2_f32
// dsmil_resnet_pack_ex / dsmil_resnet_forward_ex with precision = 1 (the opt-in reduced-precision path): the form of THIS call on
// THIS host thread; 0 = the process form below
thread_local int g_form_override = 0;
struct FormOverride {
    int saved;
    explicit FormOverride(int np) : saved(g_form_override) { g_form_override = np; }
    ~FormOverride() { g_form_override = saved; }
};
inline int wino_form() {
    if (g_form_override) return g_form_override;
#ifdef DSMIL_EXPERIMENTS
    static const int form = [] {
        const char* e = getenv("DSMIL_WINO");
        if (e && !strcmp(e, "f32")) return 0;
        if (e && !strcmp(e, "s9")) return 9;
        if (e && !strcmp(e, "s6")) return 6;
        return NPD;
    }();
    return form;
#else
    return NPD;   // the product library has one form; the alternatives are selectable in experiment builds only
#endif
}
inline bool wino_s3() { return wino_form() != 0; }
#ifndef WIDE_UC
#define WIDE_UC false
#endif
#define WIDE_UD (WIDE_UC ? 3 : 2)
inline bool wino_wide() {   // expt builds: DSMIL_WINO_NARROW=1 keeps the 64-cout workgroups everywhere (A/B)
#ifdef DSMIL_EXPERIMENTS
    static const int narrow = expt_env("DSMIL_WINO_NARROW");
    return !narrow;
#else
    return true;
#endif
}
#ifdef DSMIL_EXPERIMENTS
// experiment builds: DSMIL_WINO_KERNEL = unit (k_conv_wino_s3 everywhere) | w1 (default: k_conv_wino_w1 on the 128-cout layers) |
// pp (persistent role-split k_conv_wino_pp) | alt (k_conv_wino_alt, 128-cout layers)
inline int wino_expt_kernel() {
    static const int k = [] {
        const char* e = getenv("DSMIL_WINO_KERNEL");
        return (e && !strcmp(e, "pp")) ? 1 : (e && !strcmp(e, "alt")) ? 2 : (e && !strcmp(e, "w1")) ? 3 : (e && !strcmp(e, "unit")) ? 4 : 0;
    }();
    return k;
}
#endif
// the packed Winograd weights are tiled ([Cout/32][C/16][16][3][32][16]) for k_conv_wino_w1 and k_conv_wino_s3; the two
// experiment kernels (wino_variants.h) keep the position-major layout
inline bool wino_tiled() {
#ifdef DSMIL_EXPERIMENTS
    return wino_expt_kernel() != 1 && wino_expt_kernel() != 2;
#else
    return true;
#endif
}
// which Winograd convs run on k_conv_wino_w1 (one wave per SIMD, tiled weights): the packing and the launch must agree
inline bool use_w1(const ConvSpec& s) {
#ifdef DSMIL_EXPERIMENTS
    if (wino_expt_kernel() != 0 && wino_expt_kernel() != 3) return false;   // DSMIL_WINO_KERNEL = unit | pp | alt: the older kernels
#endif
    return wino_s3() && s.cout % 128 == 0;
}
#ifdef DSMIL_TRACE
constexpr size_t WINO_TRACE_WORDS = 4 * 2 * 256 * 8;
inline unsigned long long* wino_trace_buffer() {
    static unsigned long long* buf = [] {
        unsigned long long* p = nullptr;
        (void)hipMalloc(&p, WINO_TRACE_WORDS * 8);
        (void)hipMemset(p, 0, WINO_TRACE_WORDS * 8);
        return p;
    }();
    return buf;
}
#endif
// Experiment builds: DSMIL_CONV = h3 (default) | s6 | f32: MFMA form of the DIRECT convs (strided 3x3, 1x1) and the stem: h3 =
// fp16 MFMA over two-plane cuts, three products; s6 = bf16 MFMA over exact three-plane cuts, 6 plane products (both k_conv_s6 /
// k_stem_s6; weights cut at pack time); f32 = v_mfma_f32_32x32x2_f32 (k_conv / k_stem).  Read once per process; the packed
// weights and the kernels must agree.
inline int conv_np() {   // plane products of the direct convs (0 = the f32 form)
    if (g_form_override) return g_form_override;
#ifdef DSMIL_EXPERIMENTS
    static const int np = [] {
        const char* e = getenv("DSMIL_CONV");
        if (e && !strcmp(e, "f32")) return 0;
        if (e && !strcmp(e, "s6")) return 6;
        return NPD;
    }();
    return np;
#else
    return NPD;
#endif
}
inline bool conv_s6() { return conv_np() != 0; }
// floats of conv i in the packed buffer: 16 transform positions for Winograd convs (x 3 bf16 planes = 1.5
// floats per weight in the s3 form), ks*ks taps otherwise
inline long long wsize(const Arch& A, int i) {
    const ConvSpec& s = A.specs[i];
    if (use_wino(s)) return (long long)s.cout * s.cin * (wino_s3() ? 24 : 16);
    const long long n = (long long)s.cout * s.cin * s.ks * s.ks;
    return conv_s6() ? n * 3 / 2 : n;   // three bf16 planes = 1.5 floats per weight
}
inline size_t pack_offset(const Arch& A, int i) {  // floats; conv 0 (stem) is used unpacked
    size_t o = 0;
    for (int j = 1; j < i; ++j) o += (size_t)wsize(A, j);
    return o;
}
// offset of conv i's norm in the concatenated per-channel arrays of the frozen-statistics variant
inline int norm_offset(const Arch& A, int i) {
    int o = 0;
    for (int j = 0; j < i; ++j) o += A.specs[j].cout;
    return o;
}
inline int outdim(int x, int ks, int s, int p) { return (x + 2 * p - ks) / s + 1; }

struct Dims { int H1, W1, Hp, Wp, h[5], w[5]; };
Dims dims_for(int H, int W) {
    Dims d;
    d.H1 = outdim(H, 7, 2, 3); d.W1 = outdim(W, 7, 2, 3);
    d.Hp = outdim(d.H1, 3, 2, 1); d.Wp = outdim(d.W1, 3, 2, 1);
    d.h[1] = d.Hp; d.w[1] = d.Wp;
    for (int l = 2; l <= 4; ++l) { d.h[l] = outdim(d.h[l - 1], 3, 2, 1); d.w[l] = outdim(d.w[l - 1], 3, 2, 1); }
    return d;
}

// Winograd unit shape: IB images x TYB x TXB tiles with IB*TYB*TXB <= 32 and a raw input region of
// at most WRAW_MAX pixels, maximising the fraction of the 32 MFMA rows that carry real tiles.
// cb > 0: the conv runs on k_conv_wino_w1 with cb 128-cout blocks and `nchunks` 16-channel chunks — ONE workgroup per CU, so
// what counts is the number of ROUNDS of workgroups and the per-workgroup overhead, which grows with the images per unit
// (the statistics partials are formed per image: stamps 13.4 / 15.4 / 23 k cycles of epilogue at 1 / 2 / 4 images, 8 k of
// prologue, 6.1 k per chunk).  E.g. 14 x 14 maps at bs 256: 4 images x 1 x 7 tiles (448 units, 87 % of the slots) and
// 1 image x 4 x 7 (512 units, 77 %) both take 4 rounds of two cout blocks; the second has the 10 k shorter epilogue.
void wino_shape(int B, int TY, int TX, int& IB, int& TYB, int& TXB, int cb = 0, int nchunks = 0) {
    constexpr int ncu = 256;   // a CONSTANT (unpartitioned MI355X), not the visible CU count: the unit shape fixes which tiles share a
                               // statistics partial, i.e. the fp32 summation order — results must not depend on the box
    double best = -1.0;
    IB = 1; TYB = 1; TXB = TX < WTT ? TX : WTT;
    for (int txb = 1; txb <= TX && txb <= WTT; ++txb) {
        if (txb != TX && txb != (TX + 1) / 2 && txb != (TX + 2) / 3 && txb != (TX + 3) / 4 && txb != 32 && txb != 16) continue;
        for (int tyb = 1; tyb <= TY && tyb * txb <= WTT; ++tyb)
            for (int ib = 1; ib <= 16 && ib * tyb * txb <= WTT && ib <= B; ++ib) {
                if (ib > 1 && (tyb != TY && txb != TX) ) continue;  // several images per unit only for whole rows
                if ((long long)ib * (2 * tyb + 2) * (2 * txb + 2) > WRAW_MAX) continue;
                const double units = (double)((B + ib - 1) / ib) * ((TY + tyb - 1) / tyb) * ((TX + txb - 1) / txb);
                const double halo = 1e-4 * (double)(2 * tyb + 2) * (2 * txb + 2) / (tyb * txb);
                double score;
                if (cb > 0) {
                    const double rounds = ceil(units * cb / ncu);
                    const double wg = 6.1 * nchunks + 8.0 + 13.4 + 3.2 * (ib - 1);   // k cycles per workgroup
                    score = 1e6 / (rounds * wg) - halo;
                } else {
                    const double util = (double)B * TY * TX / (units * (double)WTT);
                    // prefer higher utilisation, then fewer raw pixels per tile (less halo)
                    score = util - halo;
                }
                if (score > best) { best = score; IB = ib; TYB = tyb; TXB = txb; }
            }
    }
}

struct RWs {
    size_t y0, buf[5], stat[4][2], part, total;  // stat[k] = {mean, rstd}
    long long act_elems, part_elems;
};
RWs rws_layout(int B, int H, int W, int depth = 18) {
    const Dims d = dims_for(H, W);
    const Arch& A = *arch_of(depth);
    const int exp = A.bottleneck ? 4 : 1;   // channel expansion of a block's output
    RWs r;
    size_t o = 0;
    const long long y0e = (long long)B * d.H1 * d.W1 * 64;
    r.act_elems = (long long)B * d.Hp * d.Wp * 64 * exp;  // largest activation behind the pool (layer-1 block output)
    r.y0 = o; o = al256(o + (size_t)y0e * 4);
    for (int i = 0; i < 5; ++i) { r.buf[i] = o; o = al256(o + (size_t)r.act_elems * 4); }
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 2; ++j) { r.stat[k][j] = o; o = al256(o + (size_t)B * 512 * exp * 4); }
    // partials: stem (n, tiles*4, 64, 3) or flat (tiles32, nslots, C, 2); take the max over layers
    // per (image, tile, wave): 8 x 16-pixel tiles x 4 waves (k_stem) or 16 x 16 x 8 waves (k_stem_s6), whichever is more
    const long long stem_rows = std::max<long long>(((d.H1 + 7) / 8) * 4, ((d.H1 + SS_TR - 1) / SS_TR) * 8);
    const long long stem_parts = (long long)B * stem_rows * ((d.W1 + 15) / 16) * 64 * 3;
    long long mx = stem_parts;
    for (int l = 1; l <= 4; ++l) {
        const int Cw = 64 << (l - 1);
        // direct convs of layer l: outputs at this layer's resolution (up to Cw*exp channels) and, for a Bottleneck's
        // conv1 in a strided block, at the previous layer's resolution (Cw channels)
        for (int prev = 0; prev < (A.bottleneck && l > 1 ? 2 : 1); ++prev) {
            const int HW = d.h[l - prev] * d.w[l - prev];
            const int C = prev ? Cw : Cw * exp;
            const long long M = (long long)B * HW;
            const int nslots = 31 / HW + 2;
            const long long e = ((M + 31) / 32) * nslots * C * 2;
            if (e > mx) mx = e;
        }
        const int TYl = (d.h[l] + 1) / 2, TXl = (d.w[l] + 1) / 2;
        for (int v = 0; v < 2; ++v) {   // the unit shape of both Winograd kernels (wino_shape: utilisation / rounds of workgroups)
            int ib, tyb, txb;
            wino_shape(B, TYl, TXl, ib, tyb, txb, v ? (Cw >= 128 ? Cw / 128 : 1) : 0, v ? Cw / 16 : 0);
            const long long PBl = (long long)((TYl + tyb - 1) / tyb) * ((TXl + txb - 1) / txb);
            const long long ew = (long long)B * PBl * 2 * Cw * 3;  // Winograd (cnt, mean, M2) partials
            if (ew > mx) mx = ew;
        }
    }
    r.part_elems = mx;
    r.part = o; o = al256(o + (size_t)mx * 4);
    r.total = o;
    return r;
}

// Frozen-statistics norm (eval-mode BatchNorm2d): y = (x - m[c]) * r[c] with m, r folded from the
// running statistics and the affine by the caller.  The per-(image, channel) arrays the consumers
// read are simply filled with the per-channel values — every other kernel stays as it is.
__global__ void k_fill_stats(const float* __restrict__ m_c, const float* __restrict__ r_c, float* __restrict__ mean,
                             float* __restrict__ rstd, int B, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * C) { mean[i] = m_c[i % C]; rstd[i] = r_c[i % C]; }
}

int fill_stats(hipStream_t st, const float* m_c, const float* r_c, float* mean, float* rstd, int B, int C) {
    hipLaunchKernelGGL(k_fill_stats, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, st, m_c, r_c, mean, rstd, B, C);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

int run_conv(hipStream_t st, const float* x, const float* wpk, const float* in_mean, const float* in_rstd,
             float* y, float* part, float* mean, float* rstd, int B, int H, int W, const ConvSpec& s,
             const float* bn_m = nullptr, const float* bn_r = nullptr) {
    if (use_wino(s)) {
        WinoArgs wa;
        wa.x = x; wa.u = wpk; wa.in_mean = in_mean; wa.in_rstd = in_rstd; wa.y = y; wa.part = part;
        wa.B = B; wa.H = H; wa.W = W; wa.C = s.cin; wa.Cout = s.cout;
        wa.TY = (H + 1) / 2; wa.TX = (W + 1) / 2;
        // (keyed on the layer, not on the kernel: DSMIL_WINO_KERNEL=unit of experiment builds runs the same units through
        // k_conv_wino_s3, which is what the bit-identity test of the two kernels compares)
        if (wino_s3() && s.cout % 128 == 0 && s.cin >= 64) wino_shape(B, wa.TY, wa.TX, wa.IB, wa.TYB, wa.TXB, s.cout / 128, s.cin / 16);
        else wino_shape(B, wa.TY, wa.TX, wa.IB, wa.TYB, wa.TXB);
#ifdef DSMIL_EXPERIMENTS
        static const int wexpt = expt_env("DSMIL_WINO_EXPT");
        wa.expt = wexpt;
#else
        wa.expt = 0;
#endif
        wa.nby = (wa.TY + wa.TYB - 1) / wa.TYB; wa.nbx = (wa.TX + wa.TXB - 1) / wa.TXB; wa.PB = wa.nby * wa.nbx;
#ifdef DSMIL_TRACE
        {   // DSMIL_WINO_TRACE = k: the k-th Winograd launch of the process records stamps (tools_stamp_wino.py)
            static const int want = expt_env("DSMIL_WINO_TRACE");
            static int launch = 0;
            wa.trace = (++launch == want) ? wino_trace_buffer() : nullptr;
        }
#endif
        const size_t lds = wino_s3() ? (size_t)(SV_DW + WRAW_MAX * SRLD) * sizeof(float)
                                     : (size_t)(2 * WTILE + 2 * WRAW_MAX * WLD) * sizeof(float);
        dim3 grid((unsigned)(((B + wa.IB - 1) / wa.IB) * wa.PB), (unsigned)(s.cout / 64));
        const int slot = dsmil_prof::begin(dsmil_prof::CH_CONV, st);
        if (wino_s3()) {
            // product configuration: weights two positions ahead (UD = 2), producer statistics staged through LDS (LS,
            // +4 KiB: 80 KiB per workgroup, still two per CU), NP = 6 | 9 by DSMIL_WINO
            const size_t lds_ls = lds + 4096;
#ifdef DSMIL_EXPERIMENTS
            const bool np9 = wino_form() == 9, np6 = wino_form() == 6;
#define DSMIL_IF_NP9 if (np9)
#define DSMIL_IF_NP6 if (np6)
#else
#define DSMIL_IF_NP9 if constexpr (false)   // the product library has the fp16 three-product form only: the bf16 forms are
#define DSMIL_IF_NP6 if constexpr (false)   // not instantiated
#endif
            auto go = [&](auto kern, size_t l) {
                allow_lds((const void*)kern, l);
                hipLaunchKernelGGL(kern, grid, dim3(256), l, st, wa);
            };
            const bool half1 = wino_form() == 1;   // the opt-in one-plane path (dsmil_resnet_forward_ex)
#ifdef DSMIL_EXPERIMENTS
            static const int ncu = [] {
                int dev = 0, n = 0;
                (void)hipGetDevice(&dev);
                (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
                return n > 0 ? n : 256;
            }();
            const int nunits = (int)grid.x, nitems = (int)(grid.x * grid.y);
            const size_t lds_pp = (size_t)(2 * SV_DW + 2 * WRAW_MAX * SRLD + 1024) * sizeof(float);
            auto go_pp = [&](auto kern) {   // persistent: one workgroup per CU walks the (unit, cout tile) items
                allow_lds((const void*)kern, lds_pp);
                hipLaunchKernelGGL(kern, dim3((unsigned)(nitems < ncu ? nitems : ncu)), dim3(512), lds_pp, st, wa, nunits, nitems);
            };
            const dim3 grida(grid.x, (unsigned)(s.cout / 128));
            auto goa = [&](auto kern) {
                allow_lds((const void*)kern, lds_pp);
                hipLaunchKernelGGL(kern, grida, dim3(512), lds_pp, st, wa);
            };
            int which = wino_expt_kernel();         // DSMIL_WINO_KERNEL = pp | alt | w1
            if ((which == 1 || which == 2) && !(np9 || np6)) which = 0;   // the two rejected restructurings exist in the bf16 forms only
            const int w1_abl = wa.expt;
            if (which == 3) wa.expt = 0;            // with w1 named explicitly the ablation bits address that kernel only
#else
            constexpr int which = 0;
#endif
            if (half1) {
                if (use_w1(s)) {
                    const dim3 gridw(grid.x * (unsigned)(s.cout / 128));
                    const size_t lds_w1 = (size_t)(2 * SV_DW + 2 * WRAW_MAX * SRLD + 256 + 1024) * sizeof(float);
                    auto gow1 = [&](auto kern) {
                        allow_lds((const void*)kern, lds_w1);
                        hipLaunchKernelGGL(kern, gridw, dim3(256), lds_w1, st, wa);
                    };
                    if (in_mean) gow1(k_conv_wino_w1<true, 1>); else gow1(k_conv_wino_w1<false, 1>);
                } else if (in_mean) go(k_conv_wino_s3<true, 2, true, 1>, lds_ls);
                else go(k_conv_wino_s3<false, 2, false, 1>, lds);
            } else if (which == 1) {
#ifdef DSMIL_EXPERIMENTS
                if (in_mean) { if (np9) go_pp(k_conv_wino_pp<true, 9, PP_UD>); else go_pp(k_conv_wino_pp<true, 6, PP_UD>); }
                else { if (np9) go_pp(k_conv_wino_pp<false, 9, PP_UD>); else go_pp(k_conv_wino_pp<false, 6, PP_UD>); }
#endif
            } else if (which == 2 && s.cout % 128 == 0) {
#ifdef DSMIL_EXPERIMENTS
                if (in_mean) { if (np9) goa(k_conv_wino_alt<true, 9>); else goa(k_conv_wino_alt<true, 6>); }
                else { if (np9) goa(k_conv_wino_alt<false, 9>); else goa(k_conv_wino_alt<false, 6>); }
#endif
            } else if (use_w1(s)) {
                // one wave per SIMD: 256 threads, 128 couts, all 16 positions per wave (wino_w1.h)
                const dim3 gridw(grid.x * (unsigned)(s.cout / 128));   // 1-D: the kernel maps it to (unit, cout block) per XCD
                const size_t lds_w1 = (size_t)(2 * SV_DW + 2 * WRAW_MAX * SRLD + 256 + 1024) * sizeof(float);
                auto gow1 = [&](auto kern) {
                    allow_lds((const void*)kern, lds_w1);
                    hipLaunchKernelGGL(kern, gridw, dim3(256), lds_w1, st, wa);
                };
#ifdef DSMIL_EXPERIMENTS
                switch (w1_abl) {   // DSMIL_WINO_EXPT: compile-time ablations of the w1 kernel (timing only)
#define W1_ABL(n) case n: if (in_mean) gow1(k_conv_wino_w1<true, NPD, n>); else gow1(k_conv_wino_w1<false, NPD, n>); break;
                    W1_ABL(3) W1_ABL(4) W1_ABL(15) W1_ABL(16) W1_ABL(32) W1_ABL(64) W1_ABL(96) W1_ABL(128) W1_ABL(143)
#undef W1_ABL
                    default:
                        if (in_mean) { if (np9) gow1(k_conv_wino_w1<true, 9>); else if (np6) gow1(k_conv_wino_w1<true, 6>); else gow1(k_conv_wino_w1<true, NPD>); }
                        else { if (np9) gow1(k_conv_wino_w1<false, 9>); else if (np6) gow1(k_conv_wino_w1<false, 6>); else gow1(k_conv_wino_w1<false, NPD>); }
                }
#else
                if (in_mean) gow1(k_conv_wino_w1<true, NPD>); else gow1(k_conv_wino_w1<false, NPD>);
#endif
#ifdef DSMIL_EXPERIMENTS
            } else if (s.cout % 128 == 0 && wino_wide()) {
                // (DSMIL_WINO_KERNEL=unit) 128 couts per workgroup (512 threads, one workgroup per CU): round 2's form of layers 2-4
                const dim3 gridw(grid.x, (unsigned)(s.cout / 128));
                auto gow = [&](auto kern, size_t l) {
                    allow_lds((const void*)kern, l);
                    hipLaunchKernelGGL(kern, gridw, dim3(512), l, st, wa);
                };
                if (in_mean) {
                    if (np9) gow(k_conv_wino_s3<true, WIDE_UD, true, 9, 4, WIDE_UC>, lds_ls);
                    else if (np6) gow(k_conv_wino_s3<true, WIDE_UD, true, 6, 4, WIDE_UC>, lds_ls);
                    else gow(k_conv_wino_s3<true, WIDE_UD, true, NPD, 4, WIDE_UC>, lds_ls);
                } else {
                    if (np9) gow(k_conv_wino_s3<false, WIDE_UD, false, 9, 4, WIDE_UC>, lds);
                    else if (np6) gow(k_conv_wino_s3<false, WIDE_UD, false, 6, 4, WIDE_UC>, lds);
                    else gow(k_conv_wino_s3<false, WIDE_UD, false, NPD, 4, WIDE_UC>, lds);
                }
#endif
            } else if (in_mean) {
                DSMIL_IF_NP9 go(k_conv_wino_s3<true, 2, true, 9>, lds_ls);
                else DSMIL_IF_NP6 go(k_conv_wino_s3<true, 2, true, 6>, lds_ls);
                else go(k_conv_wino_s3<true, 2, true, NPD>, lds_ls);
            } else {
                DSMIL_IF_NP9 go(k_conv_wino_s3<false, 2, false, 9>, lds);
                else DSMIL_IF_NP6 go(k_conv_wino_s3<false, 2, false, 6>, lds);
                else go(k_conv_wino_s3<false, 2, false, NPD>, lds);
            }
        }
#ifdef DSMIL_EXPERIMENTS   // DSMIL_WINO=f32: the f32-MFMA Winograd unit
        else if (in_mean) hipLaunchKernelGGL((k_conv_wino<true>), grid, dim3(256), lds, st, wa);
        else hipLaunchKernelGGL((k_conv_wino<false>), grid, dim3(256), lds, st, wa);
#endif
        dsmil_prof::end(dsmil_prof::CH_CONV, slot, st);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
        if (bn_m) return fill_stats(st, bn_m, bn_r, mean, rstd, B, s.cout);
        hipLaunchKernelGGL(k_in_finalize_cnt, dim3((unsigned)B, (unsigned)((s.cout + 63) / 64)), dim3(256), 0, st, part, mean, rstd, wa.PB * 2, s.cout);
        return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
    }
    ConvArgs a;
    a.x = x; a.w = wpk; a.in_mean = in_mean; a.in_rstd = in_rstd; a.y = y; a.part = part;
    a.B = B; a.H = H; a.W = W; a.Cin = s.cin;
    a.Ho = outdim(H, s.ks, s.stride, s.pad); a.Wo = outdim(W, s.ks, s.stride, s.pad);
    a.Cout = s.cout; a.ks = s.ks; a.stride = s.stride; a.pad = s.pad;
    const int HW = a.Ho * a.Wo;
    a.nslots = 31 / HW + 2;
    a.Mtot = (long long)B * HW;
    const bool norm = in_mean != nullptr;
    const int slot = dsmil_prof::begin(dsmil_prof::CH_CONV, st);
    const long long blocks128 = ((a.Mtot + 127) / 128) * (s.cout / 128 > 0 ? s.cout / 128 : 1);
    if (conv_s6()) {
        // tile = BM pixels x TN output channels per 256-thread workgroup: 128 x 128, 128 x 64, 64 x 256 or 64 x 128
        auto go = [&](auto kern, int BM, int TN) {
            const size_t lds = (size_t)2 * (BM + TN) * S6LD * 4;
            allow_lds((const void*)kern, lds);
            dim3 grid((unsigned)((a.Mtot + BM - 1) / BM), (unsigned)(s.cout / TN));
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
        };
        // Measured per layer (tools/s6_shapes.sh, bs 256): the activation rows are the expensive operand to stage (im2col
        // gather, producer's IN + ReLU, plane cut; the weights are a plain copy), so the tile is as short in pixels and as wide
        // in channels as the layer allows — 64 x 256, else 64 x 128 — which also quantises better (l4.0.conv1: 392 workgroups
        // in ONE round instead of 784 on 768 slots; 195 us against 288).  Six direct convs of ResNet-18: 850 -> 726 us.
        (void)blocks128;
        int shape = s.cout % 256 == 0 ? 24 : s.cout % 128 == 0 ? 22 : 42;
#ifdef DSMIL_EXPERIMENTS
        {   // DSMIL_S6_TILE = 44 | 42 | 24 | 22 forces a tile shape wherever the channel count allows it (A/B)
            static const int force = expt_env("DSMIL_S6_TILE");
            if (force == 44 && s.cout % 128 == 0) shape = 44;
            if (force == 42) shape = 42;
            if (force == 24 && s.cout % 256 == 0) shape = 24;
            if (force == 22 && s.cout % 128 == 0) shape = 22;
        }
#endif
        if (conv_np() == 1) {   // the opt-in one-plane path
            switch (shape) {
                case 24: if (norm) go(k_conv_s6<2, 4, true, 1>, 64, 256); else go(k_conv_s6<2, 4, false, 1>, 64, 256); break;
                case 22: if (norm) go(k_conv_s6<2, 2, true, 1>, 64, 128); else go(k_conv_s6<2, 2, false, 1>, 64, 128); break;
                default: if (norm) go(k_conv_s6<4, 2, true, 1>, 128, 64); else go(k_conv_s6<4, 2, false, 1>, 128, 64); break;
            }
        } else
#ifdef DSMIL_EXPERIMENTS
        if (conv_np() == 6) {   // DSMIL_CONV=s6: the bf16 three-plane form
            switch (shape) {
                case 44: if (norm) go(k_conv_s6<4, 4, true, 6>, 128, 128); else go(k_conv_s6<4, 4, false, 6>, 128, 128); break;
                case 24: if (norm) go(k_conv_s6<2, 4, true, 6>, 64, 256); else go(k_conv_s6<2, 4, false, 6>, 64, 256); break;
                case 22: if (norm) go(k_conv_s6<2, 2, true, 6>, 64, 128); else go(k_conv_s6<2, 2, false, 6>, 64, 128); break;
                default: if (norm) go(k_conv_s6<4, 2, true, 6>, 128, 64); else go(k_conv_s6<4, 2, false, 6>, 128, 64); break;
            }
        } else
#endif
        switch (shape) {
#ifdef DSMIL_EXPERIMENTS
            case 44: if (norm) go(k_conv_s6<4, 4, true, NPD>, 128, 128); else go(k_conv_s6<4, 4, false, NPD>, 128, 128); break;
#endif
            case 24: if (norm) go(k_conv_s6<2, 4, true, NPD>, 64, 256); else go(k_conv_s6<2, 4, false, NPD>, 64, 256); break;
            case 22: if (norm) go(k_conv_s6<2, 2, true, NPD>, 64, 128); else go(k_conv_s6<2, 2, false, NPD>, 64, 128); break;
            default: if (norm) go(k_conv_s6<4, 2, true, NPD>, 128, 64); else go(k_conv_s6<4, 2, false, NPD>, 128, 64); break;
        }
        dsmil_prof::end(dsmil_prof::CH_CONV, slot, st);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
        if (bn_m) return fill_stats(st, bn_m, bn_r, mean, rstd, B, s.cout);
        hipLaunchKernelGGL(k_in_finalize_flat, dim3((unsigned)B, (unsigned)((s.cout + 63) / 64)), dim3(256), 0, st, part, mean, rstd,
                           B, HW, s.cout, a.nslots, a.Mtot);
        return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
    }
#ifndef DSMIL_EXPERIMENTS
    (void)blocks128;
    return DSMIL_E_UNSUPPORTED;   // (unreachable: the product library has the bf16-MFMA form only)
#else
    // DSMIL_CONV=f32.  Tile choice: 128x64 for Cout = 64; 128x128 while that yields >= 4 workgroups per CU;
    // 64x64 (finer units, less tail quantisation) for the small late-layer maps
    if (s.cout == 64) {
        const size_t lds = (size_t)(2 * 128 * LDK + 2 * 64 * LDK) * 4;
        dim3 grid((unsigned)((a.Mtot + 127) / 128), 1);
        if (norm) hipLaunchKernelGGL((k_conv<4, 2, true>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((k_conv<4, 2, false>), grid, dim3(256), lds, st, a);
    } else if (blocks128 >= 1024) {
        const size_t lds = (size_t)(2 * 128 * LDK + 2 * 128 * LDK) * 4;
        dim3 grid((unsigned)((a.Mtot + 127) / 128), (unsigned)(s.cout / 128));
        if (norm) hipLaunchKernelGGL((k_conv<4, 4, true>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((k_conv<4, 4, false>), grid, dim3(256), lds, st, a);
    } else {
        const size_t lds = (size_t)(2 * 64 * LDK + 2 * 64 * LDK) * 4;
        dim3 grid((unsigned)((a.Mtot + 63) / 64), (unsigned)(s.cout / 64));
        if (norm) hipLaunchKernelGGL((k_conv<2, 1, true>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((k_conv<2, 1, false>), grid, dim3(256), lds, st, a);
    }
    dsmil_prof::end(dsmil_prof::CH_CONV, slot, st);
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    if (bn_m) return fill_stats(st, bn_m, bn_r, mean, rstd, B, s.cout);
    hipLaunchKernelGGL(k_in_finalize_flat, dim3((unsigned)B, (unsigned)((s.cout + 63) / 64)), dim3(256), 0, st, part, mean, rstd,
                       B, HW, s.cout, a.nslots, a.Mtot);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
#endif
}

void set_conv_attrs() {   // per (device, kernel): lds_attr.h.  Only the f32 forms of experiment builds: the others set theirs at launch
#ifdef DSMIL_EXPERIMENTS
    const int l4 = (2 * 128 * LDK + 2 * 128 * LDK) * 4, l2 = (2 * 128 * LDK + 2 * 64 * LDK) * 4;
    allow_lds((const void*)k_conv<4, 4, true>, l4);
    allow_lds((const void*)k_conv<4, 4, false>, l4);
    allow_lds((const void*)k_conv<4, 2, true>, l2);
    allow_lds((const void*)k_conv<4, 2, false>, l2);
    allow_lds((const void*)k_conv_wino<true>, (2 * WTILE + 2 * WRAW_MAX * WLD) * 4);
    allow_lds((const void*)k_conv_wino<false>, (2 * WTILE + 2 * WRAW_MAX * WLD) * 4);
#endif
}

}  // namespace

extern "C" {

int dsmil_resnet_mfma_forms(int32_t* wino_products, int32_t* direct_products) {
    if (wino_products) *wino_products = wino_form();
    if (direct_products) *direct_products = conv_np();
    return DSMIL_OK;
}

int32_t dsmil_resnet_num_convs(int32_t depth) { const Arch* A = arch_of(depth); return A ? A->nconv : 0; }
int32_t dsmil_resnet_norm_channels(int32_t depth) { const Arch* A = arch_of(depth); return A ? norm_offset(*A, A->nconv) : 0; }
// the packed image ends with the stem's three bf16 planes (k_stem_s6; conv 0 keeps its raw OIHW argument for the f32 form)
size_t dsmil_resnet_packed_bytes(int32_t depth) {
    const Arch* A = arch_of(depth);
    return A ? (pack_offset(*A, A->nconv) + SS_WIMG / 2) * sizeof(float) : 0;
}

int dsmil_resnet_pack(int32_t depth, const float* const* conv_w, float* packed, void* stream) {
    const Arch* A = arch_of(depth);
    if (!A) return DSMIL_E_UNSUPPORTED;
    if (!conv_w || !packed) return DSMIL_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (!conv_w[0]) return DSMIL_E_INVALID;
    hipLaunchKernelGGL(k_pack_stem_s6, dim3(46), dim3(256), 0, st, conv_w[0],
                       (unsigned short*)(packed + pack_offset(*A, A->nconv)), conv_np());
    if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    for (int i = 1; i < A->nconv; ++i) {
        if (!conv_w[i]) return DSMIL_E_INVALID;
        const ConvSpec& s = A->specs[i];
        if (use_wino(s)) {
            long long blocks = ((long long)s.cout * s.cin + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            if (wino_s3())
                hipLaunchKernelGGL(k_pack_wino_s3, dim3((unsigned)blocks), dim3(256), 0, st, conv_w[i],
                                   (unsigned short*)(packed + pack_offset(*A, i)), s.cout, s.cin, wino_tiled() ? 1 : 0, wino_form());
#ifdef DSMIL_EXPERIMENTS
            else
                hipLaunchKernelGGL(k_pack_wino, dim3((unsigned)blocks), dim3(256), 0, st, conv_w[i],
                                   packed + pack_offset(*A, i), s.cout, s.cin);
#endif
        } else {
            const long long total = (long long)s.cout * s.cin * s.ks * s.ks;
            long long blocks = (total + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            if (conv_s6())
                hipLaunchKernelGGL(k_pack_conv_s6, dim3((unsigned)blocks), dim3(256), 0, st, conv_w[i],
                                   (unsigned short*)(packed + pack_offset(*A, i)), s.cout, s.cin, s.ks * s.ks, conv_np());
#ifdef DSMIL_EXPERIMENTS
            else
                hipLaunchKernelGGL(k_pack_conv, dim3((unsigned)blocks), dim3(256), 0, st, conv_w[i],
                                   packed + pack_offset(*A, i), s.cout, s.cin, s.ks * s.ks);
#endif
        }
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    }
    return DSMIL_OK;
}

size_t dsmil_resnet18_packed_bytes(void) { return dsmil_resnet_packed_bytes(18); }
int dsmil_resnet18_pack(const float* const* conv_w, float* packed, void* stream) {
    return dsmil_resnet_pack(18, conv_w, packed, stream);
}

size_t dsmil_resnet18_workspace_bytes(int32_t B, int32_t H, int32_t W) {
    if (B <= 0 || H < 32 || W < 32) return 0;
    return rws_layout(B, H, W).total;
}

size_t dsmil_resnet_workspace_bytes(int32_t depth, int32_t B, int32_t H, int32_t W) {
    if (!arch_of(depth) || B <= 0 || H < 32 || W < 32) return 0;
    return rws_layout(B, H, W, depth).total;
}

int32_t dsmil_resnet_feature_dim(int32_t depth) { const Arch* A = arch_of(depth); return A ? A->feat : 0; }

static int resnet18in_forward_impl(const void* x_nchw, bool u8, int32_t B, int32_t H, int32_t W, const float* conv1_w,
                                   const float* packed, const float* fc_w, const float* fc_b, int32_t C,
                                   float* feats, float* classes, void* ws, size_t ws_bytes, void* stream,
                                   const float* bn_m = nullptr, const float* bn_r = nullptr, int depth = 18) {
    const Arch* Ap = arch_of(depth);
    if (!Ap) return DSMIL_E_UNSUPPORTED;
    const Arch& A = *Ap;
    auto bm = [&](int i) { return bn_m ? bn_m + norm_offset(A, i) : nullptr; };
    auto br = [&](int i) { return bn_r ? bn_r + norm_offset(A, i) : nullptr; };
    if (!x_nchw || !conv1_w || !packed || !feats || !ws) return DSMIL_E_INVALID;
    if (B <= 0 || H < 32 || W < 32) return DSMIL_E_INVALID;
    if (classes && (!fc_w || !fc_b || C <= 0)) return DSMIL_E_INVALID;
    if (((uintptr_t)ws % 256) || ((uintptr_t)packed % 16)) return DSMIL_E_ALIGN;
    const RWs L = rws_layout(B, H, W, depth);
    if (ws_bytes < L.total) return DSMIL_E_WORKSPACE;
    const Dims d = dims_for(H, W);
    if (d.h[4] < 1 || d.w[4] < 1) return DSMIL_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    set_conv_attrs();
    char* w8 = (char*)ws;
    float* y0 = (float*)(w8 + L.y0);
    float* buf[5];
    for (int i = 0; i < 5; ++i) buf[i] = (float*)(w8 + L.buf[i]);
    float* mean[4];
    float* rstd[4];
    for (int k = 0; k < 4; ++k) { mean[k] = (float*)(w8 + L.stat[k][0]); rstd[k] = (float*)(w8 + L.stat[k][1]); }
    float* part = (float*)(w8 + L.part);

    // ---- stem: conv1 -> IN -> ReLU -> maxpool
    size_t b16_halo_bytes = 0;
    bool b16_direct = false;
    {
        const bool s6 = conv_s6();   // DSMIL_CONV: the stem follows the direct convs' MFMA form
        const int tx = (d.W1 + 15) / 16, ty = s6 ? (d.H1 + SS_TR - 1) / SS_TR : (d.H1 + 7) / 8;
        // InstanceNorm trunks: the max-pool is fused into the stem (a frozen BatchNorm may have a negative scale, for which
        // the pool needs the window MINIMUM too: that path keeps the raw map + k_norm_relu_maxpool)
        const bool fuse = s6 && !bn_m && stem_fuse();
        auto stem_go = [&](auto kern, size_t lds_, float* out, float* halo, int Hp_, int Wp_) {
            allow_lds((const void*)kern, lds_);
            hipLaunchKernelGGL(kern, dim3((unsigned)ty, (unsigned)B), dim3(512), lds_, st, x_nchw,
                               (const unsigned short*)(packed + pack_offset(A, A.nconv)), out, part, B, H, W, d.H1, d.W1, tx, ty, halo, Hp_, Wp_);
        };
#ifdef DSMIL_EXPERIMENTS
        const bool stem6 = conv_np() == 6;   // DSMIL_CONV=s6: the bf16 three-plane form
#else
        constexpr bool stem6 = false;
#endif
        const bool stem1 = conv_np() == 1;   // the opt-in one-plane path
        if (fuse) {
            // the pooled raw map goes straight to the max-pool's destination; the raw-map region holds the halo rows
            const size_t ldsp = SS_LDS + SS_POOL_LDS;
            if (stem1) {
                if (u8) stem_go(k_stem_s6<true, true, 1>, ldsp, buf[0], y0, d.Hp, d.Wp); else stem_go(k_stem_s6<false, true, 1>, ldsp, buf[0], y0, d.Hp, d.Wp);
            } else if (stem6) {
#ifdef DSMIL_EXPERIMENTS
                if (u8) stem_go(k_stem_s6<true, true, 6>, ldsp, buf[0], y0, d.Hp, d.Wp); else stem_go(k_stem_s6<false, true, 6>, ldsp, buf[0], y0, d.Hp, d.Wp);
#endif
            } else if (u8) stem_go(k_stem_s6<true, true, NPD>, ldsp, buf[0], y0, d.Hp, d.Wp);
            else stem_go(k_stem_s6<false, true, NPD>, ldsp, buf[0], y0, d.Hp, d.Wp);
        } else if (s6) {
            if (stem1) {
                if (u8) stem_go(k_stem_s6<true, false, 1>, SS_LDS, y0, nullptr, 0, 0); else stem_go(k_stem_s6<false, false, 1>, SS_LDS, y0, nullptr, 0, 0);
            } else if (stem6) {
#ifdef DSMIL_EXPERIMENTS
                if (u8) stem_go(k_stem_s6<true, false, 6>, SS_LDS, y0, nullptr, 0, 0); else stem_go(k_stem_s6<false, false, 6>, SS_LDS, y0, nullptr, 0, 0);
#endif
            } else if (u8) stem_go(k_stem_s6<true, false, NPD>, SS_LDS, y0, nullptr, 0, 0);
            else stem_go(k_stem_s6<false, false, NPD>, SS_LDS, y0, nullptr, 0, 0);
        }
#ifdef DSMIL_EXPERIMENTS   // DSMIL_CONV=f32
        else if (u8) hipLaunchKernelGGL(k_stem<true>, dim3((unsigned)ty, (unsigned)B), dim3(256), 0, st, x_nchw, conv1_w, y0,
                                        part, B, H, W, d.H1, d.W1, tx, ty);
        else hipLaunchKernelGGL(k_stem<false>, dim3((unsigned)ty, (unsigned)B), dim3(256), 0, st, x_nchw, conv1_w, y0,
                                part, B, H, W, d.H1, d.W1, tx, ty);
#endif
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
        if (bn_m) { const int rcf = fill_stats(st, bm(0), br(0), mean[0], rstd[0], B, 64); if (rcf) return rcf; }
        else hipLaunchKernelGGL(k_in_finalize_stem, dim3((unsigned)B), dim3(64 * FS_G), 0, st, part,
                                mean[0], rstd[0], B, tx * ty * (s6 ? 8 : 4));
        const long long total = (long long)B * d.Hp * d.Wp * 16;
        long long blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        // the bf16-activation trunk takes the stem's output as bf16 in its own layout, behind the halo rows of the raw-map region
        b16_halo_bytes = al256((size_t)B * ty * d.W1 * 64 * sizeof(float));
        b16_direct = g_b16_trunk && fuse && b16_halo_bytes + b16::scratch_bytes(B, d.Hp, d.Wp) <= L.buf[0] - L.y0;
        if (fuse) hipLaunchKernelGGL(k_pool_fix_norm, dim3((unsigned)blocks), dim3(256), 0, st, buf[0], y0, mean[0], rstd[0],
                                     B, d.Hp, d.Wp, d.W1, ty, b16_direct ? (unsigned short*)(w8 + L.y0 + b16_halo_bytes) : (unsigned short*)nullptr, g_b16_trunk == 2 ? 1 : 0);
        else hipLaunchKernelGGL(k_norm_relu_maxpool, dim3((unsigned)blocks), dim3(256), 0, st, y0, mean[0], rstd[0],
                                buf[0], B, d.H1, d.W1, d.Hp, d.Wp, 64);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    }
    if (g_b16_trunk) {
        // the bf16-activation trunk: its four activation buffers and statistics partials take the raw-map region (the
        // stem's pooled output sits in buf[0]; the halo rows in y0 are dead behind k_pool_fix_norm); its weight image sits
        // behind the regular packed image
        if (!b16::arch_ok(A) || bn_m) return DSMIL_E_UNSUPPORTED;
        if (b16::scratch_bytes(B, d.Hp, d.Wp) > L.buf[0] - L.y0) return DSMIL_E_UNSUPPORTED;
        const unsigned short* wpk16 = (const unsigned short*)((const char*)packed + dsmil_resnet_packed_bytes(depth));
        // (b16_direct: k_pool_fix_norm has already written the trunk's input, bf16, behind the halo rows)
        const int rc = b16::trunk(st, A, b16_direct ? nullptr : buf[0], wpk16, w8 + L.y0 + (b16_direct ? b16_halo_bytes : 0), B, d.Hp, d.Wp, feats, g_b16_trunk == 2);
        if (rc != DSMIL_OK) return rc;
        if (classes) return dsmil_fc_forward(feats, B, A.feat, C, fc_w, fc_b, classes, stream);
        return DSMIL_OK;
    }
    // ---- layers 1..4, nblk[l] BasicBlocks each.  cur = block input (materialised, normalised)
    float* cur = buf[0];
    float* y1 = buf[1];
    float* y2 = buf[2];
    float* yd = buf[3];
    float* nxt = buf[4];
    int ci = 1;  // index into A.specs / packed weights
    int Hc = d.Hp, Wc = d.Wp;
    for (int l = 1; l <= 4; ++l) {
        for (int b = 0; b < A.nblk[l - 1]; ++b) {
            const bool last = (l == 4 && b == A.nblk[3] - 1);
            int rc, Ho, Wo, Cc;
            bool down;
            float* yout;   // raw output of the block's last conv
            int sout;      // its statistics slot
            if (A.bottleneck) {
                // conv1 1x1 -> IN -> ReLU -> conv2 3x3 (stride) -> IN -> ReLU -> conv3 1x1 -> IN, + identity | downsample
                down = b == 0;
                const ConvSpec& s1 = A.specs[ci];
                const ConvSpec& s2 = A.specs[ci + 1];
                const ConvSpec& s3 = A.specs[ci + 2];
                Ho = outdim(Hc, 3, s2.stride, 1); Wo = outdim(Wc, 3, s2.stride, 1);
                rc = run_conv(st, cur, packed + pack_offset(A, ci), nullptr, nullptr, y1, part, mean[1], rstd[1], B, Hc, Wc, s1, bm(ci), br(ci));
                if (rc) return rc;
                rc = run_conv(st, y1, packed + pack_offset(A, ci + 1), mean[1], rstd[1], y2, part, mean[2], rstd[2], B, Hc, Wc, s2, bm(ci + 1), br(ci + 1));
                if (rc) return rc;
                // y1 and statistics slot 1 are free again: conv3 writes there
                rc = run_conv(st, y2, packed + pack_offset(A, ci + 2), mean[2], rstd[2], y1, part, mean[1], rstd[1], B, Ho, Wo, s3, bm(ci + 2), br(ci + 2));
                if (rc) return rc;
                if (down) {
                    rc = run_conv(st, cur, packed + pack_offset(A, ci + 3), nullptr, nullptr, yd, part, mean[3], rstd[3], B, Hc, Wc, A.specs[ci + 3], bm(ci + 3), br(ci + 3));
                    if (rc) return rc;
                }
                yout = y1; sout = 1; Cc = s3.cout;
                ci += down ? 4 : 3;
            } else {
                down = (l > 1 && b == 0);
                const ConvSpec& sa = A.specs[ci];
                const ConvSpec& sb = A.specs[ci + 1];
                Ho = outdim(Hc, sa.ks, sa.stride, sa.pad); Wo = outdim(Wc, sa.ks, sa.stride, sa.pad);
                rc = run_conv(st, cur, packed + pack_offset(A, ci), nullptr, nullptr, y1, part, mean[1], rstd[1], B, Hc, Wc, sa, bm(ci), br(ci));
                if (rc) return rc;
                rc = run_conv(st, y1, packed + pack_offset(A, ci + 1), mean[1], rstd[1], y2, part, mean[2], rstd[2], B, Ho, Wo, sb, bm(ci + 1), br(ci + 1));
                if (rc) return rc;
                if (down) {
                    rc = run_conv(st, cur, packed + pack_offset(A, ci + 2), nullptr, nullptr, yd, part, mean[3], rstd[3], B, Hc, Wc, A.specs[ci + 2], bm(ci + 2), br(ci + 2));
                    if (rc) return rc;
                }
                yout = y2; sout = 2; Cc = sa.cout;
                ci += down ? 3 : 2;
            }
            const long long npix = (long long)B * Ho * Wo;
            if (last) {
                hipLaunchKernelGGL(k_norm_add_relu_pool, dim3((unsigned)((B * Cc + 255) / 256)), dim3(256), 0, st, yout,
                                   mean[sout], rstd[sout], cur, feats, B, Ho * Wo, Cc);
            } else {
                const int c4n = Cc / 4, tpb = 256 / (c4n < 256 ? c4n : 256);
                long long blocks = (npix + tpb - 1) / tpb;
                if (blocks > 8192) blocks = 8192;
                if (down) hipLaunchKernelGGL(k_norm_add_relu<true>, dim3((unsigned)blocks), dim3(256), 0, st, yout, mean[sout], rstd[sout], yd, mean[3], rstd[3], nxt, npix, Ho * Wo, Cc);
                else hipLaunchKernelGGL(k_norm_add_relu<false>, dim3((unsigned)blocks), dim3(256), 0, st, yout, mean[sout], rstd[sout], cur, nullptr, nullptr, nxt, npix, Ho * Wo, Cc);
            }
            if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
            float* t = cur; cur = nxt; nxt = t;
            Hc = Ho; Wc = Wo;
        }
    }
    if (classes) return dsmil_fc_forward(feats, B, A.feat, C, fc_w, fc_b, classes, stream);
    return DSMIL_OK;
}

int dsmil_resnet18in_forward(const float* x_nchw, int32_t B, int32_t H, int32_t W, const float* conv1_w,
                             const float* packed, const float* fc_w, const float* fc_b, int32_t C,
                             float* feats, float* classes, void* ws, size_t ws_bytes, void* stream) {
    return resnet18in_forward_impl(x_nchw, false, B, H, W, conv1_w, packed, fc_w, fc_b, C, feats, classes, ws,
                                   ws_bytes, stream);
}

int dsmil_resnet18in_forward_u8(const uint8_t* x_nhwc, int32_t B, int32_t H, int32_t W, const float* conv1_w,
                                const float* packed, const float* fc_w, const float* fc_b, int32_t C,
                                float* feats, float* classes, void* ws, size_t ws_bytes, void* stream) {
    return resnet18in_forward_impl(x_nhwc, true, B, H, W, conv1_w, packed, fc_w, fc_b, C, feats, classes, ws,
                                   ws_bytes, stream);
}

int32_t dsmil_resnet18_norm_channels(void) { return dsmil_resnet_norm_channels(18); }

int dsmil_resnet_forward(int32_t depth, const void* x, int32_t x_is_u8_nhwc, int32_t B, int32_t H, int32_t W,
                         const float* conv1_w, const float* packed, const float* bn_mean, const float* bn_rstd,
                         const float* fc_w, const float* fc_b, int32_t C, float* feats, float* classes, void* ws,
                         size_t ws_bytes, void* stream) {
    if ((bn_mean == nullptr) != (bn_rstd == nullptr)) return DSMIL_E_INVALID;
    return resnet18in_forward_impl(x, x_is_u8_nhwc != 0, B, H, W, conv1_w, packed, fc_w, fc_b, C, feats, classes, ws,
                                   ws_bytes, stream, bn_mean, bn_rstd, depth);
}

// precision 2 = the bf16-activation trunk (its packed image is larger)
size_t dsmil_resnet_packed_bytes_ex(int32_t depth, int32_t precision) {
    const Arch* A = arch_of(depth);
    if (!A || precision < 0 || precision > 3) return 0;
    if (precision >= 2) return b16::arch_ok(*A) ? dsmil_resnet_packed_bytes(depth) + b16::packed_bytes(*A) : 0;
    return dsmil_resnet_packed_bytes(depth);
}

int dsmil_resnet_pack_ex(int32_t depth, const float* const* conv_w, float* packed, int32_t precision, void* stream) {
    if (precision < 0 || precision > 3) return DSMIL_E_INVALID;
    FormOverride fo(precision >= 1 ? 1 : 0);
    const int rc = dsmil_resnet_pack(depth, conv_w, packed, stream);
    if (rc != DSMIL_OK || precision < 2) return rc;
    const Arch* A = arch_of(depth);
    if (!b16::arch_ok(*A)) return DSMIL_E_UNSUPPORTED;
    return b16::pack_all(*A, conv_w, (unsigned short*)((char*)packed + dsmil_resnet_packed_bytes(depth)), (hipStream_t)stream, precision == 3);
}

int dsmil_resnet_forward_ex(int32_t depth, const void* x, int32_t x_is_u8_nhwc, int32_t B, int32_t H, int32_t W,
                            const float* conv1_w, const float* packed, const float* bn_mean, const float* bn_rstd,
                            const float* fc_w, const float* fc_b, int32_t C, float* feats, float* classes, void* ws,
                            size_t ws_bytes, int32_t precision, void* stream) {
    if (precision < 0 || precision > 3) return DSMIL_E_INVALID;
    if (precision >= 2 && (bn_mean || bn_rstd)) return DSMIL_E_UNSUPPORTED;
    FormOverride fo(precision >= 1 ? 1 : 0);
    struct B16Flag { int saved; explicit B16Flag(int v) : saved(g_b16_trunk) { g_b16_trunk = v; } ~B16Flag() { g_b16_trunk = saved; } } bf(precision >= 2 ? precision - 1 : 0);
    return dsmil_resnet_forward(depth, x, x_is_u8_nhwc, B, H, W, conv1_w, packed, bn_mean, bn_rstd, fc_w, fc_b, C, feats, classes,
                                ws, ws_bytes, stream);
}

int dsmil_resnet18bn_forward(const void* x, int32_t x_is_u8_nhwc, int32_t B, int32_t H, int32_t W,
                             const float* conv1_w, const float* packed, const float* bn_mean,
                             const float* bn_rstd, const float* fc_w, const float* fc_b, int32_t C,
                             float* feats, float* classes, void* ws, size_t ws_bytes, void* stream) {
    if (!bn_mean || !bn_rstd) return DSMIL_E_INVALID;
    return resnet18in_forward_impl(x, x_is_u8_nhwc != 0, B, H, W, conv1_w, packed, fc_w, fc_b, C, feats, classes, ws,
                                   ws_bytes, stream, bn_mean, bn_rstd);
}

}  // extern "C"

#ifdef DSMIL_TRACE
// trace builds only (not part of the ABI): copies the stamp buffer of the traced Winograd launch to the host
extern "C" int dsmil_debug_wino_trace(unsigned long long* out, int words) {
    if (!out || words < (int)WINO_TRACE_WORDS) return DSMIL_E_INVALID;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out, wino_trace_buffer(), WINO_TRACE_WORDS * 8, hipMemcpyDeviceToHost) == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}
#endif
