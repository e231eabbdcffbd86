// resnet_b16.h — the OPT-IN bf16-activation trunk of the patch embedder (dsmil_resnet_forward_ex, precision = 2;
// compute_feats.py:146-170 / dsmil.py:14-25 behind the stem).  EXPERIMENT builds only (-DDSMIL_EXPERIMENTS): included by resnet_fwd.hip.
//
// What it is for: BASELINE.md's "1 patch, bf16 MFMA / f32 accumulate" row.  The fp32-class trunk (precision 0) keeps fp32
// activations in HBM, cuts every operand into two fp16 planes and forms three plane products per MAC; its opt-in one-plane
// form (precision 1) drops two of the three products but keeps the fp32 activations, the cuts and the Winograd transforms —
// 1.2x.  This trunk stores the ACTIVATIONS in bf16 and multiplies them as they are: one v_mfma_f32_32x32x16_bf16 per
// 32 x 32 x 16 block, f32 accumulation, f32 InstanceNorm statistics.  It is NOT the 1e-4 parity path: features agree with the
// fp32 trunk to bf16 rounding (~1e-2 of the feature scale; tests/test_resnet_gpu.py states the bar).  BasicBlock trunks
// (ResNet-18 / 34) with InstanceNorm; everything else returns DSMIL_E_UNSUPPORTED.
//
// Data layout: every activation is bf16 NHWC with a ONE-PIXEL ZERO BORDER, [B][H + 2][W + 2][C], and a convolution works on
// the FLATTENED padded positions q = (n (H+2) + y) (W+2) + x:
//   * a 3 x 3 / stride-1 conv has the same padded grid on both sides, so tap (dy, dx) of output position q is input position
//     q + dy (W+2) + dx — no bounds logic, no im2col: a workgroup's 128 output positions need the 130 consecutive input
//     positions [q0 - 1, q0 + 129) of three input rows (dy = -1, 0, +1), each staged ONCE into LDS as a plain clamped copy and
//     used for the three dx taps by shifting the fragment address by one position;
//   * border positions are computed like any other and written as zeros, which is what keeps the border zero for the
//     next conv (the waste is (H+2)(W+2) / (H W): 7 % at 56 x 56, 65 % at 7 x 7);
//   * strided convs (3 x 3 / 2, 1 x 1 / 2) stage one tap at a time through per-position offsets.
// Kernels: k_b16_pad (fp32 NHWC stem output -> padded bf16), k_conv_b16 (implicit GEMM: M = positions, N = output channels,
// weights read from L2 in MFMA fragment order, activations through LDS with a 16-B pad per position: conflict-free
// ds_read_b128 fragments), k_stats_b16 / k_apply_b16 (InstanceNorm in two flat passes: per-(image, chunk, channel) partial
// sums in a fixed order — no atomics, bit-reproducible —, then normalise + residual + ReLU in place), k_pool_b16 (last
// block: normalise + residual + ReLU + average pool -> fp32 feature row).
#pragma once

namespace b16 {

constexpr int BM = 128;            // output positions per workgroup
constexpr int MAXC = 128;          // input channels per staged chunk

struct ConvGeo {
    int B, Hi, Wi, Cin, Ho, Wo, Cout, ks, stride;
    int cin_c;                     // channels per staged chunk: min(Cin, MAXC)
    long long M;                   // B (Ho+2) (Wo+2) output positions
    long long Min;                 // B (Hi+2) (Wi+2) input positions
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned f2bf(float f) {           // round to nearest even (no NaN handling: the inputs are finite)
    const unsigned u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return f2bf(lo) | (f2bf(hi) << 16); }
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// fp32 NHWC [B][H][W][C] (the stem's normalised, pooled output) -> bf16 padded [B][H+2][W+2][C]
__global__ __launch_bounds__(256) void k_b16_pad(const float* __restrict__ x, unsigned short* __restrict__ out, int B, int H, int W, int C) {
    const int oc = C >> 3;
    const long long total = (long long)B * (H + 2) * (W + 2) * oc;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int o = (int)(i % oc);
        const long long q = i / oc;
        const int xx = (int)(q % (W + 2));
        const long long r = q / (W + 2);
        const int yy = (int)(r % (H + 2));
        const int n = (int)(r / (H + 2));
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (yy >= 1 && yy <= H && xx >= 1 && xx <= W) {
            const float* s = x + (((long long)n * H + yy - 1) * W + xx - 1) * C + o * 8;
            const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
            v = u32x4_t{pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3])};
        }
        *reinterpret_cast<u32x4_t*>(out + q * C + o * 8) = v;
    }
}

// OIHW fp32 -> bf16 MFMA B-operand fragments: [chunk][tap][k-step of 16][32-cout block][lane][8]: lane l holds output channel
// 32 nb + (l & 31), input channels chunk cin_c + 16 ks + 8 (l >> 5) + e
__global__ void k_pack_b16(const float* __restrict__ w, unsigned short* __restrict__ out, int O, int I, int ks, int cin_c) {
    const int ntap = ks * ks, ksteps = cin_c / 16, nb_tot = O / 32, nchunk = I / cin_c;
    const long long total = (long long)nchunk * ntap * ksteps * nb_tot * 64 * 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        long long r = i >> 9;
        const int nb = (int)(r % nb_tot); r /= nb_tot;
        const int kst = (int)(r % ksteps); r /= ksteps;
        const int tap = (int)(r % ntap);
        const int chunk = (int)(r / ntap);
        const int co = nb * 32 + (lane & 31), ci = chunk * cin_c + kst * 16 + (lane >> 5) * 8 + e;
        out[i] = (unsigned short)f2bf(w[((long long)co * I + ci) * ntap + tap]);
    }
}

// ---- the convolution.  NT = output channels per workgroup (128 | 64), waves WM x WN over (positions, channels);
//      S1: 3 x 3 stride 1 (flat offsets, three dx taps per staged row window); else one tap per stage through offsets.
template <int NT, int WM, int WN, bool S1>
__global__ __launch_bounds__(256, 2) void k_conv_b16(const unsigned short* __restrict__ in, const unsigned short* __restrict__ wpk,
                                                     unsigned short* __restrict__ out, ConvGeo g) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int MBW = 4 / WM, NBW = (NT / 32) / WN;
    constexpr int NPOS = S1 ? BM + 2 : BM;                     // staged positions
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned char s_int[BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, hi = lane >> 5;
    const int cin_c = g.cin_c, PPL = cin_c >> 3, rowb = cin_c * 2 + 16, ksteps = cin_c >> 4;
    const long long q0 = (long long)blockIdx.x * BM;
    const int Wp = g.Wo + 2, Hp = g.Ho + 2, Wip = g.Wi + 2;
    const int ntap = g.ks * g.ks, nchunk = g.Cin / cin_c, nb_tot = g.Cout >> 5;
    const int nb0 = (int)blockIdx.y * (NT / 32) + wn * NBW;
    unsigned char* buf[2] = {smem, smem + NPOS * rowb};

    // interior flags of the workgroup's positions (a border position is written as zero)
    if (tid < BM) {
        const long long q = q0 + tid;
        const int r = (int)(q % ((long long)Hp * Wp)), yo = r / Wp, xo = r - yo * Wp;
        s_int[tid] = (q < g.M && yo >= 1 && yo <= g.Ho && xo >= 1 && xo <= g.Wo) ? 1 : 0;
    }
    // staging: piece (16 B) tid % PPL of positions tid / PPL + i (256 / PPL)
    constexpr int NI_MAX = (NPOS * (MAXC / 8) + 255) / 256;
    const int ppi = 256 / PPL, piece = tid % PPL;
    const int NI = (NPOS * PPL + 255) / 256;
    long long soff[NI_MAX];      // element offset of the staged position's centre tap (GEN) / flat position index (S1)
    int sdst[NI_MAX];
#pragma unroll
    for (int i = 0; i < NI_MAX; ++i) {
        int pos = i * ppi + tid / PPL;
        pos = pos < NPOS ? pos : NPOS - 1;                   // clamped duplicates write the same bytes
        sdst[i] = pos * rowb + piece * 16;
        if constexpr (S1) {
            soff[i] = q0 - 1 + pos;                          // + dy (W+2), clamped per stage
        } else {
            const long long q = q0 + pos;
            const long long n = q / ((long long)Hp * Wp);
            const int r = (int)(q - n * Hp * Wp), yo = r / Wp, xo = r - yo * Wp;
            const bool inside = q < g.M && yo >= 1 && yo <= g.Ho && xo >= 1 && xo <= g.Wo;
            const int yi = g.stride * (yo - 1) + 1, xi = g.stride * (xo - 1) + 1;
            soff[i] = inside ? ((n * (g.Hi + 2) + yi) * Wip + xi) * (long long)g.Cin : (long long)(Wip + 1) * g.Cin;
        }
    }
    // stage s: S1: (chunk, dy) — three taps; GEN: (chunk, tap)
    const int per_chunk = S1 ? 3 : ntap, nstage = nchunk * per_chunk;
    u32x4_t sreg[NI_MAX];
    auto stage_load = [&](int s) {
        const int chunk = s / per_chunk, t = s - chunk * per_chunk;
#pragma unroll
        for (int i = 0; i < NI_MAX; ++i) {
            if (i < NI) {
                long long eo;
                if constexpr (S1) {
                    long long p = soff[i] + (long long)(t - 1) * Wip;
                    p = p < 0 ? 0 : (p >= g.Min ? g.Min - 1 : p);
                    eo = p * g.Cin;
                } else {
                    const int dy = g.ks == 3 ? t / 3 - 1 : 0, dx = g.ks == 3 ? t % 3 - 1 : 0;
                    eo = soff[i] + (long long)(dy * Wip + dx) * g.Cin;
                }
                sreg[i] = *reinterpret_cast<const u32x4_t*>(in + eo + chunk * cin_c + piece * 8);
            }
        }
    };
    auto stage_store = [&](unsigned char* b) {
#pragma unroll
        for (int i = 0; i < NI_MAX; ++i)
            if (i < NI) *reinterpret_cast<u32x4_t*>(b + sdst[i]) = sreg[i];
    };

    f32x16 acc[MBW][NBW];
#pragma unroll
    for (int a = 0; a < MBW; ++a)
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const u32x4_t* wp4 = reinterpret_cast<const u32x4_t*>(wpk);
    stage_load(0);
    stage_store(buf[0]);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int sn = s + 1 < nstage ? s + 1 : s;             // (the last prefetch repeats the last stage: no conditional loads)
        stage_load(sn);
        const unsigned char* A = buf[s & 1];
        const int chunk = s / per_chunk, t = s - chunk * per_chunk;
        constexpr int NTAPS = S1 ? 3 : 1;
#pragma unroll
        for (int dxi = 0; dxi < NTAPS; ++dxi) {
            const int tap = S1 ? t * 3 + dxi : t;
            const u32x4_t* wt = wp4 + ((long long)(chunk * ntap + tap) * ksteps) * nb_tot * 64 + lane;
            const unsigned char* Ab = A + ((S1 ? dxi : 0) + wm * MBW * 32 + l31) * rowb + hi * 16;
            u32x4_t bf[2][NBW];
#pragma unroll
            for (int j = 0; j < NBW; ++j) bf[0][j] = wt[(long long)(nb0 + j) * 64];
            for (int ks = 0; ks < ksteps; ++ks) {
                const int kn = ks + 1 < ksteps ? ks + 1 : ks;
#pragma unroll
                for (int j = 0; j < NBW; ++j) bf[(ks + 1) & 1][j] = wt[((long long)kn * nb_tot + nb0 + j) * 64];
                u32x4_t af[MBW];
#pragma unroll
                for (int a = 0; a < MBW; ++a) af[a] = *reinterpret_cast<const u32x4_t*>(Ab + a * 32 * rowb + ks * 32);
#pragma unroll
                for (int a = 0; a < MBW; ++a)
#pragma unroll
                    for (int j = 0; j < NBW; ++j)
                        acc[a][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[a]),
                                                                            __builtin_bit_cast(bf16x8_t, bf[ks & 1][j]), acc[a][j], 0, 0, 0);
            }
        }
        stage_store(buf[(s + 1) & 1]);
        __syncthreads();
    }
    // epilogue: accumulator register i of lane l = position 8 (i / 4) + 4 (l >> 5) + (i & 3) of the block, channel l & 31
#pragma unroll
    for (int a = 0; a < MBW; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = (wm * MBW + a) * 32 + 8 * (i >> 2) + 4 * hi + (i & 3);
            const long long q = q0 + m;
            if (q < g.M) {
                const bool inside = s_int[m] != 0;
#pragma unroll
                for (int j = 0; j < NBW; ++j)
                    out[q * g.Cout + (nb0 + j) * 32 + l31] = inside ? (unsigned short)f2bf(acc[a][j][i]) : (unsigned short)0;
            }
        }
}

// ---- InstanceNorm statistics: partial (sum, sum of squares) per (image, pixel chunk, channel), f32, fixed order
__global__ __launch_bounds__(256) void k_stats_b16(const unsigned short* __restrict__ x, float* __restrict__ part, int H, int W, int C, int S) {
    __shared__ float sh[2][2048];
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int OC = C >> 3, PL = 256 / OC, o = tid % OC, pl = tid / OC;
    const int HW = H * W, per = (HW + S - 1) / S, p_lo = s * per, p_hi = (p_lo + per < HW) ? p_lo + per : HW;
    float sm[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sq[e] = 0.f; }
    for (int p = p_lo + pl; p < p_hi; p += PL) {
        const int y = p / W, xx = p - y * W;
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(x + (((long long)n * (H + 2) + y + 1) * (W + 2) + xx + 1) * C + o * 8);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float a = bf_lo(v[d]), b = bf_hi(v[d]);
            sm[2 * d] += a; sq[2 * d] = fmaf(a, a, sq[2 * d]);
            sm[2 * d + 1] += b; sq[2 * d + 1] = fmaf(b, b, sq[2 * d + 1]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[0][pl * C + o * 8 + e] = sm[e]; sh[1][pl * C + o * 8 + e] = sq[e]; }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < PL; ++k) { a += sh[0][k * C + c]; b += sh[1][k * C + c]; }
        float* dst = part + (((long long)n * S + s) * C + c) * 2;
        dst[0] = a;
        dst[1] = b;
    }
}

// mean / rstd of this thread's eight channels from the S partials (biased variance, eps 1e-5: nn.InstanceNorm2d)
__device__ __forceinline__ void b16_stats8(const float* __restrict__ part, int n, int S, int C, int c0, int HW, float (&mu)[8], float (&rs)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float a = 0.f, b = 0.f;
        for (int s = 0; s < S; ++s) {
            const float* p = part + (((long long)n * S + s) * C + c0 + e) * 2;
            a += p[0];
            b += p[1];
        }
        const float m = a / (float)HW;
        float var = b / (float)HW - m * m;
        var = var > 0.f ? var : 0.f;
        mu[e] = m;
        rs[e] = 1.0f / sqrtf(var + 1e-5f);
    }
}

// ---- y = [relu]( (x - mean) rstd [+ identity] ) on every padded position of an image (border -> 0), in place or not
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void k_apply_b16(const unsigned short* x, const unsigned short* __restrict__ idn,
                                                   unsigned short* y, const float* __restrict__ part, int H, int W, int C, int S) {   // (x may be y)
    const int n = blockIdx.y, tid = threadIdx.x;
    const int OC = C >> 3, PL = 256 / OC, o = tid % OC, pl = tid / OC;
    const int PP = (H + 2) * (W + 2), r0 = (int)blockIdx.x * PL * 8;
    float mu[8], rs[8];
    b16_stats8(part, n, S, C, o * 8, H * W, mu, rs);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int pos = r0 + it * PL + pl;
        if (pos >= PP) break;
        const int yy = pos / (W + 2), xx = pos - yy * (W + 2);
        const long long eo = ((long long)n * PP + pos) * C + o * 8;
        u32x4_t out = {0u, 0u, 0u, 0u};
        if (yy >= 1 && yy <= H && xx >= 1 && xx <= W) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(x + eo);
            u32x4_t iv = {0u, 0u, 0u, 0u};
            if constexpr (RES) iv = *reinterpret_cast<const u32x4_t*>(idn + eo);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float a = (bf_lo(v[d]) - mu[2 * d]) * rs[2 * d], b = (bf_hi(v[d]) - mu[2 * d + 1]) * rs[2 * d + 1];
                if constexpr (RES) { a += bf_lo(iv[d]); b += bf_hi(iv[d]); }
                if constexpr (RELU) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
                out[d] = pack2(a, b);
            }
        }
        *reinterpret_cast<u32x4_t*>(y + eo) = out;
    }
}

// ---- last block: feats[n][c] = mean over pixels of relu((x - mean) rstd + identity)   (dsmil.py:21-23's flatten(avgpool))
__global__ __launch_bounds__(256) void k_pool_b16(const unsigned short* __restrict__ x, const unsigned short* __restrict__ idn,
                                                  const float* __restrict__ part, float* __restrict__ feats, int H, int W, int C, int S) {
    const int n = blockIdx.y, c = (int)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int HW = H * W;
    float a = 0.f, b = 0.f;
    for (int s = 0; s < S; ++s) {
        const float* p = part + (((long long)n * S + s) * C + c) * 2;
        a += p[0];
        b += p[1];
    }
    const float m = a / (float)HW;
    float var = b / (float)HW - m * m;
    var = var > 0.f ? var : 0.f;
    const float r = 1.0f / sqrtf(var + 1e-5f);
    float acc = 0.f;
    for (int yy = 1; yy <= H; ++yy)
        for (int xx = 1; xx <= W; ++xx) {
            const long long eo = (((long long)n * (H + 2) + yy) * (W + 2) + xx) * C + c;
            const float v = (__uint_as_float((unsigned)x[eo] << 16) - m) * r + __uint_as_float((unsigned)idn[eo] << 16);
            acc += fmaxf(v, 0.f);
        }
    feats[(long long)n * C + c] = acc / (float)HW;
}

// ---- host side ------------------------------------------------------------------------------------------------------
inline int chunk_of(int cin) { return cin < MAXC ? cin : MAXC; }
inline size_t conv_packed_elems(const ConvSpec& s) { return (size_t)s.cout * s.cin * s.ks * s.ks; }   // bf16 elements

inline bool arch_ok(const Arch& A) { return !A.bottleneck; }
inline size_t packed_bytes(const Arch& A) {
    size_t e = 0;
    for (int i = 1; i < A.nconv; ++i) e += (conv_packed_elems(A.specs[i]) + 127) & ~(size_t)127;
    return e * 2;
}
inline size_t pack_off(const Arch& A, int ci) {   // bytes
    size_t e = 0;
    for (int i = 1; i < ci; ++i) e += (conv_packed_elems(A.specs[i]) + 127) & ~(size_t)127;
    return e * 2;
}
inline int pack_all(const Arch& A, const float* const* conv_w, unsigned short* dst, hipStream_t st) {
    for (int i = 1; i < A.nconv; ++i) {
        const ConvSpec& s = A.specs[i];
        if (s.cin % 16 || s.cout % 64) return DSMIL_E_UNSUPPORTED;
        hipLaunchKernelGGL(k_pack_b16, dim3(512), dim3(256), 0, st, conv_w[i], (unsigned short*)((char*)dst + pack_off(A, i)), s.cout, s.cin, s.ks, chunk_of(s.cin));
    }
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

inline int run_conv(hipStream_t st, const unsigned short* in, const unsigned short* wpk, unsigned short* out, int B, int Hi, int Wi, const ConvSpec& s, int* Ho_, int* Wo_) {
    ConvGeo g;
    g.B = B; g.Hi = Hi; g.Wi = Wi; g.Cin = s.cin; g.Cout = s.cout; g.ks = s.ks; g.stride = s.stride;
    g.Ho = (Hi + 2 * s.pad - s.ks) / s.stride + 1;
    g.Wo = (Wi + 2 * s.pad - s.ks) / s.stride + 1;
    g.cin_c = chunk_of(s.cin);
    g.M = (long long)B * (g.Ho + 2) * (g.Wo + 2);
    g.Min = (long long)B * (Hi + 2) * (Wi + 2);
    *Ho_ = g.Ho; *Wo_ = g.Wo;
    if ((s.ks != 3 && s.ks != 1) || (s.ks == 3 && s.pad != 1) || (s.ks == 1 && s.pad != 0) || s.cin % g.cin_c || g.cin_c % 16) return DSMIL_E_UNSUPPORTED;
    const bool s1 = s.ks == 3 && s.stride == 1;
    const size_t lds = (size_t)2 * (s1 ? BM + 2 : BM) * (g.cin_c * 2 + 16);
    const unsigned gx = (unsigned)((g.M + BM - 1) / BM);
    if (s.cout % 128 == 0) {
        if (s1) { allow_lds((const void*)k_conv_b16<128, 2, 2, true>, lds); hipLaunchKernelGGL((k_conv_b16<128, 2, 2, true>), dim3(gx, s.cout / 128), dim3(256), lds, st, in, wpk, out, g); }
        else { allow_lds((const void*)k_conv_b16<128, 2, 2, false>, lds); hipLaunchKernelGGL((k_conv_b16<128, 2, 2, false>), dim3(gx, s.cout / 128), dim3(256), lds, st, in, wpk, out, g); }
    } else {
        if (s1) { allow_lds((const void*)k_conv_b16<64, 4, 1, true>, lds); hipLaunchKernelGGL((k_conv_b16<64, 4, 1, true>), dim3(gx, s.cout / 64), dim3(256), lds, st, in, wpk, out, g); }
        else { allow_lds((const void*)k_conv_b16<64, 4, 1, false>, lds); hipLaunchKernelGGL((k_conv_b16<64, 4, 1, false>), dim3(gx, s.cout / 64), dim3(256), lds, st, in, wpk, out, g); }
    }
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

inline int stat_chunks(int HW) { return HW >= 2048 ? 8 : HW >= 512 ? 4 : HW >= 128 ? 2 : 1; }

inline int run_stats(hipStream_t st, const unsigned short* x, float* part, int B, int H, int W, int C) {
    const int S = stat_chunks(H * W);
    hipLaunchKernelGGL(k_stats_b16, dim3((unsigned)S, (unsigned)B), dim3(256), 0, st, x, part, H, W, C, S);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}
inline int run_apply(hipStream_t st, const unsigned short* x, const unsigned short* idn, unsigned short* y, const float* part, int B, int H, int W, int C, bool relu) {
    const int S = stat_chunks(H * W), PL = 256 / (C / 8), PP = (H + 2) * (W + 2);
    const dim3 grid((unsigned)((PP + PL * 8 - 1) / (PL * 8)), (unsigned)B);
    if (idn) hipLaunchKernelGGL((k_apply_b16<true, true>), grid, dim3(256), 0, st, x, idn, y, part, H, W, C, S);
    else if (relu) hipLaunchKernelGGL((k_apply_b16<false, true>), grid, dim3(256), 0, st, x, idn, y, part, H, W, C, S);
    else hipLaunchKernelGGL((k_apply_b16<false, false>), grid, dim3(256), 0, st, x, idn, y, part, H, W, C, S);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

// bytes the trunk needs behind the stem (four activation buffers of the largest padded map + the statistics partials)
inline size_t act_bytes(int B, int Hp, int Wp) { return al256((size_t)B * (Hp + 2) * (Wp + 2) * 64 * 2); }
inline size_t part_bytes(int B) { return al256((size_t)B * 8 * 512 * 2 * 4); }
inline size_t scratch_bytes(int B, int Hp, int Wp) { return 4 * act_bytes(B, Hp, Wp) + part_bytes(B); }

// The trunk behind the stem.  x0: the stem's normalised pooled output, fp32 NHWC [B][Hp][Wp][64]; scratch: scratch_bytes().
inline int trunk(hipStream_t st, const Arch& A, const float* x0, const unsigned short* wpk, void* scratch, int B, int Hp, int Wp, float* feats) {
    char* s8 = (char*)scratch;
    const size_t ab = act_bytes(B, Hp, Wp);
    unsigned short* bufs[4];
    for (int i = 0; i < 4; ++i) bufs[i] = (unsigned short*)(s8 + i * ab);
    float* part = (float*)(s8 + 4 * ab);
    unsigned short* cur = bufs[0];
    unsigned short* r1 = bufs[1];
    unsigned short* r2 = bufs[2];
    unsigned short* rd = bufs[3];
    {
        const long long total = (long long)B * (Hp + 2) * (Wp + 2) * 8;
        long long blocks = (total + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(k_b16_pad, dim3((unsigned)blocks), dim3(256), 0, st, x0, cur, B, Hp, Wp, 64);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    }
    int ci = 1, Hc = Hp, Wc = Wp;
    for (int l = 1; l <= 4; ++l) {
        for (int b = 0; b < A.nblk[l - 1]; ++b) {
            const bool last = (l == 4 && b == A.nblk[3] - 1), down = l > 1 && b == 0;
            const ConvSpec& s1 = A.specs[ci];
            const ConvSpec& s2 = A.specs[ci + 1];
            int Ho, Wo, H2, W2, rc;
            // conv1 -> IN -> ReLU (in place)
            if ((rc = run_conv(st, cur, (const unsigned short*)((const char*)wpk + pack_off(A, ci)), r1, B, Hc, Wc, s1, &Ho, &Wo))) return rc;
            if ((rc = run_stats(st, r1, part, B, Ho, Wo, s1.cout))) return rc;
            if ((rc = run_apply(st, r1, nullptr, r1, part, B, Ho, Wo, s1.cout, true))) return rc;
            // downsample branch: 1x1 stride 2 -> IN (no ReLU), in place
            const unsigned short* idn = cur;
            if (down) {
                const ConvSpec& sd = A.specs[ci + 2];
                int Hd, Wd;
                if ((rc = run_conv(st, cur, (const unsigned short*)((const char*)wpk + pack_off(A, ci + 2)), rd, B, Hc, Wc, sd, &Hd, &Wd))) return rc;
                if (Hd != Ho || Wd != Wo) return DSMIL_E_UNSUPPORTED;
                if ((rc = run_stats(st, rd, part, B, Hd, Wd, sd.cout))) return rc;
                if ((rc = run_apply(st, rd, nullptr, rd, part, B, Hd, Wd, sd.cout, false))) return rc;
                idn = rd;
            }
            // conv2 -> IN, + identity, ReLU
            if ((rc = run_conv(st, r1, (const unsigned short*)((const char*)wpk + pack_off(A, ci + 1)), r2, B, Ho, Wo, s2, &H2, &W2))) return rc;
            if ((rc = run_stats(st, r2, part, B, H2, W2, s2.cout))) return rc;
            if (last) {
                const int S = stat_chunks(H2 * W2);
                hipLaunchKernelGGL(k_pool_b16, dim3((unsigned)((s2.cout + 255) / 256), (unsigned)B), dim3(256), 0, st, r2, idn, part, feats, H2, W2, s2.cout, S);
                if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
            } else {
                if ((rc = run_apply(st, r2, idn, r2, part, B, H2, W2, s2.cout, true))) return rc;
                unsigned short* t = cur; cur = r2; r2 = t;     // the block's output becomes the next input; its old input is free
            }
            ci += down ? 3 : 2;
            Hc = H2; Wc = W2;
        }
    }
    return DSMIL_OK;
}

}  // namespace b16
