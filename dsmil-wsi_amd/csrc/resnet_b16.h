// resnet_b16.h — the OPT-IN bf16-activation trunk of the patch embedder (dsmil_resnet_forward_ex, precision = 2;
// compute_feats.py:146-170 / dsmil.py:14-25 behind the stem).  Included by resnet_fwd.hip.
//
// What it is for: BASELINE.md's "1 patch, bf16 MFMA / f32 accumulate" row.  The fp32-class trunk (precision 0) keeps fp32
// activations in HBM, cuts every operand into two fp16 planes and forms three plane products per MAC; its opt-in one-plane
// form (precision 1) drops two of the three products but keeps the fp32 activations, the cuts and the Winograd transforms —
// 1.2x.  This trunk stores the ACTIVATIONS in bf16 and multiplies them as they are: one v_mfma_f32_32x32x16_bf16 per
// 32 x 32 x 16 block, f32 accumulation, f32 InstanceNorm statistics: 1.55-1.6x the fp32-class trunk.  It is NOT the 1e-4 parity
// path: features agree with the fp32 trunk to bf16 rounding (max 2e-2, mean 3e-3 of features of magnitude ~1;
// tests/test_resnet_gpu.py states the bar).  BasicBlock trunks (ResNet-18 / 34) with InstanceNorm; everything else returns
// DSMIL_E_UNSUPPORTED.
//
// Data layout: every activation is bf16 NHWC with SHARED zero borders: image n owns rows n (H+1) .. n (H+1) + H of a flat
// [rows][W + 1][C] array — its row 0 is zero (the row above the image AND the row below the image in front), column W of
// every row is zero (the pixel right of a row AND the pixel left of the next one) — and one more zero row closes the last
// image: (H+1)(W+1) positions per image instead of (H+2)(W+2) with private borders (31 % border work at 7 x 7 instead of
// 65 %).  A convolution works on the FLATTENED positions q = (n (H+1) + y)(W+1) + x:
//   * a 3 x 3 / stride-1 conv has the same grid on both sides, so tap (dy, dx) of output position q is input position
//     q + dy (W+1) + dx — no bounds logic, no im2col: a workgroup's BM output positions need the BM + 2 consecutive input
//     positions [q0 - 1, q0 + BM + 1) of three input rows (dy = -1, 0, +1), each staged ONCE into LDS as a plain clamped copy
//     and used for the three dx taps by shifting the fragment address by one position;
//   * border positions are computed like any other and written as zeros, which is what keeps the borders zero for the
//     next conv;
//   * strided convs (3 x 3 / 2, 1 x 1 / 2) stage one tap at a time through per-position offsets.
// Kernels: k_b16_pad / k_b16_borders (the stem's output into the layout); k_conv_b16w / k_conv_b16g (implicit GEMM, M =
// positions, N = output channels: BOTH operands go through LDS — activations with a 16-B pad per position, weights in MFMA
// fragment order, both conflict-free ds_read_b128 —, staged one stage ahead by plain loads whose only wait is the ds_write at
// the top of the next stage, so the in-order vmcnt queue never couples a fragment read to a staging load; 128 x 64 wave tiles
// (6 fragment reads per 8 MFMAs); ONE LDS buffer and TWO workgroups per CU: one multiplies while the other stores its next
// stage and sits at its barriers; the 3 x 3 / 1 form stages the workgroup's whole input WINDOW once per 16-channel chunk and
// runs all nine taps on it; k_conv_b16n: the 64-channel layer with the window's 64 channels staged once per workgroup); k_stats_b16 / k_apply_b16 (InstanceNorm in two flat passes: per-(image, chunk, channel) partial sums
// in a fixed order — no atomics, bit-reproducible —, then normalise + residual + ReLU in place); k_pool_b16 (last block:
// normalise + residual + ReLU + average pool -> fp32 feature row).
// History of the conv kernel (bs 256, per 3 x 3 / 1 conv at 128+ channels; DESIGN.md §4 "Round 6"): 64 x 64 wave tiles with the
// weight fragments loaded per lane from L2 inside the loop: 306 us (the fragment loads shared the vmcnt queue with the
// staging loads: every stage began with an HBM round trip); both operands through LDS, 128 x 64 wave tiles, stage = (32
// channels, one tap row), one workgroup per CU with two LDS buffers: 142 us; one buffer, two workgroups per CU: 96 us; shared
// borders, stage = (16 channels, all nine taps on one staged window): 70-77 us at 14 x 14 / 28 x 28.  What holds it now: layer 1 (64 channels) is
// memory-bound (110 MB in, 110 MB out per conv: 45 us of 112), a 7 x 7 conv is a chain of 48 stages whose staging loads have
// one stage's MFMAs to cross a memory round trip, and two thirds of a stage are not MFMA time.  Private borders
// ([H+2][W+2] per image) -> shared borders: 111.8 k -> 120.4 k patches/s on three streams.
#pragma once

namespace b16 {


struct ConvGeo {
    int B, Hi, Wi, Cin, Ho, Wo, Cout, ks, stride;
    int cin_c;                     // channels per staged chunk (32: stride-1 form, 64: strided form)
    long long M;                   // output positions: B (Ho+1) (Wo+1) + (Wo+1)
    long long Mint;                // B (Ho+1) (Wo+1): the positions that belong to an image
    long long Min;                 // input positions
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int B16N_NIA = 16;    // k_conv_b16n: (256 + 2 (W+1) + 2) positions x 8 pieces <= 16 x 256  (maps up to ~126 pixels wide)
constexpr int B16W_NIA = 6;     // k_conv_b16w: 16-B window pieces per thread: (BM + 2 (W+1) + 2) positions x 2 pieces <= 6 x 256

// positions of a [B][H][W] map in the shared-border layout (see the header): (H+1)(W+1) per image + one closing zero row
__host__ __device__ __forceinline__ long long npos(int B, int H, int W) { return (long long)B * (H + 1) * (W + 1) + (W + 1); }

// 16-bit element type of the trunk: bf16 (F16 = false: fp32's range, 8 significant bits) or fp16 (F16 = true: 11 significant bits,
// values must stay inside +-65504 — InstanceNorm outputs and the conv sums of ordinary weights do; the binding checks the first
// forward of a weight set for non-finite features).  Same kernels, same layout, same MFMA rate.
template <bool F16>
__device__ __forceinline__ unsigned cvt16(float f) {          // round to nearest even (no NaN handling: the inputs are finite)
    if constexpr (F16) {
        const _Float16 h = (_Float16)f;
        return (unsigned)__builtin_bit_cast(unsigned short, h);
    } else {
        const unsigned u = __float_as_uint(f);
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    }
}
template <bool F16> __device__ __forceinline__ unsigned pack2(float lo, float hi) { return cvt16<F16>(lo) | (cvt16<F16>(hi) << 16); }
template <bool F16> __device__ __forceinline__ float el_lo(unsigned w) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
    else return __uint_as_float(w << 16);
}
template <bool F16> __device__ __forceinline__ float el_hi(unsigned w) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
    else return __uint_as_float(w & 0xffff0000u);
}
template <bool F16> __device__ __forceinline__ f32x16 mfma16(const u32x4_t& a, const u32x4_t& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// fp32 NHWC [B][H][W][C] (the stem's normalised, pooled output) -> the bf16 shared-border layout
template <bool F16>
__global__ __launch_bounds__(256) void k_b16_pad(const float* __restrict__ x, unsigned short* __restrict__ out, int B, int H, int W, int C) {
    const int oc = C >> 3;
    const long long total = npos(B, H, W) * oc;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int o = (int)(i % oc);
        const long long q = i / oc;
        const int xx = (int)(q % (W + 1));
        const long long r = q / (W + 1);
        const int yy = (int)(r % (H + 1));
        const int n = (int)(r / (H + 1));
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (n < B && yy >= 1 && xx < W) {
            const float* s = x + (((long long)n * H + yy - 1) * W + xx) * C + o * 8;
            const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
            v = u32x4_t{pack2<F16>(a[0], a[1]), pack2<F16>(a[2], a[3]), pack2<F16>(b[0], b[1]), pack2<F16>(b[2], b[3])};
        }
        *reinterpret_cast<u32x4_t*>(out + q * C + o * 8) = v;
    }
}

// the zeros of the shared-border layout around data another kernel wrote (k_pool_fix_norm): row 0 of every image, column W of
// every row, the closing row
__global__ __launch_bounds__(256) void k_b16_borders(unsigned short* __restrict__ out, int B, int H, int W, int C) {
    const int oc = C >> 3, per_img = (W + 1) + H;                 // row 0 (W + 1 positions) + column W of rows 1..H
    const long long total = ((long long)B * per_img + (W + 1)) * oc;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int o = (int)(i % oc);
        const long long b = i / oc;
        long long q;
        if (b >= (long long)B * per_img) {
            q = (long long)B * (H + 1) * (W + 1) + (b - (long long)B * per_img);          // the closing row
        } else {
            const int n = (int)(b / per_img), k = (int)(b - (long long)n * per_img);
            q = k <= W ? (long long)n * (H + 1) * (W + 1) + k                                    // row 0
                       : ((long long)n * (H + 1) + (k - W)) * (W + 1) + W;                          // column W of row k - W
        }
        *reinterpret_cast<u32x4_t*>(out + q * C + o * 8) = u32x4_t{0u, 0u, 0u, 0u};
    }
}

// OIHW fp32 -> bf16 MFMA B-operand fragments: [chunk][tap][k-step of 16][32-cout block][lane][8]: lane l holds output channel
// 32 nb + (l & 31), input channels chunk cin_c + 16 ks + 8 (l >> 5) + e
template <bool F16>
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
        out[i] = (unsigned short)cvt16<F16>(w[((long long)co * I + ci) * ntap + tap]);
    }
}

// ---- the 3 x 3 / stride-1 convolution, WINDOW form (k_conv_b16w): stage = a 16-channel chunk with ALL NINE taps.  The
//      workgroup's whole input window — the BMv + 2 (W+1) + 2 consecutive positions [q0 - (W+1) - 1, q0 + BMv + (W+1) + 1) —
//      is staged ONCE per chunk (48 B per position: 32 B + 16 B pad) instead of once per (chunk, dy): every input byte crosses
//      the CU's port ~1.2-1.5x instead of 3x, a conv has Cin / 16 stages instead of 3 Cin / 32, and a stage holds 72 MFMAs per
//      wave instead of 48 for its staging loads to land under.  Tap (dy, dx) of output m is window position m + dy (W+1) + dx.
//      Weight image: k_pack_b16 with cin_c = 16 ([chunk16][tap][1][Cout / 32][64][8]).
template <int BMv, int NT, int WM, int WN, bool F16>
__global__ __launch_bounds__(256, 2) void k_conv_b16w(const unsigned short* __restrict__ in, const unsigned short* __restrict__ wpk,
                                                      unsigned short* __restrict__ out, ConvGeo g) {
    static_assert(WM * WN == 4 && BMv == WM * 128 && NT == WN * 64, "wave tile 128 positions x 64 channels");
    constexpr int ROWB = 48;
    constexpr int B_PIECES = 9 * (NT / 32) * 64, B_BYTES = B_PIECES * 16;
    constexpr int NIB = (B_PIECES + 255) / 256;
    constexpr int NIA = B16W_NIA;                             // window pieces per thread (the host checks that they suffice)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned char s_int[BMv];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, hi = lane >> 5;
    const long long q0 = (long long)blockIdx.x * BMv;
    const int Wp = g.Wo + 1, Hp = g.Ho + 1;
    const int npw = BMv + 2 * Wp + 2, A_BYTES = (npw * ROWB + 15) / 16 * 16;
    const int nstage = g.Cin >> 4, nb_tot = g.Cout >> 5, nb_wg = (int)blockIdx.y * (NT / 32);
    for (int m = tid; m < BMv; m += 256) {
        const long long q = q0 + m;
        const int r = (int)(q % ((long long)Hp * Wp)), yo = r / Wp, xo = r - yo * Wp;
        s_int[m] = (q < g.Mint && yo >= 1 && xo < g.Wo) ? 1 : 0;
    }
    long long aoff[NIA];                 // element offset of the window piece (clamped into the tensor: the same for every stage)
    int adst[NIA];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        int p = i * 256 + tid;
        p = p < npw * 2 ? p : npw * 2 - 1;                    // clamped duplicates write the same bytes
        const int pos = p >> 1, piece = p & 1;
        long long q = q0 - Wp - 1 + pos;
        q = q < 0 ? 0 : (q >= g.Min ? g.Min - 1 : q);
        aoff[i] = q * g.Cin + piece * 8;
        adst[i] = pos * ROWB + piece * 16;
    }
    const u32x4_t* wp4 = reinterpret_cast<const u32x4_t*>(wpk);
    u32x4_t ra[NIA], rb[NIB];
    auto stage_load = [&](int s) {
#pragma unroll
        for (int i = 0; i < NIA; ++i) ra[i] = *reinterpret_cast<const u32x4_t*>(in + aoff[i] + s * 16);
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            int pidx = i * 256 + tid;
            pidx = pidx < B_PIECES ? pidx : B_PIECES - 1;
            constexpr int per = (NT / 32) * 64;
            const int tap = pidx / per, rest = pidx - tap * per;
            rb[i] = wp4[((long long)(s * 9 + tap) * nb_tot + nb_wg) * 64 + rest];
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int i = 0; i < NIA; ++i) *reinterpret_cast<u32x4_t*>(smem + adst[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            int pidx = i * 256 + tid;
            pidx = pidx < B_PIECES ? pidx : B_PIECES - 1;
            *reinterpret_cast<u32x4_t*>(smem + A_BYTES + pidx * 16) = rb[i];
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    stage_load(0);
    for (int s = 0; s < nstage; ++s) {
        __syncthreads();                                       // the fragments of stage s - 1 have been read by every wave
        stage_store();
        __syncthreads();
        stage_load(s + 1 < nstage ? s + 1 : s);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const unsigned char* Ab = smem + (dy * Wp + dx + wm * 128 + l31) * ROWB + hi * 16;
            const unsigned char* Bb = smem + A_BYTES + ((tap * (NT / 32) + wn * 2) * 64 + lane) * 16;
            u32x4_t af[4], bf[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = *reinterpret_cast<const u32x4_t*>(Ab + a * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const u32x4_t*>(Bb + j * 1024);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[a][j] = mfma16<F16>(af[a], bf[j], acc[a][j]);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = wm * 128 + a * 32 + 8 * (i >> 2) + 4 * hi + (i & 3);
            const long long q = q0 + m;
            if (q < g.M) {
                const bool inside = s_int[m] != 0;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    out[q * g.Cout + (nb_wg + wn * 2 + j) * 32 + l31] = inside ? (unsigned short)cvt16<F16>(acc[a][j][i]) : (unsigned short)0;
            }
        }
}

// ---- the window form for NARROW layers (Cin = Cout = 64: layer 1): the whole window with ALL 64 input channels is staged ONCE
//      per workgroup (144 B per position: 128 B + 16 B pad) and only the weights change per 16-channel stage.  k_conv_b16w's
//      16-channel stages touch every 128-B position row once each, four times per conv, and the rows do not stay in L2 between
//      them (504 MB of fabric traffic per conv against 212 MB algorithmic, profiles/r06_emb16).  256 positions x 64 channels per
//      workgroup, 64 x 64 wave tiles (four fragment reads per four MFMAs), two workgroups per CU (one loads its window while
//      the other multiplies).  Weight image as for k_conv_b16w (cin_c = 16).
template <bool F16>
__global__ __launch_bounds__(256, 2) void k_conv_b16n(const unsigned short* __restrict__ in, const unsigned short* __restrict__ wpk,
                                                      unsigned short* __restrict__ out, ConvGeo g) {
    constexpr int BMv = 256, NT = 64, CIN = 64, ROWB = CIN * 2 + 16;
    constexpr int B_PIECES = 9 * (NT / 32) * 64, B_BYTES = B_PIECES * 16;       // one 16-channel stage: 18 KiB
    constexpr int NIB = (B_PIECES + 255) / 256;
    constexpr int NIA = B16N_NIA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned char s_int[BMv];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;   // wave tile: positions 64 (wm) x channels 32 (wn) ... x2 blocks each way
    const long long q0 = (long long)blockIdx.x * BMv;
    const int Wp = g.Wo + 1, Hp = g.Ho + 1;
    const int npw = BMv + 2 * Wp + 2, A_BYTES = (npw * ROWB + 15) / 16 * 16;
    const int nb_tot = g.Cout >> 5;
    {
        const long long q = q0 + tid;
        const int r = (int)(q % ((long long)Hp * Wp)), yo = r / Wp, xo = r - yo * Wp;
        s_int[tid] = (q < g.Mint && yo >= 1 && xo < g.Wo) ? 1 : 0;
    }
    // the window, all 64 channels: once
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        int p = i * 256 + tid;
        p = p < npw * 8 ? p : npw * 8 - 1;
        const int pos = p >> 3, piece = p & 7;
        long long q = q0 - Wp - 1 + pos;
        q = q < 0 ? 0 : (q >= g.Min ? g.Min - 1 : q);
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(in + q * CIN + piece * 8);
        *reinterpret_cast<u32x4_t*>(smem + pos * ROWB + piece * 16) = v;
    }
    const u32x4_t* wp4 = reinterpret_cast<const u32x4_t*>(wpk);
    u32x4_t rb[NIB];
    auto w_load = [&](int s) {
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            int pidx = i * 256 + tid;
            pidx = pidx < B_PIECES ? pidx : B_PIECES - 1;
            constexpr int per = (NT / 32) * 64;
            const int tap = pidx / per, rest = pidx - tap * per;
            rb[i] = wp4[((long long)(s * 9 + tap) * nb_tot) * 64 + rest];
        }
    };
    auto w_store = [&]() {
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            int pidx = i * 256 + tid;
            pidx = pidx < B_PIECES ? pidx : B_PIECES - 1;
            *reinterpret_cast<u32x4_t*>(smem + A_BYTES + pidx * 16) = rb[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    w_load(0);
    constexpr int NSTAGE = CIN / 16;
    for (int s = 0; s < NSTAGE; ++s) {
        __syncthreads();                                       // (first pass: the window is written; later: stage s - 1 has been read)
        w_store();
        __syncthreads();
        w_load(s + 1 < NSTAGE ? s + 1 : s);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const unsigned char* Ab = smem + (dy * Wp + dx + wm * 128 + l31) * ROWB + s * 32 + hi * 16;
            const unsigned char* Bb = smem + A_BYTES + ((tap * (NT / 32)) * 64 + lane) * 16;
            u32x4_t af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const u32x4_t*>(Ab + (wn * 64 + a * 32) * ROWB);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const u32x4_t*>(Bb + j * 1024);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][j] = mfma16<F16>(af[a], bf[j], acc[a][j]);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = wm * 128 + wn * 64 + a * 32 + 8 * (i >> 2) + 4 * hi + (i & 3);
            const long long q = q0 + m;
            if (q < g.M) {
                const bool inside = s_int[m] != 0;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    out[q * g.Cout + j * 32 + l31] = inside ? (unsigned short)cvt16<F16>(acc[a][j][i]) : (unsigned short)0;
            }
        }
}

// ---- strided convolutions (3 x 3 / 2, 1 x 1 / 2) in the same two-workgroups-per-CU form: stage = (64-channel chunk, tap), the
//      256 positions of the workgroup gathered through per-position offsets (144 B per position: 128 B + 16 B pad), the tap's
//      weight fragments of 128 output channels; 128 x 64 wave tiles.  Weight image: k_pack_b16 with cin_c = 64.
template <bool F16>
__global__ __launch_bounds__(256, 2) void k_conv_b16g(const unsigned short* __restrict__ in, const unsigned short* __restrict__ wpk,
                                                      unsigned short* __restrict__ out, ConvGeo g) {
    constexpr int BMv = 256, NT = 128, ROWB = 144, A_BYTES = BMv * ROWB, NIA = 8, NIB = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned char s_int[BMv];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const long long q0 = (long long)blockIdx.x * BMv;
    const int Wp = g.Wo + 1, Hp = g.Ho + 1, Wip = g.Wi + 1;
    const int ntap = g.ks * g.ks, nchunk = g.Cin >> 6, nb_tot = g.Cout >> 5, nb_wg = (int)blockIdx.y * (NT / 32);
    {
        const long long q = q0 + tid;
        const int r = (int)(q % ((long long)Hp * Wp)), yo = r / Wp, xo = r - yo * Wp;
        s_int[tid] = (q < g.Mint && yo >= 1 && xo < g.Wo) ? 1 : 0;
    }
    int soff[NIA];                                   // element offset of the position's centre tap + this thread's piece
    const int piece = tid & 7;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const int pos = i * 32 + (tid >> 3);
        const long long q = q0 + pos;
        const long long n = q / ((long long)Hp * Wp);
        const int r = (int)(q - n * Hp * Wp), yo = r / Wp, xo = r - yo * Wp;
        const bool inside = q < g.Mint && yo >= 1 && xo < g.Wo;
        const int yi = g.stride * (yo - 1) + 1, xi = g.stride * xo;        // rows 1-based (row 0 of an image is zero), columns 0-based
        soff[i] = (inside ? (int)(((n * (g.Hi + 1) + yi) * Wip + xi) * g.Cin) : (Wip + 1) * g.Cin) + piece * 8;
    }
    const u32x4_t* wp4 = reinterpret_cast<const u32x4_t*>(wpk);
    u32x4_t ra[NIA], rb[NIB];
    const int nstage = nchunk * ntap;
    auto stage_load = [&](int s) {
        const int chunk = s / ntap, t = s - chunk * ntap;
        const int dy = g.ks == 3 ? t / 3 - 1 : 0, dx = g.ks == 3 ? t % 3 - 1 : 0;
        const int eo = (dy * Wip + dx) * g.Cin + chunk * 64;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            // (image 0, output (1, 0), tap (-1, -1) is position -1: the zero at the end of the row in front of the tensor's first —
            // read position 0 instead, another zero)
            const int e = soff[i] + eo;
            ra[i] = *reinterpret_cast<const u32x4_t*>(in + (e < 0 ? piece * 8 : e));
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) rb[i] = wp4[(((long long)(chunk * ntap + t) * 4 + i) * nb_tot + nb_wg) * 64 + tid];
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int i = 0; i < NIA; ++i) *reinterpret_cast<u32x4_t*>(smem + (i * 32 + (tid >> 3)) * ROWB + piece * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < NIB; ++i) *reinterpret_cast<u32x4_t*>(smem + A_BYTES + (i * 256 + tid) * 16) = rb[i];
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    stage_load(0);
    for (int s = 0; s < nstage; ++s) {
        __syncthreads();
        stage_store();
        __syncthreads();
        stage_load(s + 1 < nstage ? s + 1 : s);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned char* Ab = smem + (wm * 128 + l31) * ROWB + ks * 32 + hi * 16;
            const unsigned char* Bb = smem + A_BYTES + ((ks * (NT / 32) + wn * 2) * 64 + lane) * 16;
            u32x4_t af[4], bf[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = *reinterpret_cast<const u32x4_t*>(Ab + a * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const u32x4_t*>(Bb + j * 1024);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[a][j] = mfma16<F16>(af[a], bf[j], acc[a][j]);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = wm * 128 + a * 32 + 8 * (i >> 2) + 4 * hi + (i & 3);
            const long long q = q0 + m;
            if (q < g.M) {   // (two channels per store through a lane exchange: measured 6-12 % SLOWER than these 2-byte stores)
                const bool inside = s_int[m] != 0;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    out[q * g.Cout + (nb_wg + wn * 2 + j) * 32 + l31] = inside ? (unsigned short)cvt16<F16>(acc[a][j][i]) : (unsigned short)0;
            }
        }
}

// ---- InstanceNorm statistics: partial (sum, sum of squares) per (image, pixel chunk, channel), f32, fixed order
template <bool F16>
__global__ __launch_bounds__(256) void k_stats_b16(const unsigned short* __restrict__ x, float* __restrict__ part, int H, int W, int C, int S) {
    __shared__ float sh[2][2048];
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int OC = C >> 3, PL = 256 / OC, o = tid % OC, pl = tid / OC;
    const int HW = H * W, per = (HW + S - 1) / S, p_lo = s * per, p_hi = (p_lo + per < HW) ? p_lo + per : HW;
    float sm[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sq[e] = 0.f; }
    // four pixels per step, their loads in front of the adds (clamped re-reads past the end are not added): same order of adds
    for (int p0 = p_lo + pl; p0 < p_hi; p0 += 4 * PL) {
        u32x4_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int p = p0 + k * PL;
            p = p < p_hi ? p : p_hi - 1;
            const int y = p / W, xx = p - y * W;
            v[k] = *reinterpret_cast<const u32x4_t*>(x + (((long long)n * (H + 1) + y + 1) * (W + 1) + xx) * C + o * 8);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (p0 + k * PL < p_hi) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float a = el_lo<F16>(v[k][d]), b = el_hi<F16>(v[k][d]);
                    sm[2 * d] += a; sq[2 * d] = fmaf(a, a, sq[2 * d]);
                    sm[2 * d + 1] += b; sq[2 * d + 1] = fmaf(b, b, sq[2 * d + 1]);
                }
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

// mean / rstd of this thread's eight channels from the S partials (biased variance, eps 1e-5: nn.InstanceNorm2d).  The eight
// channels' (sum, sum of squares) pairs of one partial are 64 contiguous bytes: four 16-B loads per partial, all S partials
// requested before the first add (c0 is a multiple of 8: 16-B aligned)
__device__ __forceinline__ void b16_stats8(const float* __restrict__ part, int n, int S, int C, int c0, int HW, float (&mu)[8], float (&rs)[8]) {
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.f; b[e] = 0.f; }
    for (int s0 = 0; s0 < S; s0 += 4) {          // S is 1, 2, 4 or 8
        f32x4 v[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = s0 + k < S ? s0 + k : S - 1;         // (clamped re-read, not added)
            const f32x4* p = reinterpret_cast<const f32x4*>(part + (((long long)n * S + s) * C + c0) * 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[k][q] = p[q];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (s0 + k < S) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[2 * q] += v[k][q][0]; b[2 * q] += v[k][q][1];
                    a[2 * q + 1] += v[k][q][2]; b[2 * q + 1] += v[k][q][3];
                }
            }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float m = a[e] / (float)HW;
        float var = b[e] / (float)HW - m * m;
        var = var > 0.f ? var : 0.f;
        mu[e] = m;
        rs[e] = 1.0f / sqrtf(var + 1e-5f);
    }
}

// ---- y = [relu]( (x - mean) rstd [+ identity] ) on every padded position of an image (border -> 0), in place or not
template <bool RES, bool RELU, bool F16>
__global__ __launch_bounds__(256) void k_apply_b16(const unsigned short* x, const unsigned short* __restrict__ idn,
                                                   unsigned short* y, const float* __restrict__ part, int H, int W, int C, int S) {   // (x may be y)
    const int n = blockIdx.y, tid = threadIdx.x;
    const int OC = C >> 3, PL = 256 / OC, o = tid % OC, pl = tid / OC;
    const int PP = (H + 1) * (W + 1), r0 = (int)blockIdx.x * PL * 8;
    float mu[8], rs[8];
    b16_stats8(part, n, S, C, o * 8, H * W, mu, rs);
    // all eight positions' loads go out before the first use (clamped, unconditional: one round trip per workgroup instead of
    // eight dependent ones), then eight normalise / store groups
    u32x4_t v[8], iv[8];
    long long eo[8];
    bool live[8], inside[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        int pos = r0 + it * PL + pl;
        live[it] = pos < PP;
        pos = live[it] ? pos : PP - 1;
        const int yy = pos / (W + 1), xx = pos - yy * (W + 1);
        inside[it] = yy >= 1 && xx < W;
        eo[it] = ((long long)n * PP + pos) * C + o * 8;
        v[it] = *reinterpret_cast<const u32x4_t*>(x + eo[it]);
        if constexpr (RES) iv[it] = *reinterpret_cast<const u32x4_t*>(idn + eo[it]);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        u32x4_t out = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float a = (el_lo<F16>(v[it][d]) - mu[2 * d]) * rs[2 * d], b = (el_hi<F16>(v[it][d]) - mu[2 * d + 1]) * rs[2 * d + 1];
            if constexpr (RES) { a += el_lo<F16>(iv[it][d]); b += el_hi<F16>(iv[it][d]); }
            if constexpr (RELU) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
            out[d] = inside[it] ? pack2<F16>(a, b) : 0u;
        }
        if (live[it]) *reinterpret_cast<u32x4_t*>(y + eo[it]) = out;
    }
}

// ---- last block: feats[n][c] = mean over pixels of relu((x - mean) rstd + identity)   (dsmil.py:21-23's flatten(avgpool))
template <bool F16>
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
        for (int xx = 0; xx < W; ++xx) {
            const long long eo = (((long long)n * (H + 1) + yy) * (W + 1) + xx) * C + c;
            const float v = (el_lo<F16>((unsigned)x[eo]) - m) * r + el_lo<F16>((unsigned)idn[eo]);
            acc += fmaxf(v, 0.f);
        }
    feats[(long long)n * C + c] = acc / (float)HW;
}

// ---- host side ------------------------------------------------------------------------------------------------------
inline bool v2_conv(const ConvSpec& s) { return s.ks == 3 && s.stride == 1 && s.pad == 1 && s.cin % 32 == 0 && s.cout % 64 == 0; }
inline bool g2_conv(const ConvSpec& s) { return !v2_conv(s) && s.cin % 64 == 0 && s.cout % 128 == 0; }
inline int chunk_for(const ConvSpec& s) { return v2_conv(s) ? 16 : 64; }
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
inline int pack_all(const Arch& A, const float* const* conv_w, unsigned short* dst, hipStream_t st, bool f16) {
    for (int i = 1; i < A.nconv; ++i) {
        const ConvSpec& s = A.specs[i];
        if (!v2_conv(s) && !g2_conv(s)) return DSMIL_E_UNSUPPORTED;
        if (f16) hipLaunchKernelGGL(k_pack_b16<true>, dim3(512), dim3(256), 0, st, conv_w[i], (unsigned short*)((char*)dst + pack_off(A, i)), s.cout, s.cin, s.ks, chunk_for(s));
        else hipLaunchKernelGGL(k_pack_b16<false>, dim3(512), dim3(256), 0, st, conv_w[i], (unsigned short*)((char*)dst + pack_off(A, i)), s.cout, s.cin, s.ks, chunk_for(s));
    }
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

template <bool F16>
inline int run_conv(hipStream_t st, const unsigned short* in, const unsigned short* wpk, unsigned short* out, int B, int Hi, int Wi, const ConvSpec& s, int* Ho_, int* Wo_) {
    ConvGeo g;
    g.B = B; g.Hi = Hi; g.Wi = Wi; g.Cin = s.cin; g.Cout = s.cout; g.ks = s.ks; g.stride = s.stride;
    g.Ho = (Hi + 2 * s.pad - s.ks) / s.stride + 1;
    g.Wo = (Wi + 2 * s.pad - s.ks) / s.stride + 1;
    g.cin_c = chunk_for(s);
    g.M = npos(B, g.Ho, g.Wo);
    g.Mint = (long long)B * (g.Ho + 1) * (g.Wo + 1);
    g.Min = npos(B, Hi, Wi);
    *Ho_ = g.Ho; *Wo_ = g.Wo;
    if ((s.ks != 3 && s.ks != 1) || (s.ks == 3 && s.pad != 1) || (s.ks == 1 && s.pad != 0) || s.cin % g.cin_c) return DSMIL_E_UNSUPPORTED;
    const int pslot = dsmil_prof::begin(dsmil_prof::CH_CONV, st);
    struct ProfEnd { int slot; hipStream_t st; ~ProfEnd() { dsmil_prof::end(dsmil_prof::CH_CONV, slot, st); } } prof_end{pslot, st};
    if (v2_conv(s)) {
        g.cin_c = 16;
        const int Wp = g.Wo + 1;
        if (s.cout % 128 == 0) {
            constexpr int BMv = 256, NT = 128;
            if ((BMv + 2 * Wp + 2) * 2 > B16W_NIA * 256) return DSMIL_E_UNSUPPORTED;        // (maps wider than ~250 pixels)
            const size_t lds = (size_t)(((BMv + 2 * Wp + 2) * 48 + 15) / 16 * 16) + 9 * (NT / 32) * 1024;
            allow_lds((const void*)k_conv_b16w<BMv, NT, 2, 2, F16>, lds);
            hipLaunchKernelGGL((k_conv_b16w<BMv, NT, 2, 2, F16>), dim3((unsigned)((g.M + BMv - 1) / BMv), s.cout / NT), dim3(256), lds, st, in, wpk, out, g);
        } else if (s.cin == 64 && s.cout == 64 && (256 + 2 * Wp + 2) * 8 <= B16N_NIA * 256) {
            const size_t lds = (size_t)(((256 + 2 * Wp + 2) * 144 + 15) / 16 * 16) + 9 * 2 * 1024;
            allow_lds((const void*)k_conv_b16n<F16>, lds);
            hipLaunchKernelGGL(k_conv_b16n<F16>, dim3((unsigned)((g.M + 255) / 256), 1), dim3(256), lds, st, in, wpk, out, g);
        } else {
            constexpr int BMv = 512, NT = 64;
            if ((BMv + 2 * Wp + 2) * 2 > B16W_NIA * 256) return DSMIL_E_UNSUPPORTED;
            const size_t lds = (size_t)(((BMv + 2 * Wp + 2) * 48 + 15) / 16 * 16) + 9 * (NT / 32) * 1024;
            allow_lds((const void*)k_conv_b16w<BMv, NT, 4, 1, F16>, lds);
            hipLaunchKernelGGL((k_conv_b16w<BMv, NT, 4, 1, F16>), dim3((unsigned)((g.M + BMv - 1) / BMv), s.cout / NT), dim3(256), lds, st, in, wpk, out, g);
        }
        return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
    }
    if (g2_conv(s) && g.Min * s.cin < 0x7fffffffLL) {
        g.cin_c = 64;
        const size_t lds = 256 * 144 + 4 * 4 * 1024;
        allow_lds((const void*)k_conv_b16g<F16>, lds);
        hipLaunchKernelGGL(k_conv_b16g<F16>, dim3((unsigned)((g.M + 255) / 256), s.cout / 128), dim3(256), lds, st, in, wpk, out, g);
        return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
    }
    return DSMIL_E_UNSUPPORTED;   // (every conv of a BasicBlock trunk is one of the two forms above)
}

inline int stat_chunks(int HW) { return HW >= 2048 ? 8 : HW >= 512 ? 4 : HW >= 128 ? 2 : 1; }

template <bool F16>
inline int run_stats(hipStream_t st, const unsigned short* x, float* part, int B, int H, int W, int C) {
    const int S = stat_chunks(H * W);
    hipLaunchKernelGGL(k_stats_b16<F16>, dim3((unsigned)S, (unsigned)B), dim3(256), 0, st, x, part, H, W, C, S);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}
template <bool F16>
inline int run_apply(hipStream_t st, const unsigned short* x, const unsigned short* idn, unsigned short* y, const float* part, int B, int H, int W, int C, bool relu) {
    const int S = stat_chunks(H * W), PL = 256 / (C / 8), PP = (H + 1) * (W + 1);
    const dim3 grid((unsigned)((PP + PL * 8 - 1) / (PL * 8)), (unsigned)B);
    if (idn) hipLaunchKernelGGL((k_apply_b16<true, true, F16>), grid, dim3(256), 0, st, x, idn, y, part, H, W, C, S);
    else if (relu) hipLaunchKernelGGL((k_apply_b16<false, true, F16>), grid, dim3(256), 0, st, x, idn, y, part, H, W, C, S);
    else hipLaunchKernelGGL((k_apply_b16<false, false, F16>), grid, dim3(256), 0, st, x, idn, y, part, H, W, C, S);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}

// bytes the trunk needs behind the stem (four activation buffers of the largest padded map + the statistics partials)
inline size_t act_bytes(int B, int Hp, int Wp) { return al256((size_t)npos(B, Hp, Wp) * 64 * 2); }
inline size_t part_bytes(int B) { return al256((size_t)B * 8 * 512 * 2 * 4); }
inline size_t scratch_bytes(int B, int Hp, int Wp) { return 4 * act_bytes(B, Hp, Wp) + part_bytes(B); }

// The trunk behind the stem.  x0: the stem's normalised pooled output, fp32 NHWC [B][Hp][Wp][64] — or nullptr when the stem
// has already written it as bf16 into the first activation buffer (k_pool_fix_norm); scratch: scratch_bytes().
template <bool F16>
inline int trunk_t(hipStream_t st, const Arch& A, const float* x0, const unsigned short* wpk, void* scratch, int B, int Hp, int Wp, float* feats) {
    char* s8 = (char*)scratch;
    const size_t ab = act_bytes(B, Hp, Wp);
    unsigned short* bufs[4];
    for (int i = 0; i < 4; ++i) bufs[i] = (unsigned short*)(s8 + i * ab);
    float* part = (float*)(s8 + 4 * ab);
    unsigned short* cur = bufs[0];
    unsigned short* r1 = bufs[1];
    unsigned short* r2 = bufs[2];
    unsigned short* rd = bufs[3];
    if (x0) {
        const long long total = npos(B, Hp, Wp) * 8;
        long long blocks = (total + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(k_b16_pad<F16>, dim3((unsigned)blocks), dim3(256), 0, st, x0, cur, B, Hp, Wp, 64);
        if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
    } else {   // the stem's k_pool_fix_norm wrote the interior of `cur`: its borders
        const long long total = ((long long)B * (Wp + 1 + Hp) + Wp + 1) * 8;
        hipLaunchKernelGGL(k_b16_borders, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, cur, B, Hp, Wp, 64);
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
            if ((rc = run_conv<F16>(st, cur, (const unsigned short*)((const char*)wpk + pack_off(A, ci)), r1, B, Hc, Wc, s1, &Ho, &Wo))) return rc;
            if ((rc = run_stats<F16>(st, r1, part, B, Ho, Wo, s1.cout))) return rc;
            if ((rc = run_apply<F16>(st, r1, nullptr, r1, part, B, Ho, Wo, s1.cout, true))) return rc;
            // downsample branch: 1x1 stride 2 -> IN (no ReLU), in place
            const unsigned short* idn = cur;
            if (down) {
                const ConvSpec& sd = A.specs[ci + 2];
                int Hd, Wd;
                if ((rc = run_conv<F16>(st, cur, (const unsigned short*)((const char*)wpk + pack_off(A, ci + 2)), rd, B, Hc, Wc, sd, &Hd, &Wd))) return rc;
                if (Hd != Ho || Wd != Wo) return DSMIL_E_UNSUPPORTED;
                if ((rc = run_stats<F16>(st, rd, part, B, Hd, Wd, sd.cout))) return rc;
                if ((rc = run_apply<F16>(st, rd, nullptr, rd, part, B, Hd, Wd, sd.cout, false))) return rc;
                idn = rd;
            }
            // conv2 -> IN, + identity, ReLU
            if ((rc = run_conv<F16>(st, r1, (const unsigned short*)((const char*)wpk + pack_off(A, ci + 1)), r2, B, Ho, Wo, s2, &H2, &W2))) return rc;
            if ((rc = run_stats<F16>(st, r2, part, B, H2, W2, s2.cout))) return rc;
            if (last) {
                const int S = stat_chunks(H2 * W2);
                hipLaunchKernelGGL(k_pool_b16<F16>, dim3((unsigned)((s2.cout + 255) / 256), (unsigned)B), dim3(256), 0, st, r2, idn, part, feats, H2, W2, s2.cout, S);
                if (hipGetLastError() != hipSuccess) return DSMIL_E_LAUNCH;
            } else {
                if ((rc = run_apply<F16>(st, r2, idn, r2, part, B, H2, W2, s2.cout, true))) return rc;
                unsigned short* t = cur; cur = r2; r2 = t;     // the block's output becomes the next input; its old input is free
            }
            ci += down ? 3 : 2;
            Hc = H2; Wc = W2;
        }
    }
    return DSMIL_OK;
}

inline int trunk(hipStream_t st, const Arch& A, const float* x0, const unsigned short* wpk, void* scratch, int B, int Hp, int Wp, float* feats, bool f16) {
    return f16 ? trunk_t<true>(st, A, x0, wpk, scratch, B, Hp, Wp, feats) : trunk_t<false>(st, A, x0, wpk, scratch, B, Hp, Wp, feats);
}

}  // namespace b16
