// wino_w1.h — k_conv_wino_w1: the Winograd F(2x2,3x3) unit of k_conv_wino_s3 (same packed weights, same V layout, same
// arithmetic — bit-identical outputs) re-cut for ONE wave per SIMD.  Included by resnet_fwd.hip behind k_conv_wino_s3,
// whose constants and helpers it uses; not a translation unit of its own.
//
// Why.  k_conv_wino_s3 runs two waves per SIMD (256 registers each): eight accumulator tiles take half of them, so weight
// fragments can be requested only two positions ahead, the staging / transform phases need registers the MFMA phase holds,
// and the measured phases are ADDITIVE (bare MFMA loop 1.75 of 4.7 ms).  k_attend_bf16_res (agg_res.h) showed
// what one wave per SIMD buys on gfx950: 512 registers per lane — the sixteen accumulator tiles of ALL sixteen transform
// positions live in the 256 AGPRs, the 256 VGPRs hold a weight ring that is two position PAIRS deep, next pair's V
// fragments, the raw pixels of the chunk after next and the transform's temporaries at the same time — and one instruction
// stream in which the staging and transform work of chunk c+1 is laid into the MFMA shadows of chunk c.
//
// Unit: 32 tiles x 128 output channels (wave w owns channels 32w..32w+31 and all 16 positions: 16 x 6 = 96 MFMAs per
// 16-channel chunk, consecutive MFMAs alternate between the two positions of a pair, so none waits for the one before).
// LDS: V double-buffered 2 x 56 KB, raw double-buffered 2 x 20 KB, producer statistics 4 KB, tile table 1 KB = 157 KB:
// one workgroup per CU.
// One wave per SIMD hides about FIVE other instructions behind each MFMA (MI355X_MICROARCH.md) — 480 per chunk, and the
// chunk needs ~540 (48 weight loads, 48 V reads, the transform's 32 window reads, ~250 VALU ops of B^T d B and plane cuts,
// 24 plane writes, the raw staging).  So the work of the chunks ahead is spread EVENLY over the chunk's 8 pair-blocks of 12
// MFMAs, one piece behind each MFMA, straight-line code (no branch: the last iterations redo clamped work):
//   block P       : half a transform row of chunk c+1 (xi = P/2, output X or Y): raw[(c+1)&1] -> planes -> V[(c+1)&1]
//   blocks 4-7    : raw(c+2) registers -> IN + ReLU + padding -> raw[c&1], one float4 per block
//                   (loaded in iteration c-1) and the load of the same float4 of raw(c+3) into the freed register
//   block 0 / 6   : statistics(c+3) global -> register / register -> LDS
//   block 7       : ONE barrier (V[(c+1)&1], raw[c&1], statistics complete); first V fragments of c+1
//   every block   : weight fragments of the pair two blocks ahead (ring of 4 pair-sets, runs across chunks), V fragments of
//                   the next pair.
// The inverse transform needs no exchange: a lane holds all sixteen positions of its (tile, channel) in registers.
// ABL (experiment builds; timing only, wrong results): 1 no transform, 2 no raw staging, 4 weight fragments loaded once,
// 8 V fragments read once, 16 no MFMAs, 32 no output stores, 64 no statistics partials, 128 no epilogue.
// plain f32 add / sub / fma the SLP vectoriser cannot fuse into v_pk_*_f32: beside MFMAs a packed op costs ~13 cycles more
// than the two scalar ops it replaces (MI355X_MICROARCH.md, per-instruction constants)
__device__ __forceinline__ float w1_add(float x, float y) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float w1_sub(float x, float y) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ f32x4 w1_add4(const f32x4& x, const f32x4& y) { return f32x4{w1_add(x[0], y[0]), w1_add(x[1], y[1]), w1_add(x[2], y[2]), w1_add(x[3], y[3])}; }
__device__ __forceinline__ f32x4 w1_sub4(const f32x4& x, const f32x4& y) { return f32x4{w1_sub(x[0], y[0]), w1_sub(x[1], y[1]), w1_sub(x[2], y[2]), w1_sub(x[3], y[3])}; }

template <bool NORM, int NP = 6, int ABL = 0>
__global__ __launch_bounds__(256, 1) void k_conv_wino_w1(WinoArgs a) {
    constexpr int NT = 256;
    constexpr int PLN = emb_planes<NP>();               // operand planes: 2 (fp16 form, NP = 3) or 3 (bf16 forms)
    constexpr int RPT = (WRAW_MAX * 4 + NT - 1) / NT;   // raw float4 per thread per chunk (4)
    constexpr int R_DW = WRAW_MAX * SRLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sV = reinterpret_cast<unsigned*>(smem);   // [2][SV_DW]
    float* sR = smem + 2 * SV_DW;                       // [2][WRAW_MAX][SRLD]
    float* sS = sR + 2 * R_DW + 256;                    // [2 buffers][16 images][2 (mean, rstd)][16 ch]
    unsigned* sT = reinterpret_cast<unsigned*>(sR + 2 * R_DW);   // [32 tile slots][4]: output offset (lo, hi), flags, -
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // 1-D grid over (unit, 128-cout block).  Workgroup L runs on XCD L % 8 (dispatch order; used for speed only): all the
    // workgroups of an XCD take the SAME cout block(s), so its 4 MiB L2 holds one block's weights (the 3.1-6.3 MB slices of
    // layers 3-4), not all of them.  CB cout blocks, G = 8 / CB XCDs per block.
    const int CB = a.Cout >> 7;
    int bid, cbk;
    {
        const int L = blockIdx.x, total = gridDim.x;
        if ((CB == 2 || CB == 4) && (total & 7) == 0) { cbk = (L & 7) % CB; bid = (L >> 3) * (8 / CB) + (L & 7) / CB; }
        else { cbk = L % CB; bid = L / CB; }
    }
    const int n0 = cbk * 128;
    const int nchunks = a.C / SK;
    const int bid_u = bid;   // unit number (trace builds)
    (void)bid_u;
    const int bx = bid % a.nbx; bid /= a.nbx;
    const int by = bid % a.nby; bid /= a.nby;
    const int img0 = bid * a.IB;
    const int ty0 = by * a.TYB, tx0 = bx * a.TXB;
    const int pb = by * a.nbx + bx;
    const int RH = 2 * a.TYB + 2, RW = 2 * a.TXB + 2, RP = RH * RW;
    const int tpi = a.TYB * a.TXB;
    const int iy_org = 2 * ty0 - 1, ix_org = 2 * tx0 - 1;

    // ---- raw staging role (as k_conv_wino_s3): element e = tid + 256 q -> (pixel, 4-channel group)
    // (pixel of element q = pixel of element 0 + 64 q < 256: a slot past the unit's region stages zeros into its own,
    // unused, row of the raw buffer — no branch, no dummy target)
    int roff[RPT], rsto[RPT];
    const int rlds0 = ((tid & 7) | ((tid >> 5) << 3)) * SRLD + ((tid >> 3) & 3) * 4;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int e = tid + NT * q, px = (e & 7) | ((e >> 5) << 3), gg = (e >> 3) & 3;
        roff[q] = -1; rsto[q] = gg * 4;
        if (px < a.IB * RP) {
            const int il = px / RP, rem = px - il * RP, ry = rem / RW, rx = rem - ry * RW;
            const int n = img0 + il, iy = iy_org + ry, ix = ix_org + rx;
            if (n < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                roff[q] = ((n * a.H + iy) * a.W + ix) * a.C + gg * 4;
                rsto[q] = il * 32 + gg * 4;
            }
        }
    }
    // producer statistics of a chunk: thread t fetches 4 channels of (mean | rstd) of local image (t & 127) >> 3 (the upper
    // half of the workgroup repeats the lower half's work: no branch)
    f32x4 sreg = {0.f, 0.f, 0.f, 0.f};
    const int st_il = (tid & 127) >> 3, st_which = (tid >> 2) & 1, st_c4 = tid & 3;
    const int st_n = img0 + st_il < a.B ? img0 + st_il : a.B - 1;
    auto stat_load = [&](int cc) {
        if constexpr (NORM)
            sreg = *reinterpret_cast<const f32x4*>((st_which ? a.in_rstd : a.in_mean) + (long long)st_n * a.C + cc * SK + st_c4 * 4);
    };
    auto stat_write = [&](int cc) {
        if constexpr (NORM) *reinterpret_cast<f32x4*>(sS + (cc & 1) * 512 + st_il * 32 + st_which * 16 + st_c4 * 4) = sreg;
    };
    f32x4 rreg[RPT];
    auto raw_load = [&](int cc) {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const unsigned off_b = roff[q] < 0 ? 0u : (unsigned)roff[q] * 4u;
            rreg[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.x + cc * SK) + off_b);
        }
    };
    auto raw_write_q = [&](int cc, int q) {   // producer's IN + ReLU and the zero padding, once per staged pixel -> raw[cc & 1]
        f32x4 x = rreg[q];
        const bool ok = roff[q] >= 0;
        if constexpr (NORM) {
            const f32x4 mu = *reinterpret_cast<const f32x4*>(sS + (cc & 1) * 512 + rsto[q]);
            const f32x4 rs = *reinterpret_cast<const f32x4*>(sS + (cc & 1) * 512 + rsto[q] + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = fmaxf((x[e] - mu[e]) * rs[e], 0.f);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = ok ? x[e] : 0.f;
        *reinterpret_cast<f32x4*>(sR + (cc & 1) * R_DW + rlds0 + q * (64 * SRLD)) = x;
    };
    // ---- transform role (as the 256-thread k_conv_wino_s3): channel group g, tile slot ts, column half h (wave-uniform)
    const int g = tid & 3, ts = ((lane >> 5) & 1) | (((lane >> 2) & 7) << 1) | ((wave & 1) << 4);
    const int h = (wave >> 1) & 1;
    const int sil = ts / tpi, srem = ts - sil * tpi, styl = srem / a.TXB, stxl = srem - styl * a.TXB;
    const int praw = (ts < a.IB * tpi) ? ((sil * RH + 2 * styl) * RW + 2 * stxl + h) * SRLD + g * 4 : g * 4;
    // columns of the window are patch columns h..h+2.  With T_c = (B^T d)[xi][column c]:
    //   h = 0: nu 0 = T0 - T2, nu 1 = T1 + T2;   h = 1: nu 2 = T1 - T0, nu 3 = T0 - T2
    // i.e. X = T0 - T2 goes to nu (h ? 3 : 0) and Y = T1 + sz * Tz, (sz, z) = h ? (-1, 0) : (+1, 2), to nu (h ? 2 : 1):
    // the choice is an LDS address and a wave-uniform sign, not a branch (x * +-1 is exact, the fma rounds once like the add).
    const float sz = h ? -1.f : 1.f;
    const int zcol = (h ? 0 : 2) * SRLD;
    const int nuX = h ? 3 : 0, nuY = h ? 2 : 1;
    auto tr_xi = [&](int xi, const float* rwin, unsigned* vdst) {   // prologue form: one whole row, unpipelined
        // (B^T d) row xi = Ra +- Rb:  xi 0: R0 - R2,  1: R1 + R2,  2: R2 - R1,  3: R1 - R3
        const int ra = xi == 0 ? 0 : xi == 2 ? 2 : 1, rb = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
        const float* pa = rwin + ra * RW * SRLD;
        const float* pb_ = rwin + rb * RW * SRLD;
        f32x4 T[3], Tz;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4 A = *reinterpret_cast<const f32x4*>(pa + c * SRLD);
            const f32x4 Bv = *reinterpret_cast<const f32x4*>(pb_ + c * SRLD);
            T[c] = xi == 1 ? A + Bv : A - Bv;
        }
        {
            const f32x4 A = *reinterpret_cast<const f32x4*>(pa + zcol);
            const f32x4 Bv = *reinterpret_cast<const f32x4*>(pb_ + zcol);
            Tz = xi == 1 ? A + Bv : A - Bv;
        }
        const f32x4 X = T[0] - T[2];
        f32x4 Y;
#pragma unroll
        for (int e = 0; e < 4; ++e) Y[e] = __builtin_fmaf(sz, Tz[e], T[1][e]);
        u32x2_t ph, pm, pl;
        unsigned* dX = vdst + ((xi * 4 + nuX) * WTT + ts) * SVLD + g * 2;
        cut4<NP>(X, ph, pm, pl);
        *reinterpret_cast<u32x2_t*>(dX) = ph;
        if constexpr (PLN >= 2) *reinterpret_cast<u32x2_t*>(dX + 8) = pm;
        if constexpr (PLN == 3) *reinterpret_cast<u32x2_t*>(dX + 16) = pl;
        unsigned* dY = vdst + ((xi * 4 + nuY) * WTT + ts) * SVLD + g * 2;
        cut4<NP>(Y, ph, pm, pl);
        *reinterpret_cast<u32x2_t*>(dY) = ph;
        if constexpr (PLN >= 2) *reinterpret_cast<u32x2_t*>(dY + 8) = pm;
        if constexpr (PLN == 3) *reinterpret_cast<u32x2_t*>(dY + 16) = pl;
    };
    // ---- weights, TILED for this kernel (k_pack_wino_s3, tiled = 1): [Cout/32][C/16][16 pos][3 planes][32 couts][16] bf16 —
    // the 48 fragments a wave needs for one chunk are 48 consecutive KiB.  Buffer loads: one resource for the tensor, the
    // lane's 16 bytes as the only address VGPR, a wave-uniform soffset per 4 fragments and the 12-bit immediate for the
    // fragment among them: no per-load address arithmetic on the VALU (the per-lane L2 loads of k_conv_wino_s3 cost one
    // 64-bit add each — an issue slot this kernel does not have).
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u), 0, (int)((long long)a.C * a.Cout * 96), 0x00020000);
    const int ulane_b = l31 * 32 + hi * 16;             // the lane's 16 B inside a 1 KiB fragment
    const int utile_b = (((n0 >> 5) + wave) * nchunks) * (48 * 1024);   // wave-uniform
    union Frag { u32x4_t u; bf16x8_t v; };
    Frag uw[4][2][3];                                   // ring of 4 pair-sets; two pairs in flight beside the one in use
    auto uload_pair = [&](int P, int cc, Frag (&w)[2][3]) {
        const int c2 = cc < nchunks ? cc : nchunks - 1;
        const int sbase = utile_b + c2 * (48 * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < PLN; ++pl) {
                const int f = (2 * P + j) * 3 + pl;
                w[j][pl].u = __builtin_amdgcn_raw_buffer_load_b128(urs, ulane_b + (f & 3) * 1024, sbase + (f >> 2) * 4096, 0);
            }
    };
    Frag vq[2][2][3];                                   // V fragments: the pair in use and the next one
    const int vfo = l31 * SVLD + 4 * hi;                // dwords
    auto vread_pair = [&](int P, int buf, Frag (&v)[2][3]) {
        const unsigned* b = sV + buf * SV_DW + vfo + (2 * P) * WTT * SVLD;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < PLN; ++pl) v[j][pl].u = *reinterpret_cast<const u32x4_t*>(b + j * WTT * SVLD + pl * 8);
    };

#ifdef DSMIL_TRACE
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // ---- prologue: raw(0) -> raw[0] -> V[0]; raw(1) -> raw[1]; raw(2) in registers; statistics(2) in LDS; weight pairs 0, 1;
    //      the epilogue's tile table; V pair 0.  Every global load of the three chunks goes out FIRST (one memory round
    //      trip, not three: nothing hides the prologue of the only workgroup on a CU).
    const int cl1 = 1 < nchunks ? 1 : nchunks - 1, cl2 = 2 < nchunks ? 2 : nchunks - 1;
    f32x4 rr1[RPT], rr2[RPT], sr1 = {0.f, 0.f, 0.f, 0.f}, sr2 = {0.f, 0.f, 0.f, 0.f};
    raw_load(0);
    stat_load(0);
    {
        const f32x4 s0 = sreg;
        f32x4 keep[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) keep[q] = rreg[q];
        raw_load(cl1); stat_load(cl1); sr1 = sreg;
#pragma unroll
        for (int q = 0; q < RPT; ++q) rr1[q] = rreg[q];
        raw_load(cl2); stat_load(cl2); sr2 = sreg;
#pragma unroll
        for (int q = 0; q < RPT; ++q) { rr2[q] = rreg[q]; rreg[q] = keep[q]; }
        sreg = s0;
    }
    uload_pair(0, 0, uw[0]);
    uload_pair(1, 0, uw[1]);
    if (tid < 32) {
        // where the epilogue stores tile slot `tid` (the divisions once per slot here, not once per accumulator row there):
        // element offset of pixel (2 ty, 2 tx) of its image, flags 1 = slot holds a tile, 2 = column 2 tx + 1 exists,
        // 4 = row 2 ty + 1 exists, local image number from bit 8
        const int slot = tid;
        const int il = slot / tpi, rem = slot - il * tpi, tyl = rem / a.TXB, txl = rem - tyl * a.TXB;
        const int n = img0 + il, ty = ty0 + tyl, tx = tx0 + txl;
        const bool ok = slot < a.IB * tpi && n < a.B && ty < a.TY && tx < a.TX;
        const long long off = ok ? ((long long)(n * a.H + 2 * ty) * a.W + 2 * tx) * a.Cout : 0;
        const unsigned fl = ok ? (1u | (2 * tx + 1 < a.W ? 2u : 0u) | (2 * ty + 1 < a.H ? 4u : 0u) | ((unsigned)il << 8)) : 0u;
        sT[slot * 4 + 0] = (unsigned)off;
        sT[slot * 4 + 1] = (unsigned)(off >> 32);
        sT[slot * 4 + 2] = fl;
    }
    stat_write(0);                                       // statistics(0) -> sS[0]
    sreg = sr1;
    stat_write(1);                                       // statistics(1) -> sS[1]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < RPT; ++q) raw_write_q(0, q);
#pragma unroll
    for (int q = 0; q < RPT; ++q) rreg[q] = rr1[q];
#pragma unroll
    for (int q = 0; q < RPT; ++q) raw_write_q(1, q);
#pragma unroll
    for (int q = 0; q < RPT; ++q) rreg[q] = rr2[q];      // raw(2): staged by blocks 4-7 of iteration 0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // raw[0], raw[1] complete; every read of sS[0] is done
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) tr_xi(xi, sR + praw, sV);
    sreg = sr2;
    stat_write(2);                                       // statistics(2) -> sS[0]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    vread_pair(0, 0, vq[0]);

#ifdef DSMIL_TRACE
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
    for (int cc = 0; cc < nchunks; ++cc) {
        const int c3 = cc + 3 < nchunks ? cc + 3 : nchunks - 1;   // clamped: the last iterations redo work nobody reads
        const int buf = cc & 1;
        unsigned* vnext = sV + (buf ^ 1) * SV_DW;
        const float* rwin = sR + (buf ^ 1) * R_DW + praw;        // raw[(cc+1) & 1]
#ifdef DSMIL_TRACE
        unsigned long long stamp[8];
#endif
        // One pair-block = 2 NP MFMAs; behind each MFMA one PIECE of the other work, pinned there by sched_barrier (left to
        // itself — or to sched_group_barrier patterns — hipcc issues the MFMAs back to back and the VALU work in 60-100
        // instruction clumps: the pipe idles for the length of every clump).  Pieces are ~4 VALU ops; the plane cut runs
        // stage by stage over the four elements of an output (and x4, sub x4, ...: as one chain per element every op
        // waits for the one before).
        auto block = [&](auto Pc) {
            constexpr int P = decltype(Pc)::value;
#ifdef DSMIL_TRACE
            stamp[P] = __builtin_amdgcn_s_memtime();   // kept in SGPRs, written out behind block 7
#endif
            Frag (&va)[2][3] = vq[P & 1];
            Frag (&wb)[2][3] = uw[P & 3];
            using PP = PlaneProducts<NP>;
            // half a transform row: xi = P / 2; outputs X = T0 - T2 (even blocks) or Y = T1 + sz Tz (odd blocks),
            // T_c = (B^T d)[xi][c] = Ra[c] +- Rb[c]:  xi 0: R0 - R2,  1: R1 + R2,  2: R2 - R1,  3: R1 - R3
            constexpr int xi = P >> 1;
            constexpr bool isY = P & 1;
            constexpr int ra = xi == 0 ? 0 : xi == 2 ? 2 : 1, rb = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
            f32x4 wA0, wB0, wA1, wB1, Ta, Tb, O;           // window rows of the two columns, their row transforms, the output
            unsigned u_[4], r1_[4], r2_[4];                // plane cut, stage by stage
            f32x4 qx, mu, rs;                              // a raw float4 on its way to LDS
            auto step = [&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                // MFMA slots: 2 NP per pair-block, the two positions alternating (no MFMA waits for the one before).  NP = 6 / 9:
                // one per step (12 / 18 steps).  NP = 3 (fp16 form): six MFMAs in the block's twelve steps — steps 0 1, 4 5, 8 9 —
                // so that the pieces of the other work still sit between MFMAs
                // (an MFMA behind every OTHER step instead — 0 2 4 .. 10 — measured the same: 63.2 k against 62.6 k patches/s)
                constexpr bool has_mfma = NP == 3 ? (I % 4) < 2 : (NP == 1 ? I < 2 : true);   // (NP = 1, the reduced-precision path: two)
                constexpr int k = NP == 3 ? I / 4 : (NP == 1 ? 0 : I / 2), j = I & 1;
                if constexpr (has_mfma) {
                    if constexpr (!(ABL & 16))
                        acc[2 * P + j] = plane_mfma<NP>(va[j][PP::X[k]].u, wb[j][PP::W[k]].u, acc[2 * P + j]);
                    else asm volatile("" ::"v"(va[j][PP::X[k]].u), "v"(wb[j][PP::W[k]].u));
                }
                // every block: weight fragments of the pair two blocks ahead; V fragments of the next pair in two halves
                if constexpr (I == 0 && !(ABL & 4)) uload_pair((P + 2) & 7, cc + ((P + 2) >> 3), uw[(P + 2) & 3]);
                if constexpr ((I == 8 || I == 9) && !(ABL & 8) && P < 7) {
                    const unsigned* b = sV + buf * SV_DW + vfo + (2 * (P + 1) + (I - 8)) * WTT * SVLD;
#pragma unroll
                    for (int pl = 0; pl < PLN; ++pl) vq[(P + 1) & 1][I - 8][pl].u = *reinterpret_cast<const u32x4_t*>(b + pl * 8);
                }
                if constexpr (!(ABL & 1)) {                 // transform(c+1), half row
                    if constexpr (I == 1) {
                        const float* pa = rwin + ra * RW * SRLD;
                        const float* pb_ = rwin + rb * RW * SRLD;
                        const int c0 = isY ? SRLD : 0, c1 = isY ? zcol : 2 * SRLD;   // Y: columns 1 and z; X: columns 0 and 2
                        wA0 = *reinterpret_cast<const f32x4*>(pa + c0);
                        wB0 = *reinterpret_cast<const f32x4*>(pb_ + c0);
                        wA1 = *reinterpret_cast<const f32x4*>(pa + c1);
                        wB1 = *reinterpret_cast<const f32x4*>(pb_ + c1);
                    }
                    if constexpr (I == 3) Ta = xi == 1 ? w1_add4(wA0, wB0) : w1_sub4(wA0, wB0);
                    if constexpr (I == 4) Tb = xi == 1 ? w1_add4(wA1, wB1) : w1_sub4(wA1, wB1);
                    if constexpr (I == 5) {
                        if constexpr (!isY) O = w1_sub4(Ta, Tb);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) O[e] = __builtin_fmaf(sz, Tb[e], Ta[e]);
                        }
                    }
                    if constexpr (NP == 1) {                // one fp16 plane
                        if constexpr (I == 6) u_[0] = cut2h1(O[0], O[1]);
                        if constexpr (I == 7) u_[1] = cut2h1(O[2], O[3]);
                        if constexpr (I == 10) {
                            unsigned* d = vnext + ((xi * 4 + (isY ? nuY : nuX)) * WTT + ts) * SVLD + g * 2;
                            *reinterpret_cast<u32x2_t*>(d) = u32x2_t{u_[0], u_[1]};
                        }
                    }
                    if constexpr (NP == 3) {                // fp16 form: two v_fma_mix per value, the planes come out packed
                        if constexpr (I == 6) cut2h(O[0], O[1], u_[0], r1_[0]);
                        if constexpr (I == 7) cut2h(O[2], O[3], u_[1], r1_[1]);
                        if constexpr (I == 10) {
                            unsigned* d = vnext + ((xi * 4 + (isY ? nuY : nuX)) * WTT + ts) * SVLD + g * 2;
                            *reinterpret_cast<u32x2_t*>(d) = u32x2_t{u_[0], u_[1]};
                            *reinterpret_cast<u32x2_t*>(d + 8) = u32x2_t{r1_[0], r1_[1]};
                        }
                    }
                    if constexpr (NP != 3 && NP != 1 && I == 6) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { u_[e] = __float_as_uint(O[e]); r1_[e] = __float_as_uint(O[e] - __uint_as_float(u_[e] & 0xFFFF0000u)); }
                    }
                    if constexpr (NP != 3 && NP != 1 && I == 7) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) r2_[e] = __float_as_uint(__uint_as_float(r1_[e]) - __uint_as_float(r1_[e] & 0xFFFF0000u));
                    }
                    if constexpr (NP != 3 && NP != 1 && I == 10) {
                        u32x2_t ph, pm, pl;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            ph[i] = __builtin_amdgcn_perm(u_[2 * i + 1], u_[2 * i], 0x07060302u);
                            pm[i] = __builtin_amdgcn_perm(r1_[2 * i + 1], r1_[2 * i], 0x07060302u);
                            pl[i] = __builtin_amdgcn_perm(r2_[2 * i + 1], r2_[2 * i], 0x07060302u);
                        }
                        unsigned* d = vnext + ((xi * 4 + (isY ? nuY : nuX)) * WTT + ts) * SVLD + g * 2;
                        *reinterpret_cast<u32x2_t*>(d) = ph;
                        *reinterpret_cast<u32x2_t*>(d + 8) = pm;
                        *reinterpret_cast<u32x2_t*>(d + 16) = pl;
                    }
                }
                if constexpr (P >= 4 && !(ABL & 2)) {       // raw(c+2): float4 q = P - 4 -> IN + ReLU + padding -> raw[c & 1]
                    constexpr int q = P - 4;
                    if constexpr (I == 2 && NORM) {
                        mu = *reinterpret_cast<const f32x4*>(sS + buf * 512 + rsto[q]);
                        rs = *reinterpret_cast<const f32x4*>(sS + buf * 512 + rsto[q] + 16);
                    }
                    if constexpr (I == 8) {
                        qx = rreg[q];
                        if constexpr (NORM) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) qx[e] = (qx[e] - mu[e]) * rs[e];
                        }
                    }
                    if constexpr (I == 9) {
                        const bool ok = roff[q] >= 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) qx[e] = ok ? (NORM ? fmaxf(qx[e], 0.f) : qx[e]) : 0.f;
                        *reinterpret_cast<f32x4*>(sR + buf * R_DW + rlds0 + q * (64 * SRLD)) = qx;
                    }
                    if constexpr (I == 11) {                // the register is free: the same float4 of raw(c+3), used 8 blocks on
                        const unsigned off_b = roff[q] < 0 ? 0u : (unsigned)roff[q] * 4u;
                        rreg[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.x + c3 * SK) + off_b);
                    }
                }
                if constexpr (P == 0 && I == 2 && !(ABL & 2)) stat_load(c3);         // six blocks before it is written to LDS
                if constexpr (P == 6 && I == 11 && !(ABL & 2)) stat_write(cc + 3);
                if constexpr (P == 7 && I == 10) {          // everything the next chunk reads is in LDS (or about to be waited for)
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    vread_pair(0, buf ^ 1, vq[(P + 1) & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{});
            step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            if constexpr (NP == 9) {
                step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{});
                step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
                step(std::integral_constant<int, 16>{}); step(std::integral_constant<int, 17>{});
            }
        };
        block(std::integral_constant<int, 0>{});
        block(std::integral_constant<int, 1>{});
        block(std::integral_constant<int, 2>{});
        block(std::integral_constant<int, 3>{});
        block(std::integral_constant<int, 4>{});
        block(std::integral_constant<int, 5>{});
        block(std::integral_constant<int, 6>{});
        block(std::integral_constant<int, 7>{});
#ifdef DSMIL_TRACE
        if (a.trace && bid_u < 4 && n0 == 0 && lane == 0 && (wave == 0 || wave == 3) && cc < 256) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a.trace[(((long long)bid_u * 2 + (wave != 0)) * 256 + cc) * 8 + i] = stamp[i];
        }
#endif
    }

    if constexpr (ABL & 128) {   // ablation: no epilogue at all
        float keep = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[p][r];
        if (keep == 123.456f) a.y[0] = keep;
        return;
    }
#ifdef DSMIL_TRACE
    const unsigned long long t_epi = __builtin_amdgcn_s_memtime();
#endif
    // ---- epilogue: inverse transform in registers (the arithmetic order of wino_epilogue, so the outputs are bit-identical
    // to k_conv_wino_s3), raw NHWC store, (cnt, mean, M2) statistics partials per output-row parity
    const int co = n0 + wave * 32 + l31;
    u32x4_t tsl[16];              // the table entries of this lane's 16 accumulator rows
#pragma unroll
    for (int r = 0; r < 16; ++r) tsl[r] = *reinterpret_cast<const u32x4_t*>(sT + drow(r, hi) * 4);
    float ya[2][16], yb[2][16];   // [output row parity][accumulator row]: pixels (oy + parity, ox) and (oy + parity, ox + 1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float ra[4], rb[4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            ra[xi] = acc[4 * xi][r] + acc[4 * xi + 1][r] + acc[4 * xi + 2][r];
            rb[xi] = acc[4 * xi + 1][r] - acc[4 * xi + 2][r] - acc[4 * xi + 3][r];
        }
        ya[0][r] = (ra[0] + ra[1]) + ra[2];
        yb[0][r] = (rb[0] + rb[1]) + rb[2];
        ya[1][r] = (-ra[2] - ra[3]) + ra[1];
        yb[1][r] = (-rb[2] - rb[3]) + rb[1];
        if constexpr (emb_f16<NP>()) {                      // the weights carry 2^EMB_WSHIFT (exact to undo)
            ya[0][r] *= EMB_OSCALE; yb[0][r] *= EMB_OSCALE; ya[1][r] *= EMB_OSCALE; yb[1][r] *= EMB_OSCALE;
        }
    }
    const long long rowstep = (long long)a.W * a.Cout;
#pragma unroll
    for (int wp = 0; wp < 2; ++wp) {
        unsigned vmask[16];
        int rimg[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned fl = tsl[r][2];
            unsigned vm = 0;
            rimg[r] = -1;
            if ((fl & 1u) && (wp == 0 || (fl & 4u))) {
                const long long off = (long long)(((unsigned long long)tsl[r][1] << 32) | tsl[r][0]);
                float* o = a.y + off + wp * rowstep + co;
                if constexpr (!(ABL & 32)) o[0] = ya[wp][r];
                vm = 1u;
                if (fl & 2u) { if constexpr (!(ABL & 32)) o[a.Cout] = yb[wp][r]; vm |= 2u; }
                rimg[r] = (int)(fl >> 8);
            }
            vmask[r] = vm;
        }
        for (int il = 0; il < ((ABL & 64) ? 0 : a.IB); ++il) {
            const int n = img0 + il;
            if (n >= a.B) break;
            float sum = 0.f, cnt = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned vm = (rimg[r] == il) ? vmask[r] : 0u;
                sum += ((vm & 1u) ? ya[wp][r] : 0.f) + ((vm & 2u) ? yb[wp][r] : 0.f);
                cnt += (float)__popc(vm);
            }
            sum += __shfl_xor(sum, 32, 64);
            cnt += __shfl_xor(cnt, 32, 64);
            const float mean = cnt > 0.f ? sum / cnt : 0.f;
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned vm = (rimg[r] == il) ? vmask[r] : 0u;
                const float d0 = ya[wp][r] - mean, d1 = yb[wp][r] - mean;
                q += ((vm & 1u) ? d0 * d0 : 0.f) + ((vm & 2u) ? d1 * d1 : 0.f);
            }
            q += __shfl_xor(q, 32, 64);
            if (hi == 0) {
                float* o = a.part + ((((long long)n * a.PB + pb) * 2 + wp) * a.Cout + co) * 3;
                o[0] = cnt; o[1] = mean; o[2] = q;
            }
        }
    }
#ifdef DSMIL_TRACE
    // slot 255 of the trace: kernel start, loop start, epilogue start, end (per workgroup 0-3, waves 0 and 3)
    if (a.trace && bid_u < 4 && n0 == 0 && lane == 0 && (wave == 0 || wave == 3)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* t = a.trace + (((long long)bid_u * 2 + (wave != 0)) * 256 + 255) * 8;
        t[0] = t_start; t[1] = t_loop; t[2] = t_epi; t[3] = __builtin_amdgcn_s_memtime();
    }
#endif
}
