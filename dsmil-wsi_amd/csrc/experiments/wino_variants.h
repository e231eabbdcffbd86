// experiments/wino_variants.h — the two measured-and-lost restructurings of the Winograd unit (k_conv_wino_pp: persistent,
// role-split; k_conv_wino_alt: half-step phase offset between the cout halves), see DESIGN.md §4 and profiles/README.md.
// NOT part of the product library: resnet_fwd.hip includes this file only in experiment builds
// (python dsmil-wsi_amd/build.py --variant expt -DDSMIL_EXPERIMENTS; selected with DSMIL_WINO_KERNEL=pp|alt).  It relies on
// the helpers defined above its include point in resnet_fwd.hip and is not a translation unit of its own.
// --------------------------------------------------------------------------------------------
// k_conv_wino_alt — the 128-cout Winograd unit with the two HALVES of the workgroup out of phase: 512 threads, waves 0-3
// (positions 0-7, group A) and waves 4-7 (positions 8-15, group B) alternate between the MFMA job and the staging job,
// half a step apart, so at every moment each SIMD holds one wave that multiplies and one that stages:
//     half-step 2c   : A  MFMAs of chunk c on V[c&1]           | B  stages: its half of transform(c+1) -> V[(c+1)&1],
//     half-step 2c+1 : A  stages (the same, its half)          |            raw(c+2) regs -> R[c&1], loads raw(c+3)
//                                                              | B  MFMAs of chunk c
// The transform splits by xi rows (A: xi = 0, 1; B: xi = 2, 3), the raw staging by element; V and raw are double buffered
// (2 x 56 KB + 2 x 20 KB + 4 KB statistics = 156 KB, one workgroup per CU); one workgroup barrier per half-step.
// Why: stamps of the in-phase forms (profiles/README.md) — every wave of k_conv_wino_s3 spends ~48 % of a step in its
// MFMA phase and ~47 % staging, and the phases of co-resident waves do not overlap; the role-split k_conv_wino_pp
// overlaps them but leaves ALL the staging to four waves (4300 cycles against 2320 of MFMAs).  Here the two jobs are
// the same size on every wave, and each wave's activation loads are issued a whole staging job before it next waits on
// a weight fragment.
// --------------------------------------------------------------------------------------------
template <bool NORM, int NP>
__global__ __launch_bounds__(512, 2) void k_conv_wino_alt(WinoArgs a) {
    constexpr int UD = 2;                               // (UD = 3 spills inside the staging job: 9500-cycle steps against 7700)
    constexpr int R_DW = WRAW_MAX * SRLD;
    constexpr int RPT = 2;                              // raw float4 per thread per chunk (1024 elements / 512 threads)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sV = reinterpret_cast<unsigned*>(smem);   // [2][SV_DW]
    float* sR = smem + 2 * SV_DW;                       // [2][R_DW]
    float* sS = sR + 2 * R_DW;                          // [2][16 images][2 (mean, rstd)][16 ch]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                          // 0 = A, 1 = B: position half of the MFMAs, xi half of the transform
    const int wn = wave & 3, wp = grp;
    const int l31 = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.y * 128;
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

    // ---- raw staging role: element e = tid + 512 q -> (pixel, channel group), as in k_conv_wino_s3
    int roff[RPT], rlds[RPT], rsto[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int e = tid + 512 * q, px = (e & 7) | ((e >> 5) << 3), gg = (e >> 3) & 3;
        roff[q] = -2; rlds[q] = 0; rsto[q] = gg * 4;
        if (px < a.IB * RP) {
            const int il = px / RP, rem = px - il * RP, ry = rem / RW, rx = rem - ry * RW;
            const int n = img0 + il, iy = iy_org + ry, ix = ix_org + rx;
            roff[q] = -1;
            if (n < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) roff[q] = ((n * a.H + iy) * a.W + ix) * a.C + gg * 4;
            rsto[q] = il * 32 + gg * 4;
            rlds[q] = px * SRLD + gg * 4;
        }
    }
    f32x4 sreg = {0.f, 0.f, 0.f, 0.f};
    auto stat_load = [&](int cc) {                      // threads 0..127 (group A)
        if constexpr (NORM) {
            if (tid < 128) {
                const int il = tid >> 3, which = (tid >> 2) & 1, c4 = tid & 3;
                const int n = img0 + il < a.B ? img0 + il : a.B - 1;
                const int c = cc < nchunks ? cc : nchunks - 1;
                sreg = *reinterpret_cast<const f32x4*>((which ? a.in_rstd : a.in_mean) + (long long)n * a.C + c * SK + c4 * 4);
            }
        }
    };
    auto stat_write = [&](int cc) {
        if constexpr (NORM) {
            if (tid < 128) {
                const int il = tid >> 3, which = (tid >> 2) & 1, c4 = tid & 3;
                *reinterpret_cast<f32x4*>(sS + (cc & 1) * 512 + il * 32 + which * 16 + c4 * 4) = sreg;
            }
        }
    };
    // ---- transform role (g, ts, h) of the thread inside its group; the group is the xi half
    const int w4 = wave & 3;
    const int g = tid & 3, ts = ((lane >> 5) & 1) | (((lane >> 2) & 7) << 1) | ((w4 & 1) << 4);
    const int h = w4 >> 1;
    const int sil = ts / tpi, srem = ts - sil * tpi, styl = srem / a.TXB, stxl = srem - styl * a.TXB;
    const int praw = (ts < a.IB * tpi) ? ((sil * RH + 2 * styl) * RW + 2 * stxl + h) * SRLD + g * 4 : g * 4;

    f32x4 rreg[RPT];
    auto raw_load = [&](int cc) {
        const int c = cc < nchunks ? cc : nchunks - 1;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int off = roff[q] < 0 ? 0 : roff[q];
            rreg[q] = *reinterpret_cast<const f32x4*>(a.x + (long long)off + c * SK);
        }
    };
    auto raw_write = [&](int cc) {                      // -> R[cc & 1], statistics S[cc & 1]
        float* dst = sR + (cc & 1) * R_DW;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            f32x4 x = rreg[q];
            if constexpr (NORM) {
                const f32x4 mu = *reinterpret_cast<const f32x4*>(sS + (cc & 1) * 512 + rsto[q]);
                const f32x4 rs = *reinterpret_cast<const f32x4*>(sS + (cc & 1) * 512 + rsto[q] + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = fmaxf((x[e] - mu[e]) * rs[e], 0.f);
            }
            const bool ok = roff[q] >= 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = ok ? x[e] : 0.f;
            if (roff[q] != -2) *reinterpret_cast<f32x4*>(dst + rlds[q]) = x;
        }
    };
    auto xi_out = [&](unsigned* vdst, int xi, const f32x4 (&T)[3]) {
        f32x4 o0, o1;
        if (h == 0) { o0 = T[0] - T[2]; o1 = T[1] + T[2]; }
        else { o0 = T[1] - T[0]; o1 = T[0] - T[2]; }
        const int pos = xi * 4 + 2 * h;
        u32x2_t ph, pm, pl;
        unsigned* d0 = vdst + (pos * WTT + ts) * SVLD + g * 2;
        cut4(o0, ph, pm, pl);
        *reinterpret_cast<u32x2_t*>(d0) = ph;
        *reinterpret_cast<u32x2_t*>(d0 + 8) = pm;
        *reinterpret_cast<u32x2_t*>(d0 + 16) = pl;
        unsigned* d1 = d0 + WTT * SVLD;
        cut4(o1, ph, pm, pl);
        *reinterpret_cast<u32x2_t*>(d1) = ph;
        *reinterpret_cast<u32x2_t*>(d1 + 8) = pm;
        *reinterpret_cast<u32x2_t*>(d1 + 16) = pl;
    };
    // half (xi rows {2 half, 2 half + 1}) of the transform of chunk cc: R[cc & 1] -> V[cc & 1]; column by column, so only
    // the two rows' column sums stay live (24 registers), not the 3 x 3 window
    auto transform_half = [&](int cc, int half) {
        const float* r = sR + (cc & 1) * R_DW + praw;
        unsigned* vdst = sV + (cc & 1) * SV_DW;
        f32x4 TA[3], TB[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {                   // rows half, half+1, half+2 of the 4 x 3 window
            const f32x4 Ra = *reinterpret_cast<const f32x4*>(r + ((half + 0) * RW + c) * SRLD);
            const f32x4 Rb = *reinterpret_cast<const f32x4*>(r + ((half + 1) * RW + c) * SRLD);
            const f32x4 Rc = *reinterpret_cast<const f32x4*>(r + ((half + 2) * RW + c) * SRLD);
            if (half == 0) { TA[c] = Ra - Rc; TB[c] = Rb + Rc; }     // xi 0: R0 - R2, xi 1: R1 + R2
            else { TA[c] = Rb - Ra; TB[c] = Ra - Rc; }               // xi 2: R2 - R1, xi 3: R1 - R3
        }
        xi_out(vdst, 2 * half, TA);
        xi_out(vdst, 2 * half + 1, TB);
    };
    // ---- MFMA job
    const unsigned short* ub16 = reinterpret_cast<const unsigned short*>(a.u);
    const long long uplane = (long long)a.Cout * SK;
    const long long uchunk = 3 * uplane;
    const long long upos = (long long)nchunks * uchunk;
    const unsigned short* ubase = ub16 + (long long)(8 * wp) * upos + (long long)(n0 + wn * 32) * SK;
    const int ulane = l31 * SK + 8 * hi;
    auto uload = [&](int p, int cc, u32x4_t (&w)[3]) {
        const unsigned short* q = ubase + p * upos + cc * uchunk + ulane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const u32x4_t*>(q + pl * uplane);
    };
    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    const int vfo = ((8 * wp) * WTT + l31) * SVLD + 4 * hi;   // dwords
    union Frag { u32x4_t u; bf16x8_t v; };
    u32x4_t w[UD + 1][3];
    auto mfma_job = [&](int cc) {                       // the first UD fragments of chunk cc were requested at the end of the staging job
        const unsigned* vsrc = sV + (cc & 1) * SV_DW + vfo;
        Frag vq[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) vq[0][pl].u = *reinterpret_cast<const u32x4_t*>(vsrc + pl * 8);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (p + UD < 8) uload(p + UD, cc, w[(p + UD) % (UD + 1)]);
            Frag wb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                if (p < 7) vq[(p + 1) & 1][pl].u = *reinterpret_cast<const u32x4_t*>(vsrc + (p + 1) * WTT * SVLD + pl * 8);
                wb[pl].u = w[p % (UD + 1)][pl];
            }
            __builtin_amdgcn_sched_barrier(0);
            const Frag (&v)[3] = vq[p & 1];
            if constexpr (NP == 9) {
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[2].v, wb[2].v, acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[1].v, wb[2].v, acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[2].v, wb[1].v, acc[p], 0, 0, 0);
            }
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[2].v, wb[0].v, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[0].v, wb[2].v, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[1].v, wb[1].v, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[1].v, wb[0].v, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[0].v, wb[1].v, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[0].v, wb[0].v, acc[p], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // staging job of chunk index c: raw(c+2) registers -> R[c&1] first (frees the registers), the loads of raw(c+3) next
    // (they have the whole transform to land), this group's half of transform(c+1), group A's statistics hand-over, and
    // last the first UD weight fragments of the MFMA job that follows (chunk unext)
    auto stage_job = [&](int c, int unext) {
        if (c + 2 < nchunks) raw_write(c + 2);
        if (c + 3 < nchunks) raw_load(c + 3);
        if (c + 1 < nchunks) transform_half(c + 1, grp);
        if (grp == 0) {                                  // statistics(c+3) -> S[(c+3)&1] (last read by raw_write(c+1)), request (c+4)
            if (c + 3 < nchunks) stat_write(c + 3);
            if (c + 4 < nchunks) stat_load(c + 4);
        }
        if (unext < nchunks) {
#pragma unroll
            for (int p = 0; p < UD; ++p) uload(p, unext, w[p]);
        }
    };

    // ---- prologue (both groups in phase): V[0] = T(0), R[1] = raw(1), registers = raw(2), S[0] = statistics(2), A: (3) pending
    raw_load(0); stat_load(0); stat_write(0);
    __syncthreads();
    raw_write(0);
    if (nchunks > 1) { raw_load(1); stat_load(1); stat_write(1); }
    __syncthreads();
    transform_half(0, grp);
    if (nchunks > 1) raw_write(1);
    if (nchunks > 2) { raw_load(2); stat_load(2); stat_write(2); }   // S[0] was last read before the barrier above
    if (nchunks > 3) stat_load(3);
    if (grp == 0) {
#pragma unroll
        for (int p = 0; p < UD; ++p) uload(p, 0, w[p]);
    }
    __syncthreads();
    // two straight-line programs (no control-flow join inside the loop: joined once per half-step the register allocator
    // sees the SUM of both jobs' live ranges and spills ~400 registers), the same barrier sequence on both
#ifdef DSMIL_TRACE
#define ALT_STAMP(slot)                                                                                             \
    do {                                                                                                            \
        if (a.trace && blockIdx.x < 4 && blockIdx.y == 0 && lane == 0 && (wave & 3) == 0 && c < 256)                \
            a.trace[(((long long)blockIdx.x * 2 + grp) * 256 + c) * 8 + (slot)] = __builtin_amdgcn_s_memtime();     \
    } while (0)
#else
#define ALT_STAMP(slot) do { } while (0)
#endif
    if (grp == 0) {
        for (int c = 0; c < nchunks; ++c) {
            ALT_STAMP(0);
            mfma_job(c);                // half-step 2c
            ALT_STAMP(1);
            __syncthreads();
            ALT_STAMP(2);
            stage_job(c, c + 1);        // half-step 2c+1
            ALT_STAMP(3);
            __syncthreads();
            ALT_STAMP(4);
        }
    } else {
        for (int c = 0; c < nchunks; ++c) {
            ALT_STAMP(0);
            stage_job(c, c);            // half-step 2c
            ALT_STAMP(1);
            __syncthreads();
            ALT_STAMP(2);
            mfma_job(c);                // half-step 2c+1
            ALT_STAMP(3);
            __syncthreads();
            ALT_STAMP(4);
        }
    }
    wino_epilogue(a, acc, smem, lane, wn, wp, n0, img0, ty0, tx0, tpi, pb);
}

// --------------------------------------------------------------------------------------------
// k_conv_wino_pp — the Winograd unit as a PERSISTENT, role-split, software-pipelined workgroup (512 threads, one per CU):
//   waves 0-3  multiply: the MFMAs of step s (8 positions x NP plane products, V fragments from LDS one position
//              ahead, weight fragments straight from L2 UD positions ahead in a register ring that runs on across
//              chunk, unit and cout-tile boundaries) and the unit epilogue; their s_waitcnt vmcnt stream holds
//              nothing but weight loads, so no activation load with HBM latency ever queues in front of a fragment;
//   waves 4-7  stage: transform(s+1) raw -> V planes, raw(s+2) registers -> IN + ReLU + padding -> LDS, global loads of
//              raw(s+3), the producer's statistics of step s+3 — a whole step of slack on every dependent chain.
// A step is one 16-channel chunk of one (unit, cout tile) item; each workgroup walks items first, first+G, ... so the
// stream of steps never drains between units: the per-unit prologue (two global round trips + a transform) that
// k_conv_wino_s3 pays with the MFMA pipe idle exists once per LAUNCH here.  V and raw are double buffered
// (2 x 56 KB + 2 x 20 KB + 4 KB statistics = 156 KB); ONE workgroup barrier per step; at the end of a unit two more
// around the epilogue's cross-wave exchange (it borrows the V buffer the step just finished with).
// --------------------------------------------------------------------------------------------
#ifndef PP_UD
#define PP_UD 3
#endif
template <bool NORM, int NP, int UD>
__global__ __launch_bounds__(512, 2) void k_conv_wino_pp(WinoArgs a, int nunits, int nitems) {
    static_assert(UD == 3 || UD == 7, "the fragment ring (UD + 1 slots) must divide the 8 positions of a step");
    constexpr int RING = UD + 1;                        // weight prefetch distance in positions: ring of UD + 1 slots
    constexpr int R_DW = WRAW_MAX * SRLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sV = reinterpret_cast<unsigned*>(smem);   // [2][SV_DW]
    float* sR = smem + 2 * SV_DW;                       // [2][R_DW]
    float* sS = sR + 2 * R_DW;                          // [2][16 images][2 (mean, rstd)][16 ch]
    const int tid = threadIdx.x & 255, lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool stager = wave8 >= 4;
    const int wave = wave8 & 3;
    const int nchunks = a.C / SK;
    const int first = blockIdx.x, stride = gridDim.x;
    const int nmine = (nitems - first + stride - 1) / stride;
    const int S = nmine * nchunks;                      // steps of this workgroup
    const int RH = 2 * a.TYB + 2, RW = 2 * a.TXB + 2, RP = RH * RW;
    const int tpi = a.TYB * a.TXB;
    struct Item { int img0, ty0, tx0, pb, n0; };
    auto decode = [&](int j) {                          // j-th item of this workgroup (uniform)
        j = j < nmine ? j : nmine - 1;
        const int i = first + j * stride;
        const int ct = i / nunits;
        int u = i - ct * nunits;
        const int bx = u % a.nbx; u /= a.nbx;
        const int by = u % a.nby; u /= a.nby;
        return Item{u * a.IB, by * a.TYB, bx * a.TXB, by * a.nbx + bx, ct * 64};
    };

    if (stager) {
        // ---- raw staging role: element e = tid + 256 q -> (pixel, channel group), as in k_conv_wino_s3
        int rgeo[SRPT], rlds[SRPT], rsto[SRPT], roff[SRPT];
#pragma unroll
        for (int q = 0; q < SRPT; ++q) {
            const int e = tid + 256 * q, px = (e & 7) | ((e >> 5) << 3), gg = (e >> 3) & 3;
            rgeo[q] = -1; rlds[q] = px * SRLD + gg * 4; rsto[q] = gg * 4;
            if (px < a.IB * RP) {
                const int il = px / RP, rem = px - il * RP, ry = rem / RW, rx = rem - ry * RW;
                rgeo[q] = (il << 20) | (ry << 10) | rx;
                rsto[q] = il * 32 + gg * 4;
            }
        }
        auto set_item = [&](const Item& it) {           // global element offsets of this thread's pixels in the item's region
#pragma unroll
            for (int q = 0; q < SRPT; ++q) {
                const int e = tid + 256 * q, gg = (e >> 3) & 3;
                roff[q] = -1;
                if (rgeo[q] >= 0) {
                    const int n = it.img0 + (rgeo[q] >> 20), iy = 2 * it.ty0 - 1 + ((rgeo[q] >> 10) & 1023),
                              ix = 2 * it.tx0 - 1 + (rgeo[q] & 1023);
                    if (n < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) roff[q] = ((n * a.H + iy) * a.W + ix) * a.C + gg * 4;
                }
            }
        };
        // transform role (see k_conv_wino_s3 for the lane -> (g, ts) assignment)
        const int g = tid & 3, ts = ((lane >> 5) & 1) | (((lane >> 2) & 7) << 1) | ((wave & 1) << 4);
        const int h = wave >> 1;
        const int sil = ts / tpi, srem = ts - sil * tpi, styl = srem / a.TXB, stxl = srem - styl * a.TXB;
        const int praw = (ts < a.IB * tpi) ? ((sil * RH + 2 * styl) * RW + 2 * stxl + h) * SRLD + g * 4 : g * 4;

        f32x4 rreg[SRPT];
        unsigned rok = 0;                               // validity of rreg's pixels (captured when they were requested)
        f32x4 sreg = {0.f, 0.f, 0.f, 0.f};
        int jL = 0, ccL = 0;                            // item / chunk of the step the NEXT raw_load requests
        Item itL = decode(0);
        set_item(itL);
        auto advance = [&]() {                          // to the next step of the stream (stays on the last one at the end)
            if (ccL + 1 < nchunks) { ++ccL; return; }
            if (jL + 1 < nmine) { ++jL; ccL = 0; itL = decode(jL); set_item(itL); }
        };
        auto raw_load = [&]() {
            rok = 0;
#pragma unroll
            for (int q = 0; q < SRPT; ++q) {
                const int off = roff[q] < 0 ? 0 : roff[q];
                rok |= roff[q] >= 0 ? (1u << q) : 0u;
                rreg[q] = *reinterpret_cast<const f32x4*>(a.x + (long long)off + ccL * SK);
            }
        };
        auto stat_load = [&]() {
            if constexpr (NORM) {
                if (tid < 128) {
                    const int il = tid >> 3, which = (tid >> 2) & 1, c4 = tid & 3;
                    const int n = itL.img0 + il < a.B ? itL.img0 + il : a.B - 1;
                    sreg = *reinterpret_cast<const f32x4*>((which ? a.in_rstd : a.in_mean) + (long long)n * a.C + ccL * SK + c4 * 4);
                }
            }
        };
        auto stat_write = [&](int b) {
            if constexpr (NORM) {
                if (tid < 128) {
                    const int il = tid >> 3, which = (tid >> 2) & 1, c4 = tid & 3;
                    *reinterpret_cast<f32x4*>(sS + b * 512 + il * 32 + which * 16 + c4 * 4) = sreg;
                }
            }
        };
        auto raw_write = [&](int rb, int sb) {          // rreg -> IN + ReLU + zero padding -> raw buffer rb (statistics buffer sb)
            float* dst = sR + rb * R_DW;
#pragma unroll
            for (int q = 0; q < SRPT; ++q) {
                f32x4 x = rreg[q];
                if constexpr (NORM) {
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(sS + sb * 512 + rsto[q]);
                    const f32x4 rs = *reinterpret_cast<const f32x4*>(sS + sb * 512 + rsto[q] + 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = fmaxf((x[e] - mu[e]) * rs[e], 0.f);
                }
                const bool ok = (rok >> q) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = ok ? x[e] : 0.f;
                if (rgeo[q] >= 0) *reinterpret_cast<f32x4*>(dst + rlds[q]) = x;
            }
        };
        auto transform = [&](int rb, int vb) {          // raw buffer rb -> V buffer vb (planes)
            const float* r = sR + rb * R_DW + praw;
            unsigned* vdst = sV + vb * SV_DW;
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
                f32x4 o0, o1;
                if (h == 0) { o0 = T[xi][0] - T[xi][2]; o1 = T[xi][1] + T[xi][2]; }
                else { o0 = T[xi][1] - T[xi][0]; o1 = T[xi][0] - T[xi][2]; }
                const int pos = xi * 4 + 2 * h;
                u32x2_t ph, pm, pl;
                unsigned* d0 = vdst + (pos * WTT + ts) * SVLD + g * 2;
                cut4(o0, ph, pm, pl);
                *reinterpret_cast<u32x2_t*>(d0) = ph;
                *reinterpret_cast<u32x2_t*>(d0 + 8) = pm;
                *reinterpret_cast<u32x2_t*>(d0 + 16) = pl;
                unsigned* d1 = d0 + WTT * SVLD;
                cut4(o1, ph, pm, pl);
                *reinterpret_cast<u32x2_t*>(d1) = ph;
                *reinterpret_cast<u32x2_t*>(d1 + 8) = pm;
                *reinterpret_cast<u32x2_t*>(d1 + 16) = pl;
            }
        };
        // ---- prologue: state at the top of step 0 = { V[0] = T(0), R[1] = raw(1), rreg = raw(2), S[0] = statistics(2) }
        raw_load(); stat_load(); stat_write(0);
        __syncthreads();                                // P0
        raw_write(0, 0);
        advance(); raw_load(); stat_load(); stat_write(1);
        __syncthreads();                                // P1
        transform(0, 0);
        raw_write(1, 1);
        advance(); raw_load(); stat_load(); stat_write(0);   // S[0] was last read before P1
        __syncthreads();                                // P2
        int ccC = 0;
        for (int s = 0; s < S; ++s) {
            const int b = s & 1;
            PP_STAMP(0);
            advance();                                  // -> step s+3
            stat_load();
            transform(b ^ 1, b ^ 1);                    // raw(s+1) -> V[(s+1)&1]
            PP_STAMP(1);
            raw_write(b, b);                            // raw(s+2): R[s&1], statistics S[s&1]
            PP_STAMP(2);
            raw_load();                                 // raw(s+3)
            stat_write(b ^ 1);                          // statistics(s+3) -> S[(s+1)&1] (last read in step s-1)
            PP_STAMP(3);
            __syncthreads();
            PP_STAMP(4);
            if (++ccC == nchunks) {                     // the multiply waves' epilogue exchange
                ccC = 0;
                __syncthreads();
                __syncthreads();
            }
        }
        return;
    }

    // ================= multiply waves =========================================================================
    const int wn = wave & 1, wp = wave >> 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned short* ub16 = reinterpret_cast<const unsigned short*>(a.u);
    const long long uplane = (long long)a.Cout * SK;
    const long long uchunk = 3 * uplane;
    const long long upos = (long long)nchunks * uchunk;
    const int ulane = l31 * SK + 8 * hi;
    auto ustep = [&](const Item& it, int cc) {          // this wave's fragment base of a step
        return ub16 + (long long)(8 * wp) * upos + (long long)(it.n0 + wn * 32) * SK + cc * uchunk + ulane;
    };
    auto uload = [&](const unsigned short* base, int p, u32x4_t (&w)[3]) {
        const unsigned short* q = base + p * upos;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const u32x4_t*>(q + pl * uplane);
    };
    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    u32x4_t w[RING][3];
    int jC = 0, ccC = 0;
    Item itC = decode(0);
    const unsigned short* up0 = ustep(itC, 0);
#pragma unroll
    for (int p = 0; p < UD; ++p) uload(up0, p, w[p]);
    const int vfo = ((8 * wp) * WTT + l31) * SVLD + 4 * hi;   // dwords
    union Frag { u32x4_t u; bf16x8_t v; };
    __syncthreads();                                    // P0
    __syncthreads();                                    // P1
    __syncthreads();                                    // P2
    for (int s = 0; s < S; ++s) {
        // the step after this one (for the fragments requested late in this one); stays on the last step at the end
        int jN = jC, ccN = ccC + 1;
        Item itN = itC;
        if (ccN == nchunks) {
            if (jC + 1 < nmine) { jN = jC + 1; ccN = 0; itN = decode(jN); } else ccN = ccC;
        }
        PP_STAMP(0);
        const unsigned short* up1 = ustep(itN, ccN);
        const unsigned* vsrc = sV + (s & 1) * SV_DW + vfo;
        Frag va[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) va[0][pl].u = *reinterpret_cast<const u32x4_t*>(vsrc + pl * 8);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            // ring slot (p + UD) % RING = the one position p - 1 just released
            if (p + UD < 8) uload(up0, p + UD, w[(p + UD) % RING]); else uload(up1, p + UD - 8, w[(p + UD) % RING]);
            if (p < 7) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) va[(p + 1) & 1][pl].u = *reinterpret_cast<const u32x4_t*>(vsrc + (p + 1) * WTT * SVLD + pl * 8);
            }
            // keep the requests where they are written: left alone, the scheduler sinks each load to just before its use
            // (prefetch distance ~1 position, every MFMA group behind an s_waitcnt)
            __builtin_amdgcn_sched_barrier(0);
            Frag wb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wb[pl].u = w[p % RING][pl];
            const Frag (&v)[3] = va[p & 1];
            // smallest products first: (l,l) (m,l) (l,m) | (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
            if constexpr (NP == 9) {
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[2].v, wb[2].v, acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[1].v, wb[2].v, acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[2].v, wb[1].v, acc[p], 0, 0, 0);
            }
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[2].v, wb[0].v, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[0].v, wb[2].v, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[1].v, wb[1].v, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[1].v, wb[0].v, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[0].v, wb[1].v, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[0].v, wb[0].v, acc[p], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (p == 3) PP_STAMP(1);
        }
        PP_STAMP(2);
        __syncthreads();
        PP_STAMP(3);
        if (ccC + 1 == nchunks) {                       // unit done: inverse transform, exchange through V[s&1], store
            wino_epilogue(a, acc, reinterpret_cast<float*>(sV + (s & 1) * SV_DW), lane, wn, wp, itC.n0, itC.img0, itC.ty0, itC.tx0,
                          tpi, itC.pb);
            __syncthreads();                            // the exchange buffer is V again
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
            PP_STAMP(4);
        }
        jC = jN; ccC = ccN; itC = itN; up0 = up1;
    }
}

