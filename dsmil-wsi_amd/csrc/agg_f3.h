// agg_f3.h — k_attend_f3: k_attend_f2's arithmetic (fp32 bags, two fp16 planes by round-to-nearest per operand, three plane
// products on v_mfma_f32_32x32x16_f16, every feature byte read ONCE) with the query weights RESIDENT IN REGISTERS.
//
// Why (round 5, DESIGN.md §3 "what bounds the tile").  k_attend_f2 pulls its weight planes from L2 for every 64-row tile:
// 320 KB of weights next to 128 KB of features through the CU's 64 B/clk vector L1 — the tile's bound, whatever the grid size
// (the launch on 64 workgroups takes the same time per tile as on 256).  Taller tiles would amortise the weights but do not
// fit LDS.  k_attend_bf16_res (agg_res.h) showed the way out for bf16 bags: one wave per SIMD may hold 512 registers, so
//   * one 256-thread workgroup per CU; wave w keeps the A fragments of ITS 32 hidden / query units for the whole launch:
//     W1 (two fp16 planes x K <= 512: up to 256 registers) in the ACCUMULATOR half of the register file, W2's first plane in
//     VGPRs (its second plane too up to K = 256; at K = 512 it is re-read from L2 once per tile, under the hidden-layer
//     exchange: 32 KB).  The MFMAs are inline asm with "a" weight operands and VGPR accumulators (k_attend_bf16_res's recipe;
//     f3_mfma_* below say why).  No W1 byte moves after the prologue, so a short tile costs nothing:
//   * tile = 32 rows, TWO plane buffers in LDS (row-major [row][plane][K] fp16 with a 16-B pad: B fragments, the value
//     sum's k-contiguous reads and the cut's writes are all conflict-free).  While tile t is in GEMM 1, every wave cuts
//     ITS eight rows of tile t+1 out of a register ring (filled a tile earlier, as k_attend_f2's cutters do) into the other
//     buffer and refills the ring with tile t+2 — the cut in two-instruction pieces BETWEEN the 96 MFMAs (a wave issues in
//     order: a whole group behind three queued MFMAs left the matrix pipe idle), then the plane writes, then the loads;
//   * a workgroup owns a CONTIGUOUS run of tiles (k_attend_bf16_res's scheme): consecutive tiles belong to the same bag, the
//     softmax reference is a constant of the bag (tanh bounds the queries: |s| <= sum_j |q_max[j]| / sqrt(128) — no tile
//     maximum, no rescaling), the value sum accumulates ACROSS tiles in eight registers per class (lane = k-octet, wave =
//     eight rows of every tile: no cross-lane reduction per tile) and ONE partial per (workgroup, bag) is written:
//     slot = blockIdx.x + bag, merged by k_finish's segment mode.
//   * the hidden planes are scaled by a BOUND of the row's hidden maximum (weight norm x the row's feature maximum; the norm
//     rides in the packed image's trailer) instead of the maximum itself, which would have to be exchanged between the waves:
//     a power-of-two scale leaves fp16 significands alone (tests/test_f16_planes.py), so the results do not change.
// Per-row power-of-two feature scales, the cut, the three products, the packed weight image (k_pack_agg_f2) and the error class
// are k_attend_f2's (agg_f2.h); so are the tests (tests/test_agg_gpu.py::test_batch_form_*).
// Only the two-layer query (dsmil.py:31-32 nonlinear, the default) and C <= 2: everything else stays on k_attend_f2.
// LDS: planes 2 x 32 x (4 K + 16) B (129 KiB at K = 512) | hidden planes 32 x 528 B | 6.3 KiB scratch = 151.8 KiB.
// Barriers per 32-row tile: S (planes of this tile complete, the other buffer and the scratch released), B2 (hidden
// planes), T1 (partial scores).
#pragma once
#include "agg_f2.h"

namespace {

constexpr int F3_BM = 32;                   // rows per tile
constexpr int F3_THREADS = 256;             // one wave per SIMD
constexpr int F3_MAX_WG = 1024;             // (= RS_MAX_WG: the workspace holds that many + n_bags partial slots)
constexpr int F3_HROW = 528;                // bytes per row of the hidden planes: 2 planes x 256 B + 16 pad
constexpr int F3_SCR = 1600;                // floats of scratch
__host__ __device__ constexpr int f3_row_bytes(int K) { return 4 * K + 16; }
__host__ __device__ constexpr int f3_lds_bytes(int K) { return 2 * F3_BM * f3_row_bytes(K) + F3_BM * F3_HROW + F3_SCR * 4; }

// v_mfma_f32_32x32x16_f16 as inline asm (k_attend_bf16_res's recipe, agg_res.h): the A operand in the accumulator file ("a":
// the resident W1) or in a VGPR (W2), the accumulator in VGPRs.  With the builtin hipcc put the ACCUMULATORS into the
// accumulator file, moved W1 fragments out through v_mov copies and — its scheduler in minimum-pressure mode — issued every LDS
// read right in front of its MFMA (stamps: 5 200 cycles for the 96 MFMAs of a tile).  Inside an asm statement nothing is padded:
// *0 variants start a chain from the inline constant 0, F3_NOP() stands between the last MFMA and the VALU reads of its result.
__device__ __forceinline__ void f3_mfma_a(f32x16& acc, const f32x4& a_agpr, const f32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a_agpr), "v"(b));
}
__device__ __forceinline__ void f3_mfma_a0(f32x16& acc, const f32x4& a_agpr, const f32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(a_agpr), "v"(b));
}
__device__ __forceinline__ void f3_mfma_v(f32x16& acc, const f32x4& a, const f32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void f3_mfma_v0(f32x16& acc, const f32x4& a, const f32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
#define F3_NOP() asm volatile("s_nop 15" ::: "memory")   // 16 states >= the 12 an 8-pass MFMA result needs before a VALU read

// f(integral_constant<int, I>) for I = B .. E-1: every index inside the body is a constant expression (register arrays stay
// registers whatever the optimiser's pass order)
template <int B, int E, class F>
__device__ __forceinline__ void f3_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        f3_static_for<B + 1, E>(f);
    }
}

struct F3Work {
    int bag;
    long long off0, Nb, row0;
};

// position in the (bag, tile) item list: item = b * tiles_per_bag + tile, kept WITHOUT a division per step (the quotient by a
// run-time divisor is ~40 instructions; the lookahead runs once per tile in front of a barrier)
struct F3Cur {
    int item, b, tile;
};
// (tiles_per_bag == 0: a ragged batch — the items are the real tiles, a bag ends where its rows end: f3_fetch moves on)
__device__ __forceinline__ F3Cur f3_next(F3Cur c, int tiles_per_bag) {
    ++c.item;
    if (++c.tile == tiles_per_bag) { c.tile = 0; ++c.b; }
    return c;
}

// first item at or behind c, below `end`, whose tile lies inside its bag (evaluated identically by every wave); c is left on it
// (hint = the item in front: consecutive tiles mostly share the bag, whose record then needs no load at all)
__device__ __forceinline__ bool f3_fetch(const AttendArgs& a, int tiles_per_bag, int end, F3Cur& c, F3Work& w, const F3Work* hint = nullptr) {
    // (through the constant address space: scalar loads.  As plain loads behind the tile's stores hipcc makes them VECTOR loads
    // with a uniform address, and their vmcnt(0) then waits for every refill of the feature ring issued before them)
    const __attribute__((address_space(4))) long long* offs = (const __attribute__((address_space(4))) long long*)(uintptr_t)a.offsets;
    while (c.item < end) {
        const int bag = a.bag0 + c.b;
        long long off0, Nb;
        if (hint && hint->bag == bag) { off0 = hint->off0; Nb = hint->Nb; }
        else { off0 = offs[bag]; Nb = offs[bag + 1] - off0; }
        const long long row0 = (long long)c.tile * F3_BM;
        if (row0 < Nb) {
            w.bag = bag; w.off0 = off0; w.Nb = Nb; w.row0 = row0;
            return true;
        }
        if (tiles_per_bag) c.item += tiles_per_bag - c.tile;   // the rest of this bag's items lie behind its end (ragged: there are none)
        c.tile = 0;
        ++c.b;
    }
    return false;
}

// NK1 = K / 32 in {4, 8, 12, 16}; a.wpk = the image of k_pack_agg_f2; rowmax[logical row] = max_k |x| (k_logits_stream);
// a.nonlinear, a.C <= 2, vals == feats.  Workgroup g owns the tile items [g per_wg, (g + 1) per_wg) of the (bag, tile) list
// (tiles_per_bag items per bag) and writes partial slot g + bag for every bag it touches.
template <int NK1, bool TWO, int DBG = 0>   // DBG 1 (experiment builds, with DSMIL_EXPT=64: k_finish skipped): wave 0 stores s_memtime stamps
                                            // of the phase boundaries into the tile's 32 rows of A (C = 1; tools/f3_stamps.py)
__global__ __launch_bounds__(F3_THREADS, 1) void k_attend_f3(AttendArgs a, const float* __restrict__ rowmax, int tiles_per_bag,
                                                             int n_items, int per_wg) {
    static_assert(NK1 % 4 == 0 && NK1 >= 4 && NK1 <= 16, "K a multiple of 128 up to 512");
    constexpr int K = 32 * NK1, NKS = 2 * NK1, NG = NK1 / 2;   // 16-k steps of GEMM 1; 64-k groups of the feature ring
    constexpr int RB = f3_row_bytes(K), BUF = F3_BM * RB;
    constexpr int NC = TWO ? 2 : 1;
    constexpr bool W2_STREAM = NK1 > 8;
    constexpr int NMOVE = 2;                               // ring groups refilled from GEMM 2's gaps instead of GEMM 1's (measured, kernel time relative to k_attend_f2 on the same box: 1 -> 0.84, 2 -> 0.81, 3 -> 0.83, 4 -> 0.83)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sX = reinterpret_cast<char*>(smem);              // [2 buffers][32 rows][plane 2][K] fp16 + pad
    char* sH = sX + 2 * BUF;                               // [32 rows][plane 2][128] fp16 + pad; at a flush: [4 waves][NC][K] floats
    float* scr = reinterpret_cast<float*>(sH + F3_BM * F3_HROW);
    // (partial results of the two lane halves go to LDS side by side: an exchange between the halves is a ds_bpermute round trip
    // in the middle of a dependent chain, two per tile)
    float* sS = scr;            // [4 waves][2 halves][2 classes][32 rows] partial scores
    float* sBias = scr + 768;   // [2][128]: q.0 / q.2 biases
    float* sPall = scr + 1024;  // [4 waves][2 classes][32 rows]: every wave's private value-sum weights p / row scale
    float* sInvAll = scr + 1280; // [2 buffers][32 rows]: 1 / row scale of the rows whose planes sit in that buffer
    float* sQ = scr + 1344;     // [2 classes][128]: critical queries of the current bag
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const f32x4* wimg = reinterpret_cast<const f32x4*>(a.wpk);
    const float* trailer = reinterpret_cast<const float*>(wimg + (long long)(NKS + 8) * F2_CHUNK_F4);
    const float* feats = reinterpret_cast<const float*>(a.feats);
    const float scale = 0.08838834764831845f;              // 1/sqrt(128), dsmil.py:56

    F3Cur pos;
    pos.item = (int)blockIdx.x * per_wg;
    if (a.tile_pre) {                                     // ragged batch: the real tiles only (tiles_per_bag == 0 from the host)
        const int real = ((const __attribute__((address_space(4))) int*)(uintptr_t)a.tile_pre)[a.n_bags];
        n_items = n_items < real ? n_items : real;        // (the host sized the runs from an upper bound)
        if (pos.item >= n_items) return;
        pos.b = tile_owner(a.tile_pre, a.n_bags, pos.item);
        pos.tile = pos.item - a.tile_pre[pos.b];
    } else {
        pos.b = pos.item / tiles_per_bag;
        pos.tile = pos.item - pos.b * tiles_per_bag;
    }
    const int item_end = pos.item + per_wg < n_items ? pos.item + per_wg : n_items;
    F3Work cur, nxt, nn;
    if (!f3_fetch(a, tiles_per_bag, item_end, pos, cur)) return;   // (block-uniform)

    // ---- resident weights: this wave's A fragments of every 16-k step, both planes.  W1 is pinned to the accumulator file
    //      (an MFMA may take its A operand from there; the empty asm makes the tuple live there for the whole launch), W2 and
    //      everything the VALU touches stay in VGPRs.
    F2Frag w1[NKS][2], w2[8][2];
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) w1[s][p].f = wimg[(long long)s * F2_CHUNK_F4 + (2 * wave + p) * 64 + lane];
    // (W2's second plane is NOT resident at K = 512: 2 x 256 + 64 weight registers left too few for the LDS reads of GEMM 1 to
    // run ahead of their MFMAs — hipcc issued every read right in front of its MFMA.  Its 32 KB per tile come from L2 behind
    // GEMM 1, under the hidden-layer exchange.)
    const f32x4* w2p1 = wimg + (long long)NKS * F2_CHUNK_F4 + (2 * wave + 1) * 64 + lane;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        w2[st][0].f = wimg[(long long)(NKS + st) * F2_CHUNK_F4 + (2 * wave) * 64 + lane];
        if constexpr (!W2_STREAM) w2[st][1].f = w2p1[(long long)st * F2_CHUNK_F4];
    }
    const float ia1 = trailer[0], ia2 = trailer[1];
    // Bound of a row's hidden layer: relu(W1 x + b)[j] <= ||W1[j]||_1 max|x| + |b[j]|, and max|x| < 2^14 / (row scale).  The
    // hidden planes are scaled by THIS bound (a power of two from it) instead of by the row's true maximum: no exchange of
    // partial maxima between the waves, one barrier less per tile.  fp16 is a floating-point format — a loose scale costs
    // exponent range, not significand bits: with the bound 2^6 above the true maximum every element within 2^-10 of the
    // maximum still has both planes normal (22 bits), smaller ones an absolute error <= 2^-24 of the scaled unit, i.e. < 2^-33 of
    // the row's maximum.  (The exchange + barrier were ~600 of a tile's 9 700 cycles.)
    const float hb_n1 = trailer[2] * 16384.f;
    const float hb_b = wave_max(fmaxf(fabsf(a.q0_b[lane]), fabsf(a.q0_b[lane + 64])));
    sBias[tid] = tid < QD ? a.q0_b[tid] : a.q2_b[tid - QD];
#pragma unroll
    for (int s = 0; s < NKS; ++s)                         // (behind ALL the loads: a pin waits for its tuple)
#pragma unroll
        for (int p = 0; p < 2; ++p) asm volatile("" : "+a"(w1[s][p].f));

    // ---- the feature stream of this wave: rows 8 wave .. 8 wave + 7 of every tile; lane (rr = lane & 7, o = lane >> 3) holds
    //      the k-octet o of every 64-k group of row 8 wave + rr
    const int rr = lane & 7, o = lane >> 3;
    const int myrow = 8 * wave + rr;
    f32x4 ring[NG][2];
    // row lookups of a tile for this lane: the row's max |x| and its physical row (two loads; row_done turns them into the
    // scale and the source pointer — kept apart so that the caller decides where the wait goes)
    auto row_raw = [&](const F3Work& w, float& rm, long long& phys, long long& logical) {
        long long gr = w.row0 + myrow;
        if (gr >= w.Nb) gr = w.Nb - 1;                    // rows past the bag end are cut like the last row, weight 0
        rm = rowmax[w.off0 + gr];
        // (branch-free row map: a load behind a branch is waited for at the join — with the identity map the load reads a valid
        // dummy word instead)
        const long long* mp = a.rowmap ? reinterpret_cast<const long long*>(a.rowmap) + (w.off0 + gr) : reinterpret_cast<const long long*>(a.offsets);
        phys = *mp;                                       // (the mapped row, or the dummy word)
        logical = w.off0 + gr;
    };
    auto row_done = [&](float rm, long long phys, long long logical, float& sc, float& sinv) -> const float* {
        sc = f2_scale(rm, sinv);
        return feats + (a.rowmap ? phys : logical) * (long long)K + 8 * o;
    };
    auto row_src = [&](const F3Work& w, float& sc, float& sinv) -> const float* {
        float rm;
        long long phys, logical;
        row_raw(w, rm, phys, logical);
        asm volatile("" : "+v"(phys));
        return row_done(rm, phys, logical, sc, sinv);
    };
    auto fill = [&](const float* src, int c) {
        ring[c][0] = *(const DSMIL_GLOBAL f32x4*)(src + 64 * c);
        ring[c][1] = *(const DSMIL_GLOBAL f32x4*)(src + 64 * c + 4);
    };
    // group c of the ring -> the two planes of its eight values, into plane buffer pb
    char* const cut_dst = sX + myrow * RB + 16 * o;
    auto cut_write = [&](int pb, int c, float sc) {
        F2Frag f[2];
        split2h_scaled(ring[c][0], ring[c][1], sc, f);
        char* d = cut_dst + pb * BUF + 128 * c;
        *reinterpret_cast<f32x4*>(d) = f[0].f;
        *reinterpret_cast<f32x4*>(d + 2 * K) = f[1].f;
    };

    // the same cut in pieces that fit into the gaps between the MFMAs of GEMM 1 (a wave issues in order: a 16-instruction cut
    // behind three queued MFMAs leaves the matrix pipe idle — stamps: 250 cycles per group, 2 000 per tile; four v_fma_mix
    // per gap still 150 per group): two v_fma_mix per gap, then the two plane writes, then the refill
    F2Frag cutf[2];
    auto cut_piece = [&](int pb, auto s_c, auto j_c, float sc, const float* refill_src) {   // behind MFMA j of step s
        constexpr int s_ = decltype(s_c)::value, j = decltype(j_c)::value;
        constexpr int c = s_ / 4, ph = (s_ % 4) * 3 + j;  // group c owns the twelve gaps of steps 4c .. 4c+3
        if constexpr (c < NG) {
            char* d = cut_dst + pb * BUF + 128 * c;
            if constexpr (ph < 8) {                       // pair ph / 2: its first plane (even ph), its second (odd ph) — two v_fma_mix per gap
                constexpr int i = ph / 2;
                const float va = i < 2 ? ring[c][0][(2 * i) & 3] : ring[c][1][(2 * i) & 3];
                const float vb = i < 2 ? ring[c][0][(2 * i + 1) & 3] : ring[c][1][(2 * i + 1) & 3];
                if constexpr (ph % 2 == 0) {
                    unsigned h;
                    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(h) : "v"(va), "v"(vb), "v"(sc));
                    cutf[0].u[i] = h;
                } else {
                    unsigned l;
                    const unsigned h = cutf[0].u[i];
                    asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                        : "=&v"(l) : "v"(va), "v"(vb), "v"(sc), "v"(h));
                    cutf[1].u[i] = l;
                }
            } else if constexpr (ph == 8) {
                *reinterpret_cast<f32x4*>(d) = cutf[0].f;
            } else if constexpr (ph == 9) {
                *reinterpret_cast<f32x4*>(d + 2 * K) = cutf[1].f;
            } else if constexpr (ph == 10) {
                // (the last NMOVE groups are refilled from GEMM 2's gaps: the loads of W2's second plane are issued between the
                // two GEMMs, and the in-order vmcnt wait for them would otherwise include HBM loads a few hundred cycles old)
                if constexpr (DBG != 4 && c < NG - NMOVE) fill(refill_src, c);   // (DBG 4: timing without the refills)
            }
        }
    };

    float sc_c = 1.f, sinv_c = 1.f, sc_n = 1.f, sinv_n = 1.f, sc_nn = 1.f, sinv_nn = 1.f;
    const float* src_c = row_src(cur, sc_c, sinv_c);
#pragma unroll
    for (int c = 0; c < NG; ++c) fill(src_c, c);
    F3Cur pos_n = f3_next(pos, tiles_per_bag);
    bool has_next = f3_fetch(a, tiles_per_bag, item_end, pos_n, nxt);
    const float* src_n = has_next ? row_src(nxt, sc_n, sinv_n) : src_c;
    if (!has_next) { nxt = cur; sc_n = sc_c; sinv_n = sinv_c; }
    // tile 0: cut into buffer 0, ring <- tile 1
#pragma unroll
    for (int c = 0; c < NG; ++c) {
        cut_write(0, c, sc_c);
        fill(src_n, c);
    }
    if (o == 0) sInvAll[myrow] = sinv_c;
    F3Cur pos_nn = f3_next(pos_n, tiles_per_bag);
    bool has_nn = has_next && f3_fetch(a, tiles_per_bag, item_end, pos_nn, nn);
    const float* src_nn = has_nn ? row_src(nn, sc_nn, sinv_nn) : src_n;
    if (!has_nn) { nn = nxt; sc_nn = sc_n; sinv_nn = sinv_n; }

    // running sums of the bag inside this workgroup's run
    float bacc[NC][8];
    float l_run[NC];
    float m_bag[NC];
    auto reset_acc = [&]() {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            l_run[c] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) bacc[c][e] = 0.f;
        }
    };
    reset_acc();
#pragma unroll
    for (int c = 0; c < NC; ++c) m_bag[c] = 0.f;
    int qbag = -1;
    float* sPw = sPall + wave * 64;

    unsigned long long stamps[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) stamps[i] = 0;
    auto STAMP = [&](int i) {
        if constexpr (DBG >= 1) {                         // (pinned: hipcc otherwise sinks work past the counter read)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(63) lgkmcnt(0)" ::: "memory");
            stamps[i] = __builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int t = 0;; ++t) {
        const int buf = t & 1;
        STAMP(0);
        if (cur.bag != qbag) {
            // a new bag: its critical queries -> LDS (the previous tile's readers are behind its T1; the S barrier below
            // publishes them), the softmax reference from the same values
            __syncthreads();
            if (tid < NC * QD) sQ[tid] = a.qmax[(long long)cur.bag * a.C * QD + tid];
            qbag = cur.bag;
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NC; ++c) m_bag[c] = wave_sum(fabsf(sQ[c * QD + lane]) + fabsf(sQ[c * QD + 64 + lane])) * scale;
        }
        // the tile after the one after next: its record (scalar loads), its rows' maxima and addresses (two vector loads, issued
        // BEFORE this tile's ring refills and consumed at the end of the tile: the wait then leaves the refills in flight)
        F3Cur pos_n3 = f3_next(pos_nn, tiles_per_bag);
        F3Work n3 = nn;
        float rm_n3;
        long long phys_n3, log_n3;
        const bool has_n3 = has_nn && f3_fetch(a, tiles_per_bag, item_end, pos_n3, n3, &nn);
        if (!has_n3) n3 = nn;
        row_raw(n3, rm_n3, phys_n3, log_n3);                    // (unconditional: no tile behind -> a harmless re-read of nn's rows)
        __syncthreads();                                  // S
        STAMP(1);
        const char* xb_ = sX + buf * BUF + l31 * RB + 16 * hi;
        const float* sInv = sInvAll + buf * F3_BM;
        // ---- GEMM 1: H^T[j][n] += W1[j][k] x[n][k], j = this wave's 32 units, n = the 32 rows; Behind every fourth step: one 64-k group of the NEXT tile is cut
        //      into the other buffer and its ring slot refilled with the tile after next.
        // (two accumulators taken in turn by consecutive MFMAs: back to back on ONE accumulator a 32x32x16 MFMA issues every
        // ~60 cycles instead of 32 — stamps: 5 964 cycles for the 96 MFMAs of a tile)
        f32x16 Hacc[2];
        {
            F2Frag xs[3][2];                              // B fragments, read two steps ahead of their MFMAs (order pinned below)
            auto rd = [&](int s_, F2Frag (&d)[2]) {
                d[0].f = *reinterpret_cast<const f32x4*>(xb_ + 32 * s_);
                d[1].f = *reinterpret_cast<const f32x4*>(xb_ + 2 * K + 32 * s_);
            };
            rd(0, xs[0]);
            rd(1, xs[1]);
            f3_static_for<0, NKS>([&](auto s_c) {
                constexpr int s = decltype(s_c)::value;
                using J0 = std::integral_constant<int, 0>;
                using J1 = std::integral_constant<int, 1>;
                using J2 = std::integral_constant<int, 2>;
                if constexpr (DBG != 3 && DBG != 5 && s + 2 < NKS) rd(s + 2, xs[(s + 2) % 3]);   // (DBG 3: timing without the LDS reads of GEMM 1)
                __builtin_amdgcn_sched_barrier(0);
                // (two accumulators taken in turn: back to back on ONE accumulator the MFMAs issue every ~60 cycles instead of 32)
                if constexpr (s == 0) f3_mfma_a0(Hacc[0], w1[s][1].f, xs[s % 3][0].f);
                else f3_mfma_a(Hacc[s & 1], w1[s][1].f, xs[s % 3][0].f);
                if constexpr (DBG != 2 && DBG != 5) cut_piece(buf ^ 1, s_c, J0{}, sc_n, src_nn);   // (DBG 2: timing without the cut of the next tile)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s == 0) f3_mfma_a0(Hacc[1], w1[s][0].f, xs[s % 3][1].f);
                else f3_mfma_a(Hacc[~s & 1], w1[s][0].f, xs[s % 3][1].f);
                if constexpr (DBG != 2 && DBG != 5) cut_piece(buf ^ 1, s_c, J1{}, sc_n, src_nn);
                __builtin_amdgcn_sched_barrier(0);
                f3_mfma_a(Hacc[s & 1], w1[s][0].f, xs[s % 3][0].f);
                if constexpr (DBG != 2 && DBG != 5) cut_piece(buf ^ 1, s_c, J2{}, sc_n, src_nn);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s == 7) STAMP(10);
                if constexpr (s == 15) STAMP(11);
                if constexpr (s == 23) STAMP(12);
            });
            F3_NOP();
        }
        if (o == 0) sInvAll[(buf ^ 1) * F3_BM + myrow] = sinv_n;
        if constexpr (W2_STREAM) {
#pragma unroll
            for (int st = 0; st < 8; ++st) w2[st][1].f = *(const DSMIL_GLOBAL f32x4*)(w2p1 + (long long)st * F2_CHUNK_F4);
        }
        STAMP(2);
        // ---- un-scale, bias, ReLU: reg 4q+e <-> unit 32 wave + 8q + 4hi + e, row l31
        const float rinv1 = sInv[l31];
        const float iv1 = ia1 * rinv1;
        f32x16 H;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + 32 * wave + 8 * q + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) H[4 * q + e] = fmaxf(fmaf(Hacc[0][4 * q + e] + Hacc[1][4 * q + e], iv1, bq[e]), 0.f);
        }
        STAMP(3);
        float hsc, hinv;
        hsc = f2_scale(fmaf(hb_n1, rinv1, hb_b), hinv);
        // registers 8sx .. 8sx+7 are, for row l31, the 8 hidden units of GEMM-2 step 2 wave + sx (the k permutation the packed
        // W2 carries): scale, cut, publish
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const f32x4 h0 = {H[8 * sx], H[8 * sx + 1], H[8 * sx + 2], H[8 * sx + 3]};
            const f32x4 h1 = {H[8 * sx + 4], H[8 * sx + 5], H[8 * sx + 6], H[8 * sx + 7]};
            F2Frag f[2];
            split2h_scaled(h0, h1, hsc, f);
            char* d = sH + l31 * F3_HROW + ((2 * wave + sx) * 2 + hi) * 16;
            *reinterpret_cast<f32x4*>(d) = f[0].f;
            *reinterpret_cast<f32x4*>(d + 256) = f[1].f;
        }
        __syncthreads();                                  // B2
        STAMP(4);
        f32x16 Qacc[2];
        {
            const char* hb_ = sH + l31 * F3_HROW + 16 * hi;
            F2Frag hs[3][2];
            auto rdh = [&](int st_, F2Frag (&d)[2]) {
                d[0].f = *reinterpret_cast<const f32x4*>(hb_ + 32 * st_);
                d[1].f = *reinterpret_cast<const f32x4*>(hb_ + 256 + 32 * st_);
            };
            rdh(0, hs[0]);
            rdh(1, hs[1]);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (st + 2 < 8) rdh(st + 2, hs[(st + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
                if (st == 0) {
                    f3_mfma_v0(Qacc[0], w2[st][1].f, hs[st % 3][0].f);
                    f3_mfma_v0(Qacc[1], w2[st][0].f, hs[st % 3][1].f);
                } else {
                    f3_mfma_v(Qacc[st & 1], w2[st][1].f, hs[st % 3][0].f);
                    f3_mfma_v(Qacc[~st & 1], w2[st][0].f, hs[st % 3][1].f);
                }
                f3_mfma_v(Qacc[st & 1], w2[st][0].f, hs[st % 3][0].f);
                if (DBG != 4 && st < NMOVE) fill(src_nn, NG - NMOVE + st);
                __builtin_amdgcn_sched_barrier(0);
            }
            F3_NOP();
        }
        STAMP(5);
        // ---- tanh; partial scores over this wave's 32 query units (dsmil.py:55-56)
        {
            const float iv2 = ia2 * hinv;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + QD + 32 * wave + 8 * q + 4 * hi);
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(sQ + 32 * wave + 8 * q + 4 * hi);
                f32x4 u1 = u0;
                if constexpr (TWO) u1 = *reinterpret_cast<const f32x4*>(sQ + QD + 32 * wave + 8 * q + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float qv = fast_tanh(fmaf(Qacc[0][4 * q + e] + Qacc[1][4 * q + e], iv2, bq[e]));
                    s0 = fmaf(qv, u0[e], s0);
                    if constexpr (TWO) s1 = fmaf(qv, u1[e], s1);
                }
            }
            sS[((wave * 2 + hi) * 2 + 0) * F3_BM + l31] = s0;
            if constexpr (TWO) sS[((wave * 2 + hi) * 2 + 1) * F3_BM + l31] = s1;
        }
        STAMP(6);
        __syncthreads();                                  // T1
        STAMP(7);
        // ---- scores, softmax weights relative to the bag's constant reference: every wave for itself (lane & 31 = row; the
        //      same values in every wave), weights for its own eight rows' value sum into its private strip
        {
            const long long grow = cur.row0 + l31;
            const bool valid = grow < cur.Nb;
            const float rinv = sInv[l31];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                auto part = [&](int i) { return sS[(i * 2 + c) * F3_BM + l31]; };   // i = 2 wave + half: a fixed order
                const float s = (((part(0) + part(1)) + (part(2) + part(3))) + ((part(4) + part(5)) + (part(6) + part(7)))) * scale;
                const float p = valid ? expf(s - m_bag[c]) : 0.f;
                if (hi == 0) l_run[c] += p;
                if (DBG == 0 && wave == 0 && hi == 0 && valid) a.scores[(cur.off0 + grow) * (long long)a.C + c] = s;
                if (hi == 0) sPw[c * F3_BM + l31] = p * rinv;
            }
        }
        STAMP(8);
        // ---- value sum (dsmil.py:57) from the resident planes: lane = k-octet, this wave's rows 8 wave .. 8 wave + 7
        if (lane < K / 8) {
            const char* xv = sX + buf * BUF + (8 * wave) * RB + 16 * lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                F2Frag f0, f1;
                f0.f = *reinterpret_cast<const f32x4*>(xv + j * RB);
                f1.f = *reinterpret_cast<const f32x4*>(xv + j * RB + 2 * K);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float w = sPw[c * F3_BM + 8 * wave + j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bacc[c][e] = fmaf((float)f0.v[e], w, bacc[c][e]);
                        bacc[c][e] = fmaf((float)f1.v[e], w, bacc[c][e]);
                    }
                }
            }
        }
        STAMP(9);
        if constexpr (DBG >= 1) {
            if (tid == 0 && !TWO && cur.row0 + F3_BM <= cur.Nb) {
                unsigned long long* so = reinterpret_cast<unsigned long long*>(a.scores + (cur.off0 + cur.row0));
#pragma unroll
                for (int i = 0; i < 16; ++i) so[i] = stamps[i];
            }
        }
        // ---- end of this workgroup's part of the bag: one (m, l, B) partial, slot = blockIdx.x + bag
        if (!has_next || nxt.bag != cur.bag) {
            const long long slot = (long long)blockIdx.x + cur.bag;
            float* red = reinterpret_cast<float*>(sH);    // [4 waves][NC][K] (the hidden planes are consumed)
            __syncthreads();
            if (lane < K / 8) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float* d = red + (wave * NC + c) * K + 8 * lane;
                    *reinterpret_cast<f32x4*>(d) = f32x4{bacc[c][0], bacc[c][1], bacc[c][2], bacc[c][3]};
                    *reinterpret_cast<f32x4*>(d + 4) = f32x4{bacc[c][4], bacc[c][5], bacc[c][6], bacc[c][7]};
                }
            }
            __syncthreads();
            for (int i = tid; i < NC * K; i += F3_THREADS) {
                const int c = i / K, k = i - c * K;
                const float v = (red[(0 * NC + c) * K + k] + red[(1 * NC + c) * K + k]) + (red[(2 * NC + c) * K + k] + red[(3 * NC + c) * K + k]);
                a.part_B[(slot * a.C + c) * (long long)a.Kv + k] = v;
            }
            if (wave == 0) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float lsum = wave_sum(hi == 0 ? l_run[c] : 0.f);
                    if (lane == 0) {
                        float* ml = a.part_ml + (slot * a.C + c) * 2;
                        ml[0] = m_bag[c];
                        ml[1] = lsum;
                    }
                }
            }
            reset_acc();
        }
        STAMP(13);
        if constexpr (DBG >= 1) {
            if (tid == 0 && !TWO && cur.row0 + F3_BM <= cur.Nb)
                reinterpret_cast<unsigned long long*>(a.scores + (cur.off0 + cur.row0))[13] = stamps[13];
        }
        if (!has_next) break;
        cur = nxt; nxt = nn; nn = n3;
        has_next = has_nn; has_nn = has_n3;
        pos_nn = pos_n3;
        sc_c = sc_n; sinv_c = sinv_n;
        sc_n = sc_nn; sinv_n = sinv_nn; src_n = src_nn;
        asm volatile("" : "+v"(rm_n3), "+v"(phys_n3));   // (the lookups are consumed HERE, a tile after they were issued)
        src_nn = row_done(rm_n3, phys_n3, log_n3, sc_nn, sinv_nn);
    }
}

}  // namespace
