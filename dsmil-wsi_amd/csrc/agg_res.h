// agg_res.h — k_attend_bf16_res: the bf16-storage query / attend kernel (dsmil.py:46-62 behind the instance logits,
// BASELINE configs[2]) with the feature tile RESIDENT in LDS, the query weights RESIDENT in registers and the bag's
// partial sums RESIDENT in accumulators.
//
// What it replaces: k_query_attend_bf16_dma streamed each 64-k chunk of a 128-row tile through a 3-slot LDS ring next to
// a ring of weight chunks (as many L2 -> LDS bytes for weights as HBM -> LDS bytes for features, both in one in-order
// vmcnt queue), read the whole tile a SECOND time from L2 / HBM for the value sum B = sum_n p[n] x[n,:] (dsmil.py:57) on
// the VALU, and wrote one (m, l, B) partial per tile (20 MB per 64 bags, read back by k_finish).  Here:
//   * one PERSISTENT 256-thread workgroup per CU (one wave per SIMD: 256 VGPRs + 256 AGPRs each) owns a CONTIGUOUS run of
//     64-row tiles (so that consecutive tiles belong to the same bag);
//   * wave w keeps ITS slice of the weights in AGPRs for the whole launch: W1 rows [32w, 32w+32) x K (<= 128 registers)
//     and W2 rows [32w, 32w+32) x 128 (32 registers) as ready MFMA A-fragments (hipcc never places an MFMA A operand in the
//     accumulator file, and 160 resident VGPRs left it ~90 for everything else: it serialised every read -> use chain.
//     The MFMAs are therefore inline asm with "a" weights and "v" accumulators).  GEMM 1 gives wave w the hidden units
//     [32w, 32w+32) of all 64 rows (B operand = any row of the tile, read from LDS); the ReLU'd bf16 hidden layer is
//     exchanged through 16 KiB of LDS and GEMM 2 gives wave w the query units [32w, 32w+32) of all rows.  No weight byte
//     moves after the prologue;
//   * the tile (64 rows x K <= 512 bf16 = up to 64 KiB) stays in LDS from GEMM 1 until the value sum has consumed it: ONE
//     read of every feature byte — and there are TWO tile buffers: the pieces of tile t+1 are issued one at a time
//     between the MFMA groups of tile t's GEMM 1 (a burst of LDS-DMA instructions blocks the issuing wave for as long as
//     the CU's address path needs to take them: ~48 cycles per 1 KiB piece, 6 k cycles per 128 KiB) and have a whole
//     tile's time to land;
//   * the value sum runs on the matrix pipe: out[c][k] = sum_n p[n][c] x[n][k] with the row-major tile as the B operand
//     through ds_read_b64_tr_b16 (two transposed reads per 32 rows x 16 features) and the attention weights as the A
//     operand, cut into three bf16 planes (exact fp32 p) that sit in DIFFERENT ROWS of the same 16-row A fragment —
//     one v_mfma_f32_16x16x32_bf16 per 32 x 16 block does all three plane products.  Wave w owns the features of chunks
//     w and 4 + w for all rows, so B needs no cross-wave reduction;
//   * those accumulators run ACROSS the tiles of a bag (online softmax: weights relative to the running max of the
//     workgroup's part of the bag, accumulators rescaled when it moves) and are written once per (workgroup, bag):
//     partial slot = blockIdx.x + bag (a workgroup's bags and a bag's workgroups are both contiguous runs, so the
//     staircase is collision-free).  k_finish merges <= ~3 partials per bag instead of 79.
// LDS map (153 600 B): sX [2 buffers][8 chunks][64 rows][128 B] (16-B slot c of row r holds global slot c ^ f(r), f(r) = (r&6)|((r>>4)&1)
// on the row's index inside its 32-row group: conflict-free ds_read_b128 of the MFMA fragments and conflict-free
// transposed reads), then 16 KiB sH: the hidden layer [64 rows][16 blocks x 16 B] (block b of row n at b ^ (n & 15));
// then 4 KiB of softmax scratch (partial scores, bf16 planes of p, rescale factors), then 2 KiB of constants (b1 | b2 * 2 log2 e
// | the current bag's critical queries [2][128]).
// Register / LDS budget (round 6).  The kernel holds every CU for the whole launch, so whatever else a forward needs — the
// logits pass, the critical query, the combine — waited for it, and three streams bought 5 %.  Its per-unit constants (biases,
// the bag's critical queries: 64 long-lived VGPRs) now live in 2 KiB of LDS and are read where they are used; the K = 512
// instantiations come to 240 VGPRs + 192 AGPRs (W1 128, W2 32, the value sum's accumulators 32) = 432 of a SIMD's 512
// registers and 150 KiB of LDS: ONE 80-register wave of another kernel fits beside each wave, with 10 KiB of LDS.
// k_logits_pipe, the 4-wave k_qmax launch and the lean k_finish (agg_fwd.hip, dsmil_agg_logits_form) are cut to that slot:
// with two or more streams in flight the second read of the features (the next batch's logits) runs UNDER this kernel.
// Barriers per tile (all four waves): S (tile landed for everybody; the previous tile's buffer and scratch are released),
// Bh (hidden layer written), E1 (partial scores), E3 (planes + rescale factors).
#pragma once
#include "agg_common.h"
#include "agg_split.h"

namespace {

constexpr int RS_BM = 64;                   // rows per tile
constexpr int RS_CH_F4 = 512;               // float4 (16 B) per 8 KiB feature chunk (64 rows x 128 B)
constexpr int RS_MAXCH = 8;                 // K <= 512 (the kernel is instantiated for K = 512 and K = 256)
constexpr int RS_BUF_F4 = RS_MAXCH * RS_CH_F4;   // float4 per tile buffer (64 KiB)
constexpr int RS_THREADS = 256;             // one wave per SIMD
constexpr int RS_MAX_WG = 1024;             // upper bound of the persistent grid (the workspace holds RS_MAX_WG + n_bags partial slots)
constexpr int RS_SCRATCH_BYTES = 4096;      // softmax scratch: partial scores 2 KiB, bf16 planes of p 768 B, rescale factors
constexpr int RS_CONST_BYTES = 2048;        // biases b1 | b2 * 2 log2(e) (1 KiB), the bag's critical queries [2 classes][128] (1 KiB)
constexpr int RS_LDS_BYTES = (2 * RS_BUF_F4 + RS_BM * 16) * 16 + RS_SCRATCH_BYTES + RS_CONST_BYTES;   // 2 x 64 KiB + 16 KiB + 4 KiB + 2 KiB = 153 600

typedef short rs_v4s __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x16_bf16 with the A operand AND the accumulator in the accumulator half of the register file.  A wave
// may hold 512 registers, but only 256 of them are addressable as VGPRs; the resident weight fragments (160 registers)
// left hipcc ~90 VGPRs for everything else and it serialised every read -> use chain.  "a" operands keep the weights in
// AGPRs for the whole launch (hipcc itself never places an MFMA A/B operand there).  Inside an asm statement the compiler
// pads no hazards: callers put s_nop states between a compiler-written accumulator and the first MFMA, and between the
// last MFMA and any compiler read of its result (RS_NOP).
// The accumulators stay in VGPRs ("+v"): every v_accvgpr_read / _write between the matrix pipe and the VALU phases is an
// issue slot this one-wave-per-SIMD kernel does not have.  rs_mfma0 starts a chain from the inline constant 0.
__device__ __forceinline__ void rs_mfma(f32x16& acc, const f32x4& a_agpr, const f32x4& b_vgpr) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a_agpr), "v"(b_vgpr));
}
__device__ __forceinline__ void rs_mfma0(f32x16& acc, const f32x4& a_agpr, const f32x4& b_vgpr) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(a_agpr), "v"(b_vgpr));
}
#define RS_NOP() asm volatile("s_nop 15" ::: "memory")   // 16 states >= the 12 an 8-pass MFMA result needs before a VALU read

struct RsWork {
    int bag;
    long long off0, Nb, row0;
};

// first work item in [item, end) whose tile lies inside its bag; false when the run is exhausted.  Evaluated identically
// by every wave of the workgroup.
// (a.tile_pre != nullptr: a ragged batch — the items are the real tiles, tile_pre[b] of them in front of bag b; the bag of an
// item: the bag of the item in front or the one behind it (three independent scalar loads), else by bisection)
__device__ __forceinline__ bool rs_fetch(const AttendArgs& a, int tiles_per_bag, int end, int& item, RsWork& w, int guess = -1) {
    while (item < end) {
        int b, tile;
        if (a.tile_pre) { b = tile_owner_near(a.tile_pre, a.n_bags, item, guess); tile = item - a.tile_pre[b]; }
        else { b = item / tiles_per_bag; tile = item - b * tiles_per_bag; }
        const int bag = a.bag0 + b;
        const long long off0 = a.offsets[bag];
        const long long Nb = a.offsets[bag + 1] - off0;
        const long long row0 = (long long)tile * RS_BM;
        if (row0 < Nb) {
            w.bag = bag; w.off0 = off0; w.Nb = Nb; w.row0 = row0;
            return true;
        }
        ++item;
    }
    return false;
}

template <int NCH, bool TWO, bool NL, int PV = 0>   // NCH: K / 64; TWO: C == 2 (else C == 1); NL: the two-layer query of dsmil.py:31-32;
                                                    // PV: placement of the next tile's pieces (A/B variants of experiment builds)
__global__ __launch_bounds__(RS_THREADS, 1) void k_attend_bf16_res(AttendArgs a, int tiles_per_bag, int n_items, int per_wg) {
    static_assert(NCH >= 1 && NCH <= RS_MAXCH, "feature chunks");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* sX = reinterpret_cast<f32x4*>(smem);              // [2][8][RS_CH_F4]
    f32x4* sH = sX + 2 * RS_BUF_F4;                          // [64 rows][16 blocks]: hidden layer, bf16
    float* sS = reinterpret_cast<float*>(sH + RS_BM * 16);   // own scratch (not aliased: no barrier between GEMM 2 and the scores): [4 waves][2 classes][64 rows] partial scores
    unsigned short* sPl = reinterpret_cast<unsigned short*>(sS + 4 * 2 * RS_BM);   // [3 planes][2 classes][64 rows] bf16 planes of p
    float* sF = reinterpret_cast<float*>(sPl + 3 * 2 * RS_BM);                     // [2] rescale factor of the running sums per class
    // per-unit constants live in LDS, not in registers (round 6: 64 long-lived VGPRs of this wave went back to the SIMD's file,
    // see the register budget in the header): read as broadcast ds_read_b128 where they are used
    float* sBias = reinterpret_cast<float*>(reinterpret_cast<char*>(sS) + RS_SCRATCH_BYTES);   // [128] b1, [128] b2 * 2 log2(e)
    float* sU = sBias + 2 * QD;                                                     // [2 classes][128] critical queries of the current bag
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int K = a.K, C = a.C;
    const bf16_t* feats = reinterpret_cast<const bf16_t*>(a.feats);
    const f32x4* wpk = reinterpret_cast<const f32x4*>(a.wpk);   // chunk-major fragment image (k_pack_agg_bf16)

    int item = (int)blockIdx.x * per_wg;
    if (a.tile_pre) {                                        // ragged batch: the host sized the runs from an upper bound
        const int real = a.tile_pre[a.n_bags];
        n_items = n_items < real ? n_items : real;
    }
    const int item_end = item + per_wg < n_items ? item + per_wg : n_items;
    RsWork cur, nxt;
    if (!rs_fetch(a, tiles_per_bag, item_end, item, cur)) return;
#ifdef DSMIL_TRACE
    // trace builds, DSMIL_EXPT & 64: wave 0 keeps s_memtime stamps of the phase boundaries in registers and stores them into
    // the tile's rows of A at the end of the tile (tools/stamp_res.py); k_finish is skipped.
    unsigned long long stamps[10];
    int nstamp = 0;
#define RS_STAMP() do { if (nstamp < 10) stamps[nstamp] = __builtin_readcyclecounter(); ++nstamp; } while (0)
#define RS_STAMP_FLUSH(w) do { if (DSMIL_EXPT_ON(a, 64) && tid == 0 && (w).row0 + RS_BM <= (w).Nb) { \
        unsigned long long* o_ = reinterpret_cast<unsigned long long*>(a.scores + ((w).off0 + (w).row0) * (long long)C); \
        _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) o_[i_] = stamps[i_]; } nstamp = 0; } while (0)
#else
#define RS_STAMP()
#define RS_STAMP_FLUSH(w)
#endif

    // ---- the feature stream: this wave's 16 rows of every chunk, 2 pieces of 8 rows x 128 B
    const bf16_t* src[2];
    auto set_rows = [&](const RsWork& w) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = wave * 16 + p * 8 + (lane >> 3), r32 = r & 31;
            long long gr = w.row0 + r;
            if (gr >= w.Nb) gr = w.Nb - 1;                                  // rows past the bag end get weight 0 later
            const int gslot = (lane & 7) ^ ((r32 & 6) | ((r32 >> 4) & 1));
            src[p] = feats + phys_row(a.rowmap, w.off0 + gr) * (long long)K + gslot * 8;
        }
    };
    auto issue_piece = [&](int buf, int i) {   // piece i = 2 c + p of this wave
        __builtin_amdgcn_global_load_lds((const DSMIL_GLOBAL void*)(src[i & 1] + (i >> 1) * 64),
                                         (__attribute__((address_space(3))) void*)(sX + buf * RS_BUF_F4 + (i >> 1) * RS_CH_F4 + (wave * 16 + (i & 1) * 8) * 8), 16, 0, 0);
    };
    set_rows(cur);
#pragma unroll
    for (int i = 0; i < 2 * NCH; ++i) issue_piece(0, i);

    // ---- resident weights: A fragments of this wave's 32 hidden / query units (fragment image of k_pack_agg_bf16:
    //      chunk s: [ks][t][lane] x 16 B; W2 chunks carry the k permutation of the accumulator layout), loaded STRAIGHT
    //      into the accumulator file (a VMEM load may target AGPRs on gfx950): a 128-bit tuple that is born there stays
    //      there.  The loads are invisible to hipcc's waitcnt pass and are drained by hand below.
    union Frag { f32x4 f; bf16x8 v; };
    f32x4 w1[NCH * 4], w2[8];
#pragma unroll
    for (int q = 0; q < NCH * 4; ++q) {
        const f32x4* src_w = wpk + (long long)(q >> 2) * 1024 + ((q & 3) * 4 + wave) * 64 + lane;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(w1[q]) : "v"(src_w) : "memory");
    }
    // biases of the 128 hidden / query units; b2c = b2 * 2 log2(e): folded into tanh's first fma
    constexpr float TANH_C = 2.8853900817779268f;
    if (tid < QD) sBias[tid] = a.q0_b[tid];
    else sBias[tid] = NL ? a.q2_b[tid - QD] * TANH_C : 0.f;
    if constexpr (NL) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4* src_w = wpk + (long long)(NCH + (q >> 2)) * 1024 + ((q & 3) * 4 + wave) * 64 + lane;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(w2[q]) : "v"(src_w) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the resident weights (and the first tile's pieces) have landed
    __builtin_amdgcn_sched_barrier(0);
    const int fr = (l31 & 6) | ((l31 >> 4) & 1);
    const float scale = 0.08838834764831845f;                               // 1/sqrt(128), dsmil.py:56
    int xsl[4];                                                             // (slot ^ row permutation) of the four 16-k steps of a chunk
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xsl[ks] = l31 * 8 + ((ks * 2 + hi) ^ fr);

    // ---- value-sum geometry (constant): A-fragment source, transposed-read offsets
    const int ai = lane & 15, ag = lane >> 4;
    const int ts = lane & 15, trow = 4 * ag + (ts >> 2), tq = ts & 3;     // source lane ts of 16-lane group ag: row 4 ag + (ts >> 2), features 4 tq
    int toff0[4], toff1[4];                                                // byte offsets inside a chunk, row group 0: rows trow / trow + 16
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int sl = 2 * b + (tq >> 1), f0 = (trow & 6), f1 = f0 | 1;
        toff0[b] = trow * 128 + ((sl ^ f0) * 16) + (tq & 1) * 8;
        toff1[b] = (trow + 16) * 128 + ((sl ^ f1) * 16) + (tq & 1) * 8;
    }
    // running sums of the bag in the workgroup's run: raw D layout of the value-sum MFMA (rows = plane x class), 2 chunks x 4 blocks
    f32x4 acc[2][4];
    float m_run = -INFINITY, l_run = 0.f;                                   // waves 0 (class 0) and 1 (class 1)
    auto reset_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[j][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run = -INFINITY;
        l_run = 0.f;
    };
    reset_acc();
    // critical queries of the current bag, this wave's 32 units (reloaded when the bag changes: a vector load issued
    // while the next tile's pieces are in flight would wait for them in the in-order vmcnt queue)
    int ubag = -1;
    const float* sBw = sBias + 32 * wave + 4 * hi;     // this lane's units 32 wave + 8 g + 4 hi + e: f32x4 at + 8 g
    const float* sUw = sU + 32 * wave + 4 * hi;

    for (int t = 0;; ++t) {
        const int buf = t & 1;
        int in = item + 1;
        const bool has_next = rs_fetch(a, tiles_per_bag, item_end, in, nxt, cur.bag - a.bag0);
        if (has_next) set_rows(nxt);
        const bool new_bag = cur.bag != ubag;                               // workgroup-uniform
        float uval = 0.f;
        if (new_bag) {
            // thread (class, unit) fetches one critical-query value; it goes to LDS behind the S wait below (the scores of the
            // previous tile were read in front of its barrier E1, the next reads come behind this tile's barriers S and Bh)
            if (tid < QD * (TWO ? 2 : 1)) uval = a.qmax[((long long)cur.bag * C) * QD + tid];
            ubag = cur.bag;
            if constexpr (NL) {
                // the bag's softmax reference (waves 0 / 1 = classes 0 / 1): every wave sums its class's 128 |q_max| alike
                const float* qa = a.qmax + ((long long)cur.bag * C + (wave < C ? wave : 0)) * QD;
                m_run = wave_sum(fabsf(qa[lane]) + fabsf(qa[lane + 64])) * scale;
            }
        }
        RS_STAMP();                                                         // 0: tile start
        // ---- the tile has landed: own pieces (everything this wave ever issued), then everybody's
        S3_WAIT_VM(0);
        if (new_bag) sU[tid] = uval;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                       // S
        RS_STAMP();                                                         // 1
        // ---- GEMM 1 (transposed): H^T[j][n] += W1[j][k] X[n][k]; j = this wave's 32 units, n = all 64 rows.
        // One wave per SIMD: nothing hides an LDS round trip unless the code does.  The B fragments are read TWO 16-k steps
        // THREE 16-k steps ahead into four rotating register sets, the order pinned by sched_barrier; one piece of the NEXT tile goes
        // out every other step (an LDS-DMA issue costs the wave ~45-60 cycles of its own stream here, twice that between the
        // VALU stages of the tanh phase).
        f32x16 H[2];
        {
            const f32x4* xb_ = sX + buf * RS_BUF_F4;
            Frag xs[4][2];
            auto rd = [&](int q, Frag (&d)[2]) {
#pragma unroll
                for (int r = 0; r < 2; ++r) d[r].f = xb_[(q >> 2) * RS_CH_F4 + r * 256 + xsl[q & 3]];
            };
            rd(0, xs[0]);
            rd(1, xs[1]);
            rd(2, xs[2]);
#pragma unroll
            for (int q = 0; q < NCH * 4; ++q) {
                if (q + 3 < NCH * 4) rd(q + 3, xs[(q + 3) & 3]);
                if (PV == 1 && (q & 1) == 0 && has_next) issue_piece(buf ^ 1, q >> 1);               // all 2 NCH pieces here
                if (PV == 0 && (q & 3) == 0 && has_next) issue_piece(buf ^ 1, q >> 2);               // NCH here, NCH behind the tanh stages
                if (PV == 2 && (q & 3) == 0 && has_next) issue_piece(buf ^ 1, q >> 2);               // NCH here, NCH in the H exchange
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (q == 0) rs_mfma0(H[r], w1[q], xs[q & 3][r].f);
                    else rs_mfma(H[r], w1[q], xs[q & 3][r].f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        RS_NOP();                                                           // last MFMA -> VALU reads of H
        RS_STAMP();                                                         // 2: GEMM 1 done
        f32x16 Q[2];
        if constexpr (NL) {
            // ---- bias, round to bf16, ReLU (on the packed pair: a negative bf16 is a negative int16), publish: block
            //      ((wave, sidx), hi) of row n holds the 8 hidden units that accumulator registers 8 sidx .. 8 sidx + 7 of
            //      this lane carry — a ready B fragment of GEMM 2
            f32x4 b1[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) b1[g] = *reinterpret_cast<const f32x4*>(sBw + 8 * g);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int n = 32 * r + l31;
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
                    union { unsigned u[4]; f32x4 f; } hb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i0 = 8 * sidx + 2 * e;
                        const unsigned pk = pack_bf16x2_hw(H[r][i0] + b1[i0 >> 2][i0 & 3], H[r][i0 + 1] + b1[(i0 + 1) >> 2][(i0 + 1) & 3]);
                        typedef short s2 __attribute__((ext_vector_type(2)));
                        const s2 z = {0, 0};
                        hb.u[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, pk), z));
                    }
                    sH[n * 16 + ((((wave * 2 + sidx) * 2) + hi) ^ (n & 15))] = hb.f;
                    if (PV == 2 && has_next) {
#pragma unroll
                        for (int i = 0; i < NCH / 4; ++i) issue_piece(buf ^ 1, NCH + (r * 2 + sidx) * (NCH / 4) + i);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                   // Bh
            // ---- GEMM 2 (transposed): Q^T[j2][n] += W2[j2][k] H^T[k][n]; j2 = this wave's 32 units
            {
                Frag hs[3][2];
                auto rdh = [&](int q, Frag (&d)[2]) {   // k-step q = 2 t + sidx: block (q, hi) of every row
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int n = 32 * r + l31;
                        d[r].f = sH[n * 16 + (((q * 2) + hi) ^ (n & 15))];
                    }
                };
                rdh(0, hs[0]);
                rdh(1, hs[1]);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (q + 2 < 8) rdh(q + 2, hs[(q + 2) % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                    // packed chunk c2 = q >> 2, fragment index q & 3 = 2 (t & 1) + sidx  (k_pack_agg_bf16)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        if (q == 0) rs_mfma0(Q[r], w2[q], hs[q % 3][r].f);
                        else rs_mfma(Q[r], w2[q], hs[q % 3][r].f);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                RS_NOP();
            }
        } else {
            f32x4 b1[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) b1[g] = *reinterpret_cast<const f32x4*>(sBw + 8 * g);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) Q[r][i] = H[r][i] + b1[i >> 2][i & 3];
        }
        RS_STAMP();                                                         // 3 (no barrier: the scratch below is not sH)
        // ---- tanh; partial scores over this wave's 32 query units (dsmil.py:55-56).  Written stage by stage over 16 values
        //      so that the exp / rcp chains of different values overlap.
        {
            f32x4 b2c[4], u0[4], u1[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (NL) b2c[g] = *reinterpret_cast<const f32x4*>(sBw + QD + 8 * g);
                u0[g] = *reinterpret_cast<const f32x4*>(sUw + 8 * g);
                if constexpr (TWO) u1[g] = *reinterpret_cast<const f32x4*>(sUw + QD + 8 * g);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float q[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) q[i] = Q[r][i];
                if constexpr (NL) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) q[i] = __builtin_amdgcn_exp2f(fmaf(q[i], TANH_C, b2c[i >> 2][i & 3]));
                    if (PV == 0 && has_next) {
#pragma unroll
                        for (int i = 0; i < NCH / 4; ++i) issue_piece(buf ^ 1, NCH + r * (NCH / 2) + i);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) q[i] = __builtin_amdgcn_rcpf(1.f + q[i]);
                    if (PV == 0 && has_next) {
#pragma unroll
                        for (int i = NCH / 4; i < NCH / 2; ++i) issue_piece(buf ^ 1, NCH + r * (NCH / 2) + i);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) q[i] = fmaf(q[i], -2.f, 1.f);
                } else if (PV == 0 && has_next) {
#pragma unroll
                    for (int i = 0; i < NCH / 2; ++i) issue_piece(buf ^ 1, NCH + r * (NCH / 2) + i);
                }
                float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sa[e] = fmaf(q[4 * g + e], u0[g][e], sa[e]);
                        if constexpr (TWO) sb[e] = fmaf(q[4 * g + e], u1[g][e], sb[e]);
                    }
                float s0 = (sa[0] + sa[1]) + (sa[2] + sa[3]), s1 = (sb[0] + sb[1]) + (sb[2] + sb[3]);
                s0 += __shfl_xor(s0, 32, 64);
                if constexpr (TWO) s1 += __shfl_xor(s1, 32, 64);
                if (hi == 0) {
                    sS[(wave * 2 + 0) * RS_BM + 32 * r + l31] = s0;
                    if constexpr (TWO) sS[(wave * 2 + 1) * RS_BM + 32 * r + l31] = s1;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                       // E1
        RS_STAMP();                                                         // 4
        // ---- the value sum's B operands do not depend on the attention weights: the transposed reads of this wave's first
        //      feature chunk (8 blocks x 2 reads) go out NOW, under the softmax, the second chunk's behind the first's MFMAs
        //      (into the registers they release): no LDS round trip sits between barrier E3 and the sixteen MFMAs
        rs_v4s tr0[8], tr1[8];
        auto rd_block = [&](int j, int u) {          // chunk 4 j + wave, block u = 4 rg + b
            const int c = 4 * j + wave;
            if (c < NCH) {
                const char* xc = reinterpret_cast<const char*>(sX + buf * RS_BUF_F4 + c * RS_CH_F4);
                tr0[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rs_v4s*)(xc + (u >> 2) * 4096 + toff0[u & 3]));
                tr1[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rs_v4s*)(xc + (u >> 2) * 4096 + toff1[u & 3]));
            }
        };
#pragma unroll
        for (int u = 0; u < 8; ++u) rd_block(0, u);  // the second chunk's reads follow each MFMA of the first (same registers)
        __builtin_amdgcn_sched_barrier(0);
        // ---- scores and online softmax: wave cls holds the 64 rows of class cls (lane = row).  The weight is taken
        //      relative to the running max of this workgroup's part of the bag and cut into three bf16 planes for the value
        //      sum's A fragments; the factor by which the running sums shrink goes to every wave through sF.
        if (wave < (TWO ? 2 : 1)) {
            const int cls = wave, row = lane;
            const float s = ((sS[(0 * 2 + cls) * RS_BM + row] + sS[(1 * 2 + cls) * RS_BM + row]) +
                             (sS[(2 * 2 + cls) * RS_BM + row] + sS[(3 * 2 + cls) * RS_BM + row])) * scale;
            const bool valid = cur.row0 + row < cur.Nb;
            if (valid && !DSMIL_EXPT_ON(a, 64)) a.scores[(cur.off0 + cur.row0 + row) * (long long)C + cls] = s;
            float p;
            if constexpr (NL) {
                // tanh bounds the queries: |s| <= scale * sum_j |q_max[j]| = m_run, a constant of the bag — no tile maximum,
                // no rescaling, and the row weights are summed per lane (one wave reduction per partial, not per tile)
                p = valid ? expf(s - m_run) : 0.f;
                l_run += p;
            } else {
                const float m_new = fmaxf(m_run, wave_max(valid ? s : -INFINITY));   // finite: a tile has a valid row
                const float f = expf(m_run - m_new);                                 // 0 on the first tile of a run (m_run = -inf)
                p = valid ? expf(s - m_new) : 0.f;
                l_run = l_run * f + wave_sum(p);
                m_run = m_new;
                if (lane == 0) sF[cls] = f;
            }
            const unsigned h0 = __float_as_uint(p) & 0xFFFF0000u;
            const float r1 = p - __uint_as_float(h0);
            const unsigned h1 = __float_as_uint(r1) & 0xFFFF0000u;
            const unsigned h2 = __float_as_uint(r1 - __uint_as_float(h1));
            sPl[(0 * 2 + cls) * RS_BM + row] = (unsigned short)(h0 >> 16);
            sPl[(1 * 2 + cls) * RS_BM + row] = (unsigned short)(h1 >> 16);
            sPl[(2 * 2 + cls) * RS_BM + row] = (unsigned short)(h2 >> 16);
        } else if (!TWO && wave == 1) {
            // one class: the class-1 rows of the A fragments read zeros
            sPl[(0 * 2 + 1) * RS_BM + lane] = 0; sPl[(1 * 2 + 1) * RS_BM + lane] = 0; sPl[(2 * 2 + 1) * RS_BM + lane] = 0;
            if (!NL && lane == 0) sF[1] = 1.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                       // E3
        RS_STAMP();                                                         // 5
        // ---- value sum on the matrix pipe (dsmil.py:57).  16x16x32: A[i][kk] = plane_{i>>1}(p[row(kk)][class i&1]) for
        //      i < 6 (three exact bf16 planes of the fp32 weight in different rows), B[kk][j] = x[row(kk)][f0 + j] by
        //      transposed LDS reads; kk = 8 g + e  <->  row n0 + 16 (e>>2) + 4 g + (e&3)  (a 32-lane half of a transposed read
        //      then touches 8 consecutive rows: conflict-free).  D rows 0..5 = (hi c0, hi c1, mid c0, mid c1, lo c0, lo c1).
        {
            Frag pa[2];
            {
                const unsigned short* pl = sPl + (ai < 6 ? ai : 0) * RS_BM + 4 * ag;   // plane ai >> 1, class ai & 1
#pragma unroll
                for (int rg = 0; rg < 2; ++rg) {
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(pl + 32 * rg);
                    const u32x2 up = *reinterpret_cast<const u32x2*>(pl + 32 * rg + 16);
                    union { unsigned u[4]; f32x4 f; } pk;
                    pk.u[0] = ai < 6 ? lo.x : 0u; pk.u[1] = ai < 6 ? lo.y : 0u;
                    pk.u[2] = ai < 6 ? up.x : 0u; pk.u[3] = ai < 6 ? up.y : 0u;
                    pa[rg].f = pk.f;
                }
            }
            if constexpr (!NL) {
                const float f0 = sF[0], f1 = sF[1];
                if (f0 != 1.f || f1 != 1.f) {                               // the running max moved: D rows alternate class 0 / class 1
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int b = 0; b < 4; ++b) { acc[j][b][0] *= f0; acc[j][b][1] *= f1; acc[j][b][2] *= f0; acc[j][b][3] *= f1; }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (4 * j + wave < NCH) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        bf16x8 xb;
                        xb[0] = tr0[u][0]; xb[1] = tr0[u][1]; xb[2] = tr0[u][2]; xb[3] = tr0[u][3];
                        xb[4] = tr1[u][0]; xb[5] = tr1[u][1]; xb[6] = tr1[u][2]; xb[7] = tr1[u][3];
                        acc[j][u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[u >> 2].v, xb, acc[j][u & 3], 0, 0, 0);
                        if (j == 0) rd_block(1, u);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        RS_STAMP();                                                         // 6
        // no barrier here: the next tile's S barrier is the one that releases this tile's buffer (its pieces are issued
        // behind S) and the scratch (written behind E1)
        RS_STAMP();                                                         // 7
        // ---- end of this workgroup's part of the bag: one (m, l, B) partial, slot = blockIdx.x + bag
        if (!has_next || nxt.bag != cur.bag) {
            const long long slot = (long long)blockIdx.x + cur.bag;
            // D row i sits in lane group i >> 2, register i & 3: classes (0, 1) = regs (0, 1) + regs (2, 3) of lanes 0..15
            // + regs (0, 1) of lanes 16..31
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = 4 * j + wave;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float l0 = __shfl(acc[j][b][0], (lane & 15) + 16, 64), l1 = __shfl(acc[j][b][1], (lane & 15) + 16, 64);
                    const float o0 = (acc[j][b][0] + acc[j][b][2]) + l0, o1 = (acc[j][b][1] + acc[j][b][3]) + l1;
                    if (lane < 16 && c < NCH) {
                        const int k = c * 64 + b * 16 + lane;
                        a.part_B[(slot * C + 0) * (long long)a.Kv + k] = o0;
                        if constexpr (TWO) a.part_B[(slot * C + 1) * (long long)a.Kv + k] = o1;
                    }
                }
            }
            if (wave < (TWO ? 2 : 1)) {
                const float lsum = NL ? wave_sum(l_run) : l_run;
                if (lane == 0) {
                    float* ml = a.part_ml + (slot * C + wave) * 2;
                    ml[0] = m_run;
                    ml[1] = lsum;
                }
            }
            reset_acc();
        }
        RS_STAMP_FLUSH(cur);
        if (!has_next) break;
        cur = nxt;
        item = in;
    }
}

}  // namespace
