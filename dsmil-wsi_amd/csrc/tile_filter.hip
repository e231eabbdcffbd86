// Background filters of the reference's tilers, on decoded tiles resident in HBM — byte work,
// HBM-bound: every tile byte is read once (rows staged through LDS in a 3-row ring with coalesced dword loads).
//
//   deepzoom_tiler.py:56-61   edge = tile.filter(ImageFilter.FIND_EDGES); ImageStat.Stat(edge).sum (per band);
//                             keep the tile when mean(band sums) / tile_size^2 > threshold (default 15, :255)
//       PIL's FIND_EDGES is the 3x3 kernel [-1 -1 -1; -1 8 -1; -1 -1 -1] (scale 1, offset 0) applied per band,
//       result clipped to [0,255]; the first / last row and column are copied from the input (Pillow Filter.c).
//   test_crop_single.py:17-24 thres_saturation: sat = img_as_ubyte(rgb2hsv(img)[:, :, 1]); keep when
//                             sum(sat) / (h*w) >= t.  skimage: arr = img / 255.0 (float64), v = max, delta = max - min
//       (of the float64 values), s = delta / v (0 where delta == 0), ubyte = rint(s * 255) — evaluated here with
//       the same IEEE float64 operations (hipcc keeps div / mul exact without -ffast-math).
//
// Output per tile: uint64 {edge sum band 0, band 1, band 2, saturation sum}: exact integers; the caller forms the
// two ratios in float64 (as numpy does) and compares with its thresholds — decisions identical to the reference's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dsmil_hip.h"

namespace {

constexpr int TF_MAXW = 1024;                 // widest tile row (pixels) the LDS ring holds
constexpr int TF_ROWB = TF_MAXW * 3;          // bytes per row
constexpr int TF_T = 256;

__device__ __forceinline__ unsigned sat_ubyte(unsigned M, unsigned m) {
    if (M == m) return 0u;                    // delta == 0 -> s = 0 (also covers M = 0)
    const double a = (double)M / 255.0, b = (double)m / 255.0;
    const double s = (a - b) / a;
    return (unsigned)__builtin_rint(s * 255.0);   // round half to even, like np.rint
}

__global__ __launch_bounds__(TF_T) void k_tile_stats(const uint8_t* __restrict__ tiles, unsigned long long* __restrict__ out,
                                                     int H, int W) {
    __shared__ __attribute__((aligned(16))) uint8_t rows[3][TF_ROWB + 16];
    __shared__ unsigned long long red[4][TF_T / 64];
    const int tid = threadIdx.x;
    const int rowb = W * 3;
    const uint8_t* tile = tiles + (size_t)blockIdx.x * H * rowb;
    const bool dw = (rowb % 4 == 0) && (((uintptr_t)tile) % 4 == 0);   // dword path: coalesced 4-B loads
    auto load_row = [&](int y) {
        uint8_t* dst = rows[y % 3];
        const uint8_t* src = tile + (size_t)y * rowb;
        if (dw) {
            for (int i = tid; i < rowb / 4; i += TF_T) reinterpret_cast<unsigned*>(dst)[i] = reinterpret_cast<const unsigned*>(src)[i];
        } else {
            for (int i = tid; i < rowb; i += TF_T) dst[i] = src[i];
        }
    };
    unsigned e[3] = {0u, 0u, 0u};
    unsigned long long E[3] = {0ull, 0ull, 0ull}, S = 0ull;
    unsigned sat = 0u;
    load_row(0);
    if (H > 1) load_row(1);
    __syncthreads();
    for (int y = 0; y < H; ++y) {
        const uint8_t* cur = rows[y % 3];
        const uint8_t* up = rows[(y + 2) % 3];   // row y-1 (valid for y >= 1)
        const uint8_t* dn = rows[(y + 1) % 3];   // row y+1 (valid for y <= H-2)
        const bool yborder = (y == 0) || (y == H - 1);
        for (int i = tid; i < rowb; i += TF_T) {
            const int x = i / 3, c = i - 3 * x;
            int v;
            if (yborder || x == 0 || x == W - 1) {
                v = cur[i];
            } else {
                const int nb = up[i - 3] + up[i] + up[i + 3] + cur[i - 3] + cur[i + 3] + dn[i - 3] + dn[i] + dn[i + 3];
                v = 8 * (int)cur[i] - nb;
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
            }
            e[c] += (unsigned)v;
        }
        for (int x = tid; x < W; x += TF_T) {
            const unsigned r = cur[3 * x], g = cur[3 * x + 1], b = cur[3 * x + 2];
            const unsigned M = max(r, max(g, b)), m = min(r, min(g, b));
            sat += sat_ubyte(M, m);
        }
        if ((y & 1023) == 1023) {   // spill the 32-bit partials long before they can wrap (tall tiles)
            for (int c = 0; c < 3; ++c) { E[c] += e[c]; e[c] = 0u; }
            S += sat; sat = 0u;
        }
        __syncthreads();                         // everyone is done with row y-1
        if (y + 2 < H) load_row(y + 2);          // overwrites the slot of row y-1
        __syncthreads();
    }
    for (int c = 0; c < 3; ++c) E[c] += e[c];
    S += sat;
    unsigned long long v4[4] = {E[0], E[1], E[2], S};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned long long v = v4[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((tid & 63) == 0) red[k][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 4) {
        unsigned long long v = 0ull;
        for (int w = 0; w < TF_T / 64; ++w) v += red[tid][w];
        out[(size_t)blockIdx.x * 4 + tid] = v;
    }
}

}  // namespace

extern "C" int dsmil_tile_stats(const uint8_t* tiles_nhwc, int32_t B, int32_t H, int32_t W, uint64_t* out, void* stream) {
    if (!tiles_nhwc || !out || B <= 0 || H <= 0 || W <= 0) return DSMIL_E_INVALID;
    if (W > TF_MAXW) return DSMIL_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_tile_stats, dim3((unsigned)B), dim3(TF_T), 0, (hipStream_t)stream, tiles_nhwc,
                       (unsigned long long*)out, H, W);
    return hipGetLastError() == hipSuccess ? DSMIL_OK : DSMIL_E_LAUNCH;
}
