// csv_format.hip — HOST code only: the text of the reference's feature files.
//
// compute_feats.py:80-82 writes every bag as `pd.DataFrame(feats).to_csv(path, index=False, float_format='%.4f')`: 10 000 x 512
// features are 5.1 M numbers, and pandas formats them one Python `'%.4f' % x` at a time — 5.4 s per bag on this host against
// 0.08-0.17 s to COMPUTE the bag's features on the device.  The file format is the reference's (train_tcga.py:27-32 reads it
// back with pd.read_csv), so the format stays and the formatting moves here: dsmil_csv_format_f32 produces the same bytes.
//
// '%.df' of a float32 x is the correctly rounded (ties to even) d-digit decimal of x's exact value.  x has a 24-bit
// significand and 10^d = 2^d 5^d with 5^d < 2^21 for d <= 9, so |x| 10^d has at most 45 significant bits: the double product is
// EXACT, nearbyint() of it (round-to-nearest-even, the default mode) is the exact tie-to-even integer, and the digits of that
// integer are the digits printf prints.  Values of 1e9 and more go through snprintf itself.  NaN is an empty field (pandas'
// na_rep = ''), infinities are 'inf' / '-inf', the sign of a negative zero or of a negative value that rounds to zero is kept
// ('-0.0000': what Python's % gives).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "dsmil_hip.h"

namespace {

inline char* put_uint(char* p, uint64_t v) {   // decimal digits of v, no padding
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

}  // namespace

extern "C" int64_t dsmil_csv_format_f32(const float* x, int64_t rows, int64_t cols, int64_t row_stride, int32_t decimals,
                                        char* out, int64_t cap) {
    if (!x || !out || rows < 0 || cols <= 0 || row_stride < cols || decimals < 0 || decimals > 9) return DSMIL_E_INVALID;
    static const uint64_t P10[10] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull};
    const uint64_t p10 = P10[decimals];
    const double scale = (double)p10;
    char* p = out;
    char* const end = out + cap;
    for (int64_t r = 0; r < rows; ++r) {
        const float* row = x + r * row_stride;
        for (int64_t c = 0; c < cols; ++c) {
            if (end - p < 64) return DSMIL_E_WORKSPACE;          // (a field is at most 1 + 39 + 1 + 9 characters + separator)
            const double v = (double)row[c];
            if (std::isnan(v)) {
                // empty field
            } else if (std::isinf(v)) {
                if (v < 0) *p++ = '-';
                *p++ = 'i'; *p++ = 'n'; *p++ = 'f';
            } else {
                const double a = std::fabs(v);
                if (a < 1e9) {
                    const uint64_t n = (uint64_t)std::nearbyint(a * scale);   // exact product, ties to even
                    if (std::signbit(v)) *p++ = '-';
                    p = put_uint(p, n / p10);
                    if (decimals) {
                        *p++ = '.';
                        uint64_t f = n % p10;
                        for (int d = decimals - 1; d >= 0; --d) { p[d] = (char)('0' + f % 10); f /= 10; }
                        p += decimals;
                    }
                } else {
                    p += std::snprintf(p, (size_t)(end - p), "%.*f", (int)decimals, v);
                }
            }
            *p++ = c + 1 < cols ? ',' : '\n';
        }
    }
    return (int64_t)(p - out);
}
