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

// ---------------------------------------------------------------------------------------------------------------------
// dsmil_csv_parse_f32 — the way back (train_tcga.py:27-32 `pd.read_csv(path)` -> `torch.tensor(..., dtype=torch.float32)`):
// rows of decimal fields -> float32.  A plain decimal field of at most 15 significant digits and at most 22 decimals is
// n / 10^f with n < 2^53 and 10^f exact in double: ONE correctly rounded division — the double pandas' parser produces for
// the '%.4f' fields the reference writes (its xstrtod divides by 1e4 once as well) — then the same round-to-nearest cast to
// float32 as torch's.  Everything else (exponents, more digits, 'inf', 'nan') goes through strtod; an empty field is NaN.
// Blank lines are skipped (pandas' default).  Returns the number of rows parsed, or DSMIL_E_INVALID for a field that is not a
// number / a row of another width / too many rows for `out` — the caller then lets pandas read the file.
extern "C" int64_t dsmil_csv_parse_f32(const char* text, int64_t nbytes, int64_t cols, float* out, int64_t max_rows) {
    if (!text || !out || nbytes < 0 || cols <= 0 || max_rows < 0) return DSMIL_E_INVALID;
    static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                                   1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char* p = text;
    const char* const end = text + nbytes;
    int64_t rows = 0;
    while (p < end) {
        // a blank line?
        if (*p == '\n') { ++p; continue; }
        if (*p == '\r' && p + 1 < end && p[1] == '\n') { p += 2; continue; }
        if (rows >= max_rows) return DSMIL_E_INVALID;
        float* dst = out + rows * cols;
        for (int64_t c = 0; c < cols; ++c) {
            const char* f0 = p;
            bool neg = false;
            if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
            uint64_t n = 0;
            int nd = 0, nf = 0;
            bool any = false, plain = true;
            while (p < end && *p >= '0' && *p <= '9') { if (nd < 19) { n = n * 10 + (uint64_t)(*p - '0'); } if (n || nd) ++nd; any = true; ++p; }
            if (p < end && *p == '.') {
                ++p;
                while (p < end && *p >= '0' && *p <= '9') { if (nd < 19) { n = n * 10 + (uint64_t)(*p - '0'); } if (n || nd) ++nd; ++nf; any = true; ++p; }
            }
            if (p < end && *p != ',' && *p != '\n' && *p != '\r') plain = false;     // an exponent, 'inf', 'nan', or junk
            double v;
            if (plain && any && nd <= 15 && nf <= 22) {
                v = (double)n / P10[nf];
                if (neg) v = -v;
            } else if (plain && !any && p == f0) {
                v = NAN;                                                              // an empty field
            } else {
                // the general case: the field as a C string through strtod
                const char* q = f0;
                while (q < end && *q != ',' && *q != '\n' && *q != '\r') ++q;
                char tmp[64];
                const size_t len = (size_t)(q - f0);
                if (len == 0 || len >= sizeof(tmp)) return DSMIL_E_INVALID;
                std::memcpy(tmp, f0, len);
                tmp[len] = 0;
                char* e = nullptr;
                v = std::strtod(tmp, &e);
                if (e != tmp + len) return DSMIL_E_INVALID;
                p = q;
            }
            dst[c] = (float)v;
            if (c + 1 < cols) {
                if (p >= end || *p != ',') return DSMIL_E_INVALID;
                ++p;
            }
        }
        if (p < end && *p == '\r') ++p;
        if (p < end) {
            if (*p != '\n') return DSMIL_E_INVALID;
            ++p;
        }
        ++rows;
    }
    return rows;
}

// ---------------------------------------------------------------------------------------------------------------------
// dsmil_read_files — the loader's file reads (compute_feats.py:21-56: a DataLoader worker opens every tile file): n files
// back to back into one buffer, in one call that holds no interpreter lock.  A Python thread pays ~25 us of interpreter time per
// `open` / `read` / `close`, serialised over all threads: 100 ms per 4 000-tile bag, more than the device needs to decode and
// embed it.  paths: n NUL-terminated strings, path_off[i] = offset of path i in `paths`.  out == NULL: only the sizes (fstat) —
// the caller sizes the buffer from the returned total; out != NULL: file i goes to out + (sum of sizes[0..i)), at most sizes[i]
// bytes as stat'ed now.  sizes[i] = -1 for a file that cannot be opened / read (the caller raises what `open` would).
// Returns the total number of bytes, or DSMIL_E_WORKSPACE when `cap` is too small, DSMIL_E_INVALID for bad arguments.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

extern "C" int64_t dsmil_read_files(const char* paths, const int64_t* path_off, int32_t n, uint8_t* out, int64_t cap, int64_t* sizes) {
    if (!paths || !path_off || !sizes || n < 0 || (out && cap < 0)) return DSMIL_E_INVALID;
    int64_t total = 0;
    for (int32_t i = 0; i < n; ++i) {
        const char* path = paths + path_off[i];
        const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) { sizes[i] = -1; continue; }
        struct stat st;
        if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); sizes[i] = -1; continue; }
        int64_t sz = (int64_t)st.st_size;
        if (out) {
            if (total + sz > cap) { ::close(fd); return DSMIL_E_WORKSPACE; }
            int64_t got = 0;
            while (got < sz) {
                const ssize_t r = ::read(fd, out + total + got, (size_t)(sz - got));
                if (r < 0) { got = -1; break; }
                if (r == 0) break;                       // (the file shrank under us: what was read is what there is)
                got += r;
            }
            if (got < 0) { ::close(fd); sizes[i] = -1; continue; }
            sz = got;
        }
        ::close(fd);
        sizes[i] = sz;
        total += sz;
    }
    return total;
}
