// Shared machinery of the LDS-DMA (global_load_lds) GEMM kernels with row maps:
//  * FastDiv: exact unsigned division by a runtime constant for row indices < 2^24 and
//    divisors < 2^16 (one 64-bit multiply + shift instead of the ~25-instruction software
//    divide hipcc emits for `/` and `%`; the row maps need 1-3 divisions per fetched row);
//  * RowMapDev: dist_rowmap with its divisors pre-inverted on the host;
//  * a 64-byte zero page in the code object's data segment: LDS-DMA cannot write
//    immediates, so lanes whose tap falls into the convolution's zero padding (or beyond the
//    matrix) fetch from this page instead.
#pragma once
#include "common.h"

struct FastDiv {
    unsigned d;
    unsigned long long mul;     // ceil(2^40 / d)
};
inline FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.d = d > 0 ? (unsigned)d : 1u;
    f.mul = ((1ull << 40) + f.d - 1) / f.d;
    return f;
}
DEV unsigned fd_div(const FastDiv& f, unsigned m) { return (unsigned)(((unsigned long long)m * f.mul) >> 40); }
DEV void fd_divmod(const FastDiv& f, unsigned m, unsigned& q, unsigned& r) { q = fd_div(f, m); r = m - q * f.d; }

struct RowMapDev {
    int mode, p0, p1, sign;
    FastDiv d0, d1, d2;         // SHIFT: d0 = G | SPATIAL: d0 = grid^2, d1 = grid | STRIDED: d0 = N | SKIPCLS: d0 = N
};
inline RowMapDev make_rowmap_dev(const dist_rowmap& m) {
    RowMapDev r;
    r.mode = m.mode; r.p0 = m.p0; r.p1 = m.p1; r.sign = m.sign;
    r.d0 = r.d1 = r.d2 = make_fastdiv(1);
    switch (m.mode) {
        case DIST_RM_SHIFT: r.d0 = make_fastdiv(m.p0); break;
        case DIST_RM_SPATIAL: r.d0 = make_fastdiv(m.p0 * m.p0); r.d1 = make_fastdiv(m.p0); break;
        case DIST_RM_STRIDED: r.d0 = make_fastdiv(m.p1); break;
        case DIST_RM_SKIPCLS: r.d0 = make_fastdiv(m.p0); break;
        default: break;
    }
    return r;
}
inline bool rowmap_fast_ok(const dist_rowmap& m, long rows) {
    if (rows >= (1l << 24)) return false;
    const long big = 65536;
    switch (m.mode) {
        case DIST_RM_SHIFT: return m.p0 > 0 && m.p0 < big;
        case DIST_RM_SPATIAL: return m.p0 > 0 && (long)m.p0 * m.p0 < big;
        case DIST_RM_STRIDED: return m.p1 > 0 && m.p1 < big;
        case DIST_RM_SKIPCLS: return m.p0 > 0 && m.p0 < big;
        default: return true;
    }
}
// same semantics as rowmap_src() in common.h (-1 = zero padding)
DEV int rowmap_src_fast(const RowMapDev& rm, int m, int tap, int taps) {
    switch (rm.mode) {
        case DIST_RM_SHIFT: {
            const int off = rm.sign * (tap - taps / 2) * rm.p1;
            unsigned q, r;
            fd_divmod(rm.d0, (unsigned)m, q, r);
            const int rr = (int)r + off;
            return (rr >= 0 && rr < rm.p0) ? m + off : -1;
        }
        case DIST_RM_SPATIAL: {
            const int gsz = rm.p0;
            const int t3 = tap / 3;
            const int dy = (t3 - 1) * rm.sign, dx = (tap - t3 * 3 - 1) * rm.sign;
            unsigned q, n, yy, xx;
            fd_divmod(rm.d0, (unsigned)m, q, n);
            fd_divmod(rm.d1, n, yy, xx);
            const int y = (int)yy + dy, x = (int)xx + dx;
            return (y >= 0 && y < gsz && x >= 0 && x < gsz) ? m + dy * gsz + dx : -1;
        }
        case DIST_RM_STRIDED: {
            unsigned bj, n;
            fd_divmod(rm.d0, (unsigned)m, bj, n);
            return (int)((bj * rm.p0 + tap) * rm.p1 + n);
        }
        case DIST_RM_SKIPCLS: {
            unsigned bj, n;
            fd_divmod(rm.d0, (unsigned)m, bj, n);
            return (int)(bj * (rm.p0 + 1) + 1 + n);
        }
        default: return m;
    }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// zero page (one copy per translation unit: no relocatable device code needed)
static __device__ uint4 dist_zero_page[4];
