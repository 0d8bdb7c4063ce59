// LayerNorm forward / backward over the channel axis (HBM-bound; C in {32..1024}).
// A row is owned by LPR lanes (16 for C <= 128, else 64); each lane keeps its 16-B / 32-B
// vectors of the row in registers, statistics are fp32 (reference clip.py:181-187).
// Forward can add a periodic fp32 table first (cls/positional embedding, clip.py:274-276)
// and can emit a second affine output sharing the statistics (dist.py:43-45).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXIT = 3;

// lanes per row: the smallest of {4, 16, 32, 64} that leaves at most MAXIT 8-element vectors per lane, so every
// lane has up to three independent loads in flight and the lanes of a row are evenly loaded (C = 768 -> 32 lanes x 3,
// C = 384 -> 16 x 3, C = 96 -> 4 x 3, C = 1024 -> 64 x 2)
inline int lanes_per_row(int C) {
    const int nvec = C / 8;
    if (nvec <= 4 * MAXIT) return 4;
    if (nvec <= 16 * MAXIT) return 16;
    if (nvec <= 32 * MAXIT) return 32;
    return 64;
}

template <typename T, int LPR>
__global__ __launch_bounds__(NT) void ln_fwd_kernel(const dist_ln_args p) {
    constexpr int RPB = NT / LPR;
    const int tid = threadIdx.x;
    const int slot = tid / LPR, lr = tid % LPR;
    const int C = p.C, nvec = C / 8;
    const T* __restrict__ X = static_cast<const T*>(p.x);
    T* __restrict__ Y = static_cast<T*>(p.y);
    T* __restrict__ Y2 = static_cast<T*>(p.y2);
    const float invC = 1.f / (float)C;
    const long stride = (long)gridDim.x * RPB;

    // this lane's columns are the same for every row: affine parameters live in registers
    float w[MAXIT][8], bb[MAXIT][8];
    unsigned vok = 0;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int v = lr + it * LPR;
        if (v < nvec) vok |= 1u << it;
        const int vc = min(v, nvec - 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) { w[it][e] = p.w[vc * 8 + e]; bb[it][e] = p.b[vc * 8 + e]; }
    }
    // software pipeline over this lane group's rows: the vectors of the next row are requested before the current row is
    // reduced (unconditional, clamped: no branch around the loads)
    long row = (long)blockIdx.x * RPB + slot;
    Frag<T> nx[MAXIT];
    {
        const long r0 = min(row, p.rows - 1);
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) frag_load(nx[it], X + r0 * C + min(lr + it * LPR, nvec - 1) * 8);
    }
    for (; row < p.rows; row += stride) {
        float x[MAXIT][8];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[it][e] = frag_get(nx[it], e);
        {
            const long rn = min(row + stride, p.rows - 1);
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) frag_load(nx[it], X + rn * C + min(lr + it * LPR, nvec - 1) * 8);
        }
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            if ((vok >> it) & 1u) {
                if (p.addend) {
                    const float* ad = p.addend + (row % p.addend_period) * C + (lr + it * LPR) * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[it][e] += ad[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) s += x[it][e];
            }
        }
        const float mean = wave_sum(s, LPR) * invC;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            if ((vok >> it) & 1u) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = x[it][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q, LPR) * invC + p.eps);
        if (lr == 0) {
            if (p.mean) p.mean[row] = mean;
            if (p.rstd) p.rstd[row] = rstd;
        }
        if (!Y) continue;                                   // statistics only (the consumer GEMM normalises: DIST_EPI_LNFOLD)
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int v = lr + it * LPR;
            if ((vok >> it) & 1u) {
                Frag<T> o;
#pragma unroll
                for (int e = 0; e < 8; ++e) frag_set(o, e, (x[it][e] - mean) * rstd * w[it][e] + bb[it][e]);
                frag_store_nt(o, Y + row * C + v * 8);
                if (Y2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) frag_set(o, e, (x[it][e] - mean) * rstd * p.w2[v * 8 + e] + p.b2[v * 8 + e]);
                    frag_store_nt(o, Y2 + row * C + v * 8);
                }
            }
        }
    }
}

template <typename T, int LPR, bool DUAL>
__global__ __launch_bounds__(NT) void ln_bwd_kernel(const dist_ln_bwd_args p) {
    constexpr int RPB = NT / LPR;
    __shared__ float red[4][1024];
    const int tid = threadIdx.x;
    const int slot = tid / LPR, lr = tid % LPR;
    const int C = p.C, nvec = C / 8;
    const T* __restrict__ X = static_cast<const T*>(p.x);
    const T* __restrict__ DY = static_cast<const T*>(p.dy);
    const T* __restrict__ DY2 = static_cast<const T*>(p.dy2);
    T* __restrict__ DX = static_cast<T*>(p.dx);
    const T* __restrict__ DXA = p.dx_add ? static_cast<const T*>(p.dx_add) : (p.accumulate_dx ? static_cast<const T*>(p.dx) : nullptr);
    T* __restrict__ DXC = static_cast<T*>(p.dx_copy);
    const float invC = 1.f / (float)C;
    const bool want_w = p.dw || p.db || p.dw2 || p.db2;

    // this lane's columns are the same for every row: weights in registers; all vectors of a row are requested
    // together (unconditional, clamped; validity as a mask) before the first one is consumed
    float w[MAXIT][8];
    int vcol[MAXIT];
    unsigned vok = 0;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int v = lr + it * LPR;
        if (v < nvec) vok |= 1u << it;
        vcol[it] = min(v, nvec - 1) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) w[it][e] = p.w[vcol[it] + e];
    }
    float aw[MAXIT][8], ab[MAXIT][8], aw2[DUAL ? MAXIT : 1][8], ab2[DUAL ? MAXIT : 1][8];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) aw[it][e] = ab[it][e] = 0.f;
#pragma unroll
    for (int it = 0; it < (DUAL ? MAXIT : 1); ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) aw2[it][e] = ab2[it][e] = 0.f;

    for (long row = (long)blockIdx.x * RPB + slot; row < p.rows; row += (long)gridDim.x * RPB) {
        Frag<T> fx[MAXIT], fd[MAXIT], fd2[DUAL ? MAXIT : 1], fa[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            frag_load(fx[it], X + row * C + vcol[it]);
            frag_load(fd[it], DY + row * C + vcol[it]);
            if (DUAL) frag_load(fd2[it], DY2 + row * C + vcol[it]);
        }
        if (DXA) {
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) frag_load(fa[it], DXA + row * C + vcol[it]);
        } else {
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) frag_zero(fa[it]);
        }
        const float mean = p.mean[row], rstd = p.rstd[row];
        float xh[MAXIT][8], g[MAXIT][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            if ((vok >> it) & 1u) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[it][e] = (frag_get(fx[it], e) - mean) * rstd;
                    const float dy = frag_get(fd[it], e);
                    g[it][e] = dy * w[it][e];
                    aw[it][e] += dy * xh[it][e];
                    ab[it][e] += dy;
                }
                if (DUAL) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float dy = frag_get(fd2[it], e);
                        g[it][e] += dy * p.w2[vcol[it] + e];
                        aw2[it][e] += dy * xh[it][e];
                        ab2[it][e] += dy;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1 += g[it][e]; s2 += g[it][e] * xh[it][e]; }
            }
        }
        const float c1 = wave_sum(s1, LPR) * invC, c2 = wave_sum(s2, LPR) * invC;
        if (DX) {
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                if ((vok >> it) & 1u) {
                    Frag<T> o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) frag_set(o, e, frag_get(fa[it], e) + rstd * (g[it][e] - c1 - xh[it][e] * c2));
                    frag_store_nt(o, DX + row * C + vcol[it]);
                    if (DXC) frag_store_nt(o, DXC + row * C + vcol[it]);
                }
            }
        }
    }
    if (!want_w) return;
    // block reduction of the per-lane parameter-gradient sums, one array at a time: lanes that own the same columns
    // (same lr) are first summed inside the wave by shuffles, then across the 4 waves through a [4][C] LDS image
    // (plain stores; LDS atomics on C addresses from every lane serialised thousands of updates per block)
    int which = 0;                                         // index of the array being reduced (two-phase layout)
    auto reduce_out = [&](auto& a, float* __restrict__ dst) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = a[it][e];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1)
                    if (o >= LPR) v += __shfl_xor(v, o, 64);
                a[it][e] = v;
            }
        const int wid = tid >> 6, lane = tid & 63;
        if (lane < LPR) {
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int v = lr + it * LPR;
                if (v < nvec) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) red[wid][v * 8 + e] = a[it][e];
                }
            }
        }
        __syncthreads();
        if (p.partial) {                                   // two-phase: this block's row of array `which`, plain stores
            float* prow = p.partial + ((long)which * gridDim.x + blockIdx.x) * C;
            for (int c = tid; c < C; c += NT) prow[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        } else {
            for (int c = tid; c < C; c += NT) atomicAdd(dst + c, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
        }
        __syncthreads();
        ++which;
    };
    if (p.dw) reduce_out(aw, p.dw);
    if (p.db) reduce_out(ab, p.db);
    if constexpr (DUAL) {
        if (p.dw2) reduce_out(aw2, p.dw2);
        if (p.db2) reduce_out(ab2, p.db2);
    }
}

// second phase of the parameter gradients: array a (dw, db, dw2, db2 in the order they were requested), channel c: the `nblk` per-block
// sums in block order, four independent accumulators; one thread per (array, channel)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int C, float* d0, float* d1, float* d2, float* d3) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int a = e / C, c = e - a * C;
    float* dst = a == 0 ? d0 : (a == 1 ? d1 : (a == 2 ? d2 : d3));
    if (a > 3 || !dst) return;
    const float* col = part + (long)a * nblk * C + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < nblk; b += 4) { s0 += col[(long)b * C]; s1 += col[(long)(b + 1) * C]; s2 += col[(long)(b + 2) * C]; s3 += col[(long)(b + 3) * C]; }
    for (; b < nblk; ++b) s0 += col[(long)b * C];
    dst[c] += (s0 + s1) + (s2 + s3);
}

int grid_for(long rows, int rpb) {
    long g = (rows + rpb - 1) / rpb;
    if (g > 2048) g = 2048;
    return (int)g;
}

}  // namespace

template <typename T, typename A, typename K4, typename K16, typename K32, typename K64>
void launch_by_lpr(int lpr, int grid, hipStream_t s, const A& a, K4 k4, K16 k16, K32 k32, K64 k64) {
    switch (lpr) {
        case 4: hipLaunchKernelGGL(k4, dim3(grid), dim3(NT), 0, s, a); break;
        case 16: hipLaunchKernelGGL(k16, dim3(grid), dim3(NT), 0, s, a); break;
        case 32: hipLaunchKernelGGL(k32, dim3(grid), dim3(NT), 0, s, a); break;
        default: hipLaunchKernelGGL(k64, dim3(grid), dim3(NT), 0, s, a); break;
    }
}

extern "C" int dist_op_layernorm(const dist_ln_args* a, void* stream) {
    if (!a || !a->x || !a->w || !a->b || a->rows <= 0) return DIST_ERR_ARG;
    if (!a->y && (!a->mean || !a->rstd || a->y2)) return DIST_ERR_ARG;      // y == NULL: statistics only
    if (a->C % 8 || a->C > 1024 || a->C < 8) return DIST_ERR_ARG;
    if (a->y2 && (!a->w2 || !a->b2)) return DIST_ERR_ARG;
    if (a->addend && a->addend_period <= 0) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int lpr = lanes_per_row(a->C);
    const int grid = grid_for(a->rows, NT / lpr);
    if (a->dtype == DIST_BF16)
        launch_by_lpr<bf16_t>(lpr, grid, s, *a, ln_fwd_kernel<bf16_t, 4>, ln_fwd_kernel<bf16_t, 16>, ln_fwd_kernel<bf16_t, 32>, ln_fwd_kernel<bf16_t, 64>);
    else
        launch_by_lpr<float>(lpr, grid, s, *a, ln_fwd_kernel<float, 4>, ln_fwd_kernel<float, 16>, ln_fwd_kernel<float, 32>, ln_fwd_kernel<float, 64>);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

namespace {
// one thread per row; consecutive threads read consecutive (S, Q) pairs of a slice (coalesced), slices in index order
__global__ __launch_bounds__(256) void ln_stats_from_partials_kernel(const float2* __restrict__ part, const int slices, const long rows, const float invC,
                                                                      const float eps, float* __restrict__ mean, float* __restrict__ rstd) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= rows) return;
    float s = 0.f, q = 0.f;
    int sl = 0;
    for (; sl + 4 <= slices; sl += 4) {                  // four loads in flight before the first addition
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = part[(long)(sl + u) * rows + m];
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += v[u].x; q += v[u].y; }
    }
    for (; sl < slices; ++sl) { const float2 v = part[(long)sl * rows + m]; s += v.x; q += v.y; }
    const float mu = s * invC;
    mean[m] = mu;
    rstd[m] = rsqrtf(fmaxf(q * invC - mu * mu, 0.f) + eps);
}
}  // namespace

extern "C" int dist_op_ln_stats_from_partials(const float* part, int slices, int64_t rows, int C, float eps, float* mean, float* rstd, void* stream) {
    if (!part || !mean || !rstd || slices <= 0 || rows <= 0 || C != 64 * slices) return DIST_ERR_ARG;
    hipLaunchKernelGGL(ln_stats_from_partials_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(part), slices, (long)rows, 1.f / (float)C, eps, mean, rstd);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_layernorm_bwd(const dist_ln_bwd_args* a, void* stream) {
    if (!a || !a->x || !a->mean || !a->rstd || !a->dy || !a->w || a->rows <= 0) return DIST_ERR_ARG;
    if (a->C % 8 || a->C > 1024 || a->C < 8) return DIST_ERR_ARG;
    if (a->dy2 && !a->w2) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int lpr = lanes_per_row(a->C);
    const int rpb = NT / lpr;
    long g = (a->rows + rpb - 1) / rpb;
    // every block ends with C x 2..4 global atomics on the SAME addresses: with parameter gradients requested the
    // same-line atomic traffic, not HBM, sets the time, so fewer, longer blocks win (measured, tools/bench_ln2.py:
    // dual-input 62 -> 52 us at 256 blocks, single-input 42 -> 33 us at 384); without them 1024 blocks stream best
    const bool want_w = a->dw || a->db || a->dw2 || a->db2;
    static const int gcap_env = DIST_AB_KNOB("DIST_AMD_LN_GCAP", 0);
    const bool two_phase_on = true;                        // (whoever passes `partial` asks for it; the engine does so only under DIST_AMD_LN_TWO_PHASE=1)
    // two-phase parameter gradients (round 3): per-block sums to `partial`, a second launch adds them in block order - no same-line atomics at
    // the end of every block, so the grid no longer has to be capped for them, and the sums are bit-repeatable
    const int narr = (a->dw ? 1 : 0) + (a->db ? 1 : 0) + (a->dw2 ? 1 : 0) + (a->db2 ? 1 : 0);
    long gcap = gcap_env > 0 ? gcap_env : (!want_w ? 1024 : (a->dy2 ? 256 : 384));
    bool two_phase = want_w && two_phase_on && a->partial != nullptr;
    if (two_phase) {
        static const int gcap2_env = DIST_AB_KNOB("DIST_AMD_LN_GCAP2", 0);
        const long gcap2 = gcap2_env > 0 ? gcap2_env : gcap;       // (512 blocks: +0.4 ms in the step - the caps found for the atomics stay)
        if ((g > gcap2 ? gcap2 : g) * 4 * a->C <= a->partial_elems) gcap = gcap2; else two_phase = false;
    }
    if (g > gcap) g = gcap;
    const int grid = (int)g;
    const bool dual = a->dy2 != nullptr;
    dist_ln_bwd_args local = *a;
    if (!two_phase) local.partial = nullptr;
    a = &local;
    if (a->dtype == DIST_BF16) {
        if (dual) launch_by_lpr<bf16_t>(lpr, grid, s, *a, ln_bwd_kernel<bf16_t, 4, true>, ln_bwd_kernel<bf16_t, 16, true>, ln_bwd_kernel<bf16_t, 32, true>, ln_bwd_kernel<bf16_t, 64, true>);
        else launch_by_lpr<bf16_t>(lpr, grid, s, *a, ln_bwd_kernel<bf16_t, 4, false>, ln_bwd_kernel<bf16_t, 16, false>, ln_bwd_kernel<bf16_t, 32, false>, ln_bwd_kernel<bf16_t, 64, false>);
    } else {
        if (dual) launch_by_lpr<float>(lpr, grid, s, *a, ln_bwd_kernel<float, 4, true>, ln_bwd_kernel<float, 16, true>, ln_bwd_kernel<float, 32, true>, ln_bwd_kernel<float, 64, true>);
        else launch_by_lpr<float>(lpr, grid, s, *a, ln_bwd_kernel<float, 4, false>, ln_bwd_kernel<float, 16, false>, ln_bwd_kernel<float, 32, false>, ln_bwd_kernel<float, 64, false>);
    }
    if (two_phase) {                                       // arrays in the order the kernel stored them
        float* d[4] = {nullptr, nullptr, nullptr, nullptr};
        int k = 0;
        if (a->dw) d[k++] = a->dw;
        if (a->db) d[k++] = a->db;
        if (a->dw2) d[k++] = a->dw2;
        if (a->db2) d[k++] = a->db2;
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)((narr * a->C + 255) / 256)), dim3(256), 0, s, a->partial, grid, a->C, d[0], d[1], d[2], d[3]);
    }
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int64_t dist_op_layernorm_bwd_scratch(int64_t rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    const int rpb = NT / lanes_per_row(C);
    long g = (rows + rpb - 1) / rpb;
    if (g > 512) g = 512;
    return g * 4 * (int64_t)C;
}
