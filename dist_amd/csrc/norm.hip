// LayerNorm forward / backward over the channel axis (HBM-bound; C in {32..1024}).
// A row is owned by LPR lanes (16 for C <= 128, else 64); each lane keeps its 16-B / 32-B
// vectors of the row in registers, statistics are fp32 (reference clip.py:181-187).
// Forward can add a periodic fp32 table first (cls/positional embedding, clip.py:274-276)
// and can emit a second affine output sharing the statistics (dist.py:43-45).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXIT = 2;

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

    for (long row = (long)blockIdx.x * RPB + slot; row < p.rows; row += (long)gridDim.x * RPB) {
        float x[MAXIT][8];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int v = lr + it * LPR;
            if (v < nvec) {
                Frag<T> f;
                frag_load(f, X + row * C + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[it][e] = frag_get(f, e);
                if (p.addend) {
                    const float* ad = p.addend + (row % p.addend_period) * C + v * 8;
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
            const int v = lr + it * LPR;
            if (v < nvec) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = x[it][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q, LPR) * invC + p.eps);
        if (lr == 0) {
            if (p.mean) p.mean[row] = mean;
            if (p.rstd) p.rstd[row] = rstd;
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int v = lr + it * LPR;
            if (v < nvec) {
                Frag<T> o;
#pragma unroll
                for (int e = 0; e < 8; ++e) frag_set(o, e, (x[it][e] - mean) * rstd * p.w[v * 8 + e] + p.b[v * 8 + e]);
                frag_store(o, Y + row * C + v * 8);
                if (Y2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) frag_set(o, e, (x[it][e] - mean) * rstd * p.w2[v * 8 + e] + p.b2[v * 8 + e]);
                    frag_store(o, Y2 + row * C + v * 8);
                }
            }
        }
    }
}

template <typename T, int LPR>
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
    const T* __restrict__ DXA = static_cast<const T*>(p.dx_add);
    T* __restrict__ DXC = static_cast<T*>(p.dx_copy);
    const float invC = 1.f / (float)C;
    const bool want_w = p.dw || p.db || p.dw2 || p.db2;

    for (int i = tid; i < 4 * 1024; i += NT) (&red[0][0])[i] = 0.f;
    __syncthreads();

    float aw[MAXIT][8], ab[MAXIT][8], aw2[MAXIT][8], ab2[MAXIT][8];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) aw[it][e] = ab[it][e] = aw2[it][e] = ab2[it][e] = 0.f;

    for (long row = (long)blockIdx.x * RPB + slot; row < p.rows; row += (long)gridDim.x * RPB) {
        const float mean = p.mean[row], rstd = p.rstd[row];
        float xh[MAXIT][8], g[MAXIT][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int v = lr + it * LPR;
            if (v < nvec) {
                Frag<T> fx, fd;
                frag_load(fx, X + row * C + v * 8);
                frag_load(fd, DY + row * C + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[it][e] = (frag_get(fx, e) - mean) * rstd;
                    const float dy = frag_get(fd, e);
                    g[it][e] = dy * p.w[v * 8 + e];
                    aw[it][e] += dy * xh[it][e];
                    ab[it][e] += dy;
                }
                if (DY2) {
                    frag_load(fd, DY2 + row * C + v * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float dy = frag_get(fd, e);
                        g[it][e] += dy * p.w2[v * 8 + e];
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
                const int v = lr + it * LPR;
                if (v < nvec) {
                    Frag<T> o;
                    if (DXA) frag_load(o, DXA + row * C + v * 8);
                    else if (p.accumulate_dx) frag_load(o, DX + row * C + v * 8);
                    else frag_zero(o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) frag_set(o, e, frag_get(o, e) + rstd * (g[it][e] - c1 - xh[it][e] * c2));
                    frag_store(o, DX + row * C + v * 8);
                    if (DXC) frag_store(o, DXC + row * C + v * 8);
                }
            }
        }
    }
    if (!want_w) return;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int v = lr + it * LPR;
        if (v < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                atomicAdd(&red[0][v * 8 + e], aw[it][e]);
                atomicAdd(&red[1][v * 8 + e], ab[it][e]);
                if (DY2) {
                    atomicAdd(&red[2][v * 8 + e], aw2[it][e]);
                    atomicAdd(&red[3][v * 8 + e], ab2[it][e]);
                }
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += NT) {
        if (p.dw) atomicAdd(p.dw + c, red[0][c]);
        if (p.db) atomicAdd(p.db + c, red[1][c]);
        if (DY2 && p.dw2) atomicAdd(p.dw2 + c, red[2][c]);
        if (DY2 && p.db2) atomicAdd(p.db2 + c, red[3][c]);
    }
}

int grid_for(long rows, int rpb) {
    long g = (rows + rpb - 1) / rpb;
    if (g > 2048) g = 2048;
    return (int)g;
}

}  // namespace

extern "C" int dist_op_layernorm(const dist_ln_args* a, void* stream) {
    if (!a || !a->x || !a->y || !a->w || !a->b || a->rows <= 0) return DIST_ERR_ARG;
    if (a->C % 8 || a->C > 1024 || a->C < 8) return DIST_ERR_ARG;
    if (a->y2 && (!a->w2 || !a->b2)) return DIST_ERR_ARG;
    if (a->addend && a->addend_period <= 0) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool small = a->C <= 128;
    const int rpb = small ? NT / 16 : NT / 64;
    const int grid = grid_for(a->rows, rpb);
    if (a->dtype == DIST_BF16) {
        if (small) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, 16>), dim3(grid), dim3(NT), 0, s, *a);
        else hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, 64>), dim3(grid), dim3(NT), 0, s, *a);
    } else {
        if (small) hipLaunchKernelGGL((ln_fwd_kernel<float, 16>), dim3(grid), dim3(NT), 0, s, *a);
        else hipLaunchKernelGGL((ln_fwd_kernel<float, 64>), dim3(grid), dim3(NT), 0, s, *a);
    }
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_layernorm_bwd(const dist_ln_bwd_args* a, void* stream) {
    if (!a || !a->x || !a->mean || !a->rstd || !a->dy || !a->w || a->rows <= 0) return DIST_ERR_ARG;
    if (a->C % 8 || a->C > 1024 || a->C < 8) return DIST_ERR_ARG;
    if (a->dy2 && !a->w2) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool small = a->C <= 128;
    const int rpb = small ? NT / 16 : NT / 64;
    long g = (a->rows + rpb - 1) / rpb;
    if (g > 512) g = 512;          // few blocks: each ends with C x 4 global atomics
    const int grid = (int)g;
    if (a->dtype == DIST_BF16) {
        if (small) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 16>), dim3(grid), dim3(NT), 0, s, *a);
        else hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 64>), dim3(grid), dim3(NT), 0, s, *a);
    } else {
        if (small) hipLaunchKernelGGL((ln_bwd_kernel<float, 16>), dim3(grid), dim3(NT), 0, s, *a);
        else hipLaunchKernelGGL((ln_bwd_kernel<float, 64>), dim3(grid), dim3(NT), 0, s, *a);
    }
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}
