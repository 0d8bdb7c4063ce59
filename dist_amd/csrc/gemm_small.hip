// Small-M path of dist_op_gemm_nt (M < 1024: the ada-pooling networks, the cls projections and the head work on
// b*t = 256 or b = 32 rows).  Those launches sit un-overlapped at the end of the forward and the start of the backward
// pass and are pure latency: with 128x128 tiles they ran on 3-24 workgroups and walked K in 6-24 dependent
// load -> LDS -> MFMA steps (15 us each).  Here a workgroup owns a 32 x 64 tile, its four waves split K, and every wave
// requests the MFMA fragments of up to four 32-deep K steps straight from global memory (L2) in one burst - no LDS
// staging, one memory latency per burst - before it multiplies; the four partial tiles meet in LDS for the epilogue.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NT = 256, BM = 32, BN = 64, CH = 4;

template <typename T>
__global__ __launch_bounds__(NT) void gemm_small_kernel(const dist_gemm_args p) {
    __shared__ float red[4][BM][BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int M = (int)p.M, N = p.N, K = p.K;
    const T* __restrict__ A = static_cast<const T*>(p.A);
    const T* __restrict__ B = static_cast<const T*>(p.B);

    const int ksteps = K / 32, per = (ksteps + 3) / 4;
    const int s_beg = wid * per, s_end = min(ksteps, s_beg + per);
    const T* ap[2];
    const T* bp[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) ap[i] = A + (long)min(m0 + i * 16 + li, M - 1) * p.lda + lg * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) bp[j] = B + (long)min(n0 + j * 16 + li, N - 1) * p.ldb + lg * 8;

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = s_beg; s0 < s_end; s0 += CH) {
        Frag<T> fa[CH][2], fb[CH][4];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k0 = min(s0 + c, s_end - 1) * 32;       // past the end: a clamped (unused) re-load, no branch around loads
#pragma unroll
            for (int i = 0; i < 2; ++i) frag_load(fa[c][i], ap[i] + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) frag_load(fb[c][j], bp[j] + k0);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (s0 + c < s_end) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16(fb[c][j], fa[c][i], acc[i][j]);   // swapped: lane holds 4 consecutive columns
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(&red[wid][i * 16 + li][j * 16 + lg * 4]) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    __syncthreads();

    // epilogue: thread -> (row, 8 consecutive columns); same order of operations as the tiled kernels
    const int row = tid >> 3, c8 = (tid & 7) * 8;
    const int m = m0 + row, n = n0 + c8;
    if (m >= M || n >= N) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (red[0][row][c8 + e] + red[1][row][c8 + e]) + (red[2][row][c8 + e] + red[3][row][c8 + e]);
    const int flags = p.flags;
    if (flags & DIST_EPI_BIAS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.bias[n + e] + (p.bias2 ? p.bias2[n + e] : 0.f);
    }
    if (flags & DIST_EPI_MULG) {
        Frag<T> x;
        frag_load(x, static_cast<const T*>(p.aux) + (long)m * p.ldaux + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= qgelu_grad_t<T>(frag_get(x, e));
    }
    if (flags & DIST_EPI_RES) {
        Frag<T> x;
        frag_load(x, static_cast<const T*>(p.res) + (long)m * p.ldres + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += frag_get(x, e);
    }
    Frag<T> o;
    if ((flags & DIST_EPI_ACT2) && !p.C) {
#pragma unroll
        for (int e = 0; e < 8; ++e) frag_set(o, e, qgelu_t<T>(v[e]));
        frag_store(o, static_cast<T*>(p.C2) + (long)m * p.ldc2 + n);
        return;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) frag_set(o, e, v[e]);
    frag_store(o, static_cast<T*>(p.C) + (long)m * p.ldc + n);
    if (flags & DIST_EPI_ACT2) {                             // second output = quickgelu(stored value)
#pragma unroll
        for (int e = 0; e < 8; ++e) frag_set(o, e, qgelu_t<T>(frag_get(o, e)));
        frag_store(o, static_cast<T*>(p.C2) + (long)m * p.ldc2 + n);
    }
}

}  // namespace

// returns 1 if handled, 0 if the shape does not qualify (caller falls through), <0 on error
int dist_k_gemm_small(const dist_gemm_args* a, hipStream_t s) {
    if (a->M >= 1024 || a->taps != 1 || a->amap.mode != DIST_RM_PLAIN || a->omap.mode != DIST_OM_PLAIN) return 0;
    if (a->K % 32 || a->N % 8 || a->lda % 8 || a->ldb % 8 || (a->flags & DIST_EPI_MULG_POST)) return 0;
    const int epv = 8;
    if ((a->C && a->ldc % epv) || ((a->flags & DIST_EPI_ACT2) && a->ldc2 % epv) || ((a->flags & DIST_EPI_RES) && a->ldres % epv) ||
        ((a->flags & DIST_EPI_MULG) && a->ldaux % epv)) return 0;
    const long tiles = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
    if (a->dtype == DIST_BF16) hipLaunchKernelGGL(gemm_small_kernel<bf16_t>, dim3((unsigned)tiles), dim3(NT), 0, s, *a);
    else hipLaunchKernelGGL(gemm_small_kernel<float>, dim3((unsigned)tiles), dim3(NT), 0, s, *a);
    HIP_CHECK_RET(hipGetLastError());
    return 1;
}
