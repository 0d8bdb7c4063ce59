// Per-row symmetric quantisation to OCP e4m3 (dist_op_quant_rows_fp8): the operands of the DIST_EPI_FP8 GEMM (BASELINE config 5:
// fp8 frozen spatial branch).  One wave per row: the row is read once (16-byte vectors, kept in registers), amax by a wave
// reduction, then 8 e4m3 bytes per vector.  HBM-bound: 2 (bf16) or 4 (fp32) bytes in, 1 byte out per element.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int QV = 16;                                     // vectors of 8 elements per lane: rows up to 8192 elements

template <typename T>
__global__ __launch_bounds__(NT) void quant_rows_fp8_kernel(const T* __restrict__ x, const long rows, const int K, const int ld,
                                                            unsigned char* __restrict__ q, const int ldq, float* __restrict__ scale) {
    const long row = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const T* xr = x + row * ld;
    const int nv = K / 8;
    Frag<T> f[QV];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < QV; ++i) {
        const int v = lane + i * 64;
        if (v < nv) {
            frag_load(f[i], xr + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(frag_get(f[i], e)));
        }
    }
    amax = wave_max(amax, 64);
    const float inv = amax > 0.f ? 448.f / amax : 1.f;     // IEEE division: the same factor a host restatement computes
    if (lane == 0) scale[row] = amax > 0.f ? amax / 448.f : 1.f;
    unsigned char* qr = q + row * ldq;
#pragma unroll
    for (int i = 0; i < QV; ++i) {
        const int v = lane + i * 64;
        if (v < nv) {
            int lo = 0, hi = 0;                             // v_cvt_pk_fp8_f32: two fp32 -> two e4m3 (round to nearest even) into one half of a dword
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(frag_get(f[i], 0) * inv, frag_get(f[i], 1) * inv, lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(frag_get(f[i], 2) * inv, frag_get(f[i], 3) * inv, lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(frag_get(f[i], 4) * inv, frag_get(f[i], 5) * inv, hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(frag_get(f[i], 6) * inv, frag_get(f[i], 7) * inv, hi, true);
            *reinterpret_cast<int2*>(qr + v * 8) = make_int2(lo, hi);
        }
    }
}

// out[r] = scale[r] * sum_k e4m3(q[r][k]): one wave per row, 8 bytes per lane per step, fixed order
__global__ __launch_bounds__(NT) void fp8_rowsum_kernel(const unsigned char* __restrict__ q, const float* __restrict__ scale, const long rows, const int K,
                                                        const int ldq, float* __restrict__ out) {
    const long row = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const unsigned char* qr = q + row * ldq;
    float acc = 0.f;
    for (int v = lane; v < K / 8; v += 64) {
        const int2 w = *reinterpret_cast<const int2*>(qr + v * 8);
        typedef float f2_t __attribute__((ext_vector_type(2)));
        const f2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(w.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(w.x, true);
        const f2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(w.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(w.y, true);
        acc += ((a[0] + a[1]) + (b[0] + b[1])) + ((c[0] + c[1]) + (d[0] + d[1]));
    }
    acc = wave_sum(acc, 64);
    if (lane == 0) out[row] = acc * scale[row];
}

// scale[i] = 2^ceil(log2(max(amax[i], tiny) * margin / 448)); amax[i] = 0
__global__ void fp8_scale_update_kernel(float* __restrict__ amax, float* __restrict__ scale, const int n, const float margin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = fmaxf(amax[i], 5.9604645e-8f /* 2^-24 */) * margin * (1.0f / 448.f);   // (an all-zero tensor must not give a scale whose square underflows in the attention scores)
    int e;
    const float f = frexpf(t, &e);                         // t = f * 2^e, f in [0.5, 1)
    scale[i] = ldexpf(1.0f, f == 0.5f ? e - 1 : e);        // smallest power of two >= t
    amax[i] = 0.f;
}

template <typename T>
__global__ __launch_bounds__(NT) void amax_kernel(const T* __restrict__ x, const long n, float* __restrict__ amax) {
    float a = 0.f;
    for (long v = (long)blockIdx.x * NT + threadIdx.x; v < n / 8; v += (long)gridDim.x * NT) {
        Frag<T> f;
        frag_load(f, x + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fmaxf(a, fabsf(frag_get(f, e)));
    }
    a = wave_max(a, 64);
    if ((threadIdx.x & 63) == 0 && a > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(a));
}

}  // namespace

extern "C" int dist_op_fp8_scale_update(float* amax, float* scale, int n, float margin, void* stream) {
    if (!amax || !scale || n <= 0 || !(margin >= 1.f)) return DIST_ERR_ARG;
    hipLaunchKernelGGL(fp8_scale_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), amax, scale, n, margin);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_amax(const void* x, int dtype, int64_t n, float* amax, void* stream) {
    if (!x || !amax || n <= 0 || n % 8 || (dtype != DIST_BF16 && dtype != DIST_F32)) return DIST_ERR_ARG;
    const unsigned blocks = (unsigned)min((long)((n / 8 + NT - 1) / NT), 2048l);
    if (dtype == DIST_BF16) hipLaunchKernelGGL(amax_kernel<bf16_t>, dim3(blocks), dim3(NT), 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(x), (long)n, amax);
    else hipLaunchKernelGGL(amax_kernel<float>, dim3(blocks), dim3(NT), 0, static_cast<hipStream_t>(stream), static_cast<const float*>(x), (long)n, amax);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_fp8_rowsum(const void* q, const float* scale, int64_t rows, int K, int ldq, float* out, void* stream) {
    if (!q || !scale || !out || rows <= 0 || K <= 0 || K % 8 || ldq % 8 || ldq < K) return DIST_ERR_ARG;
    hipLaunchKernelGGL(fp8_rowsum_kernel, dim3((unsigned)((rows + NT / 64 - 1) / (NT / 64))), dim3(NT), 0, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned char*>(q), scale, (long)rows, K, ldq, out);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_quant_rows_fp8(const void* x, int dtype, int64_t rows, int K, int ld, void* q, int ldq, float* scale, void* stream) {
    if (!x || !q || !scale || rows <= 0 || K <= 0 || K % 8 || ld % 8 || ldq % 8 || K > QV * 64 * 8 || ld < K || ldq < K) return DIST_ERR_ARG;
    if (dtype != DIST_BF16 && dtype != DIST_F32) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)((rows + NT / 64 - 1) / (NT / 64));
    if (dtype == DIST_BF16)
        hipLaunchKernelGGL(quant_rows_fp8_kernel<bf16_t>, dim3(blocks), dim3(NT), 0, s, static_cast<const bf16_t*>(x), (long)rows, K, ld,
                           static_cast<unsigned char*>(q), ldq, scale);
    else
        hipLaunchKernelGGL(quant_rows_fp8_kernel<float>, dim3(blocks), dim3(NT), 0, s, static_cast<const float*>(x), (long)rows, K, ld,
                           static_cast<unsigned char*>(q), ldq, scale);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}
