// Micro-benchmark (round 5): what ONE wave per SIMD can issue, against two - the question behind the ping-pong GEMM (profiles/r05_gemm_pingpong.md).
//   * bf16 MFMA issue rate, v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16, 16 independent accumulators per wave, with 0 / 1 ds_read_b128
//     per two MFMAs, 1 or 2 waves per SIMD (256- / 512-thread blocks, one block per CU: 128 KB of dynamic LDS);
//   * the cost of a block: empty 512-thread blocks with 128 KB of LDS and 256 registers, 7 rounds of 256 (what a K = 768 GEMM launch dispatches).
// hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip && ./mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int READS>   // SHAPE 16: 16x16x32, 32: 32x32x16; READS: one ds_read_b128 per two MFMAs
__global__ void mfma_loop(float* out, int iters, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    const char* rd = smem + ((threadIdx.x >> 6) * 8192 + lane * 16) % (96 << 10);
    bf16x8 f[8];
    for (int i = 0; i < 8; ++i) f[i] = a;
    __syncthreads();
    const long long t0 = clock64();
    if constexpr (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i & 7], b, acc[i], 0, 0, 0);
                if (READS && (i & 1)) f[(i >> 1) & 7] = *reinterpret_cast<const bf16x8*>(rd + (i >> 1) * 1024);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {                 // 8 x 32x32x16 = the FLOPs of 16 x 16x16x32
                acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i], b, acc[i & 3], 0, 0, 0);
                if (READS) f[i] = *reinterpret_cast<const bf16x8*>(rd + i * 1024);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ __launch_bounds__(512, 1) void empty_block(float* out, int spin) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float x[200];                                            // ~ the GEMM's register allocation
    for (int i = 0; i < 200; ++i) x[i] = (float)(threadIdx.x + i);
    asm volatile("" ::: "memory");
    if (spin < 0) { float s = 0.f; for (int i = 0; i < 200; ++i) s += x[i]; out[threadIdx.x] = s + smem[threadIdx.x]; }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    auto run = [&](auto kernel, int threads, const char* tag) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
        hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), 128 << 10, 0, out, 100, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), 128 << 10, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double mfma16_per_simd = (double)iters * 16 * (threads / 256);   // in units of one 16x16x32 (a 32x32x16 counts as two)
        const double tf = 256.0 * 4 * mfma16_per_simd * 16384 / (ms * 1e-3) / 1e12;
        printf("%-58s %2d wave(s)/SIMD: %8.1f us  %7.1f TF  %6.2f cycles of s_memtime per 16x16x32-equivalent per SIMD (clock64: %lld)\n", tag, threads / 256,
               ms * 1e3, tf, (double)c / mfma16_per_simd, c);
    };
    for (int threads : {256, 512}) {
        run(mfma_loop<16, 0>, threads, "16x16x32 bf16, no LDS reads");
        run(mfma_loop<16, 1>, threads, "16x16x32 bf16, one ds_read_b128 per two MFMAs");
        run(mfma_loop<32, 0>, threads, "32x32x16 bf16, no LDS reads");
        run(mfma_loop<32, 1>, threads, "32x32x16 bf16, one ds_read_b128 per MFMA");
    }
    // dispatch cost of a GEMM-sized block
    hipFuncSetAttribute(reinterpret_cast<const void*>(empty_block), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
    for (int blocks : {256, 1792, 256 * 28}) {
        for (int spin : {0, 64}) {
            hipLaunchKernelGGL(empty_block, dim3(blocks), dim3(512), 128 << 10, 0, out, spin);
            hipDeviceSynchronize();
            float best = 1e9f;
            for (int r = 0; r < 5; ++r) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(empty_block, dim3(blocks), dim3(512), 128 << 10, 0, out, spin);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("empty 512-thread blocks, 128 KB LDS, spin %2d: %5d blocks = %4.1f rounds of 256: %8.2f us  -> %6.2f us per round\n", spin, blocks, blocks / 256.0, best * 1e3,
                   best * 1e3 / (blocks / 256.0));
        }
    }
    return 0;
}
