// Micro-benchmark: per-CU bandwidth of L2-resident streaming via (a) LDS-DMA and (b) register loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

template <int MODE>
__global__ __launch_bounds__(512) void stream(const char* __restrict__ src, float* out, int iters, long window, long stride_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const char* base = src + (long)blockIdx.x * stride_blocks;
    float acc = 0.f;
    const long per_iter = (long)nw * 4 * 1024;          // each wave moves 4 KB per iteration
    for (int it = 0; it < iters; ++it) {
        const long off = ((long)it * per_iter) % window;
        const char* p = base + off + (wid * 4) * 1024 + lane * 16;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((gbl_ptr)(p + j * 1024), (lds_ptr)(smem + ((it & 3) * nw * 4 + wid * 4 + j) * 1024), 16, 0, 0);
            if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(p + j * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[j].x + v[j].w;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 0) { __syncthreads(); acc = reinterpret_cast<float*>(smem)[tid]; }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const long bytes = 256l << 20;
    char* d; float* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)stream<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int mode = 0; mode < 2; ++mode)
        for (int threads : {256, 512})
            for (int bpc : {1, 2})
                for (long window : {64l << 10, 1l << 20}) {
                    const int nw = threads / 64, blocks = 256 * bpc, iters = 2000;
                    const size_t smem = mode == 0 ? (size_t)4 * nw * 4 * 1024 : 0;
                    const long stride = window;       // each block has its own window (L2/MALL resident: 256*bpc*window <= 512 MB)
                    if ((long)blocks * stride > bytes) continue;
                    auto launch = [&]() {
                        if (mode == 0) hipLaunchKernelGGL(stream<0>, dim3(blocks), dim3(threads), smem, 0, d, o, iters, window, stride);
                        else hipLaunchKernelGGL(stream<1>, dim3(blocks), dim3(threads), smem, 0, d, o, iters, window, stride);
                    };
                    launch(); hipDeviceSynchronize();
                    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    const double tot = (double)blocks * iters * nw * 4 * 1024;
                    printf("%s threads=%d blocks/CU=%d window=%4ldKB: %7.2f TB/s  = %6.1f GB/s per CU\n", mode == 0 ? "LDS-DMA " : "reg-load", threads, bpc, window >> 10,
                           tot / ms / 1e9, tot / ms / 1e6 / 256);
                }
    return 0;
}
