// Micro-benchmark (round 6): how many bytes can ONE CU keep in flight, through LDS-DMA (buffer/global_load ... lds), through register loads, and through both at once?
// The GEMM's K loop runs at in-flight / latency with ~64 KB of LDS-DMA in flight (profiles/r05_gemm_pingpong.md); VERDICT r05 asked whether the B operand could travel
// straight to VGPRs instead.  That only helps if register loads have in-flight capacity of their OWN.  Few blocks (the fabric is not the bound), a region far larger than
// L2 (every request pays the fabric latency), every wave keeps `D` 1 KB requests outstanding: GB/s per CU = in-flight bytes / latency.
//   hipcc --offload-arch=gfx950 -O3 -o inflight tools/ubench/inflight.hip && ./inflight
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// MODE 0: D LDS-DMA pieces per wave in flight; MODE 1: D register loads (16 B per lane) per wave; MODE 2: D of each
template <int MODE, int D>
__global__ __launch_bounds__(512) void stream(const char* __restrict__ src, float* out, int iters, long span) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    // every wave walks its own stream of 1 KB pieces, 4 KB apart from the next wave's, the whole block `span` bytes apart from the next block
    const char* base = src + (long)blockIdx.x * span + (long)wid * 1024 + lane * 16;
    const long step = (long)nw * 1024;
    float acc = 0.f;
    float4 v[D];
    long off = 0;
    // prologue: D (or 2 D) requests
    if (MODE != 1) {
#pragma unroll
        for (int j = 0; j < D; ++j) { __builtin_amdgcn_global_load_lds((gbl_ptr)(base + off), (lds_ptr)(smem + (wid * D + j) * 1024), 16, 0, 0); off += step; }
    }
    if (MODE != 0) {
#pragma unroll
        for (int j = 0; j < D; ++j) { v[j] = *reinterpret_cast<const float4*>(base + off); off += step; }
    }
    for (int it = 0; it < iters; ++it) {
        // retire the oldest request(s), issue new one(s): the number outstanding stays D (2 D in MODE 2)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (MODE == 0) {
                wait_vm<D - 1>();
                __builtin_amdgcn_global_load_lds((gbl_ptr)(base + off), (lds_ptr)(smem + (wid * D + j) * 1024), 16, 0, 0); off += step;
            } else if (MODE == 1) {
                wait_vm<D - 1>();
                acc += v[j].x;
                v[j] = *reinterpret_cast<const float4*>(base + off); off += step;
            } else {
                wait_vm<2 * D - 2>();
                acc += v[j].x;
                __builtin_amdgcn_global_load_lds((gbl_ptr)(base + off), (lds_ptr)(smem + (wid * D + j) * 1024), 16, 0, 0); off += step;
                v[j] = *reinterpret_cast<const float4*>(base + off); off += step;
            }
            if (off + step * 2 > span) off = 0;
        }
    }
    wait_vm<0>();
    if (MODE != 1) { __syncthreads(); acc += reinterpret_cast<float*>(smem)[tid]; }
    if (MODE != 0) {
#pragma unroll
        for (int j = 0; j < D; ++j) acc += v[j].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int D>
void run(const char* d, float* o, int blocks, int threads, long span, hipEvent_t e0, hipEvent_t e1) {
    const int nw = threads / 64, iters = 400;
    const size_t smem = (size_t)nw * D * 1024 + 2048;
    hipFuncSetAttribute((const void*)stream<MODE, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto launch = [&]() { hipLaunchKernelGGL((stream<MODE, D>), dim3(blocks), dim3(threads), smem, 0, d, o, iters, span); };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = (double)(iters * D + D) * 1024 * (MODE == 2 ? 2 : 1);
    const double tot = per_wave * nw * blocks;
    const double inflight = (double)nw * D * 1024 * (MODE == 2 ? 2 : 1);
    printf("%-8s blocks=%3d waves=%d D=%2d in flight per CU %5.0f KB : %7.1f GB/s per CU  -> latency %5.2f us\n", MODE == 0 ? "LDS-DMA" : MODE == 1 ? "reg-load" : "both",
           blocks, nw, D, inflight / 1024, tot / ms / 1e6 / blocks, inflight / (tot / ms / 1e6 / blocks * 1e9) * 1e6);
}

int main() {
    const long bytes = 8l << 30;                           // 8 GB: nothing is re-read from L2 / the Infinity Cache within a launch
    char* d; float* o;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&o, 4); hipMemset(d, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {32, 256}) {
        const long span = bytes / blocks;
        run<0, 4>(d, o, blocks, 512, span, e0, e1); run<0, 8>(d, o, blocks, 512, span, e0, e1); run<0, 12>(d, o, blocks, 512, span, e0, e1); run<0, 16>(d, o, blocks, 512, span, e0, e1);
        run<1, 4>(d, o, blocks, 512, span, e0, e1); run<1, 8>(d, o, blocks, 512, span, e0, e1); run<1, 12>(d, o, blocks, 512, span, e0, e1); run<1, 16>(d, o, blocks, 512, span, e0, e1);
        run<2, 4>(d, o, blocks, 512, span, e0, e1); run<2, 8>(d, o, blocks, 512, span, e0, e1);
        run<0, 8>(d, o, blocks, 256, span, e0, e1); run<1, 8>(d, o, blocks, 256, span, e0, e1); run<2, 8>(d, o, blocks, 256, span, e0, e1);
    }
    return 0;
}
