// Micro-benchmark: per-CU global STORE rate (16 B per lane, plain / non-temporal) as a function of the number of CUs storing at the same
// time.  Every block writes 128 KB "tiles" (what one 256x256 bf16 C tile is) from registers, `iters` times, each to a fresh region.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <int NT_STORE>
__global__ __launch_bounds__(512) void stores(char* __restrict__ dst, int iters, long tile_stride, long blk_stride, int rowlen) {
    const int tid = threadIdx.x;
    char* base = dst + (long)blockIdx.x * blk_stride;
    u32x4_t v = {(unsigned)tid, 1u, 2u, 3u};
    for (int it = 0; it < iters; ++it) {
        char* t = base + (long)it * tile_stride;
#pragma unroll
        for (int j = 0; j < 16; ++j) {                     // 512 threads x 16 x 16 B = 128 KB
            const int vec = j * 512 + tid;                 // 16-byte vector index inside the tile
            // rowlen = bytes of one contiguous row piece (512 B = a 256-column bf16 tile row inside a wider matrix; 0 = fully contiguous)
            char* p = rowlen ? t + (long)(vec / (rowlen / 16)) * (rowlen * 9) + (vec % (rowlen / 16)) * 16 : t + (long)vec * 16;
            if (NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p));
            else *reinterpret_cast<u32x4_t*>(p) = v;
        }
    }
}

int main() {
    const long bytes = 6l << 30;
    char* d;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 16;
    for (int rowlen : {0, 512})
        for (int nt = 0; nt < 2; ++nt)
            for (int blocks : {1, 8, 32, 64, 128, 256, 512}) {
                const long tile = rowlen ? (128l << 10) * 9 : (128l << 10);
                const long blk_stride = tile * iters;
                if (blocks * blk_stride > bytes) continue;
                auto launch = [&]() {
                    if (nt) hipLaunchKernelGGL(stores<1>, dim3(blocks), dim3(512), 0, 0, d, iters, tile, blk_stride, rowlen);
                    else hipLaunchKernelGGL(stores<0>, dim3(blocks), dim3(512), 0, 0, d, iters, tile, blk_stride, rowlen);
                };
                launch(); hipDeviceSynchronize();
                float best = 1e9f;
                for (int r = 0; r < 3; ++r) {
                    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                const double tot = (double)blocks * iters * (128 << 10);
                const int cus = blocks < 256 ? blocks : 256;
                printf("%s %s blocks=%3d: %8.1f us  %7.2f TB/s  %6.1f GB/s per CU = %5.1f B/clk/CU at 2.4 GHz, %5.2f us per 128 KB tile per block\n",
                       rowlen ? "rows of 512 B (pitch 4608)" : "contiguous              ", nt ? "nt   " : "plain", blocks, best * 1e3, tot / best / 1e9,
                       tot / best / 1e6 / cus, tot / best / 1e6 / cus / 2.4, best * 1e3 / iters);
            }
    return 0;
}
