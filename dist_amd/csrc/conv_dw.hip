// Weight gradient of the TemporalNet's 3 x 3 frame convolution (reference models/module_zoo/branches/dist.py:54-60, c_fc2; autograd of
// runs/train.py:110), all nine taps from ONE LDS-resident frame:
//   dW[co][ci][tap] += sum_frames sum_n dp[f][n][co] * V[f][n + shift(tap)][ci],   n over the G x G plane, zero outside it
// (dist_op_gemm_tn with bmap = DIST_RM_SPATIAL, taps = 9, 96 x 96 channels, bf16).
//
// The generic kernel (gemm_tn_kernel<96, 96, 3, 2, TR, SPATIAL>) makes every tap its own tile job: both operands are staged nine times per
// row chunk (through L2), each (chunk, tap) block writes a 36 KB partial tile, and it runs at 4.6 % of the MFMA peak (profiles/r04: 143.8 us in
// situ for 16.6 GF, 1.77 x the operand bytes in traffic).  Here a block walks whole FRAMES:
//   * both operands of a frame go HBM / L2 -> LDS once, by LDS-DMA, into a PADDED-PLANE image: position (y, x) of the 14 x 14 plane sits in LDS
//     row 16 y + x (two zero columns per image row, a zero halo above and below), rows of 96 bf16 at a 224-byte pitch (conflict-free transpose
//     reads).  A tap is then a pure row shift 16 dy + dx of the V image - no masks, no validity flags: everything outside the plane is a zero the
//     buffer descriptor supplied (the descriptor covers exactly one frame, the per-lane source offset of a pad slot is out of range);
//   * a frame is two half-planes (image rows 0-7 = 4 k-blocks of 32 positions, rows 8-13 = 3 k-blocks) in two LDS sets: the next half lands
//     while the current one multiplies (one s_barrier per half: 126 KB of LDS, one block per CU);
//   * 8 waves = 2 (tap groups: taps 0-4 | 4-8) x 2 (output-channel halves) x 2 (input-channel halves), the SIMD partners one of each tap group:
//     a wave holds the 3 x 3 tiles of its FIVE taps in 180 accumulator registers over ALL frames of the block.  ONE instruction stream for every
//     wave - a tap is an address register, not an immediate: wave-dependent code paths around the MFMAs made hipcc rename the accumulators and
//     spill 700 registers, and a 12-wave split (three taps per wave, 168 registers) spilled inside the frame loop.  The centre tap is therefore
//     multiplied by both groups (90 instead of 81 MFMAs per SIMD and k-block) and stored by one.  Fragments by ds_read_b64_tr_b16 (inline
//     assembly: behind the builtin hipcc waits vmcnt(0) - the other half's DMA - in front of every transpose read);
//   * the bias gradient (column sums of dp) rides along (v_dot2 of the dp fragments against (1, 1)); each block leaves ONE 9 x 96 x 96 partial (fragment order is not
//     needed: 64-byte runs), and the second phase adds the blocks in index order (bit-repeatable) into the reference's [Co][Ci][taps] layout.
// Operand bytes are read once (38.5 MB per layer at b = 32); partial bytes = blocks x 332 KB (96 blocks: 32 MB).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int C9_G = 14, C9_N = C9_G * C9_G, C9_C = 96;
constexpr int C9_PITCH = 224;                             // bytes per LDS row: 96 bf16 + 32 bytes of padding
constexpr int C9_GRP = 32 * C9_PITCH;                     // one DMA group: 32 padded positions = 2 image rows = 7 pieces of 1 KB
constexpr int C9_FRAME_BYTES = C9_N * C9_C * 2;           // 37 632: one frame of one operand (rows of 96 bf16, dense)
// LDS sets: half 0 = k-blocks 0-3 (positions 0..127), half 1 = k-blocks 4-6 (positions 128..223); V carries a 32-row halo on both sides
constexpr int C9_A0 = 0, C9_V0 = 4 * C9_GRP, C9_A1 = 10 * C9_GRP, C9_V1 = 13 * C9_GRP, C9_LDS = 18 * C9_GRP;     // 129 024 bytes
constexpr int C9_ACC = 9 * C9_C * C9_C;                   // floats of one partial (then 96 column sums)
constexpr int C9_PART = C9_ACC + C9_C;

template <int OFF> DEV bf16x8 c9_tr8(const unsigned addr) {            // 8 k-slots of one column: rows 4g + e and 16 + 4g + (e - 4)
    static_assert(OFF >= 0 && OFF + 16 * C9_PITCH < 65536, "16-bit ds offset");
    s16x4 a, b;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a) : "v"(addr), "n"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(OFF + 16 * C9_PITCH) : "memory");
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = a; u.s[1] = b;
    return u.v;
}
DEV void c9_ready(bf16x8 (&f)[3]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) :: "memory"); }
DEV void c9_ready2(bf16x8 (&f)[3], bf16x8 (&g)[3]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(g[0]), "+v"(g[1]), "+v"(g[2]) :: "memory");
}
template <int I, int N, typename Fn> DEV void c9_for(Fn&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); c9_for<I + 1, N>(f); }
}

// SIGN: the tap direction of the row map (dy = (tap / 3 - 1) * SIGN, dx = (tap % 3 - 1) * SIGN)
template <int SIGN>
__global__ __launch_bounds__(512, 1) void conv3x3_dw_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* __restrict__ partial, const int frames) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wid >> 2, ch = (wid >> 1) & 1, ih = wid & 1;       // tap group (taps 4 tg .. 4 tg + 4), output- / input-channel half
    const int nblk = gridDim.x, blk = blockIdx.x;
    const int nf = blk < frames ? (frames - blk + nblk - 1) / nblk : 0;

    // ---- LDS-DMA: wave w < 7 moves piece w of every 32-row group.  Lane l of piece j lands at byte o = 1024 j + 16 l of the group: padded row
    // o / 224 = 16 yy + x', column byte o % 224; its source is row 14 yy + x' of the group's two image rows - or nothing (pad column, pad slot).
    unsigned pat;
    {
        const int o = 1024 * (wid < 7 ? wid : 0) + 16 * lane;
        const int row = o / C9_PITCH, cb = o - row * C9_PITCH;
        const int xp = row & 15, yy = row >> 4;
        pat = (xp < C9_G && cb < C9_C * 2) ? (unsigned)((yy * C9_G + xp) * (C9_C * 2) + cb) : 0x80000000u;
    }
    // one group (image rows 2 grp, 2 grp + 1; grp = -1 and 7 lie outside the plane: every offset is out of range and reads zero)
    auto dma_group = [&](const __amdgpu_buffer_rsrc_t r, const int lds_base, const int grp) __attribute__((always_inline)) {
        const unsigned v = pat + (unsigned)(grp * 2 * C9_G * C9_C * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(smem + lds_base + wid * 1024), 16, v, 0, 0, 0);
    };
    auto dma_half = [&](const int f, auto half_c) __attribute__((always_inline)) {
        constexpr int H = decltype(half_c)::value;
        if (wid >= 7) return;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A) + (size_t)f * C9_N * C9_C, 0, C9_FRAME_BYTES, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B) + (size_t)f * C9_N * C9_C, 0, C9_FRAME_BYTES, 0x00020000);
        if constexpr (H == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dma_group(ra, C9_A0 + q * C9_GRP, q);
#pragma unroll
            for (int q = 0; q < 6; ++q) dma_group(rb, C9_V0 + q * C9_GRP, q - 1);
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) dma_group(ra, C9_A1 + q * C9_GRP, 4 + q);
#pragma unroll
            for (int q = 0; q < 5; ++q) dma_group(rb, C9_V1 + q * C9_GRP, 3 + q);
        }
    };

    // ---- fragments: lane (li, lg) addresses row 4 lg + (li >> 2), 8 bytes at column 4 (li & 3) of a 16-column block
    const unsigned lrow = (unsigned)((4 * (lane >> 4) + ((lane & 15) >> 2)) * C9_PITCH + (lane & 3) * 8);
    const unsigned la = lrow + ch * 96, lb = lrow + ih * 96;          // this wave's three 16-column blocks start at column 48 ch / 48 ih
    // the five taps of this wave as address registers: tap 4 tg + k shifts the V image by 16 dy + dx padded positions
    unsigned lbk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int tap = 4 * tg + k;
        lbk[k] = lb + (unsigned)(((tap / 3 - 1) * SIGN * 16 + (tap % 3 - 1) * SIGN) * C9_PITCH);
    }
    f32x4 acc[5][3][3];                                   // [tap of the group][output block][input block]
    float cs[3] = {0.f, 0.f, 0.f};                         // column sums of dp (the bias gradient) over this lane's k-slots: every wave, no branch
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((ext_vector_type(2))) __bf16 c9_bf16x2;

    // one k-block (32 padded positions) x the wave's five taps.  ONE copy of this body serves both halves and every k-block (a runtime loop with
    // the set / k-block offset in a register): two instantiations - one per half - made hipcc assign the accumulators differently in each and move
    // them through scratch at every half (17 quads spilled inside the frame loop, each reload a vmcnt(0) on the other half's DMA).
    // Fragment pipeline: the V fragment of step n + 1 is requested in front of the three MFMAs of step n (counted lgkmcnt: the two newest reads may
    // stay in flight), and the NEXT k-block's dp fragments + first V fragment behind the last steps of this one (`more`), so that a wave never
    // waits a full LDS latency per fragment (the first version did: 15 x (2 reads, lgkmcnt(0), 3 MFMAs) ran the matrix pipe at ~45 %).
    bf16x8 fa[3], fan[3], fb[2];
    auto frag_a = [&](bf16x8 (&f)[3], const unsigned abase) __attribute__((always_inline)) {
        const unsigned aa = la + abase;
        c9_for<0, 3>([&](auto i_c) { constexpr int i = decltype(i_c)::value; f[i] = c9_tr8<i * 32>(aa); });
    };
    auto kblock = [&](const unsigned abase_next, const unsigned vbase, const bool more) __attribute__((always_inline)) {
        c9_for<0, 15>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value, t = n / 3, j = n % 3, cur = n & 1, nxt = cur ^ 1;
            // requests of this step: the next V fragment (this k-block's step n + 1, or the next k-block's step 0), at step 12 the next dp fragments
            if constexpr (n + 1 < 15) fb[nxt] = c9_tr8<8192 + ((n + 1) % 3) * 32>(lbk[(n + 1) / 3] + vbase);
            else if (more) fb[nxt] = c9_tr8<8192>(lbk[0] + vbase + (unsigned)C9_GRP);
            if constexpr (n == 12) { if (more) frag_a(fan, abase_next); }
            // fragment n (requested a step ago) has landed once at most this step's requests are outstanding
            if constexpr (n == 12) {
                if (more) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fb[cur]) :: "memory"); else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[cur]) :: "memory");
            } else if constexpr (n == 13) {
                if (more) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fb[cur]) :: "memory"); else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[cur]) :: "memory");
            } else if constexpr (n == 14) {                // (LDS reads return in order: fragment 14 sits behind the next dp fragments, only the next V fragment may remain)
                if (more) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[cur]) :: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[cur]) :: "memory");
            } else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[cur]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[cur], fa[i], acc[t][i][j], 0, 0, 0);   // swapped: D[ci][co]
            if constexpr (n == 0) {                        // packed pairs of the dp fragments against (1, 1): 12 v_dot2 per k-block under the MFMAs
                const c9_bf16x2 one = {(bf16_t)1.0f, (bf16_t)1.0f};
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const c9_bf16x2 pr = {fa[i][2 * e], fa[i][2 * e + 1]};
                        cs[i] = __builtin_amdgcn_fdot2_f32_bf16(pr, one, cs[i], false);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);              // (the next step's requests stay behind these MFMAs)
        });
        if (more) {                                        // the next k-block's first V fragment is the only request still in flight
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fan[0]), "+v"(fan[1]), "+v"(fan[2]), "+v"(fb[1]) :: "memory");
#pragma unroll
            for (int i = 0; i < 3; ++i) fa[i] = fan[i];
            fb[0] = fb[1];                                 // step 0 of a k-block reads fb[0]
        }
    };
    using H0 = std::integral_constant<int, 0>; using H1 = std::integral_constant<int, 1>;
    auto sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the half about to be multiplied have landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                      // everyone's have; and everyone is done with the set the next DMA overwrites
        __builtin_amdgcn_sched_barrier(0);
    };

    // half-steps: step 2k = half 0 of the block's k-th frame (4 k-blocks), step 2k + 1 = its half 1 (3 k-blocks)
    if (nf > 0) dma_half(blk, H0{});
#pragma unroll 1
    for (int st = 0; st < 2 * nf; ++st) {
        const int f = blk + (st >> 1) * nblk, h = st & 1;
        sync();
        if (h == 0) dma_half(f, H1{});
        else if (st + 1 < 2 * nf) dma_half(f + nblk, H0{});
        const int nkb = h ? 3 : 4;
        unsigned abase = h ? C9_A1 : C9_A0, vbase = (h ? C9_V1 : C9_V0) + C9_GRP - 8192;
        frag_a(fa, abase);                                 // the half's first fragments (behind the barrier: the only exposed LDS latency of a half)
        fb[0] = c9_tr8<8192>(lbk[0] + vbase);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fb[0]) :: "memory");
#pragma unroll 1
        for (int kb = 0; kb < nkb; ++kb) {
            abase += C9_GRP;
            kblock(abase, vbase, kb + 1 < nkb);
            vbase += C9_GRP;
        }
    }

    // ---- this block's partial: [tap][co][ci] fp32, then the 96 column sums.  The MFMAs ran swapped (D[ci][co]), so acc[k][i][j][r] is tap 4 tg + k,
    // co = 48 ch + 16 i + li, ci = 48 ih + 16 j + 4 lg + r: a lane's quad is 16 contiguous bytes of the partial (45 stores per lane instead of 180)
    // (lane coordinates from an OPAQUE copy of the lane id: hoisted above the frame loop, the store offsets were spilled inside it)
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int li = lo & 15, lg = lo >> 4;
    float* __restrict__ P = partial + (size_t)blk * C9_PART;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (k == 0 && tg == 1) continue;                   // the centre tap was multiplied by both groups: group 0 stores it
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                *reinterpret_cast<f32x4*>(P + ((size_t)(4 * tg + k) * C9_C + (48 * ch + 16 * i + li)) * C9_C + 48 * ih + 16 * j + 4 * lg) = acc[k][i][j];
    }
    // lane (li, lg) summed column 48 ch + 16 i + li over ITS k-slots: add the four lane groups, one wave per channel half writes
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float v = cs[i];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (tg == 0 && ih == 0 && lg == 0) P[C9_ACC + 48 * ch + 16 * i + li] = v;
    }
}

// second phase: out[(co, ci, tap)] += sum over blocks (index order), colsum[co] += sum over blocks
__global__ __launch_bounds__(256) void conv3x3_dw_reduce_kernel(const dist_gemm_tn_args p, const int nblk) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= C9_PART) return;
    // blocks in index order (bit-repeatable), eight loads in flight: a dependent load per block made this kernel 0.17 us x blocks
    float s = 0.f;
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p.partial[(size_t)(b + u) * C9_PART + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblk; ++b) s += p.partial[(size_t)b * C9_PART + e];
    if (e >= C9_ACC) { if (p.colsum) p.colsum[e - C9_ACC] += s; return; }
    const int tap = e / (C9_C * C9_C), r = e - tap * (C9_C * C9_C), co = r / C9_C, ci = r - co * C9_C;
    p.out[(long)co * p.so_i + (long)tap * p.so_tap + (long)(ci / p.inner) * p.so_outer + (ci % p.inner)] += s;
}

}  // namespace

// 1 = launched, 0 = not this kernel's call (dist_op_gemm_tn falls through to gemm_tn_kernel), < 0 = error
int dist_k_conv3x3_dw(const dist_gemm_tn_args* a, hipStream_t s) {
    static const int on = dist_knob("DIST_AMD_CONV9", 1);           // 0: the generic tap-per-tile kernel (other plane sizes, channel counts and the fp32 mode take it anyway)
    if (!on) return 0;
    if (a->dtype != DIST_BF16 || !a->use_tr || a->taps != 9 || a->amap.mode != DIST_RM_PLAIN || a->bmap.mode != DIST_RM_SPATIAL) return 0;
    if (a->bmap.p0 != C9_G || (a->bmap.sign != 1 && a->bmap.sign != -1)) return 0;
    if (a->NI != C9_C || a->K != C9_C || a->lda != C9_C || a->ldb != C9_C || a->inner <= 0 || a->out2 || a->colsum2) return 0;
    if (a->M % C9_N || a->M < 8 * C9_N || !a->partial) return 0;
    if (((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15)) return 0;
    const long frames = a->M / C9_N;
    static const int max_blocks = dist_knob("DIST_AMD_TN_BLOCKS", 96);
    long nblk = a->max_blocks > 0 ? a->max_blocks : max_blocks;
    if (nblk > 256) nblk = 256;
    if (nblk > frames) nblk = frames;
    if (nblk * (long)C9_PART > a->partial_elems) nblk = a->partial_elems / C9_PART;
    if (nblk < 1) return 0;
    static DistSmemOnce attr_p, attr_m;
    if (a->bmap.sign == 1) {
        RUN_(dist_max_smem(attr_p, reinterpret_cast<const void*>(conv3x3_dw_kernel<1>), (size_t)C9_LDS));
        hipLaunchKernelGGL(conv3x3_dw_kernel<1>, dim3((unsigned)nblk), dim3(512), (size_t)C9_LDS, s, static_cast<const bf16_t*>(a->A), static_cast<const bf16_t*>(a->B), a->partial, (int)frames);
    } else {
        RUN_(dist_max_smem(attr_m, reinterpret_cast<const void*>(conv3x3_dw_kernel<-1>), (size_t)C9_LDS));
        hipLaunchKernelGGL(conv3x3_dw_kernel<-1>, dim3((unsigned)nblk), dim3(512), (size_t)C9_LDS, s, static_cast<const bf16_t*>(a->A), static_cast<const bf16_t*>(a->B), a->partial, (int)frames);
    }
    HIP_CHECK_RET(hipGetLastError());
    hipLaunchKernelGGL(conv3x3_dw_reduce_kernel, dim3((C9_PART + 255) / 256), dim3(256), 0, s, *a, (int)nblk);
    HIP_CHECK_RET(hipGetLastError());
    return 1;
}
