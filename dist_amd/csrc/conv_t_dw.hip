// Weight gradients of the branch's TEMPORAL convolutions (reference models/module_zoo/branches/dist.py: the (3, 1, 1) Conv3d's of TemporalNet.c_fc1
// and temporal_ffn.c_fc2, lines 23-36 / 54-60, and the (5, 1, 1)-strided temporal stem, lines 178-181; autograd of runs/train.py:110):
//   dW[co][ci][tap] += sum over clips, frames f, tokens n of dY[f][n][co] * X[f + d(tap)][n][ci],   d = sign * (tap - taps / 2), zero outside the clip
// (dist_op_gemm_tn with bmap = DIST_RM_SHIFT over whole frames, taps = 3 or 5, 96 output channels, bf16).
//
// The generic kernel makes every tap its own tile job: X is staged once per tap, dY once per tap and column tile, every (chunk, tap) block leaves its
// own partial tile - 32 us alone for the 96 x 96 x 3 gradients (5.5 GF, 38.5 MB of operands), 373 us for the stem's 96 x 768 x 5 on 90 blocks, the
// LAST kernel of the backward with nothing beside it.  Here, like conv_dw.hip, every operand row is read ONCE per 96-column tile of X:
//   * work item = (clip, quarter of a frame's tokens): a block walks the item's T frames with a RING of frame slots in LDS - taps + 2 slots of X
//     (frames f - h ... f + h multiply while f + h + 1, f + h + 2 land) and three of dY - so a tap is an LDS slot, not a second read of X;
//   * a slot is 64 rows (two k-blocks of 32; a quarter of N <= 256 tokens, the rest zero) x 96 bf16 at a 224-byte pitch, filled by LDS-DMA straight
//     from the [rows][ld] operands (the buffer descriptor ends behind the quarter's last token: missing rows, and whole frames outside the clip - an
//     empty descriptor - arrive as zeros; no masks, no validity flags);
//   * 8 waves = 2 (tap groups: taps 0 ... h | h ... 2h, the centre tap in both) x 2 (output-channel halves) x 2 (input-channel halves); ONE
//     instruction stream for every wave (a tap is an address register): the lesson of conv_dw.hip.  Fragments by ds_read_b64_tr_b16 (inline
//     assembly), the next fragment requested in front of the current one's three MFMAs;
//   * the bias gradient (column sums of dY) rides along; each block leaves ONE taps x 96 x 96 partial, the second phase adds the blocks in index
//     order (bit-repeatable) into the caller's layout (so_i / so_tap / so_outer / inner: Conv3d [Co][Ci][taps] or the stem's [Co][3][taps][P][P]).
// K > 96 (the stem's 768 patch columns): one launch, 96-column tiles of X as a grid dimension (dY is re-read per tile: 8 x 19 MB against X's 154 MB).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int TD_C = 96;
constexpr int TD_PITCH = 224;                             // bytes per LDS row: 96 bf16 + 32 bytes of padding (conflict-free transpose reads)
constexpr int TD_ROWS = 64;                               // rows of a slot: two k-blocks
constexpr int TD_SLOT = TD_ROWS * TD_PITCH;               // 14 336 bytes = 14 pieces of 1 KB
constexpr int TD_KB = 32 * TD_PITCH;                      // one k-block

template <int OFF> DEV bf16x8 td_tr8(const unsigned addr) {            // 8 k-slots of one column: rows 4g + e and 16 + 4g + (e - 4)
    static_assert(OFF >= 0 && OFF + 16 * TD_PITCH < 65536, "16-bit ds offset");
    s16x4 a, b;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a) : "v"(addr), "n"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(OFF + 16 * TD_PITCH) : "memory");
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = a; u.s[1] = b;
    return u.v;
}
template <int I, int N, typename Fn> DEV void td_for(Fn&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); td_for<I + 1, N>(f); }
}

struct TdArgs {
    const bf16_t* A; const bf16_t* B; float* partial;
    int lda, ldb;                  // elements
    int clips, T, N, PT;           // frames per clip, tokens per frame, tokens per quarter
    int sign;
    int bpt;                       // blocks per 96-column tile of X (gridDim.x = bpt x column tiles)
};

template <int TAPS>
__global__ __launch_bounds__(512, 1) void conv_t_dw_kernel(const TdArgs p) {
    constexpr int H = TAPS / 2, TG = H + 1, NS = TG * 3;
    constexpr int D = 2, RA = D + 1, RB = TAPS + D;       // frames requested D steps ahead (one step is shorter than an LDS-DMA round trip: with D = 1 a step lasted
                                                          // as long as the memory latency, 1.25 us for 36 MFMAs per wave)
    constexpr int B0 = RA * TD_SLOT;                      // LDS: RA dY slots, then RB slots of X
    constexpr int ACC = TAPS * TD_C * TD_C, PART = ACC + TD_C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wid >> 2, ch = (wid >> 1) & 1, ih = wid & 1;       // tap group, output- / input-channel half
    const int ct = blockIdx.x / p.bpt, blk = blockIdx.x - ct * p.bpt;
    const int items = p.clips * 4;
    const int T = p.T, N = p.N;

    // ---- LDS-DMA: wave w moves pieces w and w + 8 (< 14) of a slot.  Lane l of piece j lands at byte o = 1024 j + 16 l: slot row o / 224, column
    // byte o % 224; its source is that row of the quarter (row pitch lda / ldb) - or nothing (pad bytes; rows behind the quarter are cut off by the descriptor)
    unsigned pa[2], pb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int o = 1024 * (wid + 8 * q) + 16 * lane;
        const int row = o / TD_PITCH, cb = o - row * TD_PITCH;
        const bool ok = wid + 8 * q < 14 && cb < TD_C * 2;
        pa[q] = ok ? (unsigned)(row * p.lda * 2 + cb) : 0x80000000u;
        pb[q] = ok ? (unsigned)(row * p.ldb * 2 + cb) : 0x80000000u;
    }
    // frame f of (clip, quarter): descriptor over the quarter's tokens (empty for a frame outside the clip)
    auto dma_slot = [&](const bf16_t* base, const int ld, const unsigned (&pat)[2], const int slot_off, const int clip, const int quarter, const int f) __attribute__((always_inline)) {
        const int tok0 = quarter * p.PT;
        const int ntok = (f >= 0 && f < T) ? min(p.PT, N - tok0) : 0;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t*>(base) + ((size_t)(clip * T + (f >= 0 && f < T ? f : 0)) * N + tok0) * ld, 0, ntok * ld * 2, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(smem + slot_off + wid * 1024), 16, pat[0], 0, 0, 0);
        if (wid < 6) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(smem + slot_off + (wid + 8) * 1024), 16, pat[1], 0, 0, 0);
    };
    const bf16_t* const Bt = p.B + ct * TD_C;             // this block's 96 columns of X

    // ---- fragments: lane (li, lg) addresses row 4 lg + (li >> 2), 8 bytes at column 4 (li & 3) of a 16-column block
    const unsigned lrow = (unsigned)((4 * (lane >> 4) + ((lane & 15) >> 2)) * TD_PITCH + (lane & 3) * 8);
    const unsigned la = lrow + ch * 96, lb = lrow + ih * 96 + B0;     // this wave's three 16-column blocks start at column 48 ch / 48 ih
    f32x4 acc[TG][3][3];                                  // [tap of the group][output block][input block]
    float cs[3] = {0.f, 0.f, 0.f};                         // column sums of dY over this lane's k-slots
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((ext_vector_type(2))) __bf16 td_bf16x2;

    // one k-block x the wave's TG taps (the body of conv_dw.hip's kblock): the X fragment of step n + 1 is requested in front of the three MFMAs of step n,
    // the next k-block's dY fragments + first X fragment behind the last steps of this one (`more`)
    bf16x8 fa[3], fan[3], fb[2];
    unsigned lbk[TG];                                     // the X slot of each tap of this wave, per frame step
    auto frag_a = [&](bf16x8 (&f)[3], const unsigned abase) __attribute__((always_inline)) {
        const unsigned aa = la + abase;
        td_for<0, 3>([&](auto i_c) { constexpr int i = decltype(i_c)::value; f[i] = td_tr8<i * 32>(aa); });
    };
    auto kblock = [&](const unsigned abase_next, const unsigned koff, const bool more) __attribute__((always_inline)) {
        td_for<0, NS>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value, t = n / 3, j = n % 3, cur = n & 1, nxt = cur ^ 1;
            if constexpr (n + 1 < NS) fb[nxt] = td_tr8<((n + 1) % 3) * 32>(lbk[(n + 1) / 3] + koff);
            else if (more) fb[nxt] = td_tr8<0>(lbk[0] + koff + (unsigned)TD_KB);
            if constexpr (n == NS - 3) { if (more) frag_a(fan, abase_next); }
            if constexpr (n == NS - 3 || n == NS - 2) {
                if (more) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fb[cur]) :: "memory"); else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[cur]) :: "memory");
            } else if constexpr (n == NS - 1) {            // (LDS reads return in order: the last fragment sits behind the next dY fragments, only the next X fragment may remain)
                if (more) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[cur]) :: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[cur]) :: "memory");
            } else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[cur]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[cur], fa[i], acc[t][i][j], 0, 0, 0);   // swapped: D[ci][co]
            if constexpr (n == 0) {                        // packed pairs of the dY fragments against (1, 1): 12 v_dot2 per k-block under the MFMAs
                const td_bf16x2 one = {(bf16_t)1.0f, (bf16_t)1.0f};
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const td_bf16x2 pr = {fa[i][2 * e], fa[i][2 * e + 1]};
                        cs[i] = __builtin_amdgcn_fdot2_f32_bf16(pr, one, cs[i], false);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fan[0]), "+v"(fan[1]), "+v"(fan[2]), "+v"(fb[NS & 1]) :: "memory");
#pragma unroll
            for (int i = 0; i < 3; ++i) fa[i] = fan[i];
            if constexpr (NS & 1) fb[0] = fb[1];           // step 0 of a k-block reads fb[0]
        }
    };
    // a step's requests of this wave: one or two pieces each of a dY slot and an X slot; the pieces of the frame about to be multiplied have landed
    // once only the D - 1 newer steps' requests are outstanding (`tail`: no newer requests exist)
    auto sync = [&](const bool tail) __attribute__((always_inline)) {
        static_assert(D == 2, "counted waits below are written for D = 2");
        if (tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (wid < 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                      // everyone's have; and everyone is done with the slots the next DMA overwrites
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- items of this block: (clip, quarter) = blk, blk + bpt, ...; one frame per step, frame f of dY against frames f - H ... f + H of X
#pragma unroll 1
    for (int it = blk; it < items; it += p.bpt) {
        const int clip = it >> 2, quarter = it & 3;
        __builtin_amdgcn_s_barrier();                      // (the previous item's last step is done with every slot)
        // request order = frame order (the counted waits rely on it): X frames -H ... H - 1, then step groups {dY g, X g + H} for g = 0 ... D - 1
#pragma unroll
        for (int d = -H; d < H; ++d) dma_slot(Bt, p.ldb, pb, B0 + ((d + RB) % RB) * TD_SLOT, clip, quarter, d);
#pragma unroll
        for (int g = 0; g < D; ++g) {
            dma_slot(p.A, p.lda, pa, g * TD_SLOT, clip, quarter, g);
            dma_slot(Bt, p.ldb, pb, B0 + ((g + H) % RB) * TD_SLOT, clip, quarter, g + H);
        }
#pragma unroll 1
        for (int f = 0; f < T; ++f) {
            sync(f + D - 1 >= T);
            if (f + D < T) {
                dma_slot(p.A, p.lda, pa, ((f + D) % RA) * TD_SLOT, clip, quarter, f + D);
                dma_slot(Bt, p.ldb, pb, B0 + ((f + H + D) % RB) * TD_SLOT, clip, quarter, f + H + D);
            }
            // tap k of this wave's group: tap index tg * H + k, frame f + sign * (tap - H)
#pragma unroll
            for (int k = 0; k < TG; ++k) {
                const int fr = f + p.sign * (tg * H + k - H);
                lbk[k] = lb + (unsigned)(((fr + RB) % RB) * TD_SLOT);
            }
            const unsigned abase = (unsigned)((f % RA) * TD_SLOT);
            frag_a(fa, abase);                             // the step's first fragments (behind the barrier: the only exposed LDS latency of a step)
            fb[0] = td_tr8<0>(lbk[0]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fb[0]) :: "memory");
            kblock(abase + TD_KB, 0u, true);
            kblock(0u, (unsigned)TD_KB, false);
        }
    }

    // ---- this block's partial: [tap][co][ci] fp32, then the 96 column sums (acc[k][i][j][r]: tap tg H + k, co = 48 ch + 16 i + li, ci = 48 ih + 16 j + 4 lg + r)
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int li = lo & 15, lg = lo >> 4;
    float* __restrict__ P = p.partial + (size_t)blockIdx.x * PART;
#pragma unroll
    for (int k = 0; k < TG; ++k) {
        if (k == 0 && tg == 1) continue;                   // the centre tap was multiplied by both groups: group 0 stores it
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                *reinterpret_cast<f32x4*>(P + ((size_t)(tg * H + k) * TD_C + (48 * ch + 16 * i + li)) * TD_C + 48 * ih + 16 * j + 4 * lg) = acc[k][i][j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float v = cs[i];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (tg == 0 && ih == 0 && lg == 0) P[ACC + 48 * ch + 16 * i + li] = v;
    }
}

// second phase: out[(co, ci of tile ct, tap)] += sum over the tile's blocks (index order), colsum[co] += sum over the blocks of tile 0
template <int TAPS>
__global__ __launch_bounds__(256) void conv_t_dw_reduce_kernel(const dist_gemm_tn_args p, const int bpt) {
    constexpr int ACC = TAPS * TD_C * TD_C, PART = ACC + TD_C;
    const int e = blockIdx.x * 256 + threadIdx.x, ct = blockIdx.y;
    if (e >= PART || (e >= ACC && ct > 0)) return;
    const float* q = p.partial + (size_t)ct * bpt * PART + e;
    float s = 0.f;
    int b = 0;
    for (; b + 8 <= bpt; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = q[(size_t)(b + u) * PART];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < bpt; ++b) s += q[(size_t)b * PART];
    if (e >= ACC) { if (p.colsum) p.colsum[e - ACC] += s; return; }
    const int tap = e / (TD_C * TD_C), r = e - tap * (TD_C * TD_C), co = r / TD_C, ci = ct * TD_C + (r - co * TD_C);
    p.out[(long)co * p.so_i + (long)tap * p.so_tap + (long)(ci / p.inner) * p.so_outer + (ci % p.inner)] += s;
}

template <int TAPS>
int td_launch(const dist_gemm_tn_args* a, const TdArgs& t, const int tiles, hipStream_t s) {
    constexpr int LDS = (3 + TAPS + 2) * TD_SLOT, PART = TAPS * TD_C * TD_C + TD_C;      // RA + RB slots (D = 2)
    static DistSmemOnce attr;
    RUN_(dist_max_smem(attr, reinterpret_cast<const void*>(conv_t_dw_kernel<TAPS>), (size_t)LDS));
    hipLaunchKernelGGL(conv_t_dw_kernel<TAPS>, dim3((unsigned)(t.bpt * tiles)), dim3(512), (size_t)LDS, s, t);
    HIP_CHECK_RET(hipGetLastError());
    hipLaunchKernelGGL(conv_t_dw_reduce_kernel<TAPS>, dim3((PART + 255) / 256, (unsigned)tiles), dim3(256), 0, s, *a, t.bpt);
    HIP_CHECK_RET(hipGetLastError());
    return 1;
}

}  // namespace

// 1 = launched, 0 = not this kernel's call (dist_op_gemm_tn falls through to gemm_tn_kernel), < 0 = error
int dist_k_conv_t_dw(const dist_gemm_tn_args* a, hipStream_t s) {
    static const int on = dist_knob("DIST_AMD_CONV9", 1);           // (one selector for the two multi-tap kernels) 0: the generic tap-per-tile kernel
    if (!on) return 0;
    if (a->dtype != DIST_BF16 || !a->use_tr || (a->taps != 3 && a->taps != 5) || a->amap.mode != DIST_RM_PLAIN || a->bmap.mode != DIST_RM_SHIFT) return 0;
    if (a->NI != TD_C || a->K % TD_C || a->K < TD_C || a->lda < TD_C || a->ldb < a->K || a->lda % 8 || a->ldb % 8 || a->inner <= 0 || a->out2 || a->colsum2) return 0;     // (a pair gradient - two outputs, two bias sums - is the generic kernel's)
    const int N = a->bmap.p1;                             // tokens per frame (the shift of one tap), p0 = rows of a clip
    if (N < 16 || N > 4 * TD_ROWS || a->bmap.p0 <= 0 || a->bmap.p0 % N || a->M % a->bmap.p0 || (a->bmap.sign != 1 && a->bmap.sign != -1)) return 0;
    const int T = a->bmap.p0 / N, PT = (N + 3) / 4;
    if (T < a->taps || N - 3 * PT < 1 || !a->partial) return 0;
    if (((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15)) return 0;
    if ((long)PT * (a->lda > a->ldb ? a->lda : a->ldb) * 2 >= (1L << 31)) return 0;
    const long clips = a->M / a->bmap.p0;
    const int tiles = a->K / TD_C;
    // measured (tools/bench_conv_t_dw.py, b = 32): the stem's 96 x 768 x 5 gradient 377 -> 137 us; the 96 x 96 x 3 gradients 46.3 -> 50.2 / 29.5 -> 33.6 us
    // (48 x 48 quadrants per wave read every dY fragment four times from LDS: LDS-bound like the generic kernel) - those stay on the generic kernel
    static const int t3 = DIST_AB_KNOB("DIST_AMD_CONVT3", 0), t5 = DIST_AB_KNOB("DIST_AMD_CONVT5", 1);
    if (a->taps == 3 && tiles == 1 ? !t3 : !t5) return 0;
    const long items = clips * 4;
    if (clips < 2 || clips > (1 << 20)) return 0;
    static const int max_blocks = dist_knob("DIST_AMD_TN_BLOCKS", 96);
    // one column tile: the caller's cap (a launch beside a critical chain); several (the stem, the last kernel of the backward): a CU per block
    // (ABI 9: max_blocks bounds the main kernel's workgroups - also here; the engine passes 0 = a CU per block for the stem)
    long bpt = tiles > 1 ? 256 / tiles : (a->max_blocks > 0 ? a->max_blocks : max_blocks);
    if (tiles > 1 && a->max_blocks > 0 && bpt > a->max_blocks / tiles) bpt = a->max_blocks / tiles > 0 ? a->max_blocks / tiles : 1;
    if (bpt > 256) bpt = 256;
    if (bpt > items) bpt = items;
    const long part = (long)a->taps * TD_C * TD_C + TD_C;
    if (bpt * tiles * part > a->partial_elems) bpt = a->partial_elems / (part * tiles);
    if (bpt < 1) return 0;
    TdArgs t;
    t.A = static_cast<const bf16_t*>(a->A); t.B = static_cast<const bf16_t*>(a->B); t.partial = a->partial;
    t.lda = a->lda; t.ldb = a->ldb; t.clips = (int)clips; t.T = T; t.N = N; t.PT = PT; t.sign = a->bmap.sign; t.bpt = (int)bpt;
    return a->taps == 3 ? td_launch<3>(a, t, tiles, s) : td_launch<5>(a, t, tiles, s);
}
