// Timing-only library only (-DDIST_AMD_MEASURE): included by gemm_fast.hip inside its anonymous namespace.
// ---------------------------------------------------------------------------------------------------------------------
// gemm_fast8q_kernel: the loop of gemm_fast8p_kernel (same phases, same LDS image, same waits) as a block that walks SEVERAL 256 x 256 tiles
// with the operand stream running THROUGH the tile seams (round 5).  What a K = 768 tile loses outside its loop is ~1-1.4 us of block
// dispatch (tools/ubench/mfma_issue.hip: 512-thread blocks with 128 KB of LDS turn over in ~1 us per round), the first K-tiles' LDS-DMA latency
// (1.2-1.4 us under load) and the epilogue.  Here
//   * K-tiles nk, nk + 1 of a tile's stream are the NEXT tile's K-tiles 0, 1 (the half-tile descriptors are per tile, selected by a scalar), so
//     the next tile's first fragments are read in this tile's last phases like any other K-tile's and its loop starts without a prologue;
//   * the epilogue of a wave runs between its two loops out of a PRIVATE 4 KB staging region above the 128 KB operand ring (four passes of
//     32 rows: convert, 16-byte row stores), with no block barrier - the ring keeps filling underneath it - and re-zeroes the accumulators.
// The epilogue's stores sit in the wave's in-order VMEM queue in front of the next tile's later pieces: the counted waits of the first phases
// also wait for their acknowledgements.  Epilogues without loads behind the first pass only (bias / LayerNorm fold / QuickGELU / head-major
// output: the ViT's in_proj and c_fc, 61 % of its GEMM time); every element sees the K-tiles in gemm_fast8p_kernel's order: same bits.
// MEASURED (profiles/r05_gemm_pingpong.md, "multi-tile blocks"): bit-identical on every case, and within +-4 % of one tile per block on the ViT's
// two shapes (in_proj with the head-major output 0.96-0.98x, c_fc 0.99-1.03x; plain epilogues +2...+9 %): what the seam saves in dispatch and first-
// tile latency it pays back in an epilogue that can no longer leave its stores behind at the end of a block.  Not shipped: -DDIST_AMD_MEASURE only
// (DIST_AMD_FAST_TILES = tiles per block).
enum { QF_LNFOLD = 1, QF_ACT = 2, QF_HEADS = 4 };
constexpr int Q8_STG = 32 * 128;                          // per-wave staging: 32 rows x 128 B
constexpr int Q8_LDS = P8_LDS + 8 * Q8_STG;               // 160 KB

template <int F>
__global__ __launch_bounds__(512, 1) void gemm_fast8q_kernel(const dist_gemm_args p, const int ngroups) {
    constexpr bool LNF = (F & QF_LNFOLD) != 0, ACT = (F & QF_ACT) != 0, HEADS = (F & QF_HEADS) != 0;
    constexpr int BN = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int li = lane & 15, lg = lane >> 4;
    const int M = (int)p.M, N = p.N;
    const int nk = p.K / P8_BK;                           // even (launcher): the buffer parity of a K-tile does not depend on the tile

    // ---- this block's tiles: XCD x owns a contiguous range of the tile order (fast_tile), its blocks interleave inside it
    const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM, nblk = tiles_m * tiles_n;
    const int G8 = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, yb = (int)blockIdx.x >> 3;
    const int q8 = nblk / 8, r8 = nblk % 8;
    const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = xcd < r8 ? q8 + 1 : q8;
    const int ntiles = yb < cnt ? (cnt - yb + G8 - 1) / G8 : 0;
    if (ntiles == 0) return;
    auto tile_pos = [&](const int j, int& m0, int& n0) __attribute__((always_inline)) {
        const int t = start + yb + j * G8;
        int tm, tn;
        if (ngroups <= 1) { tm = t / tiles_n; tn = t % tiles_n; }
        else {
            const int gq = tiles_n / ngroups, gr = tiles_n % ngroups;
            const int big = tiles_m * (gq + 1);
            int g, idg, gsz, g0;
            if (t < gr * big) { g = t / big; idg = t - g * big; gsz = gq + 1; g0 = g * (gq + 1); }
            else { const int b2 = t - gr * big; g = b2 / (tiles_m * gq); idg = b2 - g * (tiles_m * gq); gsz = gq; g0 = gr * (gq + 1) + g * gq; }
            tm = idg / gsz; tn = g0 + idg % gsz;
        }
        m0 = tm * BM; n0 = tn * BN;
    };

    // ---- LDS-DMA sources: per-lane offsets inside a tile (as gemm_fast8p_kernel), tile origin in the descriptors
    const unsigned lda2 = (unsigned)p.lda * 2u, ldb2 = (unsigned)p.ldb * 2u;
    unsigned ga[2], gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = 16 * wid + 8 * j + (lane >> 3);
        const int lc = (lane & 7) ^ ((lr >> 1) & 7);
        ga[j] = (unsigned)((lr >> 6) * 128 + (lr & 63)) * lda2 + lc * 16;
        gb[j] = (unsigned)((lr >> 5) * 64 + (lr & 31)) * ldb2 + lc * 16;
    }
    const int aq1 = 64 * (int)lda2, bq1 = 32 * (int)ldb2;
    const char* Ab = static_cast<const char*>(p.A);
    const char* Bb = static_cast<const char*>(p.B);
    auto rsrc_a = [&](const int m0) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ab + (size_t)m0 * lda2), 0, max(M - m0, 0) * (int)lda2, 0x00020000);
    };
    auto rsrc_b = [&](const int n0) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Bb + (size_t)n0 * ldb2), 0, max(N - n0, 0) * (int)ldb2, 0x00020000);
    };
    int m0, n0, m0n = 0, n0n = 0;
    tile_pos(0, m0, n0);
    __amdgpu_buffer_rsrc_t ra = rsrc_a(m0), rb = rsrc_b(n0), ran = ra, rbn = rb;
    // stage half-tile `slot` of stream K-tile kts (>= nk: the next tile's K-tile kts - nk) into buffer kts & 1
    auto stage = [&](const int slot, const int kts) __attribute__((always_inline)) {
        char* sb = smem + (kts & 1) * P8_BUF + slot + wid * 2048;
        const bool cur = kts < nk;
        const int k2 = (cur ? kts : kts - nk) * (P8_BK * 2);
        if (slot == P8_A0 || slot == P8_A1) {
            const __amdgpu_buffer_rsrc_t r = cur ? ra : ran;
            const int so = k2 + (slot == P8_A1 ? aq1 : 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)sb, 16, ga[0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(sb + 1024), 16, ga[1], so, 0, 0);
        } else {
            const __amdgpu_buffer_rsrc_t r = cur ? rb : rbn;
            const int so = k2 + (slot == P8_B1 ? bq1 : 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)sb, 16, gb[0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(sb + 1024), 16, gb[1], so, 0, 0);
        }
    };

    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int pc = ((kk * 4 + lg) ^ ((li >> 1) & 7)) << 4;
        a_rd[kk] = (wr * 64 + li) * 128 + pc;
        b_rd[kk] = P8_B0 + (wc * 32 + li) * 128 + pc;
    }
    bf16x8 fa[2][4][2], sb_[2][2][2];
    auto read_a = [&](bf16x8 (&f)[4][2], const int buf, const int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) f[i][kk] = *reinterpret_cast<const bf16x8*>(smem + (slot + buf * P8_BUF + i * 2048) + a_rd[kk]);
    };
    auto read_b = [&](bf16x8 (&f)[2][2], const int buf, const int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) f[j][kk] = *reinterpret_cast<const bf16x8*>(smem + (slot - P8_B0 + buf * P8_BUF + j * 2048) + b_rd[kk]);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma16q = [&](const int qa, const int qb, const bf16x8 (&a)[4][2], const bf16x8 (&b)[2][2]) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[qa * 4 + i][qb * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][kk], a[i][kk], acc[qa * 4 + i][qb * 2 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto close_phase = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // one K-tile (gemm_fast8p_kernel's ktile); `nxt`: a next tile follows, so the stream does not end with this tile's last K-tile.
    // `seam`: the first K-tile behind an epilogue.  Every piece issued before that epilogue is complete (its vmcnt(0) in front of the first store,
    // several barriers ago), and the only operations older than this K-tile's own pieces are the epilogue's 16 stores: the counted waits of these
    // four phases would wait for store acknowledgements and nothing else, so they are skipped - K-tile 1's first `vmcnt(10)` retires the stores.
    auto ktile = [&](auto buf_c, const int kt, const bool nxt, const bool seam) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, NB = BUF ^ 1;
        const bool s1 = kt + 1 < nk || nxt, s2 = kt + 2 < nk || nxt;
        read_b(sb_[NB], BUF, P8_B1);
        __builtin_amdgcn_sched_barrier(0);
        if (s2) { stage(P8_A0, kt + 2); if (!seam) wait_vm<10>(); } else if (s1) wait_vm<8>(); else wait_vm<0>();
        mma16q(0, 0, fa[0], sb_[BUF]);
        regs_ready(sb_[NB]);
        close_phase();
        read_a(fa[1], BUF, P8_A1);
        __builtin_amdgcn_sched_barrier(0);
        if (s2) { stage(P8_B0, kt + 2); if (!seam) wait_vm<10>(); } else if (s1) wait_vm<6>();
        mma16q(0, 1, fa[0], sb_[NB]);
        regs_ready(fa[1]);
        close_phase();
        if (s1) read_a(fa[0], NB, P8_A0);
        __builtin_amdgcn_sched_barrier(0);
        if (s2) { stage(P8_B1, kt + 2); if (!seam) wait_vm<10>(); } else if (s1) wait_vm<4>();
        mma16q(1, 1, fa[1], sb_[NB]);
        regs_ready(fa[0]);
        close_phase();
        if (s1) read_b(sb_[NB], NB, P8_B0);
        __builtin_amdgcn_sched_barrier(0);
        if (s2) { stage(P8_A1, kt + 2); if (!seam) wait_vm<10>(); } else if (s1) wait_vm<2>();
        mma16q(1, 0, fa[1], sb_[BUF]);
        regs_ready(sb_[NB]);
        close_phase();
    };

    // ---- the epilogue of one tile: this wave's 128 x 64 sub-tile in four passes of 32 rows through its private staging region
    char* const ew = smem + P8_LDS + wid * Q8_STG;
    bf16_t* __restrict__ const Cout = static_cast<bf16_t*>(ACT ? p.C2 : p.C);
    const int ldo = ACT ? p.ldc2 : p.ldc;
    auto epilogue = [&](const int em0, const int en0) __attribute__((always_inline)) {
        const int mw = em0 + wr * 128, nw = en0 + wc * 64;
        // lane-constant staging addresses from an OPAQUE copy of the lane id: hoisted to the kernel entry they are spilled around the K loop, and a
        // scratch reload waits vmcnt(0) - here, for the next tile's pieces in flight (profiles/r05_gemm_pingpong.md, compiler findings)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int li = ln & 15, lg = ln >> 4;
        const int crow = ln >> 3, cchunk = ln & 7;
        float bias4[4][4], cs4[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nw + j * 16 + lg * 4 + r;
                bias4[j][r] = p.bias ? p.bias[n] : 0.f;
                if constexpr (LNF) cs4[j][r] = p.bias2[n];
            }
        // every LDS-DMA piece this wave has issued so far is complete from here on (the compiler waits vmcnt(0) for the loads above anyway;
        // the seam K-tile relies on it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int hfr = 0, htok = 0, hpart = 0, hhead = 0;
        if constexpr (HEADS) {
            hfr = (mw + crow) / p.omap.p0; htok = (mw + crow) - hfr * p.omap.p0;
            hpart = (nw >> 6) / p.omap.p1; hhead = (nw >> 6) - hpart * p.omap.p1;
        }
        const float* __restrict__ st = static_cast<const float*>(p.aux);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * q + ii;
                float mean = 0.f, rstd = 0.f;
                if constexpr (LNF) { const int m = min(mw + i * 16 + li, M - 1); mean = st[m]; rstd = st[(long)M + m]; }
                const int rl = ii * 16 + li;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16_t* slot = reinterpret_cast<bf16_t*>(ew + rl * 128 + ((((j << 1) | (lg >> 1)) ^ (rl & 7)) << 4) + ((lg & 1) << 3));
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (LNF) v[r] = lnfold_bias(acc[i][j][r], mean, rstd, cs4[j][r], bias4[j][r]);
                        else v[r] = acc[i][j][r] + bias4[j][r];
                        if constexpr (ACT) v[r] = qgelu_t<bf16_t>(v[r]);
                    }
                    store4(slot, v);
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rl = it * 8 + crow;
                const int m = mw + q * 32 + rl, n = nw + cchunk * 8;
                const uint4 v = *reinterpret_cast<const uint4*>(ew + rl * 128 + ((cchunk ^ (rl & 7)) << 4));
                if (m < M) {
                    if constexpr (HEADS) store16_nt(Cout + ((((long)hfr * p.omap.p1 + hhead) * 3 + hpart) * p.omap.p0 + htok) * 64 + cchunk * 8, v);
                    else store16_nt(Cout + (long)m * ldo + n, v);
                }
                if constexpr (HEADS) { htok += 8; if (htok >= p.omap.p0) { htok -= p.omap.p0; ++hfr; } }
            }
        }
    };

    // ---- prologue of the block's first tile (gemm_fast8p_kernel's)
    stage(P8_A0, 0); stage(P8_B0, 0); stage(P8_B1, 0); stage(P8_A1, 0);
    stage(P8_A0, 1); stage(P8_B0, 1); stage(P8_B1, 1); stage(P8_A1, 1);
    wait_vm<10>();
    __builtin_amdgcn_s_barrier();
    read_a(fa[0], 0, P8_A0);
    read_b(sb_[0], 0, P8_B0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    regs_ready(fa[0]); regs_ready(sb_[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
#pragma unroll 1
    for (int j = 0; j < ntiles; ++j) {
        const bool nxt = j + 1 < ntiles;
        if (nxt) { tile_pos(j + 1, m0n, n0n); ran = rsrc_a(m0n); rbn = rsrc_b(n0n); }
#pragma unroll 1
        for (int kt = 0; kt < nk; kt += 2) {
            ktile(I0{}, kt, nxt, kt == 0 && j > 0);
            ktile(I1{}, kt + 1, nxt, false);
        }
        epilogue(m0, n0);
        m0 = m0n; n0 = n0n; ra = ran; rb = rbn;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();            // pairs with group 1's extra barrier
}
