// Timing-only library only (-DDIST_AMD_MEASURE): host side of gemm_fast8q_kernel, included by gemm_fast.hip.
// the multi-tile form (gemm_fast8q_kernel): -1 = not its call, else its flag set
static int fast_q_flags(const dist_gemm_args* a) {
    if (a->dtype != DIST_BF16 || a->taps != 1 || a->amap.mode != DIST_RM_PLAIN) return -1;
    if (a->flags & (DIST_EPI_MULG | DIST_EPI_MULG_POST | DIST_EPI_FP8 | DIST_EPI_OUT8 | DIST_EPI_FP8_ASCALAR | DIST_EPI_RES | DIST_EPI_ROWSTATS)) return -1;
    if (a->omap.mode != DIST_OM_PLAIN && a->omap.mode != DIST_OM_HEADS) return -1;
    int f = 0;
    if (a->flags & DIST_EPI_LNFOLD) { if (!a->aux || !a->bias2) return -1; f |= QF_LNFOLD; } else if (a->bias2) return -1;
    if (a->flags & DIST_EPI_ACT2) { if (a->C || !a->C2) return -1; f |= QF_ACT; } else if (!a->C) return -1;
    if (a->omap.mode == DIST_OM_HEADS) { if ((f & QF_ACT) || a->omap.p0 < 16 || a->omap.p1 < 1 || a->ldc != 64) return -1; f |= QF_HEADS; }
    if (a->N % 256 || a->K % (2 * P8_BK) || a->K < 4 * P8_BK) return -1;
    return f;
}

template <int F>
static int launch_fast_q(const dist_gemm_args* a, int ng, int grid, hipStream_t s) {
    static DistSmemOnce attr;
    RUN_(dist_max_smem(attr, reinterpret_cast<const void*>(gemm_fast8q_kernel<F>), (size_t)Q8_LDS));
    hipLaunchKernelGGL(gemm_fast8q_kernel<F>, dim3((unsigned)grid), dim3(512), (size_t)Q8_LDS, s, *a, ng);
    HIP_CHECK_RET(hipGetLastError());
    return 1;
}

// 1 = launched, 0 = not this form's call
static int try_fast_q(const dist_gemm_args* a, int ng, hipStream_t s) {
    static const int cap = DIST_AB_KNOB("DIST_AMD_FAST_TILES", 0);     // tiles per block at most (0: one tile per block = gemm_fast8p_kernel)
    if (cap <= 1) return 0;
    const int f = fast_q_flags(a);
    if (f < 0) return 0;
    const long tiles = ((a->M + BM - 1) / BM) * (a->N / 256);
    if (tiles < 2 * 256) return 0;                        // fewer than two tiles per CU: nothing to run through
    // whole rounds of 256 blocks (one per CU), as few as keep a block at <= cap tiles: a last round of fewer blocks would run on a mostly idle chip
    const long grid = 256 * ((tiles + 256l * cap - 1) / (256l * cap));
    switch (f) {
        case 0: return launch_fast_q<0>(a, ng, (int)grid, s);
        case QF_LNFOLD: return launch_fast_q<QF_LNFOLD>(a, ng, (int)grid, s);
        case QF_ACT: return launch_fast_q<QF_ACT>(a, ng, (int)grid, s);
        case QF_HEADS: return launch_fast_q<QF_HEADS>(a, ng, (int)grid, s);
        case QF_LNFOLD | QF_ACT: return launch_fast_q<QF_LNFOLD | QF_ACT>(a, ng, (int)grid, s);
        case QF_LNFOLD | QF_HEADS: return launch_fast_q<QF_LNFOLD | QF_HEADS>(a, ng, (int)grid, s);
        default: return 0;
    }
}
