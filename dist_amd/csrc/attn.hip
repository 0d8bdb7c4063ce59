// Attention kernels.
//
// (1) dist_op_attention: per-frame multi-head self-attention of the frozen ViT
//     (nn.MultiheadAttention in ResidualAttentionBlockMid, reference clip.py:155,166-168).
//     One workgroup per (frame, head); K (row-major) and V (transposed) of the head live in
//     LDS (L <= 257 keys x 64 dims), each wave walks 16-query tiles with an online softmax
//     over 32-key slabs.  Both products are issued "swapped":
//         S^T = K Q^T   -> a lane holds, for ONE query (lane & 15), 4 keys per 16-key tile
//         O^T = V^T P^T -> the P^T fragment a lane needs is exactly what it already holds,
//     so the softmax is lane-local plus two cross-group shuffles and P never touches LDS.
// (2) dist_op_xattn1q(+_bwd): the one-query cross attention of the ada-pooling network
//     (CrossAttentionBlockGenral, clip.py:139-147; dist.py:144,158): latency-bound, one
//     wave per (batch element, head).
#include <stdlib.h>
#include "common.h"

namespace {

// waves per workgroup x two 16-query tiles = tile slots of one pass over the keys: 7 waves (14 slots) for the 13 tiles of L = 197 (8 waves left 3 of
// them half idle); 9 waves (18 slots) for the 17 tiles of L = 257 (ViT-L/14), which 7 waves served in a second pass with four of them idle
constexpr int NT7 = 448, NT9 = 576;
constexpr int HD = 64;           // head dim
constexpr int KPAD = 8;

// 4x16 block transpose reads: lane li of a 16-lane group passes the address of 4 contiguous bf16 of block row li>>2 and
// receives column li of the block (rows = 4 consecutive keys); the second read is the block 16 keys further
DEV void v_frag_tr(Frag<bf16_t>& f, const bf16_t* p0, int stride16) {
    typedef s16x4 __attribute__((address_space(3))) * lds_v4;
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p0));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p0 + stride16));
    f.v = u.v;
}
DEV void v_frag_tr(Frag<float>&, const float*, int) {}

// 8 OCP e4m3 bytes -> a bf16 fragment (exact: every e4m3 value is a bf16 value)
DEV void frag_from_e4m3(Frag<bf16_t>& f, const int lo, const int hi) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    f.v = bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)c[0], (bf16_t)c[1], (bf16_t)d[0], (bf16_t)d[1]};
}
DEV void frag_from_e4m3(Frag<float>&, int, int) {}
DEV void frag_load_e4m3(Frag<bf16_t>& f, const unsigned char* p) {
    const int2 w = *reinterpret_cast<const int2*>(p);
    frag_from_e4m3(f, w.x, w.y);
}
DEV void frag_load_e4m3(Frag<float>&, const unsigned char*) {}

// qkv8 (bf16 instantiation, head-major layout only): q | k | v arrive as e4m3 bytes with ONE scale in_scale[0] (a DIST_EPI_OUT8 image of the
// in_proj output): they are widened to bf16 on their way into LDS / the query fragments, the products stay bf16 MFMAs, the scale enters the
// scores as scale^2 and the output as scale.
// NSLAB = 0: online softmax over 32-key slabs (any L).  NSLAB = Lp / 32 (bf16): whole-row softmax - every score of a 16-query tile stays in
// registers (2 NSLAB accumulator quads), ONE row maximum, no running maximum / rescale of the output per slab: ~190 VALU + 8 NSLAB exponentials per tile
// instead of 130 VALU + 9 exponentials per SLAB (the slab loop was VALU-bound: 91 us for 155 MB at L = 197).
template <typename T, int NT, int NSLAB>
__global__ __launch_bounds__(NT, (NSLAB > 0 ? 4 : 1)) void attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int L, int heads, int Lp, int hm,
                                                  unsigned char* __restrict__ out8, const float* __restrict__ out8_scale, float* __restrict__ out8_amax,
                                                  const unsigned char* __restrict__ qkv8, const float* __restrict__ in_scale, const int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KLD = HD + KPAD, VLD = Lp + KPAD;
    constexpr bool TRV = sizeof(T) == 2;         // bf16: V stays row-major and is gathered with LDS transpose reads
    // bf16: K keeps only the rows up to the next multiple of 16: the key slabs read up to Lp rows, but a score of a key >= L is
    // replaced, never used, so those reads may land in V's first rows (finite data) - V keeps all Lp rows (zeros behind L: 0 x V).
    // L = 257: (272 + 288) x 144 B = 80 640 B, and TWO 7-wave workgroups share a CU's 160 KB (82 944 B each did not)
    const int Lk = TRV ? (L + 15) / 16 * 16 : Lp;
    T* Ks = reinterpret_cast<T*>(smem);          // [Lk][KLD]
    T* Vt = Ks + Lk * KLD;                       // fp32: [HD][VLD] (transposed) | bf16: [Lp][KLD] (row-major)

    const int f = blockIdx.x / heads, h = blockIdx.x % heads;
    const int d = heads * HD;
    // rows layout: q/k/v of head h are 64-column slices of the packed [frames*L][3d] rows; head-major layout: three
    // contiguous [L][64] blocks per (frame, head)
    const int ld = hm ? HD : 3 * d;
    const T* qb = hm ? qkv + ((long)(f * heads + h) * 3) * L * HD : qkv + (long)f * L * ld + h * HD;
    const T* kb = hm ? qb + (long)L * HD : qb + d;
    const T* vb = hm ? kb + (long)L * HD : kb + d;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const bool in8 = TRV && qkv8 != nullptr;
    const unsigned char* q8 = qkv8 + ((long)(f * heads + h) * 3) * L * HD;
    const unsigned char* k8 = q8 + (long)L * HD;
    const unsigned char* v8 = k8 + (long)L * HD;
    const float s_in = in8 ? in_scale[0] : 1.f;
    // scores are kept in units of log2(e): the softmax is 2^(s - max) with one v_exp_f32 per score and no multiply in front of it
    const float sc_qk = 0.125f * 1.44269504088896341f * s_in * s_in;
    auto ldf = [&](Frag<T>& fr, const T* p, const unsigned char* p8, long off) __attribute__((always_inline)) {
        if (in8) frag_load_e4m3(fr, p8 + off); else frag_load(fr, p + off);
    };

    const int nq = (L + 15) / 16, nslab = Lp / 32;
    constexpr int NW = NT / 64;
    // the query fragments of this wave's first tile pair are requested right behind the K / V staging stores: their
    // latency hides behind the block barrier
    auto load_q = [&](int qt0, int (&qrow)[2], Frag<T> (&fq)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int qt = qt0 + u * NW;
            qrow[u] = qt < nq ? qt * 16 + li : L;          // L = "no query" (clamped load, nothing stored)
            const long qo = (long)min(qrow[u], L - 1) * ld + lg * 8;
            ldf(fq[u][0], qb, q8, qo);
            ldf(fq[u][1], qb, q8, qo + 32);
        }
    };
    int qrow[2];
    Frag<T> fq[2][2];

    // K and V: row-major, 16-B vectors, d fastest.  All loads of a thread are issued back to back (unconditional, key
    // clamped, padding rows zeroed at the LDS store): one memory latency for the whole staging phase instead of one
    // per loop iteration
    constexpr int KV_IT = 6;                                  // covers Lp * 8 vectors up to Lp = 336 (L = 257 -> Lp = 288)
    const int n_it = (Lp * (HD / 8) + NT - 1) / NT;           // block-uniform
    if (dbg & 2) {
    } else if (in8) {
        // e4m3 rows are 64 bytes: 16-byte loads carry two 8-element vectors (8-byte loads run at little more than half the rate)
        constexpr int P_IT = KV_IT / 2;
        const int np = (Lp * (HD / 16) + NT - 1) / NT;
        int4 wk[P_IT], wv[P_IT];
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            if (it < np) {
                const int pv = min(tid + it * NT, Lp * (HD / 16) - 1);
                const int key = min(pv / (HD / 16), L - 1), dp = pv % (HD / 16);
                wk[it] = *reinterpret_cast<const int4*>(k8 + (long)key * HD + dp * 16);
                wv[it] = *reinterpret_cast<const int4*>(v8 + (long)key * HD + dp * 16);
            }
        }
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int pv = tid + it * NT;
            if (it < np && pv < Lp * (HD / 16)) {
                const int key = pv / (HD / 16), dp = pv % (HD / 16);
                if (key >= L) { wk[it] = make_int4(0, 0, 0, 0); wv[it] = make_int4(0, 0, 0, 0); }
                Frag<T> a, b;
                frag_from_e4m3(a, wk[it].x, wk[it].y); frag_from_e4m3(b, wk[it].z, wk[it].w);
                if (key < Lk) { frag_store(a, Ks + key * KLD + dp * 16); frag_store(b, Ks + key * KLD + dp * 16 + 8); }
                frag_from_e4m3(a, wv[it].x, wv[it].y); frag_from_e4m3(b, wv[it].z, wv[it].w);
                frag_store(a, Vt + key * KLD + dp * 16); frag_store(b, Vt + key * KLD + dp * 16 + 8);
            }
        }
    } else {
        Frag<T> rk[KV_IT], rv[KV_IT];
#pragma unroll
        for (int it = 0; it < KV_IT; ++it) {
            if (it < n_it) {
                const int v = min(tid + it * NT, Lp * (HD / 8) - 1);
                const int key = min(v / (HD / 8), L - 1), dv = v % (HD / 8);
                ldf(rk[it], kb, k8, (long)key * ld + dv * 8);
                if (TRV) ldf(rv[it], vb, v8, (long)key * ld + dv * 8);
            }
        }
#pragma unroll
        for (int it = 0; it < KV_IT; ++it) {
            const int v = tid + it * NT;
            if (it < n_it && v < Lp * (HD / 8)) {
                const int key = v / (HD / 8), dv = v % (HD / 8);
                if (key >= L) { frag_zero(rk[it]); if (TRV) frag_zero(rv[it]); }
                if (key < Lk) frag_store(rk[it], Ks + key * KLD + dv * 8);
                if (TRV) frag_store(rv[it], Vt + key * KLD + dv * 8);
            }
        }
    }
    if (!TRV) {
        // V transposed, key fastest (conflict-free scalar LDS writes)
        for (int v = tid; v < Lp * (HD / 8); v += NT) {
            const int key = v % Lp, dv = v / Lp;
            Frag<T> fr;
            frag_zero(fr);
            if (key < L) frag_load(fr, vb + (long)key * ld + dv * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[(dv * 8 + e) * VLD + key] = from_f<T>(frag_get(fr, e));
        }
    }
    load_q(wid, qrow, fq);
    __syncthreads();
    if (dbg & 1) return;                                   // measurement knob DIST_AMD_ATTN_DBG: 1 = staging only, 2 = no K / V staging loads

    // normalise one tile's output and store it (bf16 rows / e4m3 rows / fp32 rows)
    auto finish = [&](const f32x4 (&ou)[4], const float l_in, const int qr) __attribute__((always_inline)) {
            float l = l_in;
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            const float inv = s_in / l;
            if (TRV && out8) {
                // e4m3 output (the A operand of a DIST_EPI_FP8 out-projection): e4m3(clamp(bf16(o) / scale)) with the caller's per-tensor
                // scale, INSTEAD of the bf16 rows; same lane exchange as below on one dword (4 columns) per fragment
                const float si = 1.0f / out8_scale[0];
                float am = 0.f;
                auto pk4 = [&](int j) __attribute__((always_inline)) -> int {
                    float y[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = (float)(bf16_t)(ou[j][r] * inv);
                        am = fmaxf(am, fabsf(v));
                        y[r] = fminf(fmaxf(v * si, -448.f), 448.f);
                    }
                    int w = 0;
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], w, false);
                    return __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], w, true);
                };
                const int a = pk4(0), b = pk4(1), c = pk4(2), d4 = pk4(3);
                const bool odd = lg & 1;
                const int r0 = __shfl_xor(odd ? a : c, 16, 64), r1 = __shfl_xor(odd ? b : d4, 16, 64);
                if (qr < L) {
                    unsigned char* orow = out8 + ((long)f * L + qr) * d + h * HD + (odd ? 32 : 0) + (lg >> 1) * 8;
                    *reinterpret_cast<int2*>(orow) = odd ? make_int2(r0, c) : make_int2(a, r0);
                    *reinterpret_cast<int2*>(orow + 16) = odd ? make_int2(r1, d4) : make_int2(b, r1);
                }
                if (out8_amax) {
                    am = qr < L ? am : 0.f;
                    am = wave_max(am, 64);
                    if (lane == 0 && am > *static_cast<const volatile float*>(out8_amax)) atomicMax(reinterpret_cast<unsigned*>(out8_amax), __float_as_uint(am));
                }
            } else if (TRV) {
                // a lane holds 4 consecutive columns per 16-column fragment (8 bytes of bf16): lanes lg and lg ^ 1 swap
                // two fragments each, so every lane ends up with 8 consecutive columns of two fragments and writes two
                // 16-byte pieces instead of four 8-byte ones (the epilogue is store-issue-bound)
                auto pk2 = [&](int j, int half) __attribute__((always_inline)) -> unsigned {
                    union { bf16_t b[2]; unsigned w; } cv;
                    cv.b[0] = (bf16_t)(ou[j][2 * half] * inv); cv.b[1] = (bf16_t)(ou[j][2 * half + 1] * inv);
                    return cv.w;
                };
                // scalars, not an array: with an array hipcc turns the selects below into lane-indexed stack accesses
                const unsigned a0 = pk2(0, 0), a1 = pk2(0, 1), b0 = pk2(1, 0), b1 = pk2(1, 1);
                const unsigned c0 = pk2(2, 0), c1 = pk2(2, 1), d0 = pk2(3, 0), d1 = pk2(3, 1);
                const bool odd = lg & 1;
                // even lg keeps fragments 0, 1 and sends 2, 3; odd lg keeps 2, 3 and sends 0, 1
                const unsigned r00 = __shfl_xor(odd ? a0 : c0, 16, 64), r01 = __shfl_xor(odd ? a1 : c1, 16, 64);
                const unsigned r10 = __shfl_xor(odd ? b0 : d0, 16, 64), r11 = __shfl_xor(odd ? b1 : d1, 16, 64);
                if (qr < L) {
                    T* orow = out + ((long)f * L + qr) * d + h * HD + (odd ? 32 : 0) + (lg >> 1) * 8;
                    const unsigned k00 = odd ? c0 : a0, k01 = odd ? c1 : a1, k10 = odd ? d0 : b0, k11 = odd ? d1 : b1;
                    store16_nt(orow, odd ? make_uint4(r00, r01, k00, k01) : make_uint4(k00, k01, r00, r01));
                    store16_nt(orow + 16, odd ? make_uint4(r10, r11, k10, k11) : make_uint4(k10, k11, r10, r11));
                }
            } else if (qr < L) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v4[4] = {ou[j][0] * inv, ou[j][1] * inv, ou[j][2] * inv, ou[j][3] * inv};
                    store4(out + ((long)f * L + qr) * d + h * HD + j * 16 + lg * 4, v4);
                }
            }
    };

    if constexpr (NSLAB > 0 && TRV) {
        for (int qt0 = wid; qt0 < nq; qt0 += 2 * NW) {
            if (qt0 != wid) load_q(qt0, qrow, fq);
            const bool two = qt0 + NW < nq;               // wave-uniform
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) continue;
                // scores of ALL keys for this lane's query: st[kt][r] = key 16 kt + 4 lg + r
                f32x4 st[2 * NSLAB];
#pragma unroll
                for (int kt = 0; kt < 2 * NSLAB; ++kt) {
                    const T* kp = Ks + (kt * 16 + li) * KLD + lg * 8;
                    Frag<T> k0, k1;
                    frag_load(k0, kp);
                    frag_load(k1, kp + 32);
                    st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma16(k0, fq[u][0], st[kt]);
                    mma16(k1, fq[u][1], st[kt]);
                }
                // keys >= L live in the last slab only (Lp - 32 < L <= Lp)
#pragma unroll
                for (int kt = 2 * NSLAB - 2; kt < 2 * NSLAB; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[kt][r] = (kt * 16 + lg * 4 + r) < L ? st[kt][r] : -1e30f;
                float mx = -1e30f;
#pragma unroll
                for (int kt = 0; kt < 2 * NSLAB; ++kt) mx = fmaxf(fmaxf(mx, fmaxf(st[kt][0], st[kt][1])), fmaxf(st[kt][2], st[kt][3]));
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m = mx * sc_qk;                // raw products: the scale (> 0) commutes with the maximum
                f32x4 ou[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) ou[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                float ps = 0.f;
#pragma unroll
                for (int sl = 0; sl < NSLAB; ++sl) {
                    Frag<T> fp;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = __builtin_amdgcn_exp2f(fmaf(st[2 * sl + t][r], sc_qk, -m));   // a masked key holds -1e30: exactly 0
                            ps += pv;
                            frag_set(fp, t * 4 + r, pv);
                        }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        Frag<T> fv;
                        v_frag_tr(fv, Vt + (sl * 32 + 4 * lg + (li >> 2)) * KLD + j * 16 + (li & 3) * 4, 16 * KLD);
                        mma16(fv, fp, ou[j]);
                    }
                }
                finish(ou, ps, qrow[u]);
                __builtin_amdgcn_sched_barrier(0);        // (one tile after the other: interleaved, the two tiles' scores need 256 registers)
            }
        }
        return;
    }
    // each wave walks TWO 16-query tiles at once (qt, qt + NW): two independent score -> softmax -> PV dependency
    // chains per wave, so the shuffle / exp / MFMA latencies of one tile are covered by the other
    for (int qt0 = wid; qt0 < nq; qt0 += 2 * NW) {
        if (qt0 != wid) load_q(qt0, qrow, fq);
        f32x4 o[2][4];
        float mrun[2], lrun[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            mrun[u] = -1e30f; lrun[u] = 0.f;
        }
        const bool two = qt0 + NW < nq;                   // wave-uniform

        for (int s = 0; s < nslab; ++s) {
            Frag<T> fk[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const T* kp = Ks + (s * 32 + t * 16 + li) * KLD + lg * 8;
                frag_load(fk[t][0], kp);
                frag_load(fk[t][1], kp + 32);
            }
            Frag<T> fv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // V^T fragment: row = dcol j*16+li, k-slots = keys (s*32 + 4lg + e | +16)
                if (TRV) v_frag_tr(fv[j], Vt + (s * 32 + 4 * lg + (li >> 2)) * KLD + j * 16 + (li & 3) * 4, 16 * KLD);
                else { const T* vp = Vt + (j * 16 + li) * VLD + s * 32 + lg * 4; frag_load44(fv[j], vp, vp + 16); }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) continue;
                f32x4 st[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma16(fk[t][0], fq[u][0], st[t]);
                    mma16(fk[t][1], fq[u][1], st[t]);
                }
                float mx = -1e30f;
                // (masking only in the last slab - the only one that can hold keys >= L - saves 24 VALU per slab and tile but costs 6 registers:
                // 130 VGPRs, three waves per SIMD instead of four, 93 -> 110 us)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = s * 32 + t * 16 + lg * 4 + r;
                        st[t][r] = key < L ? st[t][r] : -1e30f;   // raw products: the scale (> 0) commutes with the maximum ...
                        mx = fmaxf(mx, st[t][r]);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(mrun[u], mx * sc_qk);
                const float scale = __builtin_amdgcn_exp2f(mrun[u] - mnew);
                mrun[u] = mnew;
                Frag<T> fp;
                float ps = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(fmaf(st[t][r], sc_qk, -mnew));   // ... and folds into one fma per score; a masked key holds -1e30: exactly 0
                        ps += pv;
                        frag_set(fp, t * 4 + r, pv);
                    }
                lrun[u] = lrun[u] * scale + ps;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[u][j][r] *= scale;
                    mma16(fv[j], fp, o[u][j]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) finish(o[u], lrun[u], qrow[u]);
    }
}

// ---- one-query cross attention ---------------------------------------------------------
// q [B, C], kv [B*S, 2C] (k = cols [0,C), v = cols [C,2C)), heads = C/64; one wave per (B, head)
// Lane map of both kernels: lane = 8 * kg + vec; key group kg (0..7) walks keys kg, kg + 8, ... and lane `vec` owns the
// 8-element (16 B bf16) vector `vec` of the 64-wide head slice, so every key / value row is one coalesced 128-byte
// access and a wave moves 8 rows per instruction (the first version read 2 bytes per lane and instruction).
template <typename T>
DEV float dot8(const Frag<T>& a, const float (&b)[8]) {
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += frag_get(a, e) * b[e];
    return acc;
}
DEV float sum_vec_lanes(float v) {            // over the 8 lanes (vec) of a key group
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}
DEV float sum_key_groups(float v) {           // over the 8 key groups (same vec)
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(64) void xattn1q_fwd(const T* __restrict__ q, const T* __restrict__ kv, T* __restrict__ o,
                                                  float* __restrict__ probs, int S, int C) {
    extern __shared__ float sp[];      // [S]
    const int H = C / HD;
    const int bi = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
    const int kg = lane >> 3, vec = lane & 7;
    const T* kvb = kv + (long)bi * S * 2 * C + h * HD + vec * 8;
    float qv[8];
    {
        Frag<T> fq;
        frag_load(fq, q + (long)bi * C + h * HD + vec * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = frag_get(fq, e) * 0.125f;
    }
    float mx = -1e30f;
    for (int s = kg; s < S; s += 8) {
        Frag<T> fk;
        frag_load(fk, kvb + (long)s * 2 * C);
        const float sc = sum_vec_lanes(dot8(fk, qv));
        if (vec == 0) sp[s] = sc;
        mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx, 64);
    __syncthreads();
    float sum = 0.f;
    for (int s = lane; s < S; s += 64) { const float e = __expf(sp[s] - mx); sp[s] = e; sum += e; }
    sum = wave_sum(sum, 64);
    const float inv = 1.f / sum;
    __syncthreads();
    float* pr = probs + ((long)bi * H + h) * S;
    for (int s = lane; s < S; s += 64) { const float pv = sp[s] * inv; sp[s] = pv; pr[s] = pv; }
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int s = kg; s < S; s += 8) {
        Frag<T> fv;
        frag_load(fv, kvb + (long)s * 2 * C + C);
        const float pv = sp[s];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += pv * frag_get(fv, e);
    }
    Frag<T> fo;
#pragma unroll
    for (int e = 0; e < 8; ++e) frag_set(fo, e, sum_key_groups(acc[e]));
    if (kg == 0) frag_store(fo, o + (long)bi * C + h * HD + vec * 8);
}

template <typename T>
__global__ __launch_bounds__(64) void xattn1q_bwd(const T* __restrict__ q, const T* __restrict__ kv, const float* __restrict__ probs,
                                                  const T* __restrict__ d_o, T* __restrict__ dq, T* __restrict__ dkv, int S, int C) {
    extern __shared__ float sp[];      // [2][S]: p, ds
    float* sds = sp + S;
    const int H = C / HD;
    const int bi = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
    const int kg = lane >> 3, vec = lane & 7;
    const T* kvb = kv + (long)bi * S * 2 * C + h * HD + vec * 8;
    T* dkvb = dkv + (long)bi * S * 2 * C + h * HD + vec * 8;
    const float* pr = probs + ((long)bi * H + h) * S;
    float dov[8], qv[8];
    {
        Frag<T> f;
        frag_load(f, d_o + (long)bi * C + h * HD + vec * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dov[e] = frag_get(f, e);
        frag_load(f, q + (long)bi * C + h * HD + vec * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = frag_get(f, e);
    }
    float dot = 0.f;
    for (int s = kg; s < S; s += 8) {
        Frag<T> fv;
        frag_load(fv, kvb + (long)s * 2 * C + C);
        const float dp = sum_vec_lanes(dot8(fv, dov));
        const float pv = pr[s];
        if (vec == 0) { sp[s] = pv; sds[s] = dp; dot += pv * dp; }
    }
    dot = wave_sum(dot, 64);
    __syncthreads();
    for (int s = lane; s < S; s += 64) sds[s] = sp[s] * (sds[s] - dot) * 0.125f;   // ds / sqrt(64)
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int s = kg; s < S; s += 8) {
        Frag<T> fk, dk, dv;
        frag_load(fk, kvb + (long)s * 2 * C);
        const float ds = sds[s], pv = sp[s];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[e] += ds * frag_get(fk, e);
            frag_set(dk, e, ds * qv[e]);
            frag_set(dv, e, pv * dov[e]);
        }
        frag_store(dk, dkvb + (long)s * 2 * C);
        frag_store(dv, dkvb + (long)s * 2 * C + C);
    }
    Frag<T> fo;
#pragma unroll
    for (int e = 0; e < 8; ++e) frag_set(fo, e, sum_key_groups(acc[e]));
    if (kg == 0) frag_store(fo, dq + (long)bi * C + h * HD + vec * 8);
}

template <typename T>
int launch_attn(const void* qkv, void* out, int frames, int L, int heads, int hm, hipStream_t s, unsigned char* out8 = nullptr,
                const float* out8_scale = nullptr, float* out8_amax = nullptr, const unsigned char* qkv8 = nullptr, const float* in_scale = nullptr) {
    const int Lp = (L + 31) / 32 * 32;
    const int Lk = (L + 15) / 16 * 16;
    const size_t smem = sizeof(T) == 2 ? (size_t)(Lk + Lp) * (HD + KPAD) * sizeof(T)
                                       : ((size_t)Lp * (HD + KPAD) + (size_t)HD * (Lp + KPAD)) * sizeof(T);
    if (smem > 160 * 1024) return DIST_ERR_ARG;
    const int nq = (L + 15) / 16;
    static const int dbg_env = dist_measure_knob("DIST_AMD_ATTN_DBG", 0);
    static const int nt_env = DIST_AB_KNOB("DIST_AMD_ATTN_NT", 0);      // A/B: force 448 / 576 threads
    // nine waves only where two 7-wave workgroups do not fit a CU's LDS anyway (L = 257: 234 us with nine waves and one workgroup per CU,
    // 190 us with seven waves and two - tools/bench_attn.py)
    const bool nine = nt_env ? nt_env == NT9 : (nq > 14 && nq <= 18 && smem > 80 * 1024);
    if (nine) {
        static DistSmemOnce attr9;
        RUN_(dist_max_smem(attr9, reinterpret_cast<const void*>(attn_kernel<T, NT9, 0>), smem));
        hipLaunchKernelGGL((attn_kernel<T, NT9, 0>), dim3(frames * heads), dim3(NT9), smem, s, static_cast<const T*>(qkv), static_cast<T*>(out), L, heads, Lp, hm,
                           out8, out8_scale, out8_amax, qkv8, in_scale, dbg_env);
    } else {
        // whole-row softmax where the scores of a query tile fit the register budget of two workgroups per CU (Lp = 224: L = 193 .. 224, the ViT-B/16 plane + cls)
        static const int full_env = dist_knob("DIST_AMD_ATTN_FULLROW", 1);      // measurement knob: 0 = online softmax
        const bool full7 = sizeof(T) == 2 && Lp == 224 && full_env;
        static DistSmemOnce attr7, attr7f;
        if (full7) {
            RUN_(dist_max_smem(attr7f, reinterpret_cast<const void*>(attn_kernel<T, NT7, 7>), smem));
            hipLaunchKernelGGL((attn_kernel<T, NT7, 7>), dim3(frames * heads), dim3(NT7), smem, s, static_cast<const T*>(qkv), static_cast<T*>(out), L, heads, Lp, hm,
                               out8, out8_scale, out8_amax, qkv8, in_scale, dbg_env);
        } else {
            RUN_(dist_max_smem(attr7, reinterpret_cast<const void*>(attn_kernel<T, NT7, 0>), smem));
            hipLaunchKernelGGL((attn_kernel<T, NT7, 0>), dim3(frames * heads), dim3(NT7), smem, s, static_cast<const T*>(qkv), static_cast<T*>(out), L, heads, Lp, hm,
                               out8, out8_scale, out8_amax, qkv8, in_scale, dbg_env);
        }
    }
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

}  // namespace

extern "C" int dist_op_attention(const void* qkv, void* out, int frames, int L, int heads, int qkv_layout, int dtype, void* stream) {
    if (!qkv || !out || frames <= 0 || L <= 0 || heads <= 0) return DIST_ERR_ARG;
    if (qkv_layout != DIST_QKV_ROWS && qkv_layout != DIST_QKV_HEADS) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == DIST_BF16 ? launch_attn<bf16_t>(qkv, out, frames, L, heads, qkv_layout, s) : launch_attn<float>(qkv, out, frames, L, heads, qkv_layout, s);
}

extern "C" int dist_op_attention_out8(const void* qkv, void* out8, const float* out8_scale, float* out8_amax, int frames, int L, int heads, int qkv_layout,
                                      void* stream) {
    if (!qkv || !out8 || !out8_scale || frames <= 0 || L <= 0 || heads <= 0) return DIST_ERR_ARG;
    if (qkv_layout != DIST_QKV_ROWS && qkv_layout != DIST_QKV_HEADS) return DIST_ERR_ARG;
    return launch_attn<bf16_t>(qkv, nullptr, frames, L, heads, qkv_layout, static_cast<hipStream_t>(stream), static_cast<unsigned char*>(out8), out8_scale, out8_amax);
}

extern "C" int dist_op_attention_fp8(const void* qkv8, const float* in_scale, void* out, void* out8, const float* out8_scale, float* out8_amax,
                                     int frames, int L, int heads, void* stream) {
    if (!qkv8 || !in_scale || (!out && !out8) || (out8 && !out8_scale) || frames <= 0 || L <= 0 || heads <= 0) return DIST_ERR_ARG;
    return launch_attn<bf16_t>(nullptr, out8 ? nullptr : out, frames, L, heads, DIST_QKV_HEADS, static_cast<hipStream_t>(stream), static_cast<unsigned char*>(out8),
                               out8_scale, out8_amax, static_cast<const unsigned char*>(qkv8), in_scale);
}

extern "C" int dist_op_xattn1q(const void* q, const void* kv, void* o, float* probs, int B, int S, int C, int dtype, void* stream) {
    if (!q || !kv || !o || !probs || B <= 0 || S <= 0 || C % HD) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H = C / HD;
    const size_t smem = (size_t)S * sizeof(float);
    if (dtype == DIST_BF16)
        hipLaunchKernelGGL(xattn1q_fwd<bf16_t>, dim3(B * H), dim3(64), smem, s, (const bf16_t*)q, (const bf16_t*)kv, (bf16_t*)o, probs, S, C);
    else
        hipLaunchKernelGGL(xattn1q_fwd<float>, dim3(B * H), dim3(64), smem, s, (const float*)q, (const float*)kv, (float*)o, probs, S, C);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_xattn1q_bwd(const void* q, const void* kv, const float* probs, const void* d_o,
                                   void* dq, void* dkv, int B, int S, int C, int dtype, void* stream) {
    if (!q || !kv || !probs || !d_o || !dq || !dkv || B <= 0 || S <= 0 || C % HD) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H = C / HD;
    const size_t smem = (size_t)2 * S * sizeof(float);
    if (dtype == DIST_BF16)
        hipLaunchKernelGGL(xattn1q_bwd<bf16_t>, dim3(B * H), dim3(64), smem, s, (const bf16_t*)q, (const bf16_t*)kv, probs, (const bf16_t*)d_o,
                           (bf16_t*)dq, (bf16_t*)dkv, S, C);
    else
        hipLaunchKernelGGL(xattn1q_bwd<float>, dim3(B * H), dim3(64), smem, s, (const float*)q, (const float*)kv, probs, (const float*)d_o,
                           (float*)dq, (float*)dkv, S, C);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}
