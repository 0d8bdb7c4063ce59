// Small HBM-/latency-bound kernels of the DiST hot path: patch gather, elementwise ops,
// column sums (bias gradients), token-row helpers, cosine-logits + soft-target CE
// (forward and backward), fused multi-tensor AdamW, weight packing.
#include <algorithm>
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NT = 256;

inline int grid1d(long n, int per_block, int cap = 4096) {
    long g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- patchify: [b,3,T,H,W] fp32 -> rows (b,k,n) x cols (c,py,px), zero padded to Kp ----
template <typename T>
__global__ __launch_bounds__(NT) void patchify_kernel(const float* __restrict__ video, T* __restrict__ out,
                                                      int b, int Tn, int H, int W, int P, int Kp) {
    const int G = W / P, Gy = H / P, N = G * Gy, PP = P * P;
    const long total = (long)b * Tn * N * Kp;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int col = (int)(i % Kp);
        const long row = i / Kp;
        float v = 0.f;
        if (col < 3 * PP) {
            const int c = col / PP, py = (col % PP) / P, px = col % P;
            const int n = (int)(row % N);
            const long bk = row / N;
            const int k = (int)(bk % Tn), bi = (int)(bk / Tn);
            const int y = (n / G) * P + py, x = (n % G) * P + px;
            v = video[(((long)bi * 3 + c) * Tn + k) * H * W + (long)y * W + x];
        }
        out[i] = from_f<T>(v);
    }
}
// P % 8 == 0 (ViT-B/16): one block per (clip, frame, patch row).  The 3 x P image rows of that band are read as whole
// W-float lines (coalesced float4), parked in LDS in the output type, and the G patch rows are written as whole
// Kp-element lines (16 B per lane): the element-wise version above reads 64-byte pieces and spends ~40 integer
// operations per 2-byte output.
template <typename T>
__global__ __launch_bounds__(NT) void patchify_band_kernel(const float* __restrict__ video, T* __restrict__ out, int Tn, int H, int W, int P, int Kp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* band = reinterpret_cast<T*>(smem);                  // [3][P][W]
    const int G = W / P, Gy = H / P;
    const int gy = blockIdx.x % Gy;
    const long bk = blockIdx.x / Gy;
    const int k = (int)(bk % Tn), bi = (int)(bk / Tn);
    const int W4 = W / 4, per_c = P * W4;
    for (int v = threadIdx.x; v < 3 * per_c; v += NT) {
        const int c = v / per_c, rem = v - c * per_c, py = rem / W4, x4 = rem - py * W4;
        const float4 f = *reinterpret_cast<const float4*>(video + (((long)bi * 3 + c) * Tn + k) * H * W + (long)(gy * P + py) * W + x4 * 4);
        T* d = band + (c * P + py) * W + x4 * 4;
        d[0] = from_f<T>(f.x); d[1] = from_f<T>(f.y); d[2] = from_f<T>(f.z); d[3] = from_f<T>(f.w);
    }
    __syncthreads();
    const int vpr = Kp / 8, vp = P / 8;                    // 8-element vectors per output row / per patch line
    const long row0 = (bk * Gy + gy) * G;
    for (int v = threadIdx.x; v < G * vpr; v += NT) {
        const int gx = v / vpr, j = v - gx * vpr;
        Frag<T> o;
        if (j * 8 < 3 * P * P) {
            const int c = j / (P * vp), rem = j - c * (P * vp), py = rem / vp, h = rem - py * vp;
            frag_load(o, band + (c * P + py) * W + gx * P + h * 8);
        } else {
            frag_zero(o);
        }
        frag_store_nt(o, out + (row0 + gx) * Kp + j * 8);
    }
}

// ---- elementwise --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long nvec) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < nvec; i += (long)gridDim.x * NT) {
        Frag<T> x, y;
        frag_load(x, a + i * 8); frag_load(y, b + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) frag_set(x, e, frag_get(x, e) + frag_get(y, e));
        frag_store(x, o + i * 8);
    }
}
template <typename T>
__global__ __launch_bounds__(NT) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ dx, long nvec) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < nvec; i += (long)gridDim.x * NT) {
        Frag<T> x, y;
        frag_load(x, dy + i * 8); frag_load(y, pre + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) frag_set(x, e, frag_get(x, e) * qgelu_grad_t<T>(frag_get(y, e)));
        frag_store_nt(x, dx + i * 8);
    }
}

// ---- column sums (bias gradients): out[c] += sum_rows x[map(row)][c] ------------------------
template <typename T>
__global__ __launch_bounds__(NT) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long rows, int Ctot, int ld,
                                                    dist_rowmap map, int rows_per_block) {
    __shared__ float red[1024];
    const int tid = threadIdx.x;
    const int cbase = blockIdx.y * 1024;                  // column chunk of <= 1024
    const int C = min(1024, Ctot - cbase);
    const int cg = C / 4;                                 // column groups of 4 (<= 256)
    const int nslot = NT / cg;
    for (int i = tid; i < C; i += NT) red[i] = 0.f;
    __syncthreads();
    const int slot = tid / cg, c4 = (tid % cg) * 4;
    if (slot < nslot) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const long r0 = (long)blockIdx.x * rows_per_block;
        const long r1 = min(rows, r0 + rows_per_block);
        for (long r = r0 + slot; r < r1; r += nslot) {
            const int src = rowmap_src(map, (int)r, 0, 1);
            float v[4];
            load4(x + (long)src * ld + cbase + c4, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&red[c4 + e], acc[e]);
    }
    __syncthreads();
    for (int i = tid; i < C; i += NT) atomicAdd(out + cbase + i, red[i]);
}

// ---- token-row helpers -----------------------------------------------------------------------
// dst[(bj)*L + 0][:] = (src ? src[(bj)*L][:] : 0) + table[(bj % period)][:]     (fp32 table)
template <typename T>
__global__ __launch_bounds__(NT) void cls_rows_kernel(T* __restrict__ dst, const T* __restrict__ src, const float* __restrict__ table,
                                                      int nbj, int L, int C, int period) {
    const long total = (long)nbj * C;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % C);
        const long bj = i / C;
        const float s = src ? to_f(src[bj * L * C + c]) : 0.f;
        dst[bj * L * C + c] = from_f<T>(s + table[(bj % period) * C + c]);
    }
}
// dtable[(bj % period)][c] += sum over bj of d[(bj)*L][c]
template <typename T>
__global__ __launch_bounds__(NT) void cls_rows_bwd_kernel(const T* __restrict__ d, float* __restrict__ dtable, int nbj, int L, int C, int period) {
    const long total = (long)period * C;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % C);
        const int j = (int)(i / C);
        float acc = 0.f;                                    // blockIdx.y: a slice of the clips (one thread walking all of them was pure latency)
        for (long bj = j + (long)blockIdx.y * period; bj < nbj; bj += (long)period * gridDim.y) acc += to_f(d[bj * L * C + c]);
        atomicAdd(dtable + i, acc);
    }
}
// out[(b, j)][c] = x[(b,j)][c] + table[j][c]
template <typename T>
__global__ __launch_bounds__(NT) void add_table_kernel(const T* __restrict__ x, const float* __restrict__ table, T* __restrict__ out,
                                                       long rows, int C, int period) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % C);
        const long r = i / C;
        out[i] = from_f<T>(to_f(x[i]) + table[(r % period) * C + c]);
    }
}
// out[(bj, n)][c] = sum_a x[((bj*alpha + a), n)][c]          (backward of nearest upsample x alpha in T)
template <typename T>
__global__ __launch_bounds__(NT) void pair_sum_kernel(const T* __restrict__ x, T* __restrict__ out, long nbj, int rowlen, int alpha) {
    const long total = nbj * rowlen / 8;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long bj = (i * 8) / rowlen, off = (i * 8) % rowlen;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int a = 0; a < alpha; ++a) {
            Frag<T> f;
            frag_load(f, x + (bj * alpha + a) * rowlen + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += frag_get(f, e);
        }
        Frag<T> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) frag_set(o, e, acc[e]);
        frag_store(o, out + bj * rowlen + off);
    }
}
// out[b][c] = mean_j feat[(b*t + j)*L + 0][c]
template <typename T>
__global__ __launch_bounds__(NT) void mean_cls_kernel(const T* __restrict__ feat, T* __restrict__ out, int b, int t, int L, int C) {
    const long total = (long)b * C;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % C);
        const long bi = i / C;
        float acc = 0.f;
        for (int j = 0; j < t; ++j) acc += to_f(feat[((bi * t + j) * (long)L) * C + c]);
        out[i] = from_f<T>(acc / (float)t);
    }
}
// broadcast rows: out[r][c] = table[c]  (fp32 param -> T rows)
template <typename T>
__global__ __launch_bounds__(NT) void bcast_rows_kernel(const float* __restrict__ table, T* __restrict__ out, long rows, int C) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) out[i] = from_f<T>(table[i % C]);
}
// dtable[c] += sum_rows d[r][c]  -> use colsum

// ---- cosine logits + soft-target CE, forward and backward -------------------------------------
// one block per clip.  v [E] -> vn = v/|v|; logits[k] = exp(ls) * vn . (text_k/|text_k|)
// loss += -sum_k y_k log_softmax(logits)_k / b ; dlogits = (softmax * sum(y) - y)/b (or given)
// dv = (dvn - vn (vn.dvn)) / |v| with dvn = exp(ls) * sum_k dlogits_k tn_k ; dls += sum_k dlogits_k logits_k
template <typename T>
__global__ __launch_bounds__(NT) void logits_loss_kernel(const T* __restrict__ v, const float* __restrict__ text, const float* __restrict__ logit_scale,
                                                         const float* __restrict__ soft_target, float* __restrict__ logits, float* __restrict__ vid_norm,
                                                         float* __restrict__ loss, T* __restrict__ dv, float* __restrict__ dlogit_scale,
                                                         const float* __restrict__ dlogits_in, float* __restrict__ dlogits_out,
                                                         int b, int E, int K) {
    extern __shared__ float sm[];     // vn[E], lg[K], dl[K], red[NT]
    float* vn = sm; float* lg = vn + E; float* dl = lg + K; float* red = dl + K;
    const int bi = blockIdx.x, tid = threadIdx.x;
    auto block_sum = [&](float x) {
        red[tid] = x; __syncthreads();
        for (int o = NT / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        const float r = red[0]; __syncthreads();
        return r;
    };
    auto block_max = [&](float x) {
        red[tid] = x; __syncthreads();
        for (int o = NT / 2; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
        const float r = red[0]; __syncthreads();
        return r;
    };
    float ss = 0.f;
    for (int e = tid; e < E; e += NT) { const float x = to_f(v[(long)bi * E + e]); vn[e] = x; ss += x * x; }
    const float nrm = sqrtf(block_sum(ss));
    const float inv = 1.f / nrm;
    for (int e = tid; e < E; e += NT) { vn[e] *= inv; if (vid_norm) vid_norm[(long)bi * E + e] = vn[e]; }
    __syncthreads();
    const float sc = __expf(logit_scale[0]);
    for (int k = tid; k < K; k += NT) {
        const float* tk = text + (long)k * E;
        float dot = 0.f, tt = 0.f;
        for (int e = 0; e < E; ++e) { dot += vn[e] * tk[e]; tt += tk[e] * tk[e]; }
        const float l = sc * dot * rsqrtf(tt);
        lg[k] = l;
        if (logits) logits[(long)bi * K + k] = l;
    }
    __syncthreads();
    if (!dv && !soft_target) return;
    if (dlogits_in) {
        for (int k = tid; k < K; k += NT) dl[k] = dlogits_in[(long)bi * K + k];
    } else {
        float mx = -1e30f;
        for (int k = tid; k < K; k += NT) mx = fmaxf(mx, lg[k]);
        mx = block_max(mx);
        float se = 0.f, sy = 0.f, syl = 0.f;
        for (int k = tid; k < K; k += NT) {
            const float y = soft_target[(long)bi * K + k];
            se += __expf(lg[k] - mx); sy += y; syl += y * lg[k];
        }
        se = block_sum(se); sy = block_sum(sy); syl = block_sum(syl);
        const float lse = mx + logf(se);
        if (tid == 0 && loss) atomicAdd(loss, (sy * lse - syl) / (float)b);
        for (int k = tid; k < K; k += NT) {
            const float y = soft_target[(long)bi * K + k];
            dl[k] = (__expf(lg[k] - lse) * sy - y) / (float)b;
            if (dlogits_out) dlogits_out[(long)bi * K + k] = dl[k];
        }
    }
    if (!dv) return;
    __syncthreads();
    float dls = 0.f;
    for (int k = tid; k < K; k += NT) dls += dl[k] * lg[k];
    dls = block_sum(dls);
    if (tid == 0 && dlogit_scale) atomicAdd(dlogit_scale, dls);
    // dvn[e] = sc * sum_k dl[k] * tn[k][e]; needs |text_k| again: fold 1/|t_k| into dl
    for (int k = tid; k < K; k += NT) {
        const float* tk = text + (long)k * E;
        float tt = 0.f;
        for (int e = 0; e < E; ++e) tt += tk[e] * tk[e];
        dl[k] *= sc * rsqrtf(tt);
    }
    __syncthreads();
    float proj = 0.f;
    float dvn_loc[4];            // E <= 4*NT
    int cnt = 0;
    for (int e = tid; e < E; e += NT, ++cnt) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += dl[k] * text[(long)k * E + e];
        dvn_loc[cnt] = acc;
        proj += acc * vn[e];
    }
    proj = block_sum(proj);
    cnt = 0;
    for (int e = tid; e < E; e += NT, ++cnt) dv[(long)bi * E + e] = from_f<T>((dvn_loc[cnt] - vn[e] * proj) * inv);
}

// ---- fused multi-tensor AdamW (torch.optim.AdamW single-tensor math) ---------------------------
DEV void adamw_one(float& pp, float gr, float& mm, float& vv, const dist_adamw_seg& sg, float b1, float b2, float eps, float bc1, float bc2s) {
    pp *= (1.f - sg.lr * sg.weight_decay);
    mm = b1 * mm + (1.f - b1) * gr;
    vv = b2 * vv + (1.f - b2) * gr * gr;
    const float denom = sqrtf(vv) / bc2s + eps;
    pp -= (sg.lr / bc1) * mm / denom;
}
constexpr int ADAMW_CHUNK = NT * 4 * 4;                   // elements per block: 4 float4 per thread
__global__ __launch_bounds__(NT) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   const dist_adamw_seg* __restrict__ segs, int nseg, long n,
                                                   float b1, float b2, float eps, float bc1, float bc2s, float gscale) {
    // a block owns ADAMW_CHUNK consecutive elements; ONE binary search per block (segment of its first element), every
    // thread then walks forward from there (segments are sorted, a chunk touches one or two of them) and works on
    // float4s: the per-element search of the first version cost nine dependent loads per 4-byte element
    __shared__ int s_first;
    const long c0 = (long)blockIdx.x * ADAMW_CHUNK;
    if (threadIdx.x == 0) {
        int lo = 0, hi = nseg - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (segs[mid].end <= c0) lo = mid + 1; else hi = mid; }
        s_first = lo;
    }
    __syncthreads();
    int si = s_first;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long i = c0 + ((long)k * NT + threadIdx.x) * 4;
        if (i >= n) break;
        while (si < nseg - 1 && segs[si].end <= i) ++si;
        const dist_adamw_seg sg = segs[si];
        if (i >= sg.begin && i + 4 <= sg.end && i + 4 <= n) {
            float4 pp = *reinterpret_cast<float4*>(p + i), mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            adamw_one(pp.x, gg.x * gscale, mm.x, vv.x, sg, b1, b2, eps, bc1, bc2s);
            adamw_one(pp.y, gg.y * gscale, mm.y, vv.y, sg, b1, b2, eps, bc1, bc2s);
            adamw_one(pp.z, gg.z * gscale, mm.z, vv.z, sg, b1, b2, eps, bc1, bc2s);
            adamw_one(pp.w, gg.w * gscale, mm.w, vv.w, sg, b1, b2, eps, bc1, bc2s);
            *reinterpret_cast<float4*>(p + i) = pp; *reinterpret_cast<float4*>(m + i) = mm; *reinterpret_cast<float4*>(v + i) = vv;
        } else {                                            // a float4 that straddles a segment border (or a gap between segments)
            int sj = si;
            for (long e = i; e < i + 4 && e < n; ++e) {
                while (sj < nseg - 1 && segs[sj].end <= e) ++sj;
                const dist_adamw_seg se = segs[sj];
                if (e < se.begin || e >= se.end) continue;
                float pp = p[e], mm = m[e], vv = v[e];
                adamw_one(pp, g[e] * gscale, mm, vv, se, b1, b2, eps, bc1, bc2s);
                p[e] = pp; m[e] = mm; v[e] = vv;
            }
        }
    }
}

// ---- weight packing -------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void pack_kernel(const PackDesc* __restrict__ descs, const int* __restrict__ blk_desc, const int* __restrict__ blk_first,
                                                  int first_block, const float* __restrict__ src0, const float* __restrict__ src1, T* __restrict__ dst_base) {
    const int blk = blockIdx.x + first_block;
    const int di = blk_desc[blk];
    const PackDesc d = descs[di];
    const float* src = (d.src_kind ? src1 : src0) + d.src_off;
    T* dst = dst_base + d.dst_off;
    const long total = (long)d.rows * d.cols;
    if (d.layout == PACK_B) {
        // transposed layout dst[ci][tap*co + co] <- src[co][..ci..]: a block owns whole destination rows; the gather runs
        // with consecutive lanes on consecutive ci (contiguous in the source for Linear weights), the tile is turned in
        // LDS (odd dword row pitch: conflict-free column writes) and written out as whole rows.  (Element-wise, every
        // lane read 4 bytes of a different source row: 16x the bytes through L2.)
        extern __shared__ __attribute__((aligned(16))) char pk_smem[];
        T* tile = reinterpret_cast<T*>(pk_smem);
        const int rpb = pack_b_rows(d.cols);
        const int r0 = (blk - blk_first[di]) * rpb, nr = min(rpb, d.rows - r0);
        const int pitch = d.cols + (sizeof(T) == 2 ? 2 : 1);
        const bool fits = (long)rpb * pitch <= PACK_B_ELEMS + 2 * rpb;
        if (fits) {
            // (indices stepped by NT elements per iteration, one division per thread and loop: see the element loop below)
            const bool one_tap = d.cols <= d.co;
            {
                const int dq = NT / nr, dm = NT - dq * nr;
                int c = threadIdx.x / nr, cl = threadIdx.x - c * nr;
                for (int e = threadIdx.x; e < nr * d.cols; e += NT) {
                    const int ci = r0 + cl, tap = one_tap ? 0 : c / d.co, co = c - tap * d.co;
                    float v = 0.f;
                    if (ci < d.kin) {
                        const int og = d.inner == 1 ? ci : ci / d.inner;
                        v = src[(long)co * d.s_co + (long)tap * d.s_tap + (d.inner == 1 ? (long)ci * d.s_outer : (long)og * d.s_outer + (ci - og * d.inner))];
                    }
                    tile[cl * pitch + c] = from_f<T>(v);
                    c += dq; cl += dm;
                    if (cl >= nr) { cl -= nr; ++c; }
                }
            }
            __syncthreads();
            {
                const int dq = NT / d.cols, dm = NT - dq * d.cols;
                int cl = threadIdx.x / d.cols, c = threadIdx.x - cl * d.cols;
                for (int e = threadIdx.x; e < nr * d.cols; e += NT) {
                    dst[(long)(r0 + cl) * d.cols + c] = tile[cl * pitch + c];
                    cl += dq; c += dm;
                    if (c >= d.cols) { c -= d.cols; ++cl; }
                }
            }
            return;
        }
        for (long i = (long)r0 * d.cols + threadIdx.x; i < (long)(r0 + nr) * d.cols; i += NT) {     // very wide rows: direct
            const int ci = (int)(i / d.cols), c = (int)(i % d.cols), tap = c / d.co, co = c % d.co;
            float v = 0.f;
            if (ci < d.kin) v = src[(long)co * d.s_co + (long)tap * d.s_tap + (long)(ci / d.inner) * d.s_outer + (ci % d.inner)];
            dst[i] = from_f<T>(v);
        }
        return;
    }
    // (row, column) of a thread's first element by one division, then stepped by NT elements per iteration; taps and `inner` groups
    // only divide where a weight has them (per element: a 64-bit and up to four 32-bit divisions made this pass VALU-bound, 6x its bytes)
    const unsigned beg = (unsigned)(blk - blk_first[di]) * PACK_PER_BLOCK;
    const unsigned end = (unsigned)min(total, (long)beg + PACK_PER_BLOCK);
    const unsigned cols = (unsigned)d.cols, dr = NT / cols, dc = NT - dr * cols;
    unsigned i = beg + threadIdx.x;
    unsigned r = i / cols, c = i - r * cols;
    const bool one_tap = d.layout == PACK_F ? d.cols <= d.kpad : d.rows <= d.kpad;
    for (; i < end; i += NT) {
        int co, tap, ci;
        if (d.layout == PACK_F) { co = (int)r; tap = one_tap ? 0 : (int)(c / (unsigned)d.kpad); ci = (int)c - tap * d.kpad; }
        else { tap = one_tap ? 0 : (int)(r / (unsigned)d.kpad); ci = (int)r - tap * d.kpad; co = (int)c; }      // PACK_FT
        float v = 0.f;
        if (ci < d.kin) {
            const int og = d.inner == 1 ? ci : ci / d.inner;
            v = src[(long)co * d.s_co + (long)tap * d.s_tap + (d.inner == 1 ? (long)ci * d.s_outer : (long)og * d.s_outer + (ci - og * d.inner))];
        }
        dst[d.dpitch ? (long)r * d.dpitch + c : (long)i] = from_f<T>(v);
        r += dr; c += dc;
        if (c >= cols) { c -= cols; ++r; }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
extern "C" int dist_op_patchify(const float* video, void* patches, int b, int T, int H, int W, int P, int dtype, void* stream) {
    if (!video || !patches || b <= 0 || T <= 0 || H % P || W % P) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Kp = (3 * P * P + 7) / 8 * 8;
    const long total = (long)b * T * (H / P) * (W / P) * Kp;
    const size_t band = (size_t)3 * P * W * (dtype == DIST_BF16 ? 2 : 4);
    if (P % 8 == 0 && W % 4 == 0 && band <= 64 * 1024 && (3 * P * P) % 8 == 0) {
        const unsigned grid = (unsigned)((long)b * T * (H / P));
        if (dtype == DIST_BF16) hipLaunchKernelGGL(patchify_band_kernel<bf16_t>, dim3(grid), dim3(NT), band, s, video, (bf16_t*)patches, T, H, W, P, Kp);
        else hipLaunchKernelGGL(patchify_band_kernel<float>, dim3(grid), dim3(NT), band, s, video, (float*)patches, T, H, W, P, Kp);
        HIP_CHECK_RET(hipGetLastError());
        return DIST_OK;
    }
    if (dtype == DIST_BF16) hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid1d(total, NT * 8)), dim3(NT), 0, s, video, (bf16_t*)patches, b, T, H, W, P, Kp);
    else hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid1d(total, NT * 8)), dim3(NT), 0, s, video, (float*)patches, b, T, H, W, P, Kp);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

// frozen-ViT features handed over by the caller in the reference's layout [L][b*t][C] (sequence first, clip.py:282-300) -> the engine's token rows
// [(b*t)*L][C] in its storage type: one wave per destination row, 8 elements per lane and step.
namespace {
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void import_feat_kernel(const TI* __restrict__ src, TO* __restrict__ dst, const int bt, const int L, const int C) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)bt * L) return;
    const int bj = (int)(row / L), l = (int)(row - (long)bj * L);
    const TI* s = src + ((long)l * bt + bj) * C;
    TO* d = dst + row * C;
    for (int c = (threadIdx.x & 63) * 8; c < C; c += 512) {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (c + e < C) d[c + e] = from_f<TO>(to_f(s[c + e]));
    }
}
}  // namespace
int dist_k_import_feat(const void* src, int src_dtype, void* dst, int dst_dtype, int bt, int L, int C, hipStream_t s) {
    const dim3 grid((unsigned)(((long)bt * L + 3) / 4));
    if (src_dtype == DIST_F32 && dst_dtype == DIST_F32) hipLaunchKernelGGL((import_feat_kernel<float, float>), grid, dim3(256), 0, s, (const float*)src, (float*)dst, bt, L, C);
    else if (src_dtype == DIST_F32) hipLaunchKernelGGL((import_feat_kernel<float, bf16_t>), grid, dim3(256), 0, s, (const float*)src, (bf16_t*)dst, bt, L, C);
    else if (dst_dtype == DIST_F32) hipLaunchKernelGGL((import_feat_kernel<bf16_t, float>), grid, dim3(256), 0, s, (const bf16_t*)src, (float*)dst, bt, L, C);
    else hipLaunchKernelGGL((import_feat_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, bt, L, C);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
    if (!a || !b || !out || n <= 0 || n % 8) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == DIST_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid1d(n / 8, NT)), dim3(NT), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, (long)(n / 8));
    else hipLaunchKernelGGL(add_kernel<float>, dim3(grid1d(n / 8, NT)), dim3(NT), 0, s, (const float*)a, (const float*)b, (float*)out, (long)(n / 8));
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int dtype, void* stream) {
    if (!dy || !pre || !dx || n <= 0 || n % 8) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == DIST_BF16) hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(grid1d(n / 8, NT)), dim3(NT), 0, s, (const bf16_t*)dy, (const bf16_t*)pre, (bf16_t*)dx, (long)(n / 8));
    else hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(grid1d(n / 8, NT)), dim3(NT), 0, s, (const float*)dy, (const float*)pre, (float*)dx, (long)(n / 8));
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_colsum(const void* x, float* out, int64_t rows, int C, int ld, dist_rowmap map, int dtype, void* stream) {
    if (!x || !out || rows <= 0 || C % 4 || ld % 4) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rpb = 512;
    long grid = (rows + rpb - 1) / rpb;
    if (grid > 1024) { rpb = (int)((rows + 1023) / 1024); grid = (rows + rpb - 1) / rpb; }
    if (dtype == DIST_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3((unsigned)grid, (C + 1023) / 1024), dim3(NT), 0, s, (const bf16_t*)x, out, (long)rows, C, ld, map, rpb);
    else hipLaunchKernelGGL(colsum_kernel<float>, dim3((unsigned)grid, (C + 1023) / 1024), dim3(NT), 0, s, (const float*)x, out, (long)rows, C, ld, map, rpb);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_logits_loss(const void* v, const float* text, const float* logit_scale, const float* soft_target,
                                   float* logits, float* vid_norm, float* loss, void* dv, float* dlogit_scale,
                                   const float* dlogits_in, int b, int E, int K, int dtype, void* stream) {
    return dist_k_logits_loss(v, text, logit_scale, soft_target, logits, vid_norm, loss, dv, dlogit_scale, dlogits_in, nullptr, b, E, K, dtype, stream);
}

int dist_k_logits_loss(const void* v, const float* text, const float* logit_scale, const float* soft_target,
                       float* logits, float* vid_norm, float* loss, void* dv, float* dlogit_scale,
                       const float* dlogits_in, float* dlogits_out, int b, int E, int K, int dtype, void* stream) {
    if (!v || !text || !logit_scale || b <= 0 || E <= 0 || K <= 0 || E > 4 * NT) return DIST_ERR_ARG;
    if (dv && !dlogits_in && !soft_target) return DIST_ERR_ARG;
    if (dlogits_in && soft_target) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t smem = (size_t)(E + 2 * K + NT) * sizeof(float);
    if (dtype == DIST_BF16)
        hipLaunchKernelGGL(logits_loss_kernel<bf16_t>, dim3(b), dim3(NT), smem, s, (const bf16_t*)v, text, logit_scale, soft_target, logits, vid_norm,
                           loss, (bf16_t*)dv, dlogit_scale, dlogits_in, dlogits_out, b, E, K);
    else
        hipLaunchKernelGGL(logits_loss_kernel<float>, dim3(b), dim3(NT), smem, s, (const float*)v, text, logit_scale, soft_target, logits, vid_norm,
                           loss, (float*)dv, dlogit_scale, dlogits_in, dlogits_out, b, E, K);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_adamw(float* param, const float* grad, float* m, float* v, const dist_adamw_seg* segs_dev, int nseg,
                             int64_t n, float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
    if (!param || !grad || !m || !v || !segs_dev || nseg <= 0 || n <= 0 || step <= 0) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + ADAMW_CHUNK - 1) / ADAMW_CHUNK)), dim3(NT), 0, s, param, grad, m, v, segs_dev, nseg, (long)n, beta1, beta2, eps, bc1, bc2s, grad_scale);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

// ---- batch-mode Mixup / CutMix and the soft target (reference dataset/utils/mixup.py:18-23,212-223) ------------------
// The reference mixes the rank's batch in place on the GPU, clip i with clip b-1-i:
//   mixup  (:221-222): x_flipped = x.flip(0).mul_(1 - lam); x.mul_(lam).add_(x_flipped)   -> three fp32 roundings per element
//   cutmix (:219):     x[..., yl:yh, xl:xh] = x.flip(0)[..., yl:yh, xl:xh]                 -> the box is SWAPPED between i and b-1-i
// One thread owns element e of BOTH clips of a pair, so the in-place update needs no copy of the batch (the reference's
// flip(0) materialises 308 MB at b = 32).  Contraction is switched off in these kernels, so every product and the sum round
// to fp32 separately as in torch: the result is bit-identical to the reference.
namespace {
constexpr int MIX_NT = 256;
// a*s + c*t with the three fp32 roundings of torch's mul_, mul_, add_ (hipcc contracts a*b + c into one fma by default, and the
// __fmul_rn / __fadd_rn header inlines carry their own `contract` flag, so plain operators under contract(off) it is)
__device__ __forceinline__ float mix_rn(float a, float s, float c, float t) {
#pragma clang fp contract(off)
    const float p = a * s;
    const float q = c * t;
    return p + q;
}
__global__ __launch_bounds__(MIX_NT) void mixup_kernel(float* __restrict__ x, const int b, const long per_clip4, const float lam, const float oml) {
    const int i = blockIdx.y, j = b - 1 - i;
    float4* xi = reinterpret_cast<float4*>(x) + (long)i * per_clip4;
    float4* xj = reinterpret_cast<float4*>(x) + (long)j * per_clip4;
    for (long e = (long)blockIdx.x * MIX_NT + threadIdx.x; e < per_clip4; e += (long)gridDim.x * MIX_NT) {
        const float4 a = xi[e], c = xj[e];
        float4 oi, oj;
        oi.x = mix_rn(a.x, lam, c.x, oml); oj.x = mix_rn(c.x, lam, a.x, oml);
        oi.y = mix_rn(a.y, lam, c.y, oml); oj.y = mix_rn(c.y, lam, a.y, oml);
        oi.z = mix_rn(a.z, lam, c.z, oml); oj.z = mix_rn(c.z, lam, a.z, oml);
        oi.w = mix_rn(a.w, lam, c.w, oml); oj.w = mix_rn(c.w, lam, a.w, oml);
        xi[e] = oi;
        if (i != j) xj[e] = oj;                       // odd batch: the middle clip is mixed with itself
    }
}
// one block per (pair, plane, box row): threads run along the box row
__global__ __launch_bounds__(MIX_NT) void cutmix_kernel(float* __restrict__ x, const int b, const int planes, const int H, const int W,
                                                        const int yl, const int yh, const int xl, const int xh) {
    const int i = blockIdx.z, j = b - 1 - i;
    if (i == j) return;
    const int pl = blockIdx.y, y = yl + blockIdx.x;
    float* ri = x + (((long)i * planes + pl) * H + y) * W;
    float* rj = x + (((long)j * planes + pl) * H + y) * W;
    for (int xx = xl + threadIdx.x; xx < xh; xx += MIX_NT) {
        const float a = ri[xx], c = rj[xx];
        ri[xx] = c; rj[xx] = a;
    }
}
// mixup_target (:18-23): y1 * lam + y2 * (1 - lam), y1 / y2 = smoothed one-hot rows of target / target.flip(0)
__global__ __launch_bounds__(MIX_NT) void mixup_target_kernel(const long* __restrict__ labels, const int b, const int K, const float lam, const float oml,
                                                              const float on, const float off, float* __restrict__ soft) {
    const long n = (long)b * K;
    for (long e = (long)blockIdx.x * MIX_NT + threadIdx.x; e < n; e += (long)gridDim.x * MIX_NT) {
        const int i = (int)(e / K), k = (int)(e - (long)i * K);
        const float y1 = labels[i] == k ? on : off, y2 = labels[b - 1 - i] == k ? on : off;
        soft[e] = mix_rn(y1, lam, y2, oml);
    }
}

// ---- patch rows of the MIXED batch (round 6): what dist_op_patchify would gather after dist_op_mixup / dist_op_cutmix, without the 308 MB round trip of the
// mixed fp32 frames.  The same arithmetic (mix_rn: three fp32 roundings, then the storage-type rounding), `video` is only read.
struct MixP { int kind; float lam, oml; int yl, yh, xl, xh; };     // kind 1 = mixup, 2 = cutmix
// one block per (PAIR i <= b-1-i, frame, patch row): the two clips' bands are read once, mixed in registers, parked in LDS in the output type and written as
// whole patch rows of BOTH clips (patchify_band_kernel's layout)
template <typename T>
__global__ __launch_bounds__(NT) void patchify_band_mix_kernel(const float* __restrict__ video, T* __restrict__ out, int b, int Tn, int H, int W, int P, int Kp, const MixP m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* band0 = reinterpret_cast<T*>(smem);                 // [3][P][W] of clip i
    T* band1 = band0 + 3 * P * W;                          // ... of clip j = b - 1 - i
    const int G = W / P, Gy = H / P;
    const int gy = blockIdx.x % Gy;
    const long pk = blockIdx.x / Gy;
    const int k = (int)(pk % Tn), i = (int)(pk / Tn), j = b - 1 - i;
    const int W4 = W / 4, per_c = P * W4;
    const long clip = (long)3 * Tn * H * W;
    // four float4 pairs per thread and trip: all eight loads are in flight before the first mix (one load pair per trip left the 86 KB of a block's two bands
    // latency-bound: 125 us per launch against 76 for the plain gather of the same 462 MB)
    constexpr int U = 4;
    for (int v0 = threadIdx.x; v0 < 3 * per_c; v0 += U * NT) {
        float4 a[U], q[U];
        int cc[U], pyy[U], xx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = min(v0 + u * NT, 3 * per_c - 1);
            const int c = v / per_c, rem = v - c * per_c, py = rem / W4, x4 = rem - py * W4;
            cc[u] = c; pyy[u] = py; xx[u] = x4;
            const long off = ((long)c * Tn + k) * H * W + (long)(gy * P + py) * W + x4 * 4;
            a[u] = *reinterpret_cast<const float4*>(video + (long)i * clip + off);
            q[u] = i == j ? a[u] : *reinterpret_cast<const float4*>(video + (long)j * clip + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (v0 + u * NT >= 3 * per_c) continue;
            const int y = gy * P + pyy[u];
            float oi[4], oj[4];
            const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, qv[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
            if (m.kind == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { oi[e] = mix_rn(av[e], m.lam, qv[e], m.oml); oj[e] = mix_rn(qv[e], m.lam, av[e], m.oml); }
            } else {                                       // cutmix: the box is swapped between the two clips (the middle clip of an odd batch keeps its frames)
                const bool rowin = y >= m.yl && y < m.yh && i != j;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int x = xx[u] * 4 + e;
                    const bool in = rowin && x >= m.xl && x < m.xh;
                    oi[e] = in ? qv[e] : av[e]; oj[e] = in ? av[e] : qv[e];
                }
            }
            T* d0 = band0 + (cc[u] * P + pyy[u]) * W + xx[u] * 4;
            T* d1 = band1 + (cc[u] * P + pyy[u]) * W + xx[u] * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { d0[e] = from_f<T>(oi[e]); d1[e] = from_f<T>(oj[e]); }
        }
    }
    __syncthreads();
    const int vpr = Kp / 8, vp = P / 8;
    for (int which = 0; which < (i == j ? 1 : 2); ++which) {
        const T* band = which ? band1 : band0;
        const long row0 = (((long)(which ? j : i) * Tn + k) * Gy + gy) * G;
        for (int v = threadIdx.x; v < G * vpr; v += NT) {
            const int gx = v / vpr, jj = v - gx * vpr;
            Frag<T> o;
            if (jj * 8 < 3 * P * P) {
                const int c = jj / (P * vp), rem = jj - c * (P * vp), py = rem / vp, h = rem - py * vp;
                frag_load(o, band + (c * P + py) * W + gx * P + h * 8);
            } else {
                frag_zero(o);
            }
            frag_store_nt(o, out + (row0 + gx) * Kp + jj * 8);
        }
    }
}
// any patch size (ViT-L/14): element-wise, every output element mixes its two source pixels itself
template <typename T>
__global__ __launch_bounds__(NT) void patchify_mix_kernel(const float* __restrict__ video, T* __restrict__ out, int b, int Tn, int H, int W, int P, int Kp, const MixP m) {
    const int G = W / P, Gy = H / P, N = G * Gy, PP = P * P;
    const long total = (long)b * Tn * N * Kp;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int col = (int)(i % Kp);
        const long row = i / Kp;
        float v = 0.f;
        if (col < 3 * PP) {
            const int c = col / PP, py = (col % PP) / P, px = col % P;
            const int n = (int)(row % N);
            const long bk = row / N;
            const int k = (int)(bk % Tn), bi = (int)(bk / Tn), bj = b - 1 - bi;
            const int y = (n / G) * P + py, x = (n % G) * P + px;
            const long off = ((long)c * Tn + k) * H * W + (long)y * W + x;
            const float a = video[(long)bi * 3 * Tn * H * W + off], q = video[(long)bj * 3 * Tn * H * W + off];
            if (m.kind == 1) v = mix_rn(a, m.lam, q, m.oml);
            else v = (bi != bj && y >= m.yl && y < m.yh && x >= m.xl && x < m.xh) ? q : a;
        }
        out[i] = from_f<T>(v);
    }
}
}  // namespace

extern "C" int dist_op_mixup(float* video, int b, int64_t per_clip, float lam, float one_minus_lam, void* stream) {
    if (!video || b <= 0 || per_clip <= 0 || per_clip % 4 || (reinterpret_cast<uintptr_t>(video) & 15)) return DIST_ERR_ARG;
    const long n4 = per_clip / 4;
    const unsigned gx = (unsigned)min((n4 + MIX_NT - 1) / MIX_NT, (long)4096);
    hipLaunchKernelGGL(mixup_kernel, dim3(gx, (unsigned)((b + 1) / 2)), dim3(MIX_NT), 0, static_cast<hipStream_t>(stream), video, b, n4, lam, one_minus_lam);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}
extern "C" int dist_op_cutmix(float* video, int b, int planes, int H, int W, int yl, int yh, int xl, int xh, void* stream) {
    if (!video || b <= 0 || planes <= 0 || H <= 0 || W <= 0 || yl < 0 || yh > H || xl < 0 || xh > W || yl > yh || xl > xh) return DIST_ERR_ARG;
    if (yl == yh || xl == xh || b < 2) return DIST_OK;          // empty box: nothing moves (the reference's slice assignment is a no-op)
    hipLaunchKernelGGL(cutmix_kernel, dim3((unsigned)(yh - yl), (unsigned)planes, (unsigned)(b / 2)), dim3(MIX_NT), 0, static_cast<hipStream_t>(stream),
                       video, b, planes, H, W, yl, yh, xl, xh);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}
extern "C" int dist_op_patchify_mixed(const float* video, void* patches, int b, int T, int H, int W, int P, int dtype, int kind, float lam, float one_minus_lam,
                                      int yl, int yh, int xl, int xh, void* stream) {
    if (kind == 0) return dist_op_patchify(video, patches, b, T, H, W, P, dtype, stream);
    if (!video || !patches || b <= 0 || T <= 0 || P <= 0 || H % P || W % P || (kind != 1 && kind != 2)) return DIST_ERR_ARG;
    if (kind == 2 && (yl < 0 || yh > H || xl < 0 || xh > W || yl > yh || xl > xh)) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const MixP m{kind, lam, one_minus_lam, yl, yh, xl, xh};
    const int Kp = (3 * P * P + 7) / 8 * 8;
    const size_t band2 = (size_t)2 * 3 * P * W * (dtype == DIST_BF16 ? 2 : 4);
    if (P % 8 == 0 && W % 4 == 0 && band2 <= 64 * 1024 && (3 * P * P) % 8 == 0 && (reinterpret_cast<uintptr_t>(video) & 15) == 0) {
        const unsigned grid = (unsigned)((long)((b + 1) / 2) * T * (H / P));
        if (dtype == DIST_BF16) hipLaunchKernelGGL(patchify_band_mix_kernel<bf16_t>, dim3(grid), dim3(NT), band2, s, video, (bf16_t*)patches, b, T, H, W, P, Kp, m);
        else hipLaunchKernelGGL(patchify_band_mix_kernel<float>, dim3(grid), dim3(NT), band2, s, video, (float*)patches, b, T, H, W, P, Kp, m);
        HIP_CHECK_RET(hipGetLastError());
        return DIST_OK;
    }
    const long total = (long)b * T * (H / P) * (W / P) * Kp;
    if (dtype == DIST_BF16) hipLaunchKernelGGL(patchify_mix_kernel<bf16_t>, dim3(grid1d(total, NT * 8)), dim3(NT), 0, s, video, (bf16_t*)patches, b, T, H, W, P, Kp, m);
    else hipLaunchKernelGGL(patchify_mix_kernel<float>, dim3(grid1d(total, NT * 8)), dim3(NT), 0, s, video, (float*)patches, b, T, H, W, P, Kp, m);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}
extern "C" int dist_op_mixup_target(const int64_t* labels, int b, int K, float lam, float one_minus_lam, float on_value, float off_value, float* soft, void* stream) {
    if (!labels || !soft || b <= 0 || K <= 0) return DIST_ERR_ARG;
    static_assert(sizeof(long) == sizeof(int64_t), "LP64");
    const long n = (long)b * K;
    hipLaunchKernelGGL(mixup_target_kernel, dim3((unsigned)min((n + MIX_NT - 1) / MIX_NT, (long)1024)), dim3(MIX_NT), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long*>(labels), b, K, lam, one_minus_lam, on_value, off_value, soft);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

// ---- engine-internal launchers (kernels.h) ------------------------------------------------------------
namespace {
template <typename F> int with_type(int dtype, F&& f) { return dtype == DIST_BF16 ? f(bf16_t{}) : f(float{}); }
}

int dist_k_pack(const PackDesc* descs_dev, const int* blk_desc_dev, const int* blk_first_dev, int first_block, int nblocks,
                const float* theta, const float* visual, void* dst_base, int dtype, hipStream_t s) {
    return with_type(dtype, [&](auto tag) {
        using T = decltype(tag);
        const size_t smem = (size_t)(PACK_B_ELEMS + 2 * PACK_B_ELEMS / 8 + 64) * sizeof(T);   // rows * (cols + pad), cols >= 8
        hipLaunchKernelGGL(pack_kernel<T>, dim3(nblocks), dim3(NT), smem, s, descs_dev, blk_desc_dev, blk_first_dev, first_block, theta, visual, (T*)dst_base);
        HIP_CHECK_RET(hipGetLastError());
        return (int)DIST_OK;
    });
}
int dist_k_cls_rows(void* dst, const void* src, const float* table, int nbj, int L, int C, int period, int dtype, hipStream_t s) {
    return with_type(dtype, [&](auto tag) {
        using T = decltype(tag);
        hipLaunchKernelGGL(cls_rows_kernel<T>, dim3(grid1d((long)nbj * C, NT)), dim3(NT), 0, s, (T*)dst, (const T*)src, table, nbj, L, C, period);
        HIP_CHECK_RET(hipGetLastError());
        return (int)DIST_OK;
    });
}
int dist_k_cls_rows_bwd(const void* d, float* dtable, int nbj, int L, int C, int period, int dtype, hipStream_t s) {
    return with_type(dtype, [&](auto tag) {
        using T = decltype(tag);
        hipLaunchKernelGGL(cls_rows_bwd_kernel<T>, dim3(grid1d((long)period * C, NT), (unsigned)std::max(1, std::min(16, nbj / period / 2))), dim3(NT), 0, s, (const T*)d, dtable, nbj, L, C, period);
        HIP_CHECK_RET(hipGetLastError());
        return (int)DIST_OK;
    });
}
int dist_k_add_table(const void* x, const float* table, void* out, long rows, int C, int period, int dtype, hipStream_t s) {
    return with_type(dtype, [&](auto tag) {
        using T = decltype(tag);
        hipLaunchKernelGGL(add_table_kernel<T>, dim3(grid1d(rows * C, NT)), dim3(NT), 0, s, (const T*)x, table, (T*)out, rows, C, period);
        HIP_CHECK_RET(hipGetLastError());
        return (int)DIST_OK;
    });
}
int dist_k_pair_sum(const void* x, void* out, long nbj, int rowlen, int alpha, int dtype, hipStream_t s) {
    return with_type(dtype, [&](auto tag) {
        using T = decltype(tag);
        hipLaunchKernelGGL(pair_sum_kernel<T>, dim3(grid1d(nbj * rowlen / 8, NT)), dim3(NT), 0, s, (const T*)x, (T*)out, nbj, rowlen, alpha);
        HIP_CHECK_RET(hipGetLastError());
        return (int)DIST_OK;
    });
}
int dist_k_mean_cls(const void* feat, void* out, int b, int t, int L, int C, int dtype, hipStream_t s) {
    return with_type(dtype, [&](auto tag) {
        using T = decltype(tag);
        hipLaunchKernelGGL(mean_cls_kernel<T>, dim3(grid1d((long)b * C, NT)), dim3(NT), 0, s, (const T*)feat, (T*)out, b, t, L, C);
        HIP_CHECK_RET(hipGetLastError());
        return (int)DIST_OK;
    });
}
int dist_k_bcast_rows(const float* table, void* out, long rows, int C, int dtype, hipStream_t s) {
    return with_type(dtype, [&](auto tag) {
        using T = decltype(tag);
        hipLaunchKernelGGL(bcast_rows_kernel<T>, dim3(grid1d(rows * C, NT)), dim3(NT), 0, s, table, (T*)out, rows, C);
        HIP_CHECK_RET(hipGetLastError());
        return (int)DIST_OK;
    });
}

// ---- LayerNorm -> Linear fold (DIST_EPI_LNFOLD): one block per output row n ----------------------------------------
namespace {
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16_t* __restrict__ Wp, float* __restrict__ colsum,
                                                      float* __restrict__ bias_out, int K) {
    __shared__ float red[2][4];
    const int n = blockIdx.x, tid = threadIdx.x;
    float s = 0.f, b = 0.f;
    for (int k = tid; k < K; k += 256) {
        const float w = W[(long)n * K + k];
        const bf16_t wp = (bf16_t)(w * gamma[k]);
        Wp[(long)n * K + k] = wp;
        s += (float)wp;                                     // the sum of what the MFMA will actually multiply
        b += w * beta[k];
    }
    s = wave_sum(s, 64); b = wave_sum(b, 64);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = b; }
    __syncthreads();
    if (tid == 0) {
        colsum[n] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        bias_out[n] = (bias ? bias[n] : 0.f) + (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
}  // namespace

extern "C" int dist_op_ln_fold(const float* W, const float* bias, const float* gamma, const float* beta, void* Wp, float* colsum, float* bias_out,
                               int N, int K, void* stream) {
    if (!W || !gamma || !beta || !Wp || !colsum || !bias_out || N <= 0 || K <= 0) return DIST_ERR_ARG;
    hipLaunchKernelGGL(ln_fold_kernel, dim3(N), dim3(256), 0, static_cast<hipStream_t>(stream), W, bias, gamma, beta,
                       static_cast<bf16_t*>(Wp), colsum, bias_out, K);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}
