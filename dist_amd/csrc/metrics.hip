// Evaluation-side operators of the path: what runs on the predictions right behind the forward pass.
//   * dist_op_softmax_rows    - the head's eval activation (reference models/base/base_blocks.py:573-585: nn.Softmax(dim=-1) at eval)
//   * dist_op_topk_correct    - utils/metrics.py:100-129 topks_correct (call sites runs/train.py:167, utils/meters.py:160)
//   * dist_op_ensemble_update - utils/meters.py:82-112 TestMeter.update_stats (multi-view score ensemble, "sum" / "max")
// All three are tiny HBM-resident integer / fp32 passes; they exist so that the evaluation loop has no per-iteration host
// synchronisation (the reference moves predictions to the host and loops over clips in Python every iteration).
#include "common.h"

namespace {

constexpr int NT = 256;

__device__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = is_max ? wave_max(v, 64) : wave_sum(v, 64);
    __syncthreads();                                   // `red` may still be read from the previous reduction
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < NT / 64; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

// y[r][:] = exp(x[r][:] - max) / sum(exp(x[r][:] - max)), fp32 throughout (torch's softmax formula); one block per row
__global__ __launch_bounds__(NT) void softmax_rows_kernel(const float* __restrict__ x, const int K, float* __restrict__ y) {
    __shared__ float red[NT / 64];
    const float* xr = x + (long)blockIdx.x * K;
    float* yr = y + (long)blockIdx.x * K;
    float m = -INFINITY;
    for (int k = threadIdx.x; k < K; k += NT) m = fmaxf(m, xr[k]);
    m = block_reduce(m, red, true);
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += NT) s += expf(xr[k] - m);
    s = block_reduce(s, red, false);
    for (int k = threadIdx.x; k < K; k += NT) yr[k] = expf(xr[k] - m) / s;
}

// One wave per row.  rank(label) = #{j : p[j] > p[label]} + #{j < label : p[j] == p[label]}  (a stable descending sort; torch.topk
// leaves the order of exactly equal scores unspecified).  correct[i] += 1 when rank < ks[i]; a label outside [0, K) is never correct.
__global__ __launch_bounds__(NT) void topk_correct_kernel(const float* __restrict__ preds, const long* __restrict__ labels, const int n, const int K,
                                                          const int k0, const int k1, const int k2, const int k3, const int nk, float* __restrict__ correct) {
    const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const long lab = labels[row];
    if (lab < 0 || lab >= K) return;
    const float* p = preds + (long)row * K;
    const float s = p[lab];
    int c = 0;
    for (int j = lane; j < K; j += 64) {
        const float v = p[j];
        c += (v > s) || (v == s && j < lab);
    }
    c = (int)wave_sum((float)c, 64);                  // < 2^24: exact in fp32
    if (lane == 0) {
        const int ks[4] = {k0, k1, k2, k3};
        for (int i = 0; i < nk; ++i)
            if (c < ks[i]) atomicAdd(correct + i, 1.0f);   // integer-valued: the order of the additions does not matter
    }
}

// Thread c walks the batch's clips IN ORDER for class c (the reference's Python loop order, so "sum" rounds exactly as its
// video_preds[vid] += preds[ind]); thread 0 also keeps the labels and view counts.  err: bit 0 = two views of one video disagree on
// the label (the reference's assert, with its `label.sum() > 0` precondition), bit 1 = a clip id outside [0, V * num_clips).
__global__ __launch_bounds__(NT) void ensemble_update_kernel(float* __restrict__ video_preds, long* __restrict__ video_labels, long* __restrict__ clip_count,
                                                             const float* __restrict__ preds, const long* __restrict__ labels, const long* __restrict__ clip_ids,
                                                             const int n, const int K, const long V, const int num_clips, const int method, int* __restrict__ err) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= K) return;
    int e = 0;
    for (int i = 0; i < n; ++i) {
        const long id = clip_ids[i];
        const long vid = id / num_clips;
        if (id < 0 || vid >= V) { e |= 2; continue; }
        float* a = video_preds + vid * K + c;
        const float p = preds[(long)i * K + c];
        if (method == 0) *a += p;
        else { const float o = *a; *a = (p > o || p != p) ? p : o; }      // torch.max: NaN wins
        if (c == 0) {
            if (video_labels[vid] > 0 && video_labels[vid] != labels[i]) e |= 1;
            video_labels[vid] = labels[i];
            clip_count[vid] += 1;
        }
    }
    if (e) atomicOr(err, e);
}

}  // namespace

extern "C" int dist_op_softmax_rows(const float* x, int rows, int K, float* y, void* stream) {
    if (!x || !y || rows <= 0 || K <= 0) return DIST_ERR_ARG;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(NT), 0, static_cast<hipStream_t>(stream), x, K, y);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_topk_correct(const float* preds, const int64_t* labels, int n, int K, const int* ks, int nk, float* correct, void* stream) {
    if (!preds || !labels || !ks || !correct || n <= 0 || K <= 0 || nk <= 0 || nk > 4) return DIST_ERR_ARG;
    static_assert(sizeof(long) == sizeof(int64_t), "LP64");
    int k[4] = {0, 0, 0, 0};
    for (int i = 0; i < nk; ++i) {
        if (ks[i] <= 0) return DIST_ERR_ARG;
        k[i] = ks[i];
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_CHECK_RET(hipMemsetAsync(correct, 0, sizeof(float) * nk, s));
    hipLaunchKernelGGL(topk_correct_kernel, dim3((unsigned)((n + NT / 64 - 1) / (NT / 64))), dim3(NT), 0, s,
                       preds, reinterpret_cast<const long*>(labels), n, K, k[0], k[1], k[2], k[3], nk, correct);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_ensemble_update(float* video_preds, int64_t* video_labels, int64_t* clip_count, const float* preds, const int64_t* labels,
                                       const int64_t* clip_ids, int n, int K, int64_t num_videos, int num_clips, int method, int* err, void* stream) {
    if (!video_preds || !video_labels || !clip_count || !preds || !labels || !clip_ids || !err) return DIST_ERR_ARG;
    if (n <= 0 || K <= 0 || num_videos <= 0 || num_clips <= 0 || (method != DIST_ENSEMBLE_SUM && method != DIST_ENSEMBLE_MAX)) return DIST_ERR_ARG;
    hipLaunchKernelGGL(ensemble_update_kernel, dim3((unsigned)((K + NT - 1) / NT)), dim3(NT), 0, static_cast<hipStream_t>(stream),
                       video_preds, reinterpret_cast<long*>(video_labels), reinterpret_cast<long*>(clip_count), preds,
                       reinterpret_cast<const long*>(labels), reinterpret_cast<const long*>(clip_ids), n, K, (long)num_videos, num_clips, method, err);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}
