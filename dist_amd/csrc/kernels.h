// Engine-internal kernel launchers (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/dist_amd.h"

// weight packing: master fp32 tensors (reference layouts) -> GEMM-ready working copies
enum { PACK_F = 0, PACK_B = 1, PACK_FT = 2 };
constexpr int PACK_PER_BLOCK = 8192;
constexpr int PACK_B_ELEMS = 12288;   // PACK_B blocks own whole destination rows: max(1, PACK_B_ELEMS / cols) of them (LDS transpose)
__host__ __device__ inline int pack_b_rows(int cols) { const int r = PACK_B_ELEMS / cols; return r < 1 ? 1 : r; }
struct PackDesc {
    long src_off;        // elements into theta (src_kind 0) or visual (src_kind 1)
    long dst_off;        // elements into the packed buffer (after its header)
    int src_kind;
    int layout;          // PACK_F: [co][tap*kpad + ci] | PACK_B: [ci][tap*co_n + co] | PACK_FT: [tap*kpad + ci][co]
    int rows, cols;      // destination shape
    int co, kin, kpad;   // number of outputs, inputs per tap, padded inputs per tap
    int inner;           // source index = co*s_co + tap*s_tap + (ci/inner)*s_outer + ci%inner
    long s_co, s_tap, s_outer;
    int dpitch;          // destination row pitch in elements (0 = cols): two weights packed side by side into one [N][K1+K2] matrix
};

int dist_k_pack(const PackDesc* descs_dev, const int* blk_desc_dev, const int* blk_first_dev, int first_block, int nblocks,
                const float* theta, const float* visual, void* dst_base, int dtype, hipStream_t s);
int dist_k_cls_rows(void* dst, const void* src, const float* table, int nbj, int L, int C, int period, int dtype, hipStream_t s);
int dist_k_cls_rows_bwd(const void* d, float* dtable, int nbj, int L, int C, int period, int dtype, hipStream_t s);
int dist_k_add_table(const void* x, const float* table, void* out, long rows, int C, int period, int dtype, hipStream_t s);
int dist_k_pair_sum(const void* x, void* out, long nbj, int rowlen, int alpha, int dtype, hipStream_t s);
int dist_k_mean_cls(const void* feat, void* out, int b, int t, int L, int C, int dtype, hipStream_t s);
int dist_k_import_feat(const void* src, int src_dtype, void* dst, int dst_dtype, int bt, int L, int C, hipStream_t s);   // [L][bt][C] -> [(bt)*L][C]
int dist_k_bcast_rows(const float* table, void* out, long rows, int C, int dtype, hipStream_t s);
int dist_k_logits_loss(const void* v, const float* text, const float* logit_scale, const float* soft_target,
                       float* logits, float* vid_norm, float* loss, void* dv, float* dlogit_scale,
                       const float* dlogits_in, float* dlogits_out, int b, int E, int K, int dtype, void* stream);

// 256x256x32 LDS-DMA GEMM (gemm_fast.hip): 1 = launched, 0 = shape not eligible, <0 = error
bool dist_k_gemm_fast_eligible(const dist_gemm_args* a);
int dist_k_gemm_fast(const dist_gemm_args* a, hipStream_t s);
#ifdef DIST_AMD_MEASURE
// timing-only library: ping-pong persistent 256x256x64 GEMM (measure/gemm_pp.hip), tried first by dist_k_gemm_fast: 1 = launched, 0 = not its call, <0 = error
int dist_k_gemm_pp(const dist_gemm_args* a, int ngroups, hipStream_t s);
#endif
int dist_k_gemm_small(const dist_gemm_args* a, hipStream_t s);       // M < 1024 (gemm_small.hip)
// two-group LDS-DMA weight-gradient GEMM for the large plain gradients (gemm_tn8p.hip): 1 = launched, 0 = not its shape, <0 = error
int dist_k_gemm_tn8p(const dist_gemm_tn_args* a, hipStream_t s);
// frame-resident nine-tap weight gradient of the 3x3 frame convolution (conv_dw.hip): 1 = launched, 0 = not its call, <0 = error
int dist_k_conv3x3_dw(const dist_gemm_tn_args* a, hipStream_t s);
int dist_k_conv_t_dw(const dist_gemm_tn_args* a, hipStream_t s);        // conv_t_dw.hip: 1 launched, 0 not its call, < 0 error
// fused TemporalNet (tnet.hip): does dist_op_temporal_net_fwd take this geometry?
bool dist_k_tnet_fwd_eligible(int dtype, int Ct, int G, int tk);
// fused IntegrationNetwork forward (integ.hip): eligibility, and the all-layers form of dist_op_integration_pack (`descs_dev`: n descriptors
// written by dist_k_integ_pack_desc, in device memory)
bool dist_k_integ_eligible(int dtype, int Ci, int C4, int t, int tk);
int64_t dist_k_integ_pack_desc_bytes();
void dist_k_integ_pack_desc(const dist_integ_pack_args* a, void* out);
int dist_k_integ_pack(const void* descs_dev, const dist_integ_pack_args* one, int n, int Ci, int C4, hipStream_t s);

