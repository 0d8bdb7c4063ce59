// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Every kernel is templated on the storage type T (float = parity mode, __bf16 =
// performance mode).  Fragments are always "8 k-slots of T per lane":
//   bf16 : one v_mfma_f32_16x16x32_bf16 consumes the 8 slots at once
//   fp32 : eight v_mfma_f32_16x16x4_f32 (exact f32 FMA chain) consume slot e = 0..7
// A lane l supplies row/col (l & 15) and k-slot group g = l >> 4; WHICH k index a
// slot (g, e) stands for is free as long as the A and B fragments agree, which the
// kernels exploit (contiguous 8, or 4+4 for the transposed reads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dist_amd.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define DEV __device__ __forceinline__

template <typename T> struct Frag;           // 8 k-slots
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

DEV void frag_zero(Frag<bf16_t>& f) { f.v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
DEV void frag_zero(Frag<float>& f) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = 0.f;
}

// 8 contiguous elements, 16-byte (bf16) / 32-byte (fp32) aligned.
DEV void frag_load(Frag<bf16_t>& f, const bf16_t* p) { f.v = *reinterpret_cast<const bf16x8*>(p); }
DEV void frag_load(Frag<float>& f, const float* p) {
    const float4 a = reinterpret_cast<const float4*>(p)[0];
    const float4 b = reinterpret_cast<const float4*>(p)[1];
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
    f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
DEV void frag_store(const Frag<bf16_t>& f, bf16_t* p) { *reinterpret_cast<bf16x8*>(p) = f.v; }
DEV void frag_store(const Frag<float>& f, float* p) {
    reinterpret_cast<float4*>(p)[0] = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(f.v[4], f.v[5], f.v[6], f.v[7]);
}
// Streaming (non-temporal) stores for tensors that are written once and read by a LATER kernel (every activation is
// far larger than the 4 MB L2 of an XCD): they do not evict the operand panels other workgroups are re-reading.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));
DEV void store16_nt(void* p, const uint4& v) {
    u32x4_t x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<u32x4_t*>(p));
}
DEV void frag_store_nt(const Frag<bf16_t>& f, bf16_t* p) { __builtin_nontemporal_store(f.v, reinterpret_cast<bf16x8*>(p)); }
DEV void frag_store_nt(const Frag<float>& f, float* p) {
    f32x4v_t a = {f.v[0], f.v[1], f.v[2], f.v[3]}, b = {f.v[4], f.v[5], f.v[6], f.v[7]};
    __builtin_nontemporal_store(a, reinterpret_cast<f32x4v_t*>(p));
    __builtin_nontemporal_store(b, reinterpret_cast<f32x4v_t*>(p) + 1);
}
// two groups of 4 contiguous elements (8-byte / 16-byte aligned) -> slots 0..3 and 4..7
DEV void frag_load44(Frag<bf16_t>& f, const bf16_t* p0, const bf16_t* p1) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p0);
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(p1);
    f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
DEV void frag_load44(Frag<float>& f, const float* p0, const float* p1) {
    const float4 a = *reinterpret_cast<const float4*>(p0);
    const float4 b = *reinterpret_cast<const float4*>(p1);
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
    f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
DEV float frag_get(const Frag<bf16_t>& f, int e) { return (float)f.v[e]; }
DEV float frag_get(const Frag<float>& f, int e) { return f.v[e]; }
DEV void frag_set(Frag<bf16_t>& f, int e, float x) { f.v[e] = (bf16_t)x; }
DEV void frag_set(Frag<float>& f, int e, float x) { f.v[e] = x; }

// D(16x16) += A(16 x 8slots*4groups) * B: lane l holds A[row l&15][slots of group l>>4],
// B[slots of group l>>4][col l&15]; D: col = l&15, row = 4*(l>>4) + r.
DEV void mma16(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
DEV void mma16(const Frag<float>& a, const Frag<float>& b, f32x4& acc) {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[e], b.v[e], acc, 0, 0, 0);
}

DEV float to_f(float x) { return x; }
DEV float to_f(bf16_t x) { return (float)x; }
template <typename T> DEV T from_f(float x);
template <> DEV float from_f<float>(float x) { return x; }
template <> DEV bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }

// 4 consecutive elements
DEV void load4(const float* p, float o[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
}
DEV void load4(const bf16_t* p, float o[4]) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
    o[0] = (float)a[0]; o[1] = (float)a[1]; o[2] = (float)a[2]; o[3] = (float)a[3];
}
DEV void store4(float* p, const float o[4]) { *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
DEV void store4(bf16_t* p, const float o[4]) {
    bf16x4 a = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
    *reinterpret_cast<bf16x4*>(p) = a;
}

// QuickGELU x*sigmoid(1.702x) and its derivative (reference models/base/clip.py:199-201)
DEV float qgelu(float x) { return x / (1.f + __expf(-1.702f * x)); }
DEV float qgelu_grad(float x) {
    const float s = 1.f / (1.f + __expf(-1.702f * x));
    return s * (1.f + 1.702f * x * (1.f - s));
}
// bf16 kernels: the sigmoid's reciprocal as one v_rcp_f32 (1 ulp of fp32, far below the bf16 rounding of the result) instead
// of the IEEE division sequence (~12 VALU per element: 128 elements per lane in a 256x256 epilogue made the fc GEMM of
// the ViT MLP 45 us slower than its FLOPs).  The fp32 parity kernels keep the exact division.
template <typename T> DEV float qgelu_t(float x) { return qgelu(x); }
template <> DEV float qgelu_t<bf16_t>(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * x)); }
template <typename T> DEV float qgelu_grad_t(float x) { return qgelu_grad(x); }
template <> DEV float qgelu_grad_t<bf16_t>(float x) {
    const float s = __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * x));
    return s * (1.f + 1.702f * x * (1.f - s));
}

// The LayerNorm fold of a GEMM epilogue (DIST_EPI_LNFOLD), v = rstd * (acc - mean * colsum) + bias, with its roundings written out: one
// fma, one multiply, one add, no further contraction - every kernel that applies the fold (gemm_fast.hip, gemm_pp.hip) produces the same bits.
DEV float lnfold_bias(float acc, float mean, float rstd, float cs, float bias) {
#pragma clang fp contract(off)
    const float t = __builtin_fmaf(-mean, cs, acc);
    const float u = rstd * t;
    return u + bias;
}

DEV float wave_sum(float v, int width) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        if (o < width) v += __shfl_xor(v, o, 64);
    return v;
}
DEV float wave_max(float v, int width) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        if (o < width) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- row maps shared by the NT and TN GEMMs (conv-as-GEMM without im2col) --------------
// source row for logical row m and tap; returns -1 when the tap falls in the zero padding
DEV int rowmap_src(const dist_rowmap& rm, int m, int tap, int taps) {
    switch (rm.mode) {
        case DIST_RM_SHIFT: {      // p0 = rows per group G, p1 = rows per shift unit S
            const int off = rm.sign * (tap - taps / 2) * rm.p1;
            const int r = m % rm.p0 + off;
            return (r >= 0 && r < rm.p0) ? m + off : -1;
        }
        case DIST_RM_SPATIAL: {    // p0 = grid (3x3 taps on a grid x grid plane)
            const int gsz = rm.p0;
            const int dy = (tap / 3 - 1) * rm.sign, dx = (tap % 3 - 1) * rm.sign;
            const int n = m % (gsz * gsz);
            const int y = n / gsz + dy, x = n % gsz + dx;
            return (y >= 0 && y < gsz && x >= 0 && x < gsz) ? m + dy * gsz + dx : -1;
        }
        case DIST_RM_STRIDED: {    // p0 = alpha, p1 = N: (bj, n) -> (bj*alpha + tap, n)
            const int bj = m / rm.p1, n = m % rm.p1;
            return (bj * rm.p0 + tap) * rm.p1 + n;
        }
        case DIST_RM_SKIPCLS: {    // p0 = N: (bj, n) -> row bj*(N+1) + 1 + n
            const int bj = m / rm.p0, n = m % rm.p0;
            return bj * (rm.p0 + 1) + 1 + n;
        }
        default: return m;
    }
}

// Two-stage form for loops that walk the taps of one row: the divisions happen once per row (rowmap_prep), the per-tap step is a few
// adds and compares (rowmap_src2 == rowmap_src for every (m, tap)).
struct RowPrep { int a, b; };
DEV RowPrep rowmap_prep(const dist_rowmap& rm, int m) {
    switch (rm.mode) {
        case DIST_RM_SHIFT: return RowPrep{m % rm.p0, 0};
        case DIST_RM_SPATIAL: { const int n = m % (rm.p0 * rm.p0); return RowPrep{n / rm.p0, n % rm.p0}; }
        case DIST_RM_STRIDED: return RowPrep{m / rm.p1, m % rm.p1};
        case DIST_RM_SKIPCLS: return RowPrep{m / rm.p0, m % rm.p0};
        default: return RowPrep{0, 0};
    }
}
DEV int rowmap_src2(const dist_rowmap& rm, int m, const RowPrep q, int tap, int taps) {
    switch (rm.mode) {
        case DIST_RM_SHIFT: {
            const int off = rm.sign * (tap - taps / 2) * rm.p1;
            const int r = q.a + off;
            return (r >= 0 && r < rm.p0) ? m + off : -1;
        }
        case DIST_RM_SPATIAL: {
            const int gsz = rm.p0;
            const int dy = (tap / 3 - 1) * rm.sign, dx = (tap % 3 - 1) * rm.sign;
            const int y = q.a + dy, x = q.b + dx;
            return (y >= 0 && y < gsz && x >= 0 && x < gsz) ? m + dy * gsz + dx : -1;
        }
        case DIST_RM_STRIDED: return (q.a * rm.p0 + tap) * rm.p1 + q.b;
        case DIST_RM_SKIPCLS: return q.a * (rm.p0 + 1) + 1 + q.b;
        default: return m;
    }
}

// Stepping a row decomposition by `delta` rows without dividing (loops whose rows advance by a constant: the weight-gradient
// GEMM walks 64 rows per step).  rowmap_inc_ok: one conditional subtraction per component is enough for this map and delta.
DEV bool rowmap_inc_ok(const dist_rowmap& rm, int delta) {
    switch (rm.mode) {
        case DIST_RM_SHIFT: return delta <= rm.p0;
        case DIST_RM_SPATIAL: return delta / rm.p0 + 1 <= rm.p0;
        case DIST_RM_STRIDED: return delta <= rm.p1;
        case DIST_RM_SKIPCLS: return delta <= rm.p0;
        default: return true;
    }
}
DEV RowPrep rowmap_step(const dist_rowmap& rm, RowPrep q, int delta) {      // == rowmap_prep(rm, m + delta) given q == rowmap_prep(rm, m)
    switch (rm.mode) {
        case DIST_RM_SHIFT: { int a = q.a + delta; a -= (a >= rm.p0) ? rm.p0 : 0; return RowPrep{a, 0}; }
        case DIST_RM_SPATIAL: {
            const int g = rm.p0, dq = delta / g, dr = delta - dq * g;          // wave-uniform
            int x = q.b + dr; const int c = x >= g ? 1 : 0; x -= c ? g : 0;
            int y = q.a + dq + c; y -= (y >= g) ? g : 0;
            return RowPrep{y, x};
        }
        case DIST_RM_STRIDED: { int n = q.b + delta; const int c = n >= rm.p1 ? 1 : 0; return RowPrep{q.a + c, n - (c ? rm.p1 : 0)}; }
        case DIST_RM_SKIPCLS: { int n = q.b + delta; const int c = n >= rm.p0 ? 1 : 0; return RowPrep{q.a + c, n - (c ? rm.p0 : 0)}; }
        default: return q;
    }
}

// Environment knobs of the library, read once per process at their call sites.
//   dist_knob():         algorithm selectors - every value computes the SAME results through another kernel sequence (a fused kernel off, a tile
//                        shape, a stream layout); tests/test_fallbacks_gpu.py runs the engine tests under the ones that switch a kernel off,
//                        bench.py reports every DIST_AMD_* variable that was set ("knobs").
//   dist_measure_knob(): timing-only switches whose results are WRONG (kernels skipped, dummy work, debug phases).  They exist only in the
//                        timing-only library: `python -m dist_amd.build --measure` compiles -DDIST_AMD_MEASURE objects (*.m.o) into
//                        csrc/libdist_amd_measure.so beside the product library, `. tools/measure_build.sh` builds it, points DIST_AMD_LIB at it and
//                        asserts dist_measure_build() == 1 (tools/skip_step.sh, tools/r05_pp_dbg.sh, the ablation scripts): the shipped library
//                        cannot be driven wrong by the environment, and a measurement cannot leave a wrong-result library in its place.
//   DIST_AB_KNOB():      selectors of variants that were MEASURED AND REJECTED (profiles/: another tile shape, wave count, stream layout ...): the A/B
//                        references of those notes.  In the product library the macro is its default - a constant - and the variant's code sits
//                        behind `if constexpr (DIST_AB)` / `#ifdef DIST_AMD_MEASURE`, so neither the kernel nor the environment read ship; in the
//                        timing-only library it is dist_knob().
#include <stdlib.h>
inline int dist_knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#ifdef DIST_AMD_MEASURE
#define DIST_AB_KNOB(name, dflt) dist_knob(name, dflt)
constexpr bool DIST_AB = true;
#else
#define DIST_AB_KNOB(name, dflt) (dflt)
constexpr bool DIST_AB = false;
#endif
#ifdef DIST_AMD_MEASURE
inline int dist_measure_knob(const char* name, int dflt) { return dist_knob(name, dflt); }
#else
inline int dist_measure_knob(const char*, int dflt) { return dflt; }
#endif

#define HIP_CHECK_RET_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return -(int)e_ - 1000; } while (0)
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE attribute of a kernel: raise it once per (kernel, device) - `st` is the call
// site's static state - and again whenever a launch needs more than was set.  (A process-wide `static bool` left the second device of a process
// with the 64 KB default; two threads racing here both set the same value.)
#include <atomic>
struct DistSmemOnce { std::atomic<size_t> bytes[32]; };
inline int dist_max_smem(DistSmemOnce& st, const void* kernel, size_t smem) {
    int dev = 0;
    HIP_CHECK_RET_(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < 32;
    if (!tracked || st.bytes[dev].load(std::memory_order_relaxed) < smem) {
        HIP_CHECK_RET_(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (tracked) st.bytes[dev].store(smem, std::memory_order_relaxed);
    }
    return 0;
}

#define RUN_(x) do { const int rc_ = (x); if (rc_ != 0) return rc_; } while (0)
#define HIP_CHECK_RET(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return -(int)e_ - 1000; } while (0)
