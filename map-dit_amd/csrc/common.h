// Shared device helpers for the MaP-DiT gfx950 kernels.  MI355X / CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mapdit.h"

typedef unsigned short bf16_t;                                   // raw bfloat16 bits in HBM / LDS
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;      // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) void lds_void_t;

#define MP_SILU_DIV 0.596f
#define NORM_EPS 1e-4f

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float x) {
    __bf16 b = (__bf16)x;                                        // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(bf16_t, b);
}
// one v_cvt_pk_bf16_f32 for the pair (as two scalar conversions merged by shift + or, hipcc emitted four instructions per pair)
typedef __attribute__((ext_vector_type(2))) float md_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 md_bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const md_f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, md_bf16x2_t));
}

// ---- the 16-bit operand format of a translation unit ---------------------------------------------------------------------
// Every kernel file that reads or writes 16-bit GEMM / attention operands is compiled TWICE (csrc/Makefile): MAPDIT_DT = 0 with
// bf16 operands (entry points mapdit_<name>) and MAPDIT_DT = 1 with IEEE fp16 operands (mapdit_<name>_f16), the same source
// otherwise.  fp16 issues on the MFMA pipe at the bf16 rate and carries TF32's 10-bit mantissa: it is what puts the fast engine
// inside the reference's fp32/TF32 tolerance (mapdit.h, MAPDIT_PREC_F16).  In the shared code a 16-bit element is a `bf16_t`
// (raw bits) whatever the format; conversions go through cvt16 / up16 / pack16 / lo16 / hi16 and products through MFMA16 / MFMA32.
// f2bf / bf2f / pack2bf above stay bf16 in both builds: the two-term split operands of the fp32-accurate GEMMs are always bf16.
#ifndef MAPDIT_DT
#define MAPDIT_DT 0
#endif
#if MAPDIT_DT == 1
#define MD_SYM(name) mapdit_##name##_f16
#define MD_TU(name) name##_f16
#define MD_SYM_GEMM mapdit_gemm_f16
#define MD_SYM_GEMM_GROUP mapdit_gemm_group_tn_f16
#define MD_SYM_F32_TO_16 mapdit_f32_to_f16
#define MD_SYM_F32_TO_16_2D mapdit_f32_to_f16_2d
#define MD_SYM_MPSILU_TO_16 mapdit_mpsilu_to_f16
#else
#define MD_SYM(name) mapdit_##name
#define MD_TU(name) name
#define MD_SYM_GEMM mapdit_gemm_bf16
#define MD_SYM_GEMM_GROUP mapdit_gemm_group_tn_bf16
#define MD_SYM_F32_TO_16 mapdit_f32_to_bf16
#define MD_SYM_F32_TO_16_2D mapdit_f32_to_bf16_2d
#define MD_SYM_MPSILU_TO_16 mapdit_mpsilu_to_bf16
#endif
// kernels live in an anonymous namespace; the fp16 build adds a named level so that profiles tell the two builds' kernels apart
#if MAPDIT_DT == 1
#define MD_NS_OPEN namespace { namespace f16 {
#define MD_NS_CLOSE } using namespace f16; }
#else
#define MD_NS_OPEN namespace {
#define MD_NS_CLOSE }
#endif
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
#if MAPDIT_DT == 1
__device__ __forceinline__ bf16_t cvt16(float x) { return __builtin_bit_cast(bf16_t, (_Float16)x); }        // v_cvt_f16_f32, RNE
__device__ __forceinline__ float up16(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint32_t pack16(float lo, float hi) {                                             // v_cvt_pk_f16_f32
    const f16x2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float lo16(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w).x; }
__device__ __forceinline__ float hi16(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w).y; }       // v_cvt_f32_f16 sdwa WORD_1
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0)
#else
__device__ __forceinline__ bf16_t cvt16(float x) { return f2bf(x); }
__device__ __forceinline__ float up16(bf16_t v) { return bf2f(v); }
__device__ __forceinline__ uint32_t pack16(float lo, float hi) { return pack2bf(lo, hi); }
__device__ __forceinline__ float lo16(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi16(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over aligned groups of 2^n lanes by DPP (no index register, no LDS traffic; ds_swizzle for the 16 <-> 16 step): every lane of
// a group ends with the group's sum.
template <int CTRL> __device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum8(float d) {
    d += dpp_f<0xB1>(d);                                    // quad_perm [1,0,3,2]
    d += dpp_f<0x4E>(d);                                    // quad_perm [2,3,0,1]
    d += dpp_f<0x141>(d);                                   // row_half_mirror
    return d;
}
__device__ __forceinline__ float sum16(float d) { d = sum8(d); return d + dpp_f<0x140>(d); }          // row_mirror
__device__ __forceinline__ float sum32(float d) {
    d = sum16(d);
    return d + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, d), 0x401F));   // lane ^ 16
}
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions): these run in GEMM epilogues, where the VALU
// time of the activation is not hidden behind anything, and their results are rounded to bf16 anyway.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
// d/dx [silu(x)/0.596]
__device__ __forceinline__ float dmpsilu_f(float x) {
    float s = __builtin_amdgcn_rcpf(1.f + __expf(-x));
    return s * (1.f + x * (1.f - s)) * (1.f / MP_SILU_DIV);
}

// ---- index validation (labels, timesteps) ------------------------------------------------------------------------------
// The reference raises an IndexError for an out-of-range label / timestep (F.embedding, numpy fancy indexing).  A kernel cannot
// raise: it clamps the index (no out-of-bounds access, ever) and records a code in a device word that
// mapdit_device_error_poll() hands to the host.  One word per translation unit (no relocatable device code in this build);
// the poll entry point ORs them together.
#define MAPDIT_DEVERR_LABEL 1
#define MAPDIT_DEVERR_TIMESTEP 2
#define MAPDIT_DEFINE_DEV_ERROR(TU)                                                                    \
    __device__ int g_dev_error_##TU = 0;                                                               \
    int mapdit_dev_error_take_##TU(hipStream_t st, int* out) {                                         \
        int v = 0, zero = 0;                                                                           \
        hipError_t e = hipMemcpyFromSymbolAsync(&v, HIP_SYMBOL(g_dev_error_##TU), sizeof(int), 0, hipMemcpyDeviceToHost, st); \
        if (e == hipSuccess) e = hipStreamSynchronize(st);                                             \
        if (e == hipSuccess && v) e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_dev_error_##TU), &zero, sizeof(int), 0, hipMemcpyHostToDevice, st); \
        if (e == hipSuccess && v) e = hipStreamSynchronize(st);                                        \
        *out = v;                                                                                      \
        return e == hipSuccess ? 0 : 1;                                                                \
    }
#define MAPDIT_CHECKED_INDEX(TU, v, n, code)                                                           \
    ([&]() -> long {                                                                                   \
        long v_ = (v);                                                                                 \
        if (v_ < 0 || v_ >= (long)(n)) {                                                               \
            g_dev_error_##TU = (code);                                                                 \
            v_ = v_ < 0 ? 0 : (long)(n) - 1;                                                           \
        }                                                                                              \
        return v_;                                                                                     \
    }())
int mapdit_dev_error_take_embed(hipStream_t st, int* out);
int mapdit_dev_error_take_embed_f16(hipStream_t st, int* out);
int mapdit_dev_error_take_diffusion(hipStream_t st, int* out);
int mapdit_dev_error_take_precise(hipStream_t st, int* out);

// Error plumbing shared by the C-ABI entry points (thread-local last-error string).
void mapdit_set_error(const char* fmt, ...);
#define MD_CHECK(cond, ...)                                   \
    do {                                                      \
        if (!(cond)) {                                        \
            mapdit_set_error(__VA_ARGS__);                    \
            return MAPDIT_ERR_ARG;                            \
        }                                                     \
    } while (0)
#define MD_LAUNCH_CHECK()                                                     \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess) {                                               \
            mapdit_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return MAPDIT_ERR_HIP;                                            \
        }                                                                     \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
