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
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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
