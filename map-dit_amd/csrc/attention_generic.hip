// Generic cosine attention for the shapes the MFMA kernels do not take (patch-8 models: 16 tokens; any head_dim <= 96 that is
// neither 64 nor 72).  Same maths and interfaces as attention.hip / attention72.hip, on the fp32 VALU: one thread owns one
// query row (forward, dQ) or one key row (dK, dV) of a head and sweeps the other index with the head's K, V (or Q, dO) rows
// broadcast from LDS.  Also the reference implementation the head_dim-72 MFMA kernels are tested against.
// Layout: qn, kn, v, dqn, dkn, dv  [B*H][T][hd] bf16 (no padding); o, dO [B*T][H*hd] bf16; lse, delta [B*H][T] fp32.
#include "common.h"

MD_NS_OPEN

constexpr int GEN_MAX_HD = 96;
constexpr int GEN_MAX_T = 256;

// qkv [M, 3D] -> qn, kn (cosine-normalised: q*sqrt(hd)/(|q|+eps)), v.  One thread per (token, head).
__global__ void qkv_split_generic_kernel(const bf16_t* __restrict__ qkv, int B, int T, int H, int hd, bf16_t* __restrict__ qn,
                                         bf16_t* __restrict__ kn, bf16_t* __restrict__ v) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)B * T * H) return;
    const int h = (int)(id % H);
    const long m = id / H;
    const int t = (int)(m % T), b = (int)(m / T);
    const int D = H * hd;
    const float rt = sqrtf((float)hd);
    for (int which = 0; which < 3; ++which) {
        const bf16_t* src = qkv + m * 3 * D + which * D + h * hd;
        bf16_t* dst = (which == 0 ? qn : which == 1 ? kn : v) + (((size_t)b * H + h) * T + t) * hd;
        float s = 1.f;
        if (which < 2) {
            float ss = 0.f;
            for (int d = 0; d < hd; ++d) { const float x = up16(src[d]); ss += x * x; }
            s = rt / (sqrtf(ss) + NORM_EPS);
        }
        for (int d = 0; d < hd; ++d) dst[d] = which < 2 ? cvt16(up16(src[d]) * s) : src[d];
    }
}

// dqn, dkn, dv + saved qkv -> dqkv [M, 3D]:  dq = s (dq^ - q (dq^.q) / (n (n+eps))),  s = sqrt(hd)/(n+eps).
__global__ void qkv_merge_bwd_generic_kernel(const bf16_t* __restrict__ qkv, int B, int T, int H, int hd,
                                             const bf16_t* __restrict__ dqn, const bf16_t* __restrict__ dkn,
                                             const bf16_t* __restrict__ dv, bf16_t* __restrict__ dqkv) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)B * T * H) return;
    const int h = (int)(id % H);
    const long m = id / H;
    const int t = (int)(m % T), b = (int)(m / T);
    const int D = H * hd;
    const float rt = sqrtf((float)hd);
    for (int which = 0; which < 3; ++which) {
        const size_t moff = (size_t)m * 3 * D + which * D + h * hd;
        const bf16_t* g = (which == 0 ? dqn : which == 1 ? dkn : dv) + (((size_t)b * H + h) * T + t) * hd;
        if (which == 2) {
            for (int d = 0; d < hd; ++d) dqkv[moff + d] = g[d];
            continue;
        }
        float ss = 0.f, dot = 0.f;
        for (int d = 0; d < hd; ++d) {
            const float x = up16(qkv[moff + d]);
            ss += x * x;
            dot += x * up16(g[d]);
        }
        const float n = sqrtf(ss), s = rt / (n + NORM_EPS), c = dot / (fmaxf(n, 1e-30f) * (n + NORM_EPS));
        for (int d = 0; d < hd; ++d) dqkv[moff + d] = cvt16(s * (up16(g[d]) - up16(qkv[moff + d]) * c));
    }
}

// dqn, dkn, dv [B*H][T][hd] -> dqkv [M, 3D], nothing else (the head merge of plain scaled-dot-product attention: q, k were not
// normalised, so there is no Jacobian).  One thread per 8 consecutive elements of a (token, which, head) row.
__global__ void heads_merge_kernel(const bf16_t* __restrict__ dqn, const bf16_t* __restrict__ dkn, const bf16_t* __restrict__ dv, int B, int T,
                                   int H, int hd, bf16_t* __restrict__ dqkv) {
    const int ch = hd / 8;
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)B * T * 3 * H * ch) return;
    const int c = (int)(id % ch);
    long r = id / ch;
    const int h = (int)(r % H); r /= H;
    const int which = (int)(r % 3);
    const long m = r / 3;
    const int t = (int)(m % T), b = (int)(m / T);
    const bf16_t* src = (which == 0 ? dqn : which == 1 ? dkn : dv) + (((size_t)b * H + h) * T + t) * hd + 8 * c;
    *(uint4*)(dqkv + (size_t)m * 3 * H * hd + (size_t)which * H * hd + h * hd + 8 * c) = *(const uint4*)src;
}

// Stage `rows` rows of `hd` bf16 (row stride ld elements) into LDS as fp32 [rows][HD], zero padded to the compile-time
// row length HD (a multiple of 4), so the sweeps below are branch-free and read LDS 16 bytes at a time.
template <int HD>
__device__ __forceinline__ void stage_f32(float* dst, const bf16_t* __restrict__ src, long ld, int rows, int hd, int tid, int nth) {
    for (int i = tid; i < rows * HD; i += nth) {
        const int r = i / HD, d = i % HD;
        dst[i] = d < hd ? up16(src[(size_t)r * ld + d]) : 0.f;
    }
}
template <int HD> __device__ __forceinline__ float dot_row(const float (&a)[HD], const float* __restrict__ row) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        const float4 k = *(const float4*)(row + d);
        s += a[d] * k.x + a[d + 1] * k.y + a[d + 2] * k.z + a[d + 3] * k.w;
    }
    return s;
}
template <int HD> __device__ __forceinline__ void axpy_row(float (&acc)[HD], float w, const float* __restrict__ row) {
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        const float4 k = *(const float4*)(row + d);
        acc[d] += w * k.x; acc[d + 1] += w * k.y; acc[d + 2] += w * k.z; acc[d + 3] += w * k.w;
    }
}

// forward: o_i = sum_j softmax_j(q_i.k_j / sqrt(hd)) v_j ; cosine logits are bounded, so no running maximum.
// maxsub != 0 (mapdit_attn_sdpa_fwd: q, k NOT normalised - README.md:58 --no-use-cosine-attention, parity unpinned): the row maximum is
// found in a first sweep and subtracted, lse = max + log(sum) - the backward kernels recompute p = exp(s - lse) and need no change.
template <int HD>
__global__ __launch_bounds__(256) void attn_generic_fwd_kernel(const bf16_t* __restrict__ qn, const bf16_t* __restrict__ kn,
                                                             const bf16_t* __restrict__ v, bf16_t* __restrict__ o,
                                                             float* __restrict__ lse, int T, int H, int hd, float scale, int maxsub) {
    extern __shared__ float sm[];
    float* ks = sm;
    float* vs = sm + (size_t)T * HD;
    const size_t bh = blockIdx.x;
    stage_f32<HD>(ks, kn + bh * T * hd, hd, T, hd, threadIdx.x, blockDim.x);
    stage_f32<HD>(vs, v + bh * T * hd, hd, T, hd, threadIdx.x, blockDim.x);
    __syncthreads();
    const int b = (int)(bh / H), hh = (int)(bh % H), D = H * hd;
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        float q[HD], acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { q[d] = d < hd ? up16(qn[(bh * T + i) * hd + d]) : 0.f; acc[d] = 0.f; }
        float l = 0.f, mx = 0.f;
        if (maxsub) {
            mx = -3.0e38f;
            for (int j = 0; j < T; ++j) mx = fmaxf(mx, dot_row<HD>(q, ks + j * HD) * scale);
        }
        for (int j = 0; j < T; ++j) {
            const float p = __expf(dot_row<HD>(q, ks + j * HD) * scale - mx);
            l += p;
            axpy_row<HD>(acc, p, vs + j * HD);
        }
        const float il = 1.f / l;
#pragma unroll
        for (int d = 0; d < HD; ++d) if (d < hd) o[((size_t)b * T + i) * D + hh * hd + d] = cvt16(acc[d] * il);
        lse[bh * T + i] = mx + __logf(l);
    }
}

// backward, query-owner pass: delta_i = do_i.o_i ; dq^_i = sum_j ds_ij k_j,  ds_ij = p_ij (do_i.v_j - delta_i) scale
template <int HD>
__global__ __launch_bounds__(256) void attn_generic_dq_kernel(const bf16_t* __restrict__ qn, const bf16_t* __restrict__ kn,
                                                            const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
                                                            const bf16_t* __restrict__ O, const float* __restrict__ lse,
                                                            float* __restrict__ delta, bf16_t* __restrict__ dqn, int T, int H,
                                                            int hd, float scale) {
    extern __shared__ float sm[];
    float* ks = sm;
    float* vs = sm + (size_t)T * HD;
    const size_t bh = blockIdx.x;
    stage_f32<HD>(ks, kn + bh * T * hd, hd, T, hd, threadIdx.x, blockDim.x);
    stage_f32<HD>(vs, v + bh * T * hd, hd, T, hd, threadIdx.x, blockDim.x);
    __syncthreads();
    const int b = (int)(bh / H), hh = (int)(bh % H), D = H * hd;
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        float q[HD], g[HD], acc[HD];
        float del = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            const size_t mo = ((size_t)b * T + i) * D + hh * hd + d;
            q[d] = d < hd ? up16(qn[(bh * T + i) * hd + d]) : 0.f;
            g[d] = d < hd ? up16(dO[mo]) : 0.f;
            if (d < hd) del += g[d] * up16(O[mo]);
            acc[d] = 0.f;
        }
        const float ls = lse[bh * T + i];
        for (int j = 0; j < T; ++j) {
            const float s = dot_row<HD>(q, ks + j * HD), dp = dot_row<HD>(g, vs + j * HD);
            axpy_row<HD>(acc, __expf(s * scale - ls) * (dp - del) * scale, ks + j * HD);
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) if (d < hd) dqn[(bh * T + i) * hd + d] = cvt16(acc[d]);
        delta[bh * T + i] = del;
    }
}

// backward, key-owner pass: dv_j = sum_i p_ij do_i ; dk^_j = sum_i ds_ij q_i
template <int HD>
__global__ __launch_bounds__(256) void attn_generic_dkv_kernel(const bf16_t* __restrict__ qn, const bf16_t* __restrict__ kn,
                                                             const bf16_t* __restrict__ v, const bf16_t* __restrict__ dO,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             bf16_t* __restrict__ dkn, bf16_t* __restrict__ dv, int T, int H, int hd,
                                                             float scale) {
    extern __shared__ float sm[];
    float* qs = sm;
    float* gs = sm + (size_t)T * HD;
    float* ls = gs + (size_t)T * HD;
    float* dl = ls + T;
    const size_t bh = blockIdx.x;
    const int b = (int)(bh / H), hh = (int)(bh % H), D = H * hd;
    stage_f32<HD>(qs, qn + bh * T * hd, hd, T, hd, threadIdx.x, blockDim.x);
    stage_f32<HD>(gs, dO + (size_t)b * T * D + hh * hd, D, T, hd, threadIdx.x, blockDim.x);
    for (int i = threadIdx.x; i < T; i += blockDim.x) { ls[i] = lse[bh * T + i]; dl[i] = delta[bh * T + i]; }
    __syncthreads();
    for (int j = threadIdx.x; j < T; j += blockDim.x) {
        float k[HD], vv[HD], ak[HD], av[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            k[d] = d < hd ? up16(kn[(bh * T + j) * hd + d]) : 0.f;
            vv[d] = d < hd ? up16(v[(bh * T + j) * hd + d]) : 0.f;
            ak[d] = 0.f; av[d] = 0.f;
        }
        for (int i = 0; i < T; ++i) {
            const float s = dot_row<HD>(k, qs + i * HD), dp = dot_row<HD>(vv, gs + i * HD);
            const float p = __expf(s * scale - ls[i]);
            axpy_row<HD>(av, p, gs + i * HD);
            axpy_row<HD>(ak, p * (dp - dl[i]) * scale, qs + i * HD);
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) if (d < hd) { dkn[(bh * T + j) * hd + d] = cvt16(ak[d]); dv[(bh * T + j) * hd + d] = cvt16(av[d]); }
    }
}

int check(int T, int hd) {
    MD_CHECK(hd > 0 && hd <= GEN_MAX_HD, "generic attention: head_dim=%d unsupported (1..%d)", hd, GEN_MAX_HD);
    MD_CHECK(T > 0 && T <= GEN_MAX_T, "generic attention: %d tokens unsupported (1..%d)", T, GEN_MAX_T);
    return MAPDIT_OK;
}

MD_NS_CLOSE

#define GEN_DISPATCH(hd_, CALL)                                            \
    if (hd_ <= 32) { constexpr int HDT = 32; CALL; }                        \
    else if (hd_ <= 64) { constexpr int HDT = 64; CALL; }                   \
    else if (hd_ <= 72) { constexpr int HDT = 72; CALL; }                   \
    else { constexpr int HDT = 96; CALL; }

extern "C" int MD_SYM(qkv_split_generic)(const uint16_t* qkv, int B, int T, int H, int head_dim, uint16_t* qn, uint16_t* kn,
                                        uint16_t* v, void* stream) {
    MD_CHECK(qkv && qn && kn && v, "qkv_split_generic: null argument");
    const long n = (long)B * T * H;
    hipLaunchKernelGGL(qkv_split_generic_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, qkv, B, T, H, head_dim, qn, kn, v);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(qkv_merge_bwd_generic)(const uint16_t* qkv, int B, int T, int H, int head_dim, const uint16_t* dqn,
                                            const uint16_t* dkn, const uint16_t* dv, uint16_t* dqkv, void* stream) {
    MD_CHECK(qkv && dqn && dkn && dv && dqkv, "qkv_merge_bwd_generic: null argument");
    const long n = (long)B * T * H;
    hipLaunchKernelGGL(qkv_merge_bwd_generic_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, qkv, B, T, H, head_dim,
                       dqn, dkn, dv, dqkv);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(heads_merge_bwd)(const uint16_t* dqn, const uint16_t* dkn, const uint16_t* dv, int B, int T, int H, int head_dim,
                                      uint16_t* dqkv, void* stream) {
    MD_CHECK(dqn && dkn && dv && dqkv && B > 0 && T > 0 && H > 0, "heads_merge_bwd: null/empty argument");
    MD_CHECK(head_dim > 0 && head_dim % 8 == 0, "heads_merge_bwd: head_dim=%d must be a multiple of 8", head_dim);
    const long n = (long)B * T * 3 * H * (head_dim / 8);
    hipLaunchKernelGGL(heads_merge_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dqn, dkn, dv, B, T, H, head_dim, dqkv);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

static int generic_fwd(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, uint16_t* o, float* lse, int B, int T, int H, int head_dim,
                       int maxsub, void* stream);
extern "C" int MD_SYM(attn_generic_fwd)(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, uint16_t* o, float* lse, int B,
                                       int T, int H, int head_dim, void* stream) {
    return generic_fwd(qn, kn, v, o, lse, B, T, H, head_dim, 0, stream);
}
// (internal, C++ linkage: the generic leg of mapdit_attn_sdpa_fwd, attention.hip)
int MD_SYM(attn_generic_fwd_max)(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, int B,
                                           int T, int H, int head_dim, void* stream) {
    return generic_fwd(q, k, v, o, lse, B, T, H, head_dim, 1, stream);
}
static int generic_fwd(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, uint16_t* o, float* lse, int B, int T, int H, int head_dim,
                       int maxsub, void* stream) {
    MD_CHECK(qn && kn && v && o && lse, "attn_generic_fwd: null argument");
    if (check(T, head_dim) != MAPDIT_OK) return MAPDIT_ERR_ARG;
    const float scale = 1.f / sqrtf((float)head_dim);
    const int hdp = head_dim <= 32 ? 32 : head_dim <= 64 ? 64 : head_dim <= 72 ? 72 : 96;      // = HDT of GEN_DISPATCH
    const size_t shm = (size_t)2 * T * hdp * 4;
    const int nth = T < 256 ? ((T + 63) / 64) * 64 : 256;
    GEN_DISPATCH(head_dim, (void)hipFuncSetAttribute((const void*)attn_generic_fwd_kernel<HDT>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    GEN_DISPATCH(head_dim, hipLaunchKernelGGL((attn_generic_fwd_kernel<HDT>), dim3(B * H), dim3(nth), shm, (hipStream_t)stream, qn, kn,
                                              v, o, lse, T, H, head_dim, scale, maxsub));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(attn_generic_bwd)(const uint16_t* qn, const uint16_t* kn, const uint16_t* v, const uint16_t* dO,
                                       const uint16_t* O, const float* lse, float* delta, uint16_t* dqn, uint16_t* dkn,
                                       uint16_t* dv, int B, int T, int H, int head_dim, void* stream) {
    MD_CHECK(qn && kn && v && dO && O && lse && delta && dqn && dkn && dv, "attn_generic_bwd: null argument");
    if (check(T, head_dim) != MAPDIT_OK) return MAPDIT_ERR_ARG;
    const float scale = 1.f / sqrtf((float)head_dim);
    const int nth = T < 256 ? ((T + 63) / 64) * 64 : 256;
    const int hdp = head_dim <= 32 ? 32 : head_dim <= 64 ? 64 : head_dim <= 72 ? 72 : 96;
    const size_t shm1 = (size_t)2 * T * hdp * 4, shm2 = shm1 + (size_t)2 * T * 4;
    GEN_DISPATCH(head_dim, (void)hipFuncSetAttribute((const void*)attn_generic_dq_kernel<HDT>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm1));
    GEN_DISPATCH(head_dim, (void)hipFuncSetAttribute((const void*)attn_generic_dkv_kernel<HDT>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2));
    GEN_DISPATCH(head_dim, hipLaunchKernelGGL((attn_generic_dq_kernel<HDT>), dim3(B * H), dim3(nth), shm1, (hipStream_t)stream, qn, kn,
                                              v, dO, O, lse, delta, dqn, T, H, head_dim, scale));
    MD_LAUNCH_CHECK();
    GEN_DISPATCH(head_dim, hipLaunchKernelGGL((attn_generic_dkv_kernel<HDT>), dim3(B * H), dim3(nth), shm2, (hipStream_t)stream, qn, kn,
                                              v, dO, lse, delta, dkn, dv, T, H, head_dim, scale));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
