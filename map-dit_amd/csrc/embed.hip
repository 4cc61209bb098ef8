// Input / conditioning / output-side kernels of DiT (SURVEY.md K9, K10, K11 and their backward).
// All are small or HBM-bound; the FLOP-carrying parts go through mapdit_gemm_bf16.
#include "common.h"

#if MAPDIT_DT == 1
MAPDIT_DEFINE_DEV_ERROR(embed_f16)
#define g_dev_error_embed g_dev_error_embed_f16
#else
MAPDIT_DEFINE_DEV_ERROR(embed)
#endif

MD_NS_OPEN

#define C5 0.70710678118654752f   // mp_sum(a, b, 0.5) = (a + b) * 0.5 / sqrt(0.5)   (src/utils.py:15-16, dit.py:84,88)

// ---- patchify + ones column + x_embedder + mp_sum with pos_embed (reference src/dit.py:81-84) --------------------
// x [N,C,S,S] fp32, w [D][P+1] fp32 effective weight, pos [T][D]; out x0 [M,D] fp32; patches [M][ldp] bf16
// (zero padded to ldp, ones column at index P) kept for the dW GEMM.  Block = 64 tokens x 128 features.
template <int DT>
__global__ __launch_bounds__(256) void patch_embed_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ pos, float* __restrict__ out,
                                                            bf16_t* __restrict__ patches, int ldp, int C, int S, int p,
                                                            int D, long M, float c5) {
    extern __shared__ float sm[];
    const int P = p * p * C, P1 = P + 1, grid = S / p, T = grid * grid;
    float* ws = sm;                    // [P1][DT]
    float* ps = sm + P1 * DT;          // [64][P1]
    const long m0 = (long)blockIdx.x * 64;
    const int d0 = blockIdx.y * DT;
    if (P1 != 17)                      // (P1 = 17 reads its weight columns straight into registers, below)
    for (int i = threadIdx.x; i < P1 * DT; i += 256) {
        const int j = i / DT, d = i % DT;
        ws[i] = w[(size_t)(d0 + d) * P1 + j];
    }
    for (int i = threadIdx.x; i < 64 * P1; i += 256) {
        const int tok = i / P1, j = i % P1;
        const long m = m0 + tok;
        float v = 0.f;
        if (m < M) {
            if (j == P) v = 1.f;
            else {
                const int n = (int)(m / T), t = (int)(m % T), hy = t / grid, wx = t % grid;
                const int c = j % C, p2 = (j / C) % p, p1 = j / (C * p);
                v = x[(((size_t)n * C + c) * S + hy * p + p1) * S + wx * p + p2];
            }
        }
        ps[i] = v;
    }
    __syncthreads();
    if (patches && blockIdx.y == 0) {
        for (int i = threadIdx.x; i < 64 * ldp; i += 256) {
            const int tok = i / ldp, j = i % ldp;
            if (m0 + tok < M) patches[(m0 + tok) * ldp + j] = cvt16(j < P1 ? ps[tok * P1 + j] : 0.f);
        }
    }
    // Round 5, patch 2 with 4 channels (P1 = 17: every */2 model on 4-channel latents): a thread owns FOUR consecutive features and keeps
    // their weight column in registers (34 LDS reads per output element before, 17 broadcast reads per four now; 16-byte stores).  The sum
    // runs over j in the same order with the same contraction: the same bits.
    if (P1 == 17) {
        // ... and the block walks ALL feature tiles of its 64 tokens (grid.y = 1): the gathered patch rows are staged once, not D / DT times
        constexpr int TPR = DT / 4;                            // threads per token row
        const int d4 = (threadIdx.x % TPR) * 4;
        for (int dd = blockIdx.y * DT; dd < D; dd += gridDim.y * DT) {
            float wr[17][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 17; ++j) wr[j][k] = w[(size_t)(dd + d4 + k) * 17 + j];
            for (int tok = threadIdx.x / TPR; tok < 64; tok += 256 / TPR) {
                const long m = m0 + tok;
                if (m >= M) break;
                float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 17; ++j) {
                    const float xv = ps[tok * 17 + j];
                    a[0] += xv * wr[j][0]; a[1] += xv * wr[j][1]; a[2] += xv * wr[j][2]; a[3] += xv * wr[j][3];
                }
                const float4 pe = *(const float4*)(pos + (size_t)(m % T) * D + dd + d4);
                *(float4*)(out + m * D + dd + d4) = make_float4((a[0] + pe.x) * c5, (a[1] + pe.y) * c5, (a[2] + pe.z) * c5, (a[3] + pe.w) * c5);
            }
        }
        return;
    }
    const int d = threadIdx.x % DT;
    for (int tok = threadIdx.x / DT; tok < 64; tok += 256 / DT) {
        const long m = m0 + tok;
        if (m >= M) break;
        float a = 0.f;
        for (int j = 0; j < P1; ++j) a += ps[tok * P1 + j] * ws[j * DT + d];
        out[m * D + d0 + d] = (a + pos[(size_t)(m % T) * D + d0 + d]) * c5;
    }
}

// ---- timestep Fourier features (reference src/blocks/timestep_embedder.py:18-21) ---------------------------------
__global__ void fourier_kernel(const long* __restrict__ t, const float* __restrict__ scale, const float* __restrict__ shift,
                               bf16_t* __restrict__ out, int n, int F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * F) return;
    const int b = i / F, f = i % F;
    // torch rounds the product before the add (outer(), then +): the arguments reach ~1e4 rad, where a fused multiply-add
    // would move cos() by up to 1e-3 - keep the two roundings
    float prod = (float)t[b] * scale[f];
    asm volatile("" : "+v"(prod));     // opaque to the optimiser: no contraction into v_fma (build uses -ffp-contract=fast)
    const float arg = prod + shift[f];
    out[i] = cvt16(1.41421356237309515f * cosf(arg));
}

// ---- c = mp_sum(t_emb, y_emb, 0.5); also MPSiLU(c) and c as bf16 GEMM operands (dit.py:86-88) ------------------------
__global__ void cond_combine_kernel(const float* __restrict__ temb, const float* __restrict__ table, const long* __restrict__ y,
                                    float* __restrict__ c, bf16_t* __restrict__ c_silu, bf16_t* __restrict__ c_bf, int n, int D,
                                    int table_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const int b = i / D, d = i % D;
    const long label = MAPDIT_CHECKED_INDEX(embed, y[b], table_rows, MAPDIT_DEVERR_LABEL);
    const float v = (temb[i] + table[(size_t)label * D + d]) * C5;
    c[i] = v;
    c_silu[i] = cvt16(silu_f(v) * (1.f / MP_SILU_DIV));
    c_bf[i] = cvt16(v);
}

// Backward of the above: dc = dcs * dmpsilu(c) + dc_direct;  dtemb = C5*dc (bf16 operand);  dtable[y] += C5*dc.
// No atomics (bit-reproducible steps): the FIRST sample that carries a label owns that table row and adds the contributions
// of every sample with the same label in sample order.  Grid: (D / 256, n).
__global__ void cond_combine_bwd_kernel(const float* __restrict__ c, const float* __restrict__ dcs, const float* __restrict__ dcd,
                                        const long* __restrict__ y, bf16_t* __restrict__ dtemb, float* __restrict__ dtable,
                                        int n, int D, int table_rows) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (d >= D) return;
    const int i = b * D + d;
    const float dc = (dcs[i] * dmpsilu_f(c[i]) + dcd[i]) * C5;
    dtemb[i] = cvt16(dc);
    const long label = y[b];
    if (label < 0 || label >= table_rows) {              // the forward clamped it and flagged the call; never write out of bounds
        g_dev_error_embed = MAPDIT_DEVERR_LABEL;
        return;
    }
    for (int o = 0; o < b; ++o)
        if (y[o] == label) return;                       // an earlier sample owns this row (uniform over the block)
    float sum = dc;
    for (int o = b + 1; o < n; ++o)
        if (y[o] == label) {
            const int k = o * D + d;
            sum += (dcs[k] * dmpsilu_f(c[k]) + dcd[k]) * C5;
        }
    dtable[(size_t)label * D + d] += sum;
}

// ---- final layer tail: MPScale gates + unpatchify + concat (final_layer.py:20-22,57-59; dit.py:96-101) ---------------
// lin [M][ldl] fp32 (mean chunk at columns 0..P-1, sigma chunk at P..2P-1); a_* [N][8] fp32; out [N,2C,S,S].
__device__ __forceinline__ float gate_of(const float* a, const float* ref) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j] * ref[j];
    s *= 0.35355339059327379f;   // 1/sqrt(8)
    return 1.f / (1.f + __expf(-s));
}
__global__ void final_out_kernel(const float* __restrict__ lin, int ldl, const float* __restrict__ a_mean,
                                 const float* __restrict__ a_sigma, const float* __restrict__ ref_mean,
                                 const float* __restrict__ ref_sigma, float* __restrict__ out, int N, int C, int S, int p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * 2 * C * S * S;
    if (i >= total) return;
    const int xx = (int)(i % S), yy = (int)((i / S) % S), ch = (int)((i / ((long)S * S)) % (2 * C)), n = (int)(i / ((long)S * S * 2 * C));
    const int grid = S / p, P = p * p * C;
    const int t = (yy / p) * grid + xx / p;
    const int c = ch % C, chunk = ch / C;
    const int j = ((yy % p) * p + (xx % p)) * C + c;
    const float g = chunk == 0 ? gate_of(a_mean + n * 8, ref_mean) : gate_of(a_sigma + n * 8, ref_sigma);
    out[i] = lin[((size_t)n * grid * grid + t) * ldl + chunk * P + j] * g;
}

// Backward: one block per sample.  dlin [M][ldd] bf16 = dout*gate (patchified; columns >= 2P stay zero),
// da_*[n][8] (bf16 operand); the per-sample terms of dref_* go to dref_part [N][2][8] and are added in sample order by
// sum_dref_kernel (no atomics: bit-reproducible steps).
__global__ __launch_bounds__(256) void final_out_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ lin, int ldl,
                                                          const float* __restrict__ a_mean, const float* __restrict__ a_sigma,
                                                          const float* __restrict__ ref_mean, const float* __restrict__ ref_sigma,
                                                          bf16_t* __restrict__ dlin, int ldd, bf16_t* __restrict__ da_bf,
                                                          float* __restrict__ dref_part, int C, int S, int p, float gscale) {
    __shared__ float red[2][4];
    const int n = blockIdx.x;
    const int grid = S / p, P = p * p * C, T = grid * grid;
    const float gm = gate_of(a_mean + n * 8, ref_mean), gs = gate_of(a_sigma + n * 8, ref_sigma);
    float sm = 0.f, ss = 0.f;
    const int per = 2 * C * S * S;
    for (int e = threadIdx.x; e < per; e += 256) {
        const int xx = e % S, yy = (e / S) % S, ch = e / (S * S);
        const int t = (yy / p) * grid + xx / p, c = ch % C, chunk = ch / C;
        const int j = ((yy % p) * p + (xx % p)) * C + c;
        const float go = dout[(size_t)n * per + e];
        const float l = lin[((size_t)n * T + t) * ldl + chunk * P + j];
        if (chunk == 0) sm += go * l; else ss += go * l;
        dlin[((size_t)n * T + t) * ldd + chunk * P + j] = cvt16(gscale * go * (chunk == 0 ? gm : gs));
    }
    sm = wave_sum(sm);
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sm; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int which = threadIdx.x >> 3, j = threadIdx.x & 7;
        const float dg = red[which][0] + red[which][1] + red[which][2] + red[which][3];
        const float g = which == 0 ? gm : gs;
        const float dang = dg * g * (1.f - g) * 0.35355339059327379f;
        const float* ref = which == 0 ? ref_mean : ref_sigma;
        const float* a = (which == 0 ? a_mean : a_sigma) + n * 8;
        da_bf[((size_t)which * gridDim.x + n) * 8 + j] = cvt16(gscale * dang * ref[j]);
        dref_part[((size_t)n * 2 + which) * 8 + j] = dang * a[j];
    }
}
__global__ void sum_dref_kernel(const float* __restrict__ part, int N, float* __restrict__ dref_mean, float* __restrict__ dref_sigma) {
    const int which = threadIdx.x >> 3, j = threadIdx.x & 7;
    if (threadIdx.x >= 16) return;
    float a = 0.f;
    int n = 0;
    for (; n + 16 <= N; n += 16) {            // sample order kept (bit-reproducible); 16 loads in flight instead of a load-add chain (42 -> ~6 us)
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = part[((size_t)(n + k) * 2 + which) * 8 + j];
#pragma unroll
        for (int k = 0; k < 16; ++k) a += v[k];
    }
    for (; n < N; ++n) a += part[((size_t)n * 2 + which) * 8 + j];
    (which == 0 ? dref_mean : dref_sigma)[j] += a;
}

// forward_with_cfg tail (reference src/dit.py:113-118): eps = u + s (c - u) on the first C channels, both halves.
__global__ void cfg_combine_kernel(const float* __restrict__ in, float* __restrict__ out, int n_total, int C, int HW, float s) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)2 * C * HW;
    if (i >= (long)n_total * per) return;
    const int n = (int)(i / per);
    const long e = i % per;
    if (e >= (long)C * HW) { out[i] = in[i]; return; }
    const int half = n_total / 2, nc = n % half;
    const float cond = in[(long)nc * per + e], unc = in[(long)(nc + half) * per + e];
    out[i] = unc + s * (cond - unc);
}

MD_NS_CLOSE

extern "C" int MD_SYM(patch_embed_fwd)(const float* x, const float* w_eff, const float* pos, float* out, uint16_t* patches,
                                      int ldp, int N, int C, int S, int p, int D, float out_scale, void* stream) {
    const float c5 = out_scale > 0.f ? out_scale : C5;       // 0: mp_sum(., ., 0.5) of the snapshot (dit.py:84); 1: the plain sum
    MD_CHECK(x && w_eff && pos && out, "patch_embed_fwd: null argument");
    MD_CHECK(S % p == 0 && D % 128 == 0, "patch_embed_fwd: S=%d p=%d D=%d unsupported", S, p, D);
    const int P1 = p * p * C + 1, T = (S / p) * (S / p);
    MD_CHECK(!patches || ldp >= P1, "patch_embed_fwd: ldp=%d too small", ldp);
    const long M = (long)N * T;
    const size_t shm = (size_t)(P1 * 128 + 64 * P1) * 4;
    if (shm <= 64 * 1024) {
        hipLaunchKernelGGL(patch_embed_fwd_kernel<128>, dim3(cdiv(M, 64), P1 == 17 ? 1 : D / 128), dim3(256), shm, (hipStream_t)stream, x, w_eff,
                           pos, out, patches, ldp, C, S, p, D, M, c5);
    } else {                                   // patch-8 models: 257-wide rows, 64-feature tiles, > 64 KiB of LDS
        const size_t shm64 = (size_t)(P1 * 64 + 64 * P1) * 4;
        MD_CHECK(shm64 <= 150 * 1024, "patch_embed_fwd: patch dim %d too large", P1 - 1);
        (void)hipFuncSetAttribute((const void*)patch_embed_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm64);
        hipLaunchKernelGGL(patch_embed_fwd_kernel<64>, dim3(cdiv(M, 64), D / 64), dim3(256), shm64, (hipStream_t)stream, x, w_eff,
                           pos, out, patches, ldp, C, S, p, D, M, c5);
    }
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

#if MAPDIT_DT == 0
extern "C" int mapdit_fourier_fwd(const int64_t* t, const float* scale, const float* shift, uint16_t* out, int n, int F,
                                  void* stream) {
    MD_CHECK(t && scale && shift && out && n > 0, "fourier_fwd: null/empty argument");
    hipLaunchKernelGGL(fourier_kernel, dim3(cdiv((long)n * F, 256)), dim3(256), 0, (hipStream_t)stream, (const long*)t, scale,
                       shift, out, n, F);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
#endif

extern "C" int MD_SYM(cond_combine_fwd)(const float* temb, const float* table, const int64_t* y, float* c, uint16_t* c_silu,
                                       uint16_t* c_bf, int n, int D, int table_rows, void* stream) {
    MD_CHECK(temb && table && y && c && c_silu && c_bf && n > 0 && table_rows > 0, "cond_combine_fwd: null/empty argument");
    hipLaunchKernelGGL(cond_combine_kernel, dim3(cdiv((long)n * D, 256)), dim3(256), 0, (hipStream_t)stream, temb, table,
                       (const long*)y, c, c_silu, c_bf, n, D, table_rows);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(cond_combine_bwd)(const float* c, const float* dcs, const float* dcd, const int64_t* y, uint16_t* dtemb,
                                       float* dtable, int n, int D, int table_rows, void* stream) {
    MD_CHECK(c && dcs && dcd && y && dtemb && dtable && n > 0 && table_rows > 0, "cond_combine_bwd: null/empty argument");
    hipLaunchKernelGGL(cond_combine_bwd_kernel, dim3(cdiv(D, 256), n), dim3(256), 0, (hipStream_t)stream, c, dcs, dcd,
                       (const long*)y, dtemb, dtable, n, D, table_rows);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

#if MAPDIT_DT == 0
extern "C" int mapdit_final_out_fwd(const float* lin, int ldl, const float* a_mean, const float* a_sigma, const float* ref_mean,
                                    const float* ref_sigma, float* out, int N, int C, int S, int p, void* stream) {
    MD_CHECK(lin && a_mean && a_sigma && ref_mean && ref_sigma && out, "final_out_fwd: null argument");
    const long total = (long)N * 2 * C * S * S;
    hipLaunchKernelGGL(final_out_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, lin, ldl, a_mean, a_sigma,
                       ref_mean, ref_sigma, out, N, C, S, p);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
#endif

extern "C" int MD_SYM(final_out_bwd)(const float* dout, const float* lin, int ldl, const float* a_mean, const float* a_sigma,
                                    const float* ref_mean, const float* ref_sigma, uint16_t* dlin, int ldd, uint16_t* da_bf,
                                    float* dref_part, float* dref_mean, float* dref_sigma, float grad_scale, int N, int C, int S,
                                    int p, void* stream) {
    MD_CHECK(dout && lin && a_mean && a_sigma && ref_mean && ref_sigma && dlin && da_bf && dref_part && dref_mean && dref_sigma,
             "final_out_bwd: null argument");
    hipLaunchKernelGGL(final_out_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, dout, lin, ldl, a_mean, a_sigma,
                       ref_mean, ref_sigma, dlin, ldd, da_bf, dref_part, C, S, p, grad_scale);
    MD_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_dref_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dref_part, N, dref_mean, dref_sigma);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

#if MAPDIT_DT == 0
extern "C" int mapdit_cfg_combine(const float* model_out, float* out, int n_total, int C, int HW, float cfg_scale, void* stream) {
    MD_CHECK(model_out && out && n_total > 0 && n_total % 2 == 0, "cfg_combine: batch must be even and non-empty");
    const long total = (long)n_total * 2 * C * HW;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, model_out, out, n_total, C,
                       HW, cfg_scale);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
#endif
