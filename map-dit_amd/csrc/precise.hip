// Building blocks of the fp32-accurate forward and backward ("bf16x3" precision, mapdit_config_t.precision = MAPDIT_PREC_BF16X3).
//
// The reference computes in fp32.  The production path feeds the MFMA GEMMs bf16 operands, which leaves
// ~5e-3 .. 1e-2 between its logits and the reference's.  This mode keeps every activation in fp32 and runs THE SAME
// MFMA GEMM kernel on operands split into two bf16 terms, x = hi + lo (hi = bf16(x), lo = bf16(x - hi), 16 mantissa
// bits together):
//     A B^T  ~=  Ahi Bhi^T + Ahi Blo^T + Alo Bhi^T          (the dropped Alo Blo^T term is ~2^-18 relative)
// which is ONE bf16 GEMM with the operands concatenated along K:  A' = [Ahi | Ahi | Alo],  B' = [Bhi | Blo | Bhi],
// K' = 3K, fp32 accumulation.  Everything between the GEMMs (MP-SiLU, modulate, residual mp_sum, cosine
// normalisation, attention softmax) runs as plain fp32 kernels below with accurate expf / cosf.  The backward uses
// the same trick with the parts laid out along ITS reduction index (engine.hip backward_precise): stacked row blocks for the
// K-major operands of the NN (dX = dY W) and TN (dW = dY^T X) products.  3x the GEMM work and unfused pointwise passes: a
// parity instrument (logits within 1e-3 of the reference, measured ~1e-5; gradients ~1e-5 .. 1e-4), not the fast path.
#include "common.h"
#include "precise.h"

MAPDIT_DEFINE_DEV_ERROR(precise)

namespace {

__device__ __forceinline__ float mpsilu_exact(float x) { return x / (1.f + expf(-x)) * (1.f / MP_SILU_DIV); }

// src fp32 [rows][K] (row stride ld) -> bf16 split image; pattern 0 (A operand): hi, hi, lo;  1 (B operand): hi, lo, hi.
// stack == 0: the three parts side by side, dst [rows][3K] (reduction index = the column index: NT / row-major operands);
// stack != 0: the three parts on top of each other, dst [3 rows][ldd] (reduction index = the row index: K-major operands of
// the NN / TN products of the backward).
__global__ void split3_kernel(const float* __restrict__ src, long ld, bf16_t* __restrict__ dst, long rows, int K, int pattern,
                              int op, int stack, long ldd) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * K) return;
    const long r = i / K;
    const int k = (int)(i % K);
    float x = src[r * ld + k];
    if (op == MAPDIT_SPLIT_OP_MPSILU) x = mpsilu_exact(x);
    const bf16_t hi = f2bf(x);
    const bf16_t lo = f2bf(x - bf2f(hi));
    const bf16_t p1 = pattern ? lo : hi, p2 = pattern ? hi : lo;
    if (stack) {
        dst[r * ldd + k] = hi;
        dst[(rows + r) * ldd + k] = p1;
        dst[(2 * rows + r) * ldd + k] = p2;
    } else {
        bf16_t* d = dst + r * 3 * K + k;
        d[0] = hi;
        d[K] = p1;
        d[2 * K] = p2;
    }
}

__device__ __forceinline__ float dmpsilu_exact(float x) {
    const float sg = 1.f / (1.f + expf(-x));
    return sg * (1.f + x * (1.f - sg)) * (1.f / MP_SILU_DIV);
}

// ---- backward building blocks (all fp32, one thread per output element or per (sample, column); deterministic) ---------------
// out = a * d/dh [silu(h) / 0.596]
__global__ void dsilu32_kernel(const float* __restrict__ a, const float* __restrict__ h, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * dmpsilu_exact(h[i]);
}

// The fp32 twin of resid_mod_bwd (pointwise.hip): one thread per (sample, column), looping over the sample's tokens.
//   dx = ca dxo + ka scale dxm ;  dscale = sum_t ka x dxm ; dshift = sum_t kb dxm ; gpart[n, d] = sum_t dxm (shift - x scale) / den
//   dy_up = cb g_up dx ;  dg_up = sum_t cb y_up dx          (ka = (1-g)/den, kb = g/den, den detached: SURVEY F8)
struct Rmb32P {
    const float *dxo, *dxm, *x, *shift, *scale, *gain, *y_up, *g_up;
    float *dx, *dshift, *dscale, *gpart, *dy_up, *dg_up;
    int ldmod, ldg_up, ldd, ldd_up, N, T, D;
    float ca, cb;
};
__global__ void rmb32_kernel(Rmb32P p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)p.N * p.D) return;
    const int n = (int)(i / p.D), d = (int)(i % p.D);
    float g = 0.f, den = 1.f, sc = 0.f, sh = 0.f, gu = 0.f;
    if (p.dxm) {
        g = *p.gain;
        den = sqrtf((1.f - g) * (1.f - g) + g * g);
        sc = p.scale[(size_t)n * p.ldmod + d];
        sh = p.shift[(size_t)n * p.ldmod + d];
    }
    if (p.y_up) gu = p.g_up[(size_t)n * p.ldg_up + d];
    const float ka = (1.f - g) / den, kb = g / den;
    float a_sc = 0.f, a_sh = 0.f, a_g = 0.f, a_gu = 0.f;
    for (int t = 0; t < p.T; ++t) {
        const size_t off = ((size_t)n * p.T + t) * p.D + d;
        float dx = p.dxo ? p.ca * p.dxo[off] : 0.f;
        if (p.dxm) {
            const float dm = p.dxm[off], xx = p.x[off];
            dx += ka * sc * dm;
            a_sc += ka * xx * dm;
            a_sh += kb * dm;
            a_g += dm * (sh - xx * sc) / den;
        }
        if (p.dx) p.dx[off] = dx;
        if (p.y_up) {
            p.dy_up[off] = p.cb * gu * dx;
            a_gu += p.cb * p.y_up[off] * dx;
        }
    }
    if (p.dxm) {
        p.dscale[(size_t)n * p.ldd + d] = a_sc;
        p.dshift[(size_t)n * p.ldd + d] = a_sh;
        p.gpart[i] = a_g;
    }
    if (p.y_up) p.dg_up[(size_t)n * p.ldd_up + d] = a_gu;
}

// Attention backward with the probability matrices spelled out (scratch P, dS: [B*H][T][T] fp32 each).
// probs: one workgroup per head, thread j = key j with k^_j and v_j in registers; the queries pass through LDS 16 rows at a time
// (every thread reads the same q^_i / dO_i element: a broadcast) and every P / dS row is written as one contiguous run over j:
//   P[i][j] = exp(q^_i . k^_j scale) / sum_j(...),   dS[i][j] = P[i][j] (dO_i . v_j - dO_i . O_i) scale
constexpr int PB_QT = 16;
template <int PB_HD>                                       // register / LDS row size: the smallest of 32, 64, 96 that holds head_dim
__global__ __launch_bounds__(256) void attn32_probs_kernel(const float* __restrict__ qn, const float* __restrict__ kn,
                                                         const float* __restrict__ v, const float* __restrict__ dO,
                                                         const float* __restrict__ O, float* __restrict__ P, float* __restrict__ dS,
                                                         int T, int H, int hd, float scale) {
    __shared__ float qs[PB_QT][PB_HD], gs[PB_QT][PB_HD], part[PB_QT][4], dl[PB_QT];
    const size_t bh = blockIdx.x;
    const int b = (int)(bh / H), hh = (int)(bh % H), D = H * hd;
    const int j = threadIdx.x, lane = j & 63, wave = j >> 6;
    const bool act = j < T;
    float kr[PB_HD], vr[PB_HD];
#pragma unroll
    for (int d = 0; d < PB_HD; ++d) {
        const bool in = act && d < hd;
        kr[d] = in ? kn[(bh * T + j) * hd + d] : 0.f;
        vr[d] = in ? v[(bh * T + j) * hd + d] : 0.f;
    }
    for (int i0 = 0; i0 < T; i0 += PB_QT) {
        __syncthreads();
        for (int e = threadIdx.x; e < PB_QT * hd; e += 256) {
            const int r = e / hd, d = e - r * hd, i = i0 + r;
            qs[r][d] = i < T ? qn[(bh * T + i) * hd + d] : 0.f;
            gs[r][d] = i < T ? dO[((size_t)b * T + i) * D + hh * hd + d] : 0.f;
        }
        if (threadIdx.x < PB_QT) {
            const int i = i0 + threadIdx.x;
            float a = 0.f;
            if (i < T)
                for (int d = 0; d < hd; ++d) a += dO[((size_t)b * T + i) * D + hh * hd + d] * O[((size_t)b * T + i) * D + hh * hd + d];
            dl[threadIdx.x] = a;
        }
        __syncthreads();
        float ex[PB_QT], dp[PB_QT];
#pragma unroll
        for (int r = 0; r < PB_QT; ++r) {
            float sdot = 0.f, g = 0.f;
#pragma unroll
            for (int d = 0; d < PB_HD; ++d)
                if (d < hd) { sdot += qs[r][d] * kr[d]; g += gs[r][d] * vr[d]; }
            ex[r] = act ? expf(sdot * scale) : 0.f;
            dp[r] = g;
            const float ws = wave_sum(ex[r]);
            if (lane == 0) part[r][wave] = ws;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PB_QT; ++r) {
            const int i = i0 + r;
            if (i < T && act) {
                const float l = (part[r][0] + part[r][1]) + (part[r][2] + part[r][3]);
                const float pj = ex[r] / l;
                P[(bh * T + i) * T + j] = pj;
                dS[(bh * T + i) * T + j] = pj * (dp[r] - dl[r]) * scale;
            }
        }
    }
}
// dq^_i = sum_j dS[i][j] k^_j : one thread per (head, query, d)
__global__ void attn32_dq_kernel(const float* __restrict__ dS, const float* __restrict__ kn, float* __restrict__ dqn, int T, int hd,
                                 long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int d = (int)(id % hd);
    const long row = id / hd, bh = row / T;          // row = bh*T + i
    const float* ds = dS + row * T;
    float a = 0.f;
    for (int j = 0; j < T; ++j) a += ds[j] * kn[(bh * T + j) * hd + d];
    dqn[id] = a;
}
// dk^_j = sum_i dS[i][j] q^_i ; dv_j = sum_i P[i][j] dO_i : one thread per (head, key, d)
__global__ void attn32_dkv_kernel(const float* __restrict__ P, const float* __restrict__ dS, const float* __restrict__ qn,
                                  const float* __restrict__ dO, float* __restrict__ dkn, float* __restrict__ dv, int T, int H, int hd,
                                  long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int d = (int)(id % hd);
    const long row = id / hd, bh = row / T;
    const int j = (int)(row % T), b = (int)(bh / H), hh = (int)(bh % H), D = H * hd;
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < T; ++i) {
        const size_t pij = (bh * T + i) * T + j;
        ak += dS[pij] * qn[(bh * T + i) * hd + d];
        av += P[pij] * dO[((size_t)b * T + i) * D + hh * hd + d];
    }
    dkn[id] = ak;
    dv[id] = av;
}
// backward of qkv_split32: dq = s dq^ - q^ (dq^ . q^) / (sqrt(hd) |q|), |q| = sqrt(hd)/s - eps; v passes through.  One thread per
// (token, head); dqkv [M, 3D] fp32.
__global__ void qkv_merge_bwd32_kernel(const float* __restrict__ qn, const float* __restrict__ kn, const float* __restrict__ scales,
                                       const float* __restrict__ dqn, const float* __restrict__ dkn, const float* __restrict__ dv,
                                       float* __restrict__ dqkv, int B, int T, int H, int hd) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)B * T * H) return;
    const int h = (int)(id % H);
    const long m = id / H;
    const int t = (int)(m % T), b = (int)(m / T);
    const int D = H * hd;
    const float rt = sqrtf((float)hd);
    const size_t hr = (((size_t)b * H + h) * T + t);
    for (int which = 0; which < 3; ++which) {
        float* dst = dqkv + m * 3 * D + which * D + h * hd;
        const float* g = (which == 0 ? dqn : which == 1 ? dkn : dv) + hr * hd;
        if (which == 2) {
            for (int d = 0; d < hd; ++d) dst[d] = g[d];
            continue;
        }
        const float* xh = (which == 0 ? qn : kn) + hr * hd;
        const float s = scales[(size_t)which * B * H * T + hr], n = rt / s - NORM_EPS;
        float dot = 0.f;
        for (int d = 0; d < hd; ++d) dot += g[d] * xh[d];
        const float c = dot / (rt * fmaxf(n, 1e-30f));
        for (int d = 0; d < hd; ++d) dst[d] = s * g[d] - xh[d] * c;
    }
}

__global__ void fourier32_kernel(const long* __restrict__ t, const float* __restrict__ scale, const float* __restrict__ shift,
                                 float* __restrict__ out, int n, int F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * F) return;
    const int b = i / F, f = i % F;
    float prod = (float)t[b] * scale[f];
    asm volatile("" : "+v"(prod));      // keep torch's two roundings (see embed.hip fourier_kernel)
    out[i] = 1.41421356237309515f * cosf(prod + shift[f]);
}

// modulate (src/utils.py:11-16) in fp32: out = ((1-g) x scale + g shift) / sqrt((1-g)^2 + g^2), per-sample shift/scale
__global__ void modulate32_kernel(const float* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ scale,
                                  int ldmod, const float* __restrict__ gain, float* __restrict__ out, int T, int D, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long m = i / D;
    const int d = (int)(i % D), n = (int)(m / T);
    const float g = *gain, den = sqrtf((1.f - g) * (1.f - g) + g * g);
    const float a = x[i] * scale[(size_t)n * ldmod + d], b = shift[(size_t)n * ldmod + d];
    out[i] = (a + g * (b - a)) / den;
}

// residual mp_sum (src/blocks/dit_block.py:35-36): xout = lerp(xin, gate*y, 0.3) / sqrt(0.7^2 + 0.3^2)
__global__ void resid32_kernel(const float* __restrict__ xin, const float* __restrict__ y, const float* __restrict__ gate, int ldg,
                               float* __restrict__ xout, int T, int D, long total, float tt, float inv_den) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long m = i / D;
    const int d = (int)(i % D), n = (int)(m / T);
    const float a = xin[i], b = gate[(size_t)n * ldg + d] * y[i];
    xout[i] = (a + tt * (b - a)) * inv_den;
}

// qkv fp32 [M,3D] -> q^, k^ (cosine normalised, src/layers/attention.py:43) and v as [B*H][T][hd] fp32
__global__ void qkv_split32_kernel(const float* __restrict__ qkv, int B, int T, int H, int hd, float* __restrict__ qn,
                                   float* __restrict__ kn, float* __restrict__ v, float* __restrict__ scales) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)B * T * H) return;
    const int h = (int)(id % H);
    const long m = id / H;
    const int t = (int)(m % T), b = (int)(m / T);
    const int D = H * hd;
    const float rt = sqrtf((float)hd);
    for (int which = 0; which < 3; ++which) {
        const float* src = qkv + m * 3 * D + which * D + h * hd;
        float* dst = (which == 0 ? qn : which == 1 ? kn : v) + (((size_t)b * H + h) * T + t) * hd;
        float s = 1.f;
        if (which < 2) {
            float ss = 0.f;
            for (int d = 0; d < hd; ++d) ss += src[d] * src[d];
            s = rt / (sqrtf(ss) + NORM_EPS);
            if (scales) scales[(size_t)which * B * H * T + ((size_t)b * H + h) * T + t] = s;
        }
        for (int d = 0; d < hd; ++d) dst[d] = src[d] * s;
    }
}

// fp32 attention: one thread per query row, K/V streamed through LDS 64 keys at a time.  softmax with running maximum
// is unnecessary (|logit| <= sqrt(hd)); expf, not the fast intrinsic.
constexpr int PA_CHUNK = 64;
template <int PA_HD>                                       // as above
__global__ __launch_bounds__(256) void attn32_kernel(const float* __restrict__ qn, const float* __restrict__ kn,
                                                   const float* __restrict__ v, float* __restrict__ o, int T, int H, int hd,
                                                   float scale) {
    __shared__ float ks[PA_CHUNK][PA_HD + 1];
    __shared__ float vs[PA_CHUNK][PA_HD + 1];
    const size_t bh = blockIdx.x;
    const int b = (int)(bh / H), hh = (int)(bh % H), D = H * hd;
    const int i = threadIdx.x;                        // query row (T <= 256 = blockDim)
    float q[PA_HD], acc[PA_HD];
#pragma unroll
    for (int d = 0; d < PA_HD; ++d) { q[d] = (i < T && d < hd) ? qn[(bh * T + i) * hd + d] : 0.f; acc[d] = 0.f; }
    float l = 0.f;
    for (int j0 = 0; j0 < T; j0 += PA_CHUNK) {
        const int nj = min(PA_CHUNK, T - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < nj * hd; e += blockDim.x) {
            const int r = e / hd, d = e % hd;
            ks[r][d] = kn[(bh * T + j0 + r) * hd + d];
            vs[r][d] = v[(bh * T + j0 + r) * hd + d];
        }
        __syncthreads();
        if (i < T) {
            for (int j = 0; j < nj; ++j) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < PA_HD; ++d) if (d < hd) s += q[d] * ks[j][d];
                const float p = expf(s * scale);
                l += p;
#pragma unroll
                for (int d = 0; d < PA_HD; ++d) if (d < hd) acc[d] += p * vs[j][d];
            }
        }
    }
    if (i < T) {
        const float il = 1.f / l;
#pragma unroll
        for (int d = 0; d < PA_HD; ++d) if (d < hd) o[((size_t)b * T + i) * D + hh * hd + d] = acc[d] * il;
    }
}

// fp32 twins of the final-layer / conditioning / patch kernels of embed.hip whose bf16 outputs feed GEMMs there.
__device__ __forceinline__ float gate32(const float* a, const float* ref) {
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += a[j] * ref[j];
    return 1.f / (1.f + expf(-s * 0.35355339059327379f));
}
// one workgroup per sample: dlin [M][ldd] = dout * gate (patchified), da [2][N][8], dref accumulated over samples by thread 0
// of each sample in a fixed order via a per-sample scratch (dref_part [N][2][8]) summed by the caller's reduce kernel.
__global__ __launch_bounds__(256) void final_out_bwd32_kernel(const float* __restrict__ dout, const float* __restrict__ lin, int ldl,
                                                            const float* __restrict__ a_mean, const float* __restrict__ a_sigma,
                                                            const float* __restrict__ ref_mean, const float* __restrict__ ref_sigma,
                                                            float* __restrict__ dlin, int ldd, float* __restrict__ da,
                                                            float* __restrict__ dref_part, int C, int S, int p) {
    __shared__ float red[2][4];
    const int n = blockIdx.x;
    const int grid = S / p, P = p * p * C, T = grid * grid;
    const float gm = gate32(a_mean + n * 8, ref_mean), gs = gate32(a_sigma + n * 8, ref_sigma);
    float sm = 0.f, ss = 0.f;
    const int per = 2 * C * S * S;
    for (int e = threadIdx.x; e < per; e += 256) {
        const int xx = e % S, yy = (e / S) % S, ch = e / (S * S);
        const int t = (yy / p) * grid + xx / p, c = ch % C, chunk = ch / C;
        const int j = ((yy % p) * p + (xx % p)) * C + c;
        const float go = dout[(size_t)n * per + e];
        const float l = lin[((size_t)n * T + t) * ldl + chunk * P + j];
        if (chunk == 0) sm += go * l; else ss += go * l;
        dlin[((size_t)n * T + t) * ldd + chunk * P + j] = go * (chunk == 0 ? gm : gs);
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
        da[((size_t)which * gridDim.x + n) * 8 + j] = dang * ref[j];
        dref_part[((size_t)n * 2 + which) * 8 + j] = dang * a[j];
    }
}
// dref[which][j] = sum_n dref_part[n][which][j]  (16 threads, fixed order)
__global__ void sum_dref_kernel(const float* __restrict__ part, int N, float* __restrict__ dref_mean, float* __restrict__ dref_sigma) {
    const int which = threadIdx.x >> 3, j = threadIdx.x & 7;
    if (threadIdx.x >= 16) return;
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += part[((size_t)n * 2 + which) * 8 + j];
    (which == 0 ? dref_mean : dref_sigma)[j] = a;
}
// dc = (dcs dmpsilu(c) + dcd) C5 ; dtemb = dc (fp32) ; dtable[y[b]] += dc: the first sample that carries a label owns the row and
// adds every sample with that label in sample order (as cond_combine_bwd_kernel in embed.hip).  Grid: (D / 256, n).
__global__ void cond_combine_bwd32_kernel(const float* __restrict__ c, const float* __restrict__ dcs, const float* __restrict__ dcd,
                                          const long* __restrict__ y, float* __restrict__ dtemb, float* __restrict__ dtable, int n,
                                          int D, int table_rows) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (d >= D) return;
    const int i = b * D + d;
    const float dc = (dcs[i] * dmpsilu_exact(c[i]) + dcd[i]) * 0.70710678118654752f;
    dtemb[i] = dc;
    const long label = y[b];
    if (label < 0 || label >= table_rows) {
        g_dev_error_precise = MAPDIT_DEVERR_LABEL;
        return;
    }
    for (int o = 0; o < b; ++o)
        if (y[o] == label) return;
    float sum = dc;
    for (int o = b + 1; o < n; ++o)
        if (y[o] == label) {
            const int k = o * D + d;
            sum += (dcs[k] * dmpsilu_exact(c[k]) + dcd[k]) * 0.70710678118654752f;
        }
    dtable[(size_t)label * D + d] += sum;
}
// patchify + ones column as fp32 [M][ldp] (zero padded): the x operand of the x_embedder weight gradient
__global__ void patchify32_kernel(const float* __restrict__ x, float* __restrict__ patches, int ldp, int C, int S, int p, long M) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * ldp) return;
    const long m = i / ldp;
    const int j = (int)(i % ldp);
    const int P = p * p * C, grid = S / p, T = grid * grid;
    float v = 0.f;
    if (j == P) v = 1.f;
    else if (j < P) {
        const int n = (int)(m / T), t = (int)(m % T), hy = t / grid, wx = t % grid;
        const int c = j % C, p2 = (j / C) % p, p1 = j / (C * p);
        v = x[(((size_t)n * C + c) * S + hy * p + p1) * S + wx * p + p2];
    }
    patches[i] = v;
}
// out[i] = alpha * in[i]  /  out[i] += in[i]
__global__ void axpby32_kernel(const float* __restrict__ in, float* __restrict__ out, long n, float alpha, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = accumulate ? out[i] + alpha * in[i] : alpha * in[i];
}

}  // namespace

#define P32_LAUNCH(KERNEL, TOTAL, ...)                                                                                     \
    hipLaunchKernelGGL(KERNEL, dim3(cdiv((long)(TOTAL), 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);           \
    MD_LAUNCH_CHECK();                                                                                                     \
    return MAPDIT_OK

int mapdit_split3_stack(const float* src, long ld, uint16_t* dst, long ldd, long rows, int K, int pattern, int op, void* stream) {
    MD_CHECK(src && dst && rows > 0 && K > 0 && ld >= K && ldd >= K, "split3_stack: bad argument");
    P32_LAUNCH(split3_kernel, rows * K, src, ld, dst, rows, K, pattern, op, 1, ldd);
}
int mapdit_dsilu32(const float* a, const float* h, float* out, long n, void* stream) {
    MD_CHECK(a && h && out && n > 0, "dsilu32: bad argument");
    P32_LAUNCH(dsilu32_kernel, n, a, h, out, n);
}
int mapdit_rmb32(const mapdit_rmb32_t* a, void* stream) {
    MD_CHECK(a && (a->dxo || a->dxm), "rmb32: bad argument");
    Rmb32P p{a->dxo, a->dxm, a->x, a->shift, a->scale, a->gain, a->y_up, a->g_up, a->dx, a->dshift, a->dscale, a->gpart, a->dy_up,
             a->dg_up, a->ldmod, a->ldg_up, a->ldd, a->ldd_up, a->N, a->T, a->D, a->ca, a->cb};
    P32_LAUNCH(rmb32_kernel, (long)a->N * a->D, p);
}
int mapdit_attn32_bwd(const float* qn, const float* kn, const float* v, const float* dO, const float* O, float* P, float* dS,
                      float* dqn, float* dkn, float* dv, int B, int T, int H, int hd, void* stream) {
    MD_CHECK(qn && kn && v && dO && O && P && dS && dqn && dkn && dv && B > 0, "attn32_bwd: bad argument");
    const long rows = (long)B * H * T;
    MD_CHECK(T <= 256 && hd <= 96, "attn32_bwd: T=%d, head_dim=%d unsupported (<= 256, <= 96)", T, hd);
    if (hd <= 32)
        hipLaunchKernelGGL(attn32_probs_kernel<32>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, qn, kn, v, dO, O, P, dS, T, H, hd,
                           1.f / sqrtf((float)hd));
    else if (hd <= 64)
        hipLaunchKernelGGL(attn32_probs_kernel<64>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, qn, kn, v, dO, O, P, dS, T, H, hd,
                           1.f / sqrtf((float)hd));
    else
        hipLaunchKernelGGL(attn32_probs_kernel<96>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, qn, kn, v, dO, O, P, dS, T, H, hd,
                           1.f / sqrtf((float)hd));
    MD_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn32_dq_kernel, dim3(cdiv(rows * hd, 256)), dim3(256), 0, (hipStream_t)stream, dS, kn, dqn, T, hd, rows * hd);
    MD_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn32_dkv_kernel, dim3(cdiv(rows * hd, 256)), dim3(256), 0, (hipStream_t)stream, P, dS, qn, dO, dkn, dv, T, H, hd,
                       rows * hd);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
int mapdit_qkv_merge_bwd32(const float* qn, const float* kn, const float* scales, const float* dqn, const float* dkn, const float* dv,
                           float* dqkv, int B, int T, int H, int hd, void* stream) {
    MD_CHECK(qn && kn && scales && dqn && dkn && dv && dqkv && B > 0, "qkv_merge_bwd32: bad argument");
    P32_LAUNCH(qkv_merge_bwd32_kernel, (long)B * T * H, qn, kn, scales, dqn, dkn, dv, dqkv, B, T, H, hd);
}
int mapdit_final_out_bwd32(const float* dout, const float* lin, int ldl, const float* a_mean, const float* a_sigma, const float* ref_mean,
                           const float* ref_sigma, float* dlin, int ldd, float* da, float* dref_part, float* dref_mean,
                           float* dref_sigma, int N, int C, int S, int p, void* stream) {
    MD_CHECK(dout && lin && dlin && da && dref_part && dref_mean && dref_sigma && N > 0, "final_out_bwd32: bad argument");
    hipLaunchKernelGGL(final_out_bwd32_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, dout, lin, ldl, a_mean, a_sigma, ref_mean,
                       ref_sigma, dlin, ldd, da, dref_part, C, S, p);
    MD_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_dref_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dref_part, N, dref_mean, dref_sigma);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
int mapdit_cond_combine_bwd32(const float* c, const float* dcs, const float* dcd, const int64_t* y, float* dtemb, float* dtable, int n,
                              int D, int table_rows, void* stream) {
    MD_CHECK(c && dcs && dcd && y && dtemb && dtable && n > 0 && table_rows > 0, "cond_combine_bwd32: bad argument");
    hipLaunchKernelGGL(cond_combine_bwd32_kernel, dim3(cdiv(D, 256), n), dim3(256), 0, (hipStream_t)stream, c, dcs, dcd, (const long*)y,
                       dtemb, dtable, n, D, table_rows);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
int mapdit_patchify32(const float* x, float* patches, int ldp, int N, int C, int S, int p, void* stream) {
    MD_CHECK(x && patches && N > 0, "patchify32: bad argument");
    const long M = (long)N * (S / p) * (S / p);
    P32_LAUNCH(patchify32_kernel, M * ldp, x, patches, ldp, C, S, p, M);
}
int mapdit_axpby32(const float* in, float* out, long n, float alpha, int accumulate, void* stream) {
    MD_CHECK(in && out && n > 0, "axpby32: bad argument");
    P32_LAUNCH(axpby32_kernel, n, in, out, n, alpha, accumulate);
}

int mapdit_split3(const float* src, long ld, uint16_t* dst, long rows, int K, int pattern, int op, void* stream) {
    MD_CHECK(src && dst && rows > 0 && K > 0 && ld >= K, "split3: bad argument");
    hipLaunchKernelGGL(split3_kernel, dim3(cdiv(rows * K, 256)), dim3(256), 0, (hipStream_t)stream, src, ld, dst, rows, K, pattern, op, 0,
                       (long)0);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

int mapdit_fourier32(const int64_t* t, const float* scale, const float* shift, float* out, int n, int F, void* stream) {
    MD_CHECK(t && scale && shift && out && n > 0, "fourier32: bad argument");
    hipLaunchKernelGGL(fourier32_kernel, dim3(cdiv((long)n * F, 256)), dim3(256), 0, (hipStream_t)stream, (const long*)t, scale, shift,
                       out, n, F);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

int mapdit_modulate32(const float* x, const float* shift, const float* scale, int ldmod, const float* gain, float* out, int N, int T,
                      int D, void* stream) {
    MD_CHECK(x && shift && scale && gain && out && N > 0, "modulate32: bad argument");
    const long total = (long)N * T * D;
    hipLaunchKernelGGL(modulate32_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, shift, scale, ldmod, gain, out,
                       T, D, total);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

int mapdit_resid32(const float* xin, const float* y, const float* gate, int ldg, float* xout, int N, int T, int D, float t,
                   void* stream) {
    MD_CHECK(xin && y && gate && xout && N > 0, "resid32: bad argument");
    const long total = (long)N * T * D;
    const float inv_den = 1.f / sqrtf((1.f - t) * (1.f - t) + t * t);
    hipLaunchKernelGGL(resid32_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, xin, y, gate, ldg, xout, T, D, total,
                       t, inv_den);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

int mapdit_qkv_split32(const float* qkv, int B, int T, int H, int hd, float* qn, float* kn, float* v, float* scales, void* stream) {
    MD_CHECK(qkv && qn && kn && v && B > 0, "qkv_split32: bad argument");
    hipLaunchKernelGGL(qkv_split32_kernel, dim3(cdiv((long)B * T * H, 256)), dim3(256), 0, (hipStream_t)stream, qkv, B, T, H, hd, qn,
                       kn, v, scales);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

int mapdit_attn32(const float* qn, const float* kn, const float* v, float* o, int B, int T, int H, int hd, void* stream) {
    MD_CHECK(qn && kn && v && o && B > 0, "attn32: bad argument");
    MD_CHECK(T <= 256 && hd <= 96, "attn32: T=%d, head_dim=%d unsupported (<= 256, <= 96)", T, hd);
    if (hd <= 32) hipLaunchKernelGGL(attn32_kernel<32>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, qn, kn, v, o, T, H, hd, 1.f / sqrtf((float)hd));
    else if (hd <= 64) hipLaunchKernelGGL(attn32_kernel<64>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, qn, kn, v, o, T, H, hd, 1.f / sqrtf((float)hd));
    else hipLaunchKernelGGL(attn32_kernel<96>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, qn, kn, v, o, T, H, hd, 1.f / sqrtf((float)hd));
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
