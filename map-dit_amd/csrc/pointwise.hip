// HBM-bound token-stream kernels of the DiT block (SURVEY.md K4, K5, K7 and their backward):
// modulate, the fused residual/modulate backward with its per-sample column reductions, and the
// head split / cosine-normalise / transpose passes on either side of attention.
// Residual stream is fp32 [M, D]; GEMM operands are bf16; per-sample conditioning vectors are fp32.
#include "common.h"

MD_NS_OPEN

__device__ __forceinline__ float mp_den(float t) { return sqrtf((1.f - t) * (1.f - t) + t * t); }

// ---- modulate forward (reference src/utils.py:11-16 via dit_block.py:35-36, final_layer.py:55) -----------------
// u = ((1-g) * x * scale[n] + g * shift[n]) / sqrt((1-g)^2 + g^2)   -> bf16 GEMM operand
__global__ __launch_bounds__(256) void modulate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                         const float* __restrict__ scale, int ldmod,
                                                         const float* __restrict__ gain, bf16_t* __restrict__ out,
                                                         long total8, int D, int T) {
    const float g = *gain, den = mp_den(g);
    const float ka = (1.f - g) / den, kb = g / den;
    const int d8 = D >> 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
        const long m = i / d8;
        const int d = (int)(i % d8) * 8;
        const int n = (int)(m / T);
        const float4* xp = (const float4*)(x + m * D + d);
        const float4* sc = (const float4*)(scale + (size_t)n * ldmod + d);
        const float4* sh = (const float4*)(shift + (size_t)n * ldmod + d);
        float4 x0 = xp[0], x1 = xp[1], c0 = sc[0], c1 = sc[1], h0 = sh[0], h1 = sh[1];
        uint4 u;
        u.x = pack16(ka * x0.x * c0.x + kb * h0.x, ka * x0.y * c0.y + kb * h0.y);
        u.y = pack16(ka * x0.z * c0.z + kb * h0.z, ka * x0.w * c0.w + kb * h0.w);
        u.z = pack16(ka * x1.x * c1.x + kb * h1.x, ka * x1.y * c1.y + kb * h1.y);
        u.w = pack16(ka * x1.z * c1.z + kb * h1.z, ka * x1.w * c1.w + kb * h1.w);
        *(uint4*)(out + m * D + d) = u;
    }
}

// ---- LayerNorm in front of modulate (README.md:64 --no-use-no-layernorm: the transformer layer normalisation the snapshot disabled;
// PARITY UNPINNED - the snapshot has no such layer; restated as upstream DiT's nn.LayerNorm(D, elementwise_affine=False, eps=1e-6)) -------
//   xh = (x - mean) * rstd,  rstd = 1 / sqrt(var + 1e-6) (biased variance over the D features of a token);  u = modulate(xh, ...) as above.
// One wave per token row, the row held in registers between the two sweeps (D <= 2048).  xh (fp32) and rstd are kept for the backward
// when given (training); inference passes NULL for both.
constexpr float LN_EPS = 1e-6f;
constexpr int LN_MAXV = 8;                       // float4 chunks per lane: D <= 64 * 4 * 8
__global__ __launch_bounds__(256) void ln_modulate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                            const float* __restrict__ scale, int ldmod, const float* __restrict__ gain,
                                                            float* __restrict__ xh, float* __restrict__ rstd, bf16_t* __restrict__ out,
                                                            long rows, int D, int T) {
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    const int lane = threadIdx.x & 63;
    const float g = *gain, den = mp_den(g);
    const float ka = (1.f - g) / den, kb = g / den;
    const int n = (int)(m / T);
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane * 4 + 256 * k;
        v[k] = c < D ? *(const float4*)(x + m * D + c) : make_float4(0, 0, 0, 0);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane * 4 + 256 * k;
        if (c < D) {
            v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
            q += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
        }
    }
    const float r = 1.f / sqrtf(wave_sum(q) / (float)D + LN_EPS);
    if (rstd && lane == 0) rstd[m] = r;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane * 4 + 256 * k;
        if (c < D) {
            const float4 h = make_float4(v[k].x * r, v[k].y * r, v[k].z * r, v[k].w * r);
            if (xh) *(float4*)(xh + m * D + c) = h;
            const float4 sc = *(const float4*)(scale + (size_t)n * ldmod + c), sh = *(const float4*)(shift + (size_t)n * ldmod + c);
            uint2 u;
            u.x = pack16(ka * h.x * sc.x + kb * sh.x, ka * h.y * sc.y + kb * sh.y);
            u.y = pack16(ka * h.z * sc.z + kb * sh.z, ka * h.w * sc.w + kb * sh.w);
            *(uint2*)(out + m * D + c) = u;
        }
    }
}

// Backward of the normalisation, merged with the residual stream's pass-through:  out = ca * dxo + rstd * (g - mean(g) - xh * mean(g * xh)),
// g = grad wrt xh (what the modulate-only form of resid_mod_bwd below leaves in its dx).  dxo: fp32, or 16-bit (dxo16), or neither (the
// final layer's site: nothing arrives from downstream).  One wave per token row.
__global__ __launch_bounds__(256) void ln_bwd_merge_kernel(const float* __restrict__ g, const float* __restrict__ xh, const float* __restrict__ rstd,
                                                         const float* __restrict__ dxo, const bf16_t* __restrict__ dxo16, float ca,
                                                         float* __restrict__ out, long rows, int D) {
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    const int lane = threadIdx.x & 63;
    float4 gv[LN_MAXV], hv[LN_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane * 4 + 256 * k;
        if (c < D) {
            gv[k] = *(const float4*)(g + m * D + c);
            hv[k] = *(const float4*)(xh + m * D + c);
            s1 += (gv[k].x + gv[k].y) + (gv[k].z + gv[k].w);
            s2 += (gv[k].x * hv[k].x + gv[k].y * hv[k].y) + (gv[k].z * hv[k].z + gv[k].w * hv[k].w);
        }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D, r = rstd[m];
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane * 4 + 256 * k;
        if (c < D) {
            float4 o = make_float4(r * (gv[k].x - m1 - hv[k].x * m2), r * (gv[k].y - m1 - hv[k].y * m2), r * (gv[k].z - m1 - hv[k].z * m2),
                                   r * (gv[k].w - m1 - hv[k].w * m2));
            if (dxo) {
                const float4 d = *(const float4*)(dxo + m * D + c);
                o.x += ca * d.x; o.y += ca * d.y; o.z += ca * d.z; o.w += ca * d.w;
            } else if (dxo16) {
                const uint2 u = *(const uint2*)(dxo16 + m * D + c);
                o.x += ca * lo16(u.x); o.y += ca * hi16(u.x); o.z += ca * lo16(u.y); o.w += ca * hi16(u.y);
            }
            *(float4*)(out + m * D + c) = o;
        }
    }
}

// ---- fused backward of  x' = mp_sum(x_up, g_up*y_up, 0.3)  ->  u = modulate(x', shift, scale, gain) -------------
// Inputs : dxo  grad wrt the residual value x' arriving from downstream (fp32, may be null = 0), scaled by ca here
//          dxm  grad wrt u (bf16, may be null)
// Outputs: dx   = ca*dxo + k*scale*dxm                       (fp32, grad wrt x')
//          dscale[n,d] = sum_t k*x*dxm ; dshift[n,d] = sum_t (g/den)*dxm ; dgain partial = sum dxm*(shift - x*scale)/den
//          (mp_sum denominator is detached for the learnable gain: SURVEY F8)
//          dy_up = cb*g_up*dx (bf16) ; dg_up[n,d] = sum_t cb*y_up*dx     (backward of the residual that produced x')
//          dx_bf = bf16(dx) optionally (operand of the patch-embedding dW GEMM)
// Block = one sample x 128 columns; 32 column-lanes (4 columns each) x 8 row groups; LDS reduce over row groups.
struct RmbP {
    const float* dxo; const bf16_t* dxm; const float* x; const float* shift; const float* scale; const float* gain;
    const bf16_t* y_up; const float* g_up;
    float* dx; bf16_t* dx_bf; float* dshift; float* dscale; float* dgain_part; bf16_t* dy_up; float* dg_up;
    int ldmod, ldg_up, ldd, ldd_up;   // row strides of (shift,scale), g_up, dshift/dscale and dg_up
    int T, D; float ca, cb;
    float gscale;                     // factor on the scalar gain partials (1 / loss scale of an fp16 engine; 1 otherwise)
    int rot;                          // rotation form: u[j] = scale[j] x'[j] + shift[j] x'[j ^ 1] (scale / shift = the A / B rows of
                                      // rot_coef_fwd); dscale / dshift receive dA / dB; no gain partial
    // gridDim.z > 1: each block handles T / gridDim.z rows and parks its column sums in `part` ([z][sample][3][D]) and its gain
    // partial in dgain_part[(z * samples + n) * D/128 + column block]; rmb_finish_kernel adds the z slices in order
    float* part;
    const bf16_t* dxo16;              // the downstream gradient as a 16-bit tensor (instead of dxo)
    int ldx;                          // row stride of the [tokens, D] tensors (>= D: a column range of wider tensors)
};

// ROT: the rotation form (compile-time: the AdaLN form's loop carries no trace of it).  CL: column lanes of 4 columns each; a
// block covers 4 CL columns and 256 / CL rows per step (CL = 64: 1-KiB fp32 row segments, when D is a multiple of 256).
template <bool ROT, int CL>
__global__ __launch_bounds__(256) void resid_mod_bwd_kernel(RmbP p) {
    constexpr int RG = 256 / CL;
    __shared__ float red[RG][CL][13];
    const int n = blockIdx.x, cb0 = blockIdx.y * (4 * CL);
    const int cl = threadIdx.x & (CL - 1), rg = threadIdx.x / CL;
    const int d = cb0 + cl * 4;
    float g = 0.f, den = 1.f;
    if (p.dxm && !ROT) { g = *p.gain; den = mp_den(g); }
    const float k = (1.f - g) / den, kb = g / den, kd = 1.f / den;
    float4 sc = make_float4(0, 0, 0, 0), sh = sc, gu = sc;
    if (p.dxm) {
        sc = *(const float4*)(p.scale + (size_t)n * p.ldmod + d);
        sh = *(const float4*)(p.shift + (size_t)n * p.ldmod + d);
    }
    if (p.y_up) gu = *(const float4*)(p.g_up + (size_t)n * p.ldg_up + d);
    float a_sc[4] = {0, 0, 0, 0}, a_sh[4] = {0, 0, 0, 0}, a_g[4] = {0, 0, 0, 0}, a_gain = 0.f;
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, guv[4] = {gu.x, gu.y, gu.z, gu.w};
    const int rows_z = p.T / gridDim.z, t_beg = blockIdx.z * rows_z;
    for (int t = t_beg + rg; t < t_beg + rows_z; t += RG) {
        const size_t off = ((size_t)n * p.T + t) * p.ldx + d;
        float dx[4] = {0, 0, 0, 0};
        if (p.dxo) {
            float4 v = *(const float4*)(p.dxo + off);
            dx[0] = p.ca * v.x; dx[1] = p.ca * v.y; dx[2] = p.ca * v.z; dx[3] = p.ca * v.w;
        } else if (p.dxo16) {
            const uint2 u = *(const uint2*)(p.dxo16 + off);
            dx[0] = p.ca * lo16(u.x); dx[1] = p.ca * hi16(u.x); dx[2] = p.ca * lo16(u.y); dx[3] = p.ca * hi16(u.y);
        }
        if (p.dxm) {
            uint2 u = *(const uint2*)(p.dxm + off);
            const float dm[4] = {lo16(u.x), hi16(u.x),
                                 lo16(u.y), hi16(u.y)};
            float4 xv = *(const float4*)(p.x + off);
            const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
            if (ROT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dx[i] += scv[i] * dm[i] + shv[i ^ 1] * dm[i ^ 1];       // u[j] = A[j] x[j] + B[j] x[j^1]
                    a_sc[i] += xx[i] * dm[i];                               // dA
                    a_sh[i] += xx[i ^ 1] * dm[i];                           // dB
                }
            } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dx[i] = __builtin_fmaf(k * scv[i], dm[i], dx[i]);           // (spelled out: the same rounding as the fused GEMM epilogue, EpiRmbT)
                a_sc[i] += k * xx[i] * dm[i];
                a_sh[i] += kb * dm[i];
                a_gain += dm[i] * (shv[i] - xx[i] * scv[i]) * kd;
            }
            }
        }
        if (p.dx) *(float4*)(p.dx + off) = make_float4(dx[0], dx[1], dx[2], dx[3]);
        // (the fp32 values are pinned before their 16-bit conversion: hipcc may otherwise fold the conversion into the fma above
        // - one rounding instead of two - here and not in the GEMM epilogue that computes the same values, or the other way round)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(dx[i]));
        if (p.dx_bf) {
            uint2 u; u.x = pack16(dx[0], dx[1]); u.y = pack16(dx[2], dx[3]);
            *(uint2*)(p.dx_bf + off) = u;
        }
        if (p.y_up) {
            uint2 u = *(const uint2*)(p.y_up + off);
            const float yy[4] = {lo16(u.x), hi16(u.x),
                                 lo16(u.y), hi16(u.y)};
            float dy[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dy[i] = p.cb * guv[i] * dx[i];
                a_g[i] += p.cb * yy[i] * dx[i];
            }
            uint2 w; w.x = pack16(dy[0], dy[1]); w.y = pack16(dy[2], dy[3]);
            *(uint2*)(p.dy_up + off) = w;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[rg][cl][i] = a_sc[i]; red[rg][cl][4 + i] = a_sh[i]; red[rg][cl][8 + i] = a_g[i]; }
    red[rg][cl][12] = a_gain;
    __syncthreads();
    if (rg == 0) {
        float s[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < RG; ++r) a += red[r][cl][j];
            s[j] = a;
        }
        const bool split = gridDim.z > 1;
        float* pz = split ? p.part + ((size_t)blockIdx.z * gridDim.x + n) * 3 * p.D : nullptr;
        if (p.dxm) {
            *(float4*)(split ? pz + d : p.dscale + (size_t)n * p.ldd + d) = make_float4(s[0], s[1], s[2], s[3]);
            *(float4*)(split ? pz + p.D + d : p.dshift + (size_t)n * p.ldd + d) = make_float4(s[4], s[5], s[6], s[7]);
            float gsum = s[12];
#pragma unroll
            for (int o = CL / 2; o > 0; o >>= 1) gsum += __shfl_xor(gsum, o, 64);
            if (cl == 0) p.dgain_part[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y] = gsum * p.gscale;
        }
        if (p.y_up) *(float4*)(split ? pz + 2 * p.D + d : p.dg_up + (size_t)n * p.ldd_up + d) = make_float4(s[8], s[9], s[10], s[11]);
    }
}

// Second stage of the row-split form: out[n, d] = sum_z part[z][n][which][d], z in order (deterministic).
// The extra last block (gain_part != null) sums the scalar-gain partials exactly as reduce_partials_kernel does - one launch less.
__global__ void rmb_finish_kernel(const float* __restrict__ part, int Z, int N, int D, float* __restrict__ dscale,
                                  float* __restrict__ dshift, int ldd, float* __restrict__ dg_up, int ldd_up,
                                  const float* __restrict__ gain_part, int gain_count, float* __restrict__ dgain_out) {
    if (gain_part && blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x >= 64) return;
        float a = 0.f;
        for (int i = threadIdx.x; i < gain_count; i += 64) a += gain_part[i];
        a = wave_sum(a);
        if (threadIdx.x == 0) *dgain_out = a;
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const int n = i / D, d = i % D;
    float a[3] = {0.f, 0.f, 0.f};
    for (int z = 0; z < Z; ++z) {
        const float* pz = part + ((size_t)z * N + n) * 3 * D + d;
        a[0] += pz[0]; a[1] += pz[D]; a[2] += pz[2 * D];
    }
    if (dscale) { dscale[(size_t)n * ldd + d] = a[0]; dshift[(size_t)n * ldd + d] = a[1]; }
    if (dg_up) dg_up[(size_t)n * ldd_up + d] = a[2];
}

// ---- rotation modulation (PARITY UNPINNED: not in the reference snapshot, its README.md:1-3 only; oracle.modulate_rot) ----------
//   (y[2i], y[2i+1]) = R(g theta[n, i]) (scale[2i] x[2i], scale[2i+1] x[2i+1]),      g = the block's learnable gain
// is linear in x with per-(sample, column) coefficients:  y[j] = A[n, j] x[j] + B[n, j] x[j ^ 1],
//   A[2i] = c sc[2i]   B[2i] = -s sc[2i+1]   A[2i+1] = c sc[2i+1]   B[2i+1] = s sc[2i]      (c, s = cos, sin of g theta[n, i]).
// rot_coef_fwd builds the A / B rows once per step and branch ([samples, D]: one sincos per pair, not per token); the consumers
// are the modulate fused into the residual GEMM epilogues (gemm.hip EpiResid, rot form), rot_modulate_fwd for block 0, and the
// fused residual / modulate backward below, whose per-sample column sums become dA, dB; rot_coef_bwd turns those into the
// gradients of theta, scale and the gain.  No pass over the token stream is added to the AdaLN engine's.
__global__ void rot_coef_fwd_kernel(const float* __restrict__ theta, const float* __restrict__ scale, int ldm,
                                    const float* __restrict__ gain, float* __restrict__ A, float* __restrict__ B, int ldc, int n, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;              // one thread per (sample, pair)
    const int h = D >> 1;
    if (i >= n * h) return;
    const int b = i / h, pr = i - b * h;
    float sn, cs;
    sincosf(*gain * theta[(size_t)b * ldm + pr], &sn, &cs);
    const float2 sc = *(const float2*)(scale + (size_t)b * ldm + 2 * pr);
    *(float2*)(A + (size_t)b * ldc + 2 * pr) = make_float2(cs * sc.x, cs * sc.y);
    *(float2*)(B + (size_t)b * ldc + 2 * pr) = make_float2(-sn * sc.y, sn * sc.x);
}

// every (block, branch) slot of the model in one launch: grid (pairs / 256, slots, samples)
struct RotSlots { const float* gain[MAPDIT_ROT_MAX_SLOTS]; int theta_off[MAPDIT_ROT_MAX_SLOTS]; int scale_off[MAPDIT_ROT_MAX_SLOTS]; };
__global__ __launch_bounds__(256) void rot_coef_fwd_all_kernel(const float* __restrict__ mod, int ldm, RotSlots sl, float* __restrict__ A,
                                                              float* __restrict__ B, int ldc, int D) {
    const int pr = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y, b = blockIdx.z;
    if (pr >= (D >> 1)) return;
    const float* row = mod + (size_t)b * ldm;
    float sn, cs;
    sincosf(*sl.gain[s] * row[sl.theta_off[s] + pr], &sn, &cs);
    const float2 sc = *(const float2*)(row + sl.scale_off[s] + 2 * pr);
    const size_t o = (size_t)b * ldc + (size_t)s * D + 2 * pr;
    *(float2*)(A + o) = make_float2(cs * sc.x, cs * sc.y);
    *(float2*)(B + o) = make_float2(-sn * sc.y, sn * sc.x);
}

// dtheta[n, i] = g dphi, dscale, and one partial of dgain = sum theta dphi per workgroup (summed in order by reduce_partials), from
//   dA[n, j] = sum_t dy[j] x[j],  dB[n, j] = sum_t dy[j] x[j ^ 1]   (the rot form of resid_mod_bwd):
//   dsc[2i] = c dA[2i] + s dB[2i+1]      dsc[2i+1] = c dA[2i+1] - s dB[2i]
//   dphi    = -s (sc[2i] dA[2i] + sc[2i+1] dA[2i+1]) + c (sc[2i] dB[2i+1] - sc[2i+1] dB[2i])
// Grid: (D / 512, samples); 256 threads, one pair each.
__global__ __launch_bounds__(256) void rot_coef_bwd_kernel(const float* __restrict__ dA, const float* __restrict__ dB, int ldc,
                                                         const float* __restrict__ theta, const float* __restrict__ scale, int ldm,
                                                         const float* __restrict__ gain, float* __restrict__ dtheta,
                                                         float* __restrict__ dscale, int ldd, float* __restrict__ dgain_part,
                                                         float gscale, int D) {
    __shared__ float red[4];
    const int b = blockIdx.y, pr = blockIdx.x * 256 + threadIdx.x;
    float gs = 0.f;
    if (pr < (D >> 1)) {
        const float g = *gain, th = theta[(size_t)b * ldm + pr];
        float sn, cs;
        sincosf(g * th, &sn, &cs);
        const float2 sc = *(const float2*)(scale + (size_t)b * ldm + 2 * pr);
        const float2 a = *(const float2*)(dA + (size_t)b * ldc + 2 * pr), bb = *(const float2*)(dB + (size_t)b * ldc + 2 * pr);
        *(float2*)(dscale + (size_t)b * ldd + 2 * pr) = make_float2(cs * a.x + sn * bb.y, cs * a.y - sn * bb.x);
        const float dphi = -sn * (sc.x * a.x + sc.y * a.y) + cs * (sc.x * bb.y - sc.y * bb.x);
        dtheta[(size_t)b * ldd + pr] = g * dphi;
        gs = th * dphi;
    }
    gs = wave_sum(gs);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gs;
    __syncthreads();
    if (threadIdx.x == 0) dgain_part[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * gscale;
}

// out = 16-bit(x * A[n] + pairswap(x) * B[n]): the rotation modulate of block 0's attention branch (every later one is fused into
// the residual GEMM epilogue that produces its input)
__global__ __launch_bounds__(256) void rot_modulate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ A,
                                                             const float* __restrict__ B, int ldc, bf16_t* __restrict__ out,
                                                             long total8, int D, int T) {
    const int d8 = D >> 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
        const long m = i / d8;
        const int d = (int)(i % d8) * 8;
        const int n = (int)(m / T);
        const float4* xp = (const float4*)(x + m * D + d);
        const float4* ap = (const float4*)(A + (size_t)n * ldc + d);
        const float4* bp = (const float4*)(B + (size_t)n * ldc + d);
        const float4 x0 = xp[0], x1 = xp[1], a0 = ap[0], a1 = ap[1], b0 = bp[0], b1 = bp[1];
        uint4 u;
        u.x = pack16(__builtin_fmaf(x0.x, a0.x, x0.y * b0.x), __builtin_fmaf(x0.y, a0.y, x0.x * b0.y));
        u.y = pack16(__builtin_fmaf(x0.z, a0.z, x0.w * b0.z), __builtin_fmaf(x0.w, a0.w, x0.z * b0.w));
        u.z = pack16(__builtin_fmaf(x1.x, a1.x, x1.y * b1.x), __builtin_fmaf(x1.y, a1.y, x1.x * b1.y));
        u.w = pack16(__builtin_fmaf(x1.z, a1.z, x1.w * b1.z), __builtin_fmaf(x1.w, a1.w, x1.z * b1.w));
        *(uint4*)(out + m * D + d) = u;
    }
}

// Deterministic final reduction of per-block partials into a scalar gradient (gain_msa / gain_mlp / gain_mod).
__global__ void reduce_partials_kernel(const float* __restrict__ part, int count, float* __restrict__ out, int accumulate) {
    float a = 0.f;
    for (int i = threadIdx.x; i < count; i += 64) a += part[i];
    a = wave_sum(a);
    if (threadIdx.x == 0) *out = accumulate ? *out + a : a;
}

// ---- head split + cosine normalise -------------------------------------------------------------------------------
// qkv [M, 3D] bf16 -> qn, kn, v  [BH][T][64].  q^ = q*sqrt(hd)/(|q|+eps) (reference attention.py:44-45 via utils.py:19-23).
// Block = (64 tokens, head, batch); 4 threads per token row, 16 features each.
__global__ __launch_bounds__(256) void qkv_split_kernel(const bf16_t* __restrict__ qkv, int T, int H, bf16_t* __restrict__ qn,
                                                      bf16_t* __restrict__ kn, bf16_t* __restrict__ v) {
    const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64;
    const int row = threadIdx.x >> 2, qd = threadIdx.x & 3;
    const size_t bh = (size_t)b * H + h;
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        const bf16_t* src = qkv + ((size_t)b * T + t0 + row) * (3 * D) + which * D + h * 64 + qd * 16;
        uint4 u0 = ((const uint4*)src)[0], u1 = ((const uint4*)src)[1];
        uint32_t w[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        if (which < 2) {
            float f[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) { f[2 * i] = lo16(w[i]); f[2 * i + 1] = hi16(w[i]); }
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) ss += f[i] * f[i];
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            const float sc = 8.f / (sqrtf(ss) + NORM_EPS);
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = pack16(f[2 * i] * sc, f[2 * i + 1] * sc);
        }
        bf16_t* dst = (which == 0 ? qn : which == 1 ? kn : v) + (bh * T + t0 + row) * 64 + qd * 16;
        ((uint4*)dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
        ((uint4*)dst)[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// Backward: dqn,dkn,dv [BH][T][64] + saved qkv -> dqkv [M,3D].  dq = s*(dq^ - q (dq^.q)/(n (n+eps))), s = 8/(n+eps).
__global__ __launch_bounds__(256) void qkv_merge_bwd_kernel(const bf16_t* __restrict__ qkv, int T, int H,
                                                          const bf16_t* __restrict__ dqn, const bf16_t* __restrict__ dkn,
                                                          const bf16_t* __restrict__ dv, bf16_t* __restrict__ dqkv) {
    const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64;
    const int row = threadIdx.x >> 2, qd = threadIdx.x & 3;
    const size_t bh = (size_t)b * H + h;
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        const size_t moff = ((size_t)b * T + t0 + row) * (3 * D) + which * D + h * 64 + qd * 16;
        const bf16_t* gsrc = (which == 0 ? dqn : which == 1 ? dkn : dv) + (bh * T + t0 + row) * 64 + qd * 16;
        uint4 g0 = ((const uint4*)gsrc)[0], g1 = ((const uint4*)gsrc)[1];
        uint32_t gw[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        if (which < 2) {
            uint4 u0 = ((const uint4*)(qkv + moff))[0], u1 = ((const uint4*)(qkv + moff))[1];
            uint32_t w[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
            float f[16], gf[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f[2 * i] = lo16(w[i]); f[2 * i + 1] = hi16(w[i]);
                gf[2 * i] = lo16(gw[i]); gf[2 * i + 1] = hi16(gw[i]);
            }
            float ss = 0.f, dot = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { ss += f[i] * f[i]; dot += f[i] * gf[i]; }
            ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64);
            dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64);
            const float n = sqrtf(ss), s = 8.f / (n + NORM_EPS);
            const float c = dot / (fmaxf(n, 1e-30f) * (n + NORM_EPS));
#pragma unroll
            for (int i = 0; i < 16; ++i) gf[i] = s * (gf[i] - f[i] * c);
#pragma unroll
            for (int i = 0; i < 8; ++i) gw[i] = pack16(gf[2 * i], gf[2 * i + 1]);
        }
        ((uint4*)(dqkv + moff))[0] = make_uint4(gw[0], gw[1], gw[2], gw[3]);
        ((uint4*)(dqkv + moff))[1] = make_uint4(gw[4], gw[5], gw[6], gw[7]);
    }
}

// mp_silu on a small fp32 matrix -> bf16 GEMM operand (conditioning path: MPSiLU(c), dit_block.py:24-25).
__global__ void mpsilu_f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cvt16(silu_f(x[i]) * (1.f / MP_SILU_DIV));
}
__global__ void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, long n, float alpha) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cvt16(alpha * x[i]);
}

__global__ void f32_to_bf16_2d_kernel(const float* __restrict__ x, int ldx, bf16_t* __restrict__ out, int ldo, int rows, int cols,
                                      float alpha) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    out[(size_t)r * ldo + c] = cvt16(alpha * x[(size_t)r * ldx + c]);
}
// out[i] = alpha * x[i]
__global__ void scale_copy_kernel(float* __restrict__ out, const float* __restrict__ x, long n, float alpha) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = alpha * x[i];
}
// out[i] = sum_s slabs[s*stride + i], four elements per thread   (fixed order; n a multiple of 4, 16-byte aligned)
__global__ void reduce_slabs_kernel(float* __restrict__ out, const float* __restrict__ slabs, int nslabs, long stride, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 a = *(const float4*)(slabs + 4 * i);
    for (int s = 1; s < nslabs; ++s) {
        const float4 t = *(const float4*)(slabs + (size_t)s * stride + 4 * i);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    *(float4*)(out + 4 * i) = a;
}
// The same for up to four buffers in one launch (items by value; a block = 1,024 elements of one item): behind the grouped weight-gradient launch
// under sharded weight passes.
struct ReduceGroup { float* out[4]; const float* slabs[4]; long stride[4], n4[4]; int first[5]; int nslabs, n; };
__global__ void reduce_slabs_group_kernel(ReduceGroup g) {
    int k = 0;
    while (k + 1 < g.n && (int)blockIdx.x >= g.first[k + 1]) ++k;
    const long i = (long)(blockIdx.x - g.first[k]) * blockDim.x + threadIdx.x;
    if (i >= g.n4[k]) return;
    const float* sl = g.slabs[k];
    float4 a = *(const float4*)(sl + 4 * i);
    for (int s = 1; s < g.nslabs; ++s) {
        const float4 t = *(const float4*)(sl + (size_t)s * g.stride[k] + 4 * i);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    *(float4*)(g.out[k] + 4 * i) = a;
}
// out[i] = sum_c bf16 chunks[c*stride + i] accumulated in fp32, in chunk order (8 elements per thread): the receiving side of a 16-bit
// gradient exchange (every rank's bf16 copy of the rows this rank owns, summed in fp32)
__global__ void sum_bf16_chunks_kernel(float* __restrict__ out, const unsigned short* __restrict__ chunks, int nchunks, long stride, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nchunks; ++c) {
        const uint4 u = *(const uint4*)(chunks + (size_t)c * stride + 8 * i);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[2 * k] += __uint_as_float(w[k] << 16);
            a[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u);
        }
    }
    *(float4*)(out + 8 * i) = make_float4(a[0], a[1], a[2], a[3]);
    *(float4*)(out + 8 * i + 4) = make_float4(a[4], a[5], a[6], a[7]);
}
// acc[i] += sum_s slabs[s*stride + i]   (fixed order: deterministic split-K reduction)
__global__ void sum_slabs_kernel(float* __restrict__ acc, const float* __restrict__ slabs, int nslabs, long stride, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = acc[i];
    for (int s = 0; s < nslabs; ++s) a += slabs[(size_t)s * stride + i];
    acc[i] = a;
}

MD_NS_CLOSE

extern "C" int MD_SYM_F32_TO_16_2D(const float* x, int ldx, uint16_t* out, int ldo, int rows, int cols, float alpha, void* stream) {
    MD_CHECK(x && out && rows > 0 && cols > 0 && ldx >= cols && ldo >= cols, "f32_to_bf16_2d: bad argument");
    hipLaunchKernelGGL(f32_to_bf16_2d_kernel, dim3(cdiv((long)rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo,
                       rows, cols, alpha);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

#if MAPDIT_DT == 0          // no 16-bit operand: one copy in the library
extern "C" int mapdit_scale_copy(float* out, const float* x, long n, float alpha, void* stream) {
    MD_CHECK(out && x && n > 0, "scale_copy: bad argument");
    hipLaunchKernelGGL(scale_copy_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, x, n, alpha);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
extern "C" int mapdit_reduce_slabs(float* out, const float* slabs, int nslabs, long slab_stride, long n, void* stream) {
    MD_CHECK(out && slabs && nslabs >= 1 && n > 0 && n % 4 == 0 && slab_stride % 4 == 0 && ((((uintptr_t)out | (uintptr_t)slabs) & 15) == 0),
             "reduce_slabs: bad argument (n and slab_stride multiples of 4, 16-byte aligned buffers)");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, slabs, nslabs, slab_stride, n / 4);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
extern "C" int mapdit_reduce_slabs_group(int n, float* const* outs, const float* const* slabs, const long* slab_strides, const long* ns, int nslabs,
                                         void* stream) {
    MD_CHECK(outs && slabs && slab_strides && ns && n >= 1 && n <= 4 && nslabs >= 1, "reduce_slabs_group: bad argument (1..4 items)");
    ReduceGroup g{};
    g.n = n; g.nslabs = nslabs;
    for (int k = 0; k < n; ++k) {
        MD_CHECK(outs[k] && slabs[k] && ns[k] > 0 && ns[k] % 4 == 0 && slab_strides[k] % 4 == 0 && ((((uintptr_t)outs[k] | (uintptr_t)slabs[k]) & 15) == 0),
                 "reduce_slabs_group: item %d: n and slab_stride multiples of 4, 16-byte aligned buffers", k);
        g.out[k] = outs[k]; g.slabs[k] = slabs[k]; g.stride[k] = slab_strides[k]; g.n4[k] = ns[k] / 4;
        g.first[k + 1] = g.first[k] + (int)cdiv(ns[k] / 4, 256);
    }
    hipLaunchKernelGGL(reduce_slabs_group_kernel, dim3(g.first[n]), dim3(256), 0, (hipStream_t)stream, g);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
extern "C" int mapdit_sum_bf16_chunks(float* out, const uint16_t* chunks, int nchunks, long chunk_stride, long n, void* stream) {
    MD_CHECK(out && chunks && nchunks >= 1 && n > 0 && n % 8 == 0 && chunk_stride % 8 == 0 && ((((uintptr_t)out | (uintptr_t)chunks) & 15) == 0),
             "sum_bf16_chunks: bad argument (n and chunk_stride multiples of 8, 16-byte aligned buffers)");
    hipLaunchKernelGGL(sum_bf16_chunks_kernel, dim3(cdiv(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, out, chunks, nchunks, chunk_stride, n / 8);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
extern "C" int mapdit_sum_slabs(float* acc, const float* slabs, int nslabs, long slab_stride, long n, void* stream) {
    MD_CHECK(acc && slabs && nslabs >= 1 && n > 0, "sum_slabs: bad argument");
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, acc, slabs, nslabs, slab_stride, n);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
#endif

extern "C" int MD_SYM(modulate_fwd)(const float* x, const float* shift, const float* scale, int ldmod, const float* gain,
                                   uint16_t* out, int n_samples, int T, int D, void* stream) {
    MD_CHECK(x && shift && scale && gain && out, "modulate_fwd: null argument");
    MD_CHECK(D % 8 == 0 && ldmod % 4 == 0, "modulate_fwd: D=%d must be a multiple of 8, ldmod=%d of 4", D, ldmod);
    const long total8 = (long)n_samples * T * (D / 8);
    const int grid = (int)((total8 + 255) / 256 < 8192 ? (total8 + 255) / 256 : 8192);
    hipLaunchKernelGGL(modulate_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, shift, scale, ldmod, gain,
                       out, total8, D, T);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(ln_modulate_fwd)(const float* x, const float* shift, const float* scale, int ldmod, const float* gain, float* xhat,
                                      float* rstd, uint16_t* out, int n_samples, int T, int D, void* stream) {
    MD_CHECK(x && shift && scale && gain && out && n_samples > 0 && T > 0, "ln_modulate_fwd: null/empty argument");
    MD_CHECK(D % 4 == 0 && D > 0 && D <= 256 * LN_MAXV && ldmod % 4 == 0, "ln_modulate_fwd: D=%d must be a multiple of 4 up to %d, ldmod=%d of 4", D,
             256 * LN_MAXV, ldmod);
    const long rows = (long)n_samples * T;
    hipLaunchKernelGGL(ln_modulate_fwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, shift, scale, ldmod, gain, xhat,
                       rstd, out, rows, D, T);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(ln_bwd_merge)(const float* dxhat, const float* xhat, const float* rstd, const float* dxo, const uint16_t* dxo16, float ca,
                                   float* out, long rows, int D, void* stream) {
    MD_CHECK(dxhat && xhat && rstd && out && rows > 0, "ln_bwd_merge: null/empty argument");
    MD_CHECK(!(dxo && dxo16), "ln_bwd_merge: dxo and dxo16 are alternatives");
    MD_CHECK(D % 4 == 0 && D > 0 && D <= 256 * LN_MAXV, "ln_bwd_merge: D=%d must be a multiple of 4 up to %d", D, 256 * LN_MAXV);
    hipLaunchKernelGGL(ln_bwd_merge_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dxhat, xhat, rstd, dxo, dxo16, ca, out,
                       rows, D);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(resid_mod_bwd)(const mapdit_resid_mod_bwd_t* a, void* stream) {
    MD_CHECK(a, "resid_mod_bwd: null argument");
    MD_CHECK(a->D % 128 == 0, "resid_mod_bwd: D=%d must be a multiple of 128", a->D);
    MD_CHECK(a->dxo || a->dxo_bf || a->dxm, "resid_mod_bwd: need dxo (or dxo_bf) and/or dxm");
    MD_CHECK(!(a->dxo && a->dxo_bf), "resid_mod_bwd: dxo and dxo_bf are alternatives");
    MD_CHECK(!a->dxm || (a->x && a->shift && a->scale && a->gain && a->dshift && a->dscale && a->dgain_part),
             "resid_mod_bwd: modulate backward needs x, shift, scale, gain, dshift, dscale, dgain_part");
    MD_CHECK(!a->y_up || (a->g_up && a->dy_up && a->dg_up), "resid_mod_bwd: residual backward needs g_up, dy_up, dg_up");
    RmbP p;
    p.dxo = a->dxo; p.dxm = a->dxm; p.x = a->x; p.shift = a->shift; p.scale = a->scale; p.gain = a->gain;
    p.y_up = a->y_up; p.g_up = a->g_up; p.dx = a->dx; p.dx_bf = a->dx_bf; p.dshift = a->dshift; p.dscale = a->dscale;
    p.dgain_part = a->dgain_part; p.dy_up = a->dy_up; p.dg_up = a->dg_up;
    p.ldmod = a->ldmod; p.ldg_up = a->ldg_up; p.ldd = a->ldd; p.ldd_up = a->ldd_up; p.T = a->T; p.D = a->D; p.ca = a->ca; p.cb = a->cb;
    p.gscale = a->dgain_scale != 0.f ? a->dgain_scale : 1.f;
    p.rot = a->rot;
    p.dxo16 = (const bf16_t*)a->dxo_bf;
    p.ldx = a->ldx > 0 ? a->ldx : a->D;
    MD_CHECK(p.ldx >= a->D && p.ldx % 4 == 0, "resid_mod_bwd: ldx=%d must be >= D=%d and a multiple of 4", p.ldx, a->D);
    // Small batches: one block per (sample, 128 columns) is too few blocks to stream at the HBM rate (32 samples x 6 = 192 blocks:
    // 60 us for 25 us of traffic).  With scratch given, the rows of a sample are cut into Z pieces (grid z), the column sums
    // parked per piece and added in order by a second small kernel; the gain partials simply become Z times as many.
    int Z = 1;
    if (a->part_scratch) {
        const int blocks = a->n_samples * (a->D / 128);
        while (Z < 8 && blocks * Z < 1024 && a->T % (2 * Z * 8) == 0 && (size_t)(2 * Z) * a->n_samples * 3 * a->D * sizeof(float) <= a->part_scratch_bytes)
            Z *= 2;
    }
    p.part = Z > 1 ? a->part_scratch : nullptr;
#ifndef MAPDIT_RMB_WIDE
#define MAPDIT_RMB_WIDE 1
#endif
    const bool wide = MAPDIT_RMB_WIDE && Z == 1 && a->D % 256 == 0;          // 256 columns per block: 1-KiB row segments
    const int cblocks = wide ? a->D / 256 : a->D / 128;
    if (wide) {
        if (p.rot) hipLaunchKernelGGL((resid_mod_bwd_kernel<true, 64>), dim3(a->n_samples, cblocks, 1), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((resid_mod_bwd_kernel<false, 64>), dim3(a->n_samples, cblocks, 1), dim3(256), 0, (hipStream_t)stream, p);
    } else if (p.rot) hipLaunchKernelGGL((resid_mod_bwd_kernel<true, 32>), dim3(a->n_samples, cblocks, Z), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((resid_mod_bwd_kernel<false, 32>), dim3(a->n_samples, cblocks, Z), dim3(256), 0, (hipStream_t)stream, p);
    MD_LAUNCH_CHECK();
    const int npart = a->n_samples * cblocks * Z;
    const bool own_gain = a->dgain_out != nullptr && a->dxm != nullptr;
    if (Z > 1) {
        hipLaunchKernelGGL(rmb_finish_kernel, dim3(cdiv((long)a->n_samples * a->D, 256) + (own_gain ? 1 : 0)), dim3(256), 0,
                           (hipStream_t)stream, p.part, Z, a->n_samples, a->D, a->dxm ? a->dscale : nullptr, a->dshift, a->ldd,
                           a->y_up ? a->dg_up : nullptr, a->ldd_up, own_gain ? a->dgain_part : nullptr, npart, a->dgain_out);
        MD_LAUNCH_CHECK();
    } else if (own_gain) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->dgain_part, npart, a->dgain_out, 0);
        MD_LAUNCH_CHECK();
    }
    if (a->gain_partials_out) *a->gain_partials_out = own_gain ? 0 : npart;
    return MAPDIT_OK;
}

extern "C" int MD_SYM(rot_modulate_fwd)(const float* x, const float* A, const float* B, int ldc, uint16_t* out, int n_samples, int T,
                                        int D, void* stream) {
    MD_CHECK(x && A && B && out && n_samples > 0, "rot_modulate_fwd: null/empty argument");
    MD_CHECK(D % 8 == 0 && ldc % 4 == 0 && ((((uintptr_t)A | (uintptr_t)B)) & 15) == 0, "rot_modulate_fwd: D %% 8, ldc %% 4, 16-byte aligned rows");
    const long total8 = (long)n_samples * T * (D / 8);
    const int grid = (int)((total8 + 255) / 256 < 8192 ? (total8 + 255) / 256 : 8192);
    hipLaunchKernelGGL(rot_modulate_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, A, B, ldc, (bf16_t*)out, total8, D, T);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

#if MAPDIT_DT == 0
extern "C" int mapdit_rot_coef_fwd(const float* theta, const float* scale, int ldm, const float* gain, float* A, float* B, int ldc,
                                   int n_samples, int D, void* stream) {
    MD_CHECK(theta && scale && gain && A && B && n_samples > 0, "rot_coef_fwd: null/empty argument");
    MD_CHECK(D % 2 == 0 && ldm % 2 == 0 && ldc % 2 == 0 && ((((uintptr_t)scale | (uintptr_t)A | (uintptr_t)B)) & 7) == 0,
             "rot_coef_fwd: even D / row strides and 8-byte aligned rows");
    hipLaunchKernelGGL(rot_coef_fwd_kernel, dim3(cdiv((long)n_samples * (D / 2), 256)), dim3(256), 0, (hipStream_t)stream, theta, scale,
                       ldm, gain, A, B, ldc, n_samples, D);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int mapdit_rot_coef_fwd_all(const float* mod, int ld_mod, const int* theta_off, const int* scale_off, const float* const* gains,
                                       int n_slots, float* A, float* B, int ldc, int n_samples, int D, void* stream) {
    MD_CHECK(mod && theta_off && scale_off && gains && A && B && n_samples > 0 && n_slots > 0, "rot_coef_fwd_all: null/empty argument");
    MD_CHECK(n_slots <= MAPDIT_ROT_MAX_SLOTS, "rot_coef_fwd_all: %d slots (at most %d)", n_slots, MAPDIT_ROT_MAX_SLOTS);
    MD_CHECK(D % 2 == 0 && ld_mod % 2 == 0 && ldc % 2 == 0 && ((((uintptr_t)mod | (uintptr_t)A | (uintptr_t)B)) & 7) == 0 && n_samples <= 65535,
             "rot_coef_fwd_all: even D / row strides, 8-byte aligned rows, at most 65535 samples");
    RotSlots sl;
    for (int s = 0; s < n_slots; ++s) {
        MD_CHECK(gains[s] && scale_off[s] % 2 == 0, "rot_coef_fwd_all: slot %d: null gain or odd scale offset", s);
        sl.gain[s] = gains[s]; sl.theta_off[s] = theta_off[s]; sl.scale_off[s] = scale_off[s];
    }
    hipLaunchKernelGGL(rot_coef_fwd_all_kernel, dim3(cdiv(D / 2, 256), n_slots, n_samples), dim3(256), 0, (hipStream_t)stream, mod, ld_mod,
                       sl, A, B, ldc, D);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int mapdit_rot_coef_bwd(const float* dA, const float* dB, int ldc, const float* theta, const float* scale, int ldm,
                                   const float* gain, float* dtheta, float* dscale, int ldd, float* dgain_part, float dgain_scale,
                                   int n_samples, int D, void* stream) {
    MD_CHECK(dA && dB && theta && scale && gain && dtheta && dscale && dgain_part && n_samples > 0, "rot_coef_bwd: null/empty argument");
    MD_CHECK(D % 2 == 0 && ldm % 2 == 0 && ldc % 2 == 0 && ldd % 2 == 0, "rot_coef_bwd: even D and row strides");
    hipLaunchKernelGGL(rot_coef_bwd_kernel, dim3(cdiv(D / 2, 256), n_samples), dim3(256), 0, (hipStream_t)stream, dA, dB, ldc, theta,
                       scale, ldm, gain, dtheta, dscale, ldd, dgain_part, dgain_scale != 0.f ? dgain_scale : 1.f, D);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
#endif

#if MAPDIT_DT == 0
extern "C" int mapdit_reduce_partials(const float* part, int count, float* out, int accumulate, void* stream) {
    MD_CHECK(part && out && count > 0, "reduce_partials: null/empty argument");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, count, out, accumulate);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
#endif

extern "C" int MD_SYM(qkv_split_generic)(const uint16_t*, int, int, int, int, uint16_t*, uint16_t*, uint16_t*, void*);
extern "C" int MD_SYM(qkv_merge_bwd_generic)(const uint16_t*, int, int, int, int, const uint16_t*, const uint16_t*, const uint16_t*,
                                            uint16_t*, void*);

// head_dim 72 (DiT-XL): coalesced chunk-per-thread kernels of attention72.hip
int MD_SYM(qkv_split72)(const uint16_t*, int, int, int, uint16_t*, uint16_t*, uint16_t*, void*);
int MD_SYM(qkv_merge_bwd72)(const uint16_t*, int, int, int, const uint16_t*, const uint16_t*, const uint16_t*, uint16_t*, void*);

extern "C" int MD_SYM(qkv_split)(const uint16_t* qkv, int B, int T, int H, int head_dim, uint16_t* qn, uint16_t* kn,
                                uint16_t* v, void* stream) {
    MD_CHECK(qkv && qn && kn && v, "qkv_split: null argument");
    if (head_dim == 72 && H <= 37) return MD_SYM(qkv_split72)(qkv, B, T, H, qn, kn, v, stream);
    if (head_dim != 64 || T % 64 != 0) return MD_SYM(qkv_split_generic)(qkv, B, T, H, head_dim, qn, kn, v, stream);
    hipLaunchKernelGGL(qkv_split_kernel, dim3(T / 64, H, B), dim3(256), 0, (hipStream_t)stream, qkv, T, H, qn, kn, v);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(qkv_merge_bwd)(const uint16_t* qkv, int B, int T, int H, int head_dim, const uint16_t* dqn,
                                    const uint16_t* dkn, const uint16_t* dv, uint16_t* dqkv, void* stream) {
    MD_CHECK(qkv && dqn && dkn && dv && dqkv, "qkv_merge_bwd: null argument");
    if (head_dim == 72 && H <= 37) return MD_SYM(qkv_merge_bwd72)(qkv, B, T, H, dqn, dkn, dv, dqkv, stream);
    if (head_dim != 64 || T % 64 != 0) return MD_SYM(qkv_merge_bwd_generic)(qkv, B, T, H, head_dim, dqn, dkn, dv, dqkv, stream);
    hipLaunchKernelGGL(qkv_merge_bwd_kernel, dim3(T / 64, H, B), dim3(256), 0, (hipStream_t)stream, qkv, T, H, dqn, dkn, dv, dqkv);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM_MPSILU_TO_16(const float* x, uint16_t* out, long n, void* stream) {
    MD_CHECK(x && out && n > 0, "mpsilu_to_bf16: null/empty argument");
    hipLaunchKernelGGL(mpsilu_f32_to_bf16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, n);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM_F32_TO_16(const float* x, uint16_t* out, long n, float alpha, void* stream) {
    MD_CHECK(x && out && n > 0, "f32_to_bf16: null/empty argument");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, n, alpha);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
