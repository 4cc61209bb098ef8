// Weight-side kernels: weight normalisation forward/backward and the fused Adam + EMA update.
// All are HBM-bound streams over the parameters; one wave per weight row,
// 16-byte accesses, fp32 arithmetic.
#include "common.h"

MD_NS_OPEN

// One wave per row.  Pass 1: sum of squares.  Pass 2 (row is L1/L2-hot): rewrite the master weight
// (forced weight norm, reference mp_linear.py:38-40 / mp_embedding.py:17-19) and emit the effective
// weight  w = out_scale * Wn / (|Wn| + eps)  (mp_linear.py:44: normalize(W)/sqrt(in) == W/(|W|+eps)).
// w3 (may be NULL): the effective row as a two-term bf16 split laid out along the reduction index, [hi | lo | hi] with row stride
// 3 cols - the B operand of the fp32-accurate GEMMs (precise.hip, MAPDIT_SPLIT_B): x = hi + lo to ~2^-17.
__device__ __forceinline__ void weightnorm_fwd_row(float* __restrict__ W, int row, int rows, int cols, int forced, float out_scale,
                                                   bf16_t* __restrict__ wb, float* __restrict__ wf, float* __restrict__ inv,
                                                   bf16_t* __restrict__ w3 = nullptr) {
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    // bit 1 of `forced` (MAPDIT_WN_PLAIN; README.md:60 --no-use-weight-normalization, parity unpinned): the effective weight is
    // out_scale * W / sqrt(cols) - mp_linear.py:44 without its normalize(); the in-place rewrite of bit 0 stays what it is
    const bool plain = (forced & MAPDIT_WN_PLAIN) != 0;
    forced &= 1;
    float* w = W + (size_t)row * cols;
    float ss = 0.f;
    const bool vec = (cols & 3) == 0;
    if (vec) {
#pragma unroll 3
        for (int c = lane * 4; c < cols; c += 256) {       // (several row chunks in flight: one wave owns the row)
            float4 v = *(const float4*)(w + c);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    } else {
        for (int c = lane; c < cols; c += 64) ss += w[c] * w[c];
    }
    ss = wave_sum(ss);
    float n = sqrtf(ss);
    float f = 1.f;                       // factor applied to the stored master weight
    if (forced) {
        f = sqrtf((float)cols) / (n + NORM_EPS);
        n = n * f;                       // norm of the rewritten row
    }
    const float s = plain ? 1.f / sqrtf((float)cols) : 1.f / (n + NORM_EPS);
    const float e = f * s * out_scale;   // w_eff = W_old * f * s * out_scale
    if (inv && lane == 0) inv[row] = s;
    if (vec) {
#pragma unroll 3
        for (int c = lane * 4; c < cols; c += 256) {
            float4 v = *(const float4*)(w + c);
            if (forced) *(float4*)(w + c) = make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
            float4 o = make_float4(v.x * e, v.y * e, v.z * e, v.w * e);
            if (wf) *(float4*)(wf + (size_t)row * cols + c) = o;
            if (wb) {
                uint2 u;
                u.x = pack16(o.x, o.y);
                u.y = pack16(o.z, o.w);
                *(uint2*)(wb + (size_t)row * cols + c) = u;
            }
            if (w3) {
                const float ov[4] = {o.x, o.y, o.z, o.w};
                bf16_t hi[4], lo[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { hi[k] = f2bf(ov[k]); lo[k] = f2bf(ov[k] - bf2f(hi[k])); }
                bf16_t* d = w3 + (size_t)row * 3 * cols + c;
                const uint2 uh = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
                const uint2 ul = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
                *(uint2*)(d) = uh;
                *(uint2*)(d + cols) = ul;
                *(uint2*)(d + 2 * cols) = uh;
            }
        }
    } else {
        for (int c = lane; c < cols; c += 64) {
            const float v = w[c];
            if (forced) w[c] = v * f;
            if (wf) wf[(size_t)row * cols + c] = v * e;
            if (wb) wb[(size_t)row * cols + c] = cvt16(v * e);
            if (w3) {
                const bf16_t hi = f2bf(v * e), lo = f2bf(v * e - bf2f(hi));
                bf16_t* d = w3 + (size_t)row * 3 * cols + c;
                d[0] = hi; d[cols] = lo; d[2 * cols] = hi;
            }
        }
    }
}

__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(float* __restrict__ W, int rows, int cols, int forced,
                                                           float out_scale, bf16_t* __restrict__ wb,
                                                           float* __restrict__ wf, float* __restrict__ inv) {
    weightnorm_fwd_row(W, blockIdx.x * 4 + (threadIdx.x >> 6), rows, cols, forced, out_scale, wb, wf, inv);
}

// Every weight of the model in ONE launch: a workgroup finds its job by binary search over the jobs' first-workgroup
// prefix (the table lives in device memory, built once per binding).
__global__ __launch_bounds__(256) void weightnorm_fwd_batch_kernel(const mapdit_wn_job_t* __restrict__ jobs, int njobs, int forced) {
    int lo = 0, hi = njobs - 1;
    const int blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const mapdit_wn_job_t j = jobs[lo];
    weightnorm_fwd_row(j.W, (blk - j.first_block) * 4 + (threadIdx.x >> 6), j.rows, j.cols, (forced & 1) | (j.flags & MAPDIT_WN_PLAIN), j.out_scale, j.w_bf16, j.w_f32,
                       nullptr, j.w_split3);
}

#ifdef MAPDIT_WN_DEBUG
// (tools/dw_stress.py builds this variant: the row's reduced scalars and every lane's partial dot product, for bisecting)
__device__ float* g_wn_dbg = nullptr;       // [rows][4 + 64 + 64]: ss, gw, a1, a2, the 64 per-lane partials of gw, those of ss
extern "C" __global__ void mapdit_wn_dbg_set_kernel(float* p) { g_wn_dbg = p; }
extern "C" void mapdit_wn_dbg_set(float* p) {
    hipLaunchKernelGGL(mapdit_wn_dbg_set_kernel, dim3(1), dim3(1), 0, 0, p);
    (void)hipDeviceSynchronize();
}
#endif

// Backward of w = out_scale * W / (n + eps), n = |W|:  dW = out_scale * (G/(n+eps) - W (G.W) / (n (n+eps)^2)).
// G may arrive as `nslabs` split-K partial sums: pass 1 adds them in slab order (deterministic) and, when there is
// more than one, parks the sum in slab 0 so that pass 2 re-reads one L2-hot row instead of all slabs again.
template <bool VEC>
__device__ __forceinline__ void weightnorm_bwd_row(const float* __restrict__ W, float* __restrict__ G, int ldg, int nslabs,
                                                   long slab_stride, float* __restrict__ dW, int row, int rows, int cols,
                                                   float out_scale, int accumulate) {
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* w = W + (size_t)row * cols;
    float* g = G + (size_t)row * ldg;
    float* d = dW + (size_t)row * cols;
    float ss = 0.f, gw = 0.f;
    if (VEC) {
#pragma unroll 3
        for (int c = lane * 4; c < cols; c += 256) {
            float4 b = *(const float4*)(g + c);
            // the slabs are added in slab order (deterministic), but up to six loads are in flight at a time: one wave owns a row, a CU
            // holds ~12 such waves, and a load-wait-add chain per slab left the kernel latency-bound (61 % of the HBM rate)
            int s = 1;
            for (; s + 5 < nslabs; s += 6) {
                const float* gs = g + (size_t)s * slab_stride + c;
                float4 t[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) t[k] = *(const float4*)(gs + (size_t)k * slab_stride);
#pragma unroll
                for (int k = 0; k < 6; ++k) { b.x += t[k].x; b.y += t[k].y; b.z += t[k].z; b.w += t[k].w; }
            }
            for (; s + 1 < nslabs; s += 2) {
                const float* gs = g + (size_t)s * slab_stride + c;
                const float4 t0 = *(const float4*)(gs), t1 = *(const float4*)(gs + slab_stride);
                b.x += t0.x; b.y += t0.y; b.z += t0.z; b.w += t0.w;
                b.x += t1.x; b.y += t1.y; b.z += t1.z; b.w += t1.w;
            }
            for (; s < nslabs; ++s) {
                const float4 t = *(const float4*)(g + (size_t)s * slab_stride + c);
                b.x += t.x; b.y += t.y; b.z += t.z; b.w += t.w;
            }
            if (nslabs > 1) *(float4*)(g + c) = b;
            const float4 a = *(const float4*)(w + c);
            ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
            gw += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
    } else {
        for (int c = lane; c < cols; c += 64) {
            float b = g[c];
            for (int s = 1; s < nslabs; ++s) b += g[(size_t)s * slab_stride + c];
            if (nslabs > 1) g[c] = b;
            const float a = w[c];
            ss += a * a;
            gw += a * b;
        }
    }
#ifdef MAPDIT_WN_DEBUG
    if (g_wn_dbg) { g_wn_dbg[(size_t)row * 132 + 4 + lane] = gw; g_wn_dbg[(size_t)row * 132 + 68 + lane] = ss; }
#endif
    ss = wave_sum(ss);
    gw = wave_sum(gw);
    const float n = sqrtf(ss);
    // bit 1 of `accumulate` (MAPDIT_WN_PLAIN): w = out_scale * W / sqrt(cols), so dW = out_scale * G / sqrt(cols) and nothing is projected out
    const bool plain = (accumulate & MAPDIT_WN_PLAIN) != 0;
    accumulate &= 1;
    const float a1 = plain ? out_scale / sqrtf((float)cols) : out_scale / (n + NORM_EPS);
    const float a2 = plain ? 0.f : out_scale * gw / (fmaxf(n, 1e-30f) * (n + NORM_EPS) * (n + NORM_EPS));
#ifdef MAPDIT_WN_DEBUG
    if (g_wn_dbg && lane == 0) { float* q = g_wn_dbg + (size_t)row * 132; q[0] = ss; q[1] = gw; q[2] = a1; q[3] = a2; }
#endif
    if (VEC) {
#pragma unroll 3
        for (int c = lane * 4; c < cols; c += 256) {      // each lane re-reads exactly the columns it wrote above
            const float4 b = *(const float4*)(g + c), a = *(const float4*)(w + c);
            float4 r = make_float4(a1 * b.x - a2 * a.x, a1 * b.y - a2 * a.y, a1 * b.z - a2 * a.z, a1 * b.w - a2 * a.w);
            if (accumulate) {
                const float4 o = *(const float4*)(d + c);
                r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
            }
            *(float4*)(d + c) = r;
        }
    } else {
        for (int c = lane; c < cols; c += 64) {
            float r = a1 * g[c] - a2 * w[c];
            if (accumulate) r += d[c];
            d[c] = r;
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void weightnorm_bwd_kernel(const float* __restrict__ W, float* __restrict__ G, int ldg,
                                                           int nslabs, long slab_stride, float* __restrict__ dW, int rows,
                                                           int cols, float out_scale, int accumulate) {
    weightnorm_bwd_row<VEC>(W, G, ldg, nslabs, slab_stride, dW, blockIdx.x * 4 + (threadIdx.x >> 6), rows, cols, out_scale, accumulate);
}

// Several weights (or row ranges of weights) in ONE launch, in place: job.W = the master rows, job.w_f32 = their gradient rows G, which
// become dW (one slab, vector path).  The Jacobians of a rank's rows under sharded weight passes: 60 launches of a few microseconds of
// work each otherwise.  A workgroup finds its job by binary search over first_block, as weightnorm_fwd_batch_kernel does.
__global__ __launch_bounds__(256) void weightnorm_bwd_batch_kernel(const mapdit_wn_job_t* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    const int blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const mapdit_wn_job_t j = jobs[lo];
    weightnorm_bwd_row<true>(j.W, j.w_f32, j.cols, 1, 0, j.w_f32, (blk - j.first_block) * 4 + (threadIdx.x >> 6), j.rows, j.cols, j.out_scale,
                             j.flags & MAPDIT_WN_PLAIN);
}

// Up to four weights' Jacobians, each over its own split-K slabs, in ONE launch (round 5: behind the grouped weight-gradient launch of gemm.hip;
// the items arrive by value as kernel arguments - no device table).  The same row function as weightnorm_bwd_kernel<true>: the same bits.
struct WnBwdGroup {
    const float* W[4]; float* G[4]; float* dW[4];
    int rows[4], cols[4], ldg[4], nslabs[4], flags[4], first[5];
    long slab[4];
    float scale[4];
    int n;
};
__global__ __launch_bounds__(256) void weightnorm_bwd_group_kernel(WnBwdGroup g) {
    const int blk = blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && blk >= g.first[i + 1]) ++i;
    weightnorm_bwd_row<true>(g.W[i], g.G[i], g.ldg[i], g.nslabs[i], g.slab[i], g.dW[i], (blk - g.first[i]) * 4 + (threadIdx.x >> 6), g.rows[i], g.cols[i],
                             g.scale[i], g.flags[i]);
}

// The same pass for a SIDE STREAM (round 5): the engine runs the Jacobian of weight i beside the weight-gradient GEMM of weight i + 1.
// A GEMM workgroup holds two 228-register waves per SIMD (464 of the 512 registers per lane after the allocation granule), so a
// co-resident wave may own 48 registers and no LDS (tools/overlap_probe.py: a streaming kernel that fits beside the persistent GEMM hides
// about half of its time).  One column chunk at a time, up to four slab loads in flight (the form above unrolls three chunks of six: 76
// registers).  Same additions in the same order as weightnorm_bwd_kernel<true>: the results are the same bits.
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(48)))
void weightnorm_bwd_slim_kernel(const float* __restrict__ W, float* __restrict__ G, int ldg, int nslabs, long slab_stride,
                                float* __restrict__ dW, int rows, int cols, float out_scale, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* w = W + (size_t)row * cols;
    float* g = G + (size_t)row * ldg;
    float* d = dW + (size_t)row * cols;
    float ss = 0.f, gw = 0.f;
#pragma unroll 1
    for (int c = lane * 4; c < cols; c += 256) {
        float4 b = *(const float4*)(g + c);
        const float4 a = *(const float4*)(w + c);
        int s = 1;
        for (; s + 3 < nslabs; s += 4) {
            const float* gs = g + (size_t)s * slab_stride + c;
            float4 t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = *(const float4*)(gs + (size_t)k * slab_stride);
#pragma unroll
            for (int k = 0; k < 4; ++k) { b.x += t[k].x; b.y += t[k].y; b.z += t[k].z; b.w += t[k].w; }
        }
        for (; s + 1 < nslabs; s += 2) {
            const float* gs = g + (size_t)s * slab_stride + c;
            const float4 t0 = *(const float4*)(gs), t1 = *(const float4*)(gs + slab_stride);
            b.x += t0.x; b.y += t0.y; b.z += t0.z; b.w += t0.w;
            b.x += t1.x; b.y += t1.y; b.z += t1.z; b.w += t1.w;
        }
        for (; s < nslabs; ++s) {
            const float4 t = *(const float4*)(g + (size_t)s * slab_stride + c);
            b.x += t.x; b.y += t.y; b.z += t.z; b.w += t.w;
        }
        if (nslabs > 1) *(float4*)(g + c) = b;
        ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        gw += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    ss = wave_sum(ss);
    gw = wave_sum(gw);
    const float n = sqrtf(ss);
    const bool plain = (accumulate & MAPDIT_WN_PLAIN) != 0;      // (as weightnorm_bwd_row)
    accumulate &= 1;
    const float a1 = plain ? out_scale / sqrtf((float)cols) : out_scale / (n + NORM_EPS);
    const float a2 = plain ? 0.f : out_scale * gw / (fmaxf(n, 1e-30f) * (n + NORM_EPS) * (n + NORM_EPS));
#pragma unroll 1
    for (int c = lane * 4; c < cols; c += 256) {
        const float4 b = *(const float4*)(g + c), a = *(const float4*)(w + c);
        float4 r = make_float4(a1 * b.x - a2 * a.x, a1 * b.y - a2 * a.y, a1 * b.z - a2 * a.z, a1 * b.w - a2 * a.w);
        if (accumulate) {
            const float4 o = *(const float4*)(d + c);
            r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
        }
        *(float4*)(d + c) = r;
    }
}

// Adam + the two EMA copies on elements i .. i + 3 of the flat buffers (the arithmetic of adam_ema_kernel, shared with its multi-range form)
__device__ __forceinline__ void adam_ema_apply4(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                float* __restrict__ e1, float* __restrict__ e2, long i, float step_size, float inv_sqrt_bc2,
                                                float eb1, float eb2, float gs, float b1, float b2, float eps) {
    float4 P = *(float4*)(p + i), G = *(const float4*)(g + i), M = *(float4*)(m + i), V = *(float4*)(v + i);
    float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x, G.y, G.z, G.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float gk = gg[k] * gs;
        mm[k] = b1 * mm[k] + (1.f - b1) * gk;
        vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
        pp[k] -= step_size * mm[k] / (sqrtf(vv[k]) * inv_sqrt_bc2 + eps);
    }
    *(float4*)(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    *(float4*)(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
    *(float4*)(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (e1) {
        float4 E = *(float4*)(e1 + i);
        E.x += eb1 * (pp[0] - E.x); E.y += eb1 * (pp[1] - E.y); E.z += eb1 * (pp[2] - E.z); E.w += eb1 * (pp[3] - E.w);
        *(float4*)(e1 + i) = E;
    }
    if (e2) {
        float4 E = *(float4*)(e2 + i);
        E.x += eb2 * (pp[0] - E.x); E.y += eb2 * (pp[1] - E.y); E.z += eb2 * (pp[2] - E.z); E.w += eb2 * (pp[3] - E.w);
        *(float4*)(e2 + i) = E;
    }
}

// torch.optim.Adam (train.py:57: lr, betas (0.9, 0.99), eps 1e-8, no weight decay) fused with the two
// power-function EMA copies (src/ema.py:135-140: ema.lerp_(param, beta)).  Step-dependent scalars are read
// from a small device array so the launch is graph-replayable:
//   hp[0] = lr / (1 - b1^t)   hp[1] = 1 / sqrt(1 - b2^t)   hp[2] = ema beta (std 0.05)   hp[3] = ema beta (std 0.1)
//   hp[4] = grad scale (1 / world size for the DP mean)
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v,
                                                     float* __restrict__ e1, float* __restrict__ e2, long n,
                                                     const float* __restrict__ hp, mapdit_adam_scalars_t hs, float b1,
                                                     float b2, float eps, const int* __restrict__ status = nullptr, int step = 0) {
    // non-finite gradient guard: the check kernel recorded this step as bad - leave parameters, moments and EMA copies alone
    if (status && status[0] == step) return;
    // hyper-parameters of the step: by value (kernel arguments, nothing crosses PCIe) or, for a caller that replays one captured
    // launch with changing values, from a 5-float device buffer
    const float step_size = hp ? hp[0] : hs.step_size, inv_sqrt_bc2 = hp ? hp[1] : hs.inv_sqrt_bc2, eb1 = hp ? hp[2] : hs.ema_beta_a,
                eb2 = hp ? hp[3] : hs.ema_beta_b, gs = hp ? hp[4] : hs.grad_scale;
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) adam_ema_apply4(p, g, m, v, e1, e2, i, step_size, inv_sqrt_bc2, eb1, eb2, gs, b1, b2, eps);
}

// Many element ranges of the flat buffers in ONE launch (a rank's rows of every sharded weight + the replicated parameters under sharded
// weight passes: 85 ranges): a workgroup owns 4,096 consecutive elements of one range, found by binary search over first_block.
__global__ __launch_bounds__(256) void adam_ema_ranges_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, float* __restrict__ e1, float* __restrict__ e2,
                                                            const mapdit_range_t* __restrict__ ranges, int nranges, mapdit_adam_scalars_t hs,
                                                            float b1, float b2, float eps, const int* __restrict__ status, int step) {
    if (status && status[0] == step) return;
    int lo = 0, hi = nranges - 1;
    const long blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ranges[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const mapdit_range_t r = ranges[lo];
    const long base = r.lo + (blk - r.first_block) * 4096;
    const long end = base + 4096 < r.hi ? base + 4096 : r.hi;
    for (long i = base + threadIdx.x * 4; i + 3 < end; i += 1024)
        adam_ema_apply4(p, g, m, v, e1, e2, i, hs.step_size, hs.inv_sqrt_bc2, hs.ema_beta_a, hs.ema_beta_b, hs.grad_scale, b1, b2, eps);
}

// the non-finite check over the same ranges
__global__ __launch_bounds__(256) void grad_nonfinite_ranges_kernel(const float* __restrict__ g, const mapdit_range_t* __restrict__ ranges, int nranges,
                                                                  int* __restrict__ status, int step) {
    int lo = 0, hi = nranges - 1;
    const long blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ranges[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const mapdit_range_t r = ranges[lo];
    const long base = r.lo + (blk - r.first_block) * 4096;
    const long end = base + 4096 < r.hi ? base + 4096 : r.hi;
    bool bad = false;
    for (long i = base + threadIdx.x * 4; i + 3 < end; i += 1024) {
        const float4 x = *(const float4*)(g + i);
        const float d = (x.x - x.x) + (x.y - x.y) + (x.z - x.z) + (x.w - x.w);
        bad |= !(d == 0.f);
    }
    if (bad) {
        if (atomicExch(&status[0], step) != step) atomicAdd(&status[1], 1);
    }
}

#if 0
    {
        float4 P = *(float4*)(p + i), G = *(const float4*)(g + i), M = *(float4*)(m + i), V = *(float4*)(v + i);
        float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x, G.y, G.z, G.w}, mm[4] = {M.x, M.y, M.z, M.w},
              vv[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * gs;
            mm[k] = b1 * mm[k] + (1.f - b1) * gk;
            vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
            pp[k] -= step_size * mm[k] / (sqrtf(vv[k]) * inv_sqrt_bc2 + eps);
        }
        *(float4*)(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        *(float4*)(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
        *(float4*)(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (e1) {
            float4 E = *(float4*)(e1 + i);
            E.x += eb1 * (pp[0] - E.x); E.y += eb1 * (pp[1] - E.y); E.z += eb1 * (pp[2] - E.z); E.w += eb1 * (pp[3] - E.w);
            *(float4*)(e1 + i) = E;
        }
        if (e2) {
            float4 E = *(float4*)(e2 + i);
            E.x += eb2 * (pp[0] - E.x); E.y += eb2 * (pp[1] - E.y); E.z += eb2 * (pp[2] - E.z); E.w += eb2 * (pp[3] - E.w);
            *(float4*)(e2 + i) = E;
        }
    }
#endif

// Non-finite gradient guard (fp16 engine): any inf / NaN among the n gradients records `step` in status[0]; the first thread to do
// so for this step also counts it in status[1].  Atomics run in the overflow case only (a finite step issues none).
__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const float* __restrict__ g, long n, int* __restrict__ status, int step) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * blockDim.x * 4;
    bool bad = false;
    for (; i + 3 < n; i += stride) {
        const float4 v = *(const float4*)(g + i);
        // x - x is 0 for every finite x and NaN for inf / NaN
        const float d = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
        bad |= !(d == 0.f);
    }
    if (bad) {
        if (atomicExch(&status[0], step) != step) atomicAdd(&status[1], 1);
    }
}

MD_NS_CLOSE

extern "C" int MD_SYM(weightnorm_fwd)(float* W, int rows, int cols, int forced, float out_scale, uint16_t* w_bf16,
                                     float* w_f32, float* inv, void* stream) {
    MD_CHECK(W && rows > 0 && cols > 0, "weightnorm_fwd: null/empty argument");
    MD_CHECK((cols & 3) != 0 || (((uintptr_t)W | (uintptr_t)w_f32) & 15) == 0, "weightnorm_fwd: unaligned pointer");
    hipLaunchKernelGGL(weightnorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, W, rows, cols,
                       forced, out_scale, w_bf16, w_f32, inv);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int MD_SYM(weightnorm_fwd_batch)(const mapdit_wn_job_t* jobs_dev, int njobs, int total_blocks, int forced, void* stream) {
    MD_CHECK(jobs_dev && njobs > 0 && total_blocks > 0, "weightnorm_fwd_batch: null/empty argument");
    hipLaunchKernelGGL(weightnorm_fwd_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs, forced);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

#if MAPDIT_DT == 0
extern "C" int mapdit_weightnorm_bwd(const float* W, float* G, int ldg, int nslabs, long slab_stride, float* dW,
                                     int rows, int cols, float out_scale, int accumulate, void* stream) {
    MD_CHECK(W && G && dW && rows > 0 && cols > 0 && ldg >= cols && nslabs >= 1, "weightnorm_bwd: null/empty argument");
    const bool vec = (cols % 4 == 0) && (ldg % 4 == 0) && (slab_stride % 4 == 0) &&
                     ((((uintptr_t)W | (uintptr_t)G | (uintptr_t)dW) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(weightnorm_bwd_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, W, G, ldg,
                           nslabs, slab_stride, dW, rows, cols, out_scale, accumulate);
    else
        hipLaunchKernelGGL(weightnorm_bwd_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, W, G, ldg,
                           nslabs, slab_stride, dW, rows, cols, out_scale, accumulate);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
#endif

#if MAPDIT_DT == 0
extern "C" int mapdit_weightnorm_bwd_group(int n, const mapdit_wn_bwd_item_t* items, void* stream) {
    MD_CHECK(items && n >= 1 && n <= 4, "weightnorm_bwd_group: 1..4 items");
    WnBwdGroup g{};
    g.n = n;
    for (int i = 0; i < n; ++i) {
        const mapdit_wn_bwd_item_t& it = items[i];
        MD_CHECK(it.W && it.G && it.dW && it.rows > 0 && it.cols > 0 && it.ldg >= it.cols && it.nslabs >= 1, "weightnorm_bwd_group: item %d: null / empty", i);
        MD_CHECK(it.cols % 4 == 0 && it.ldg % 4 == 0 && it.slab_stride % 4 == 0 && !(((uintptr_t)it.W | (uintptr_t)it.G | (uintptr_t)it.dW) & 15),
                 "weightnorm_bwd_group: item %d needs 16-byte aligned rows (cols, ldg, slab_stride multiples of 4)", i);
        g.W[i] = it.W; g.G[i] = it.G; g.dW[i] = it.dW; g.rows[i] = it.rows; g.cols[i] = it.cols; g.ldg[i] = it.ldg; g.nslabs[i] = it.nslabs;
        g.flags[i] = it.flags; g.slab[i] = it.slab_stride; g.scale[i] = it.out_scale;
        g.first[i + 1] = g.first[i] + (it.rows + 3) / 4;
    }
    hipLaunchKernelGGL(weightnorm_bwd_group_kernel, dim3(g.first[n]), dim3(256), 0, (hipStream_t)stream, g);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int mapdit_weightnorm_bwd_batch(const mapdit_wn_job_t* jobs_dev, int njobs, int total_blocks, void* stream) {
    MD_CHECK(jobs_dev && njobs > 0 && total_blocks > 0, "weightnorm_bwd_batch: null/empty argument");
    hipLaunchKernelGGL(weightnorm_bwd_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
#endif

#if MAPDIT_DT == 0
// The 48-register form of mapdit_weightnorm_bwd (same bits) for a launch that is to run beside a GEMM on another stream.  Vector path
// only: cols, ldg, slab_stride multiples of 4 and 16-byte aligned buffers (every weight of the engine).
extern "C" int mapdit_weightnorm_bwd_slim(const float* W, float* G, int ldg, int nslabs, long slab_stride, float* dW,
                                          int rows, int cols, float out_scale, int accumulate, void* stream) {
    MD_CHECK(W && G && dW && rows > 0 && cols > 0 && ldg >= cols && nslabs >= 1, "weightnorm_bwd_slim: null/empty argument");
    const bool vec = (cols % 4 == 0) && (ldg % 4 == 0) && (slab_stride % 4 == 0) &&
                     ((((uintptr_t)W | (uintptr_t)G | (uintptr_t)dW) & 15) == 0);
    if (!vec) return mapdit_weightnorm_bwd(W, G, ldg, nslabs, slab_stride, dW, rows, cols, out_scale, accumulate, stream);
    hipLaunchKernelGGL(weightnorm_bwd_slim_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, W, G, ldg, nslabs,
                       slab_stride, dW, rows, cols, out_scale, accumulate);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
#endif

#if MAPDIT_DT == 0
extern "C" int mapdit_adam_ema_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema_a,
                                    float* ema_b, long n, const float* hyper, float beta1, float beta2, float eps,
                                    void* stream) {
    MD_CHECK(params && grads && exp_avg && exp_avg_sq && hyper && n > 0, "adam_ema_step: null/empty argument");
    MD_CHECK(n % 4 == 0, "adam_ema_step: n=%ld must be a multiple of 4 (pad the flat buffer)", n);
    const int grid = (int)((n / 4 + 255) / 256 < 4096 ? (n / 4 + 255) / 256 : 4096);
    mapdit_adam_scalars_t none = {0.f, 0.f, 0.f, 0.f, 0.f};
    hipLaunchKernelGGL(adam_ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, ema_a, ema_b, n, hyper, none, beta1, beta2, eps, (const int*)nullptr, 0);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
#endif

#if MAPDIT_DT == 0
extern "C" int mapdit_adam_ema_step_scalars(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema_a,
                                            float* ema_b, long n, const mapdit_adam_scalars_t* hyper, float beta1, float beta2,
                                            float eps, void* stream) {
    MD_CHECK(params && grads && exp_avg && exp_avg_sq && hyper && n > 0, "adam_ema_step_scalars: null/empty argument");
    MD_CHECK(n % 4 == 0, "adam_ema_step_scalars: n=%ld must be a multiple of 4 (pad the flat buffer)", n);
    const int grid = (int)((n / 4 + 255) / 256 < 4096 ? (n / 4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(adam_ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, ema_a, ema_b, n, (const float*)nullptr, *hyper, beta1, beta2, eps, (const int*)nullptr, 0);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int mapdit_adam_ema_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema_a,
                                            float* ema_b, long n, const mapdit_adam_scalars_t* hyper, float beta1, float beta2,
                                            float eps, const int* status, int step, void* stream) {
    MD_CHECK(params && grads && exp_avg && exp_avg_sq && hyper && n > 0, "adam_ema_step_guarded: null/empty argument");
    MD_CHECK(n % 4 == 0, "adam_ema_step_guarded: n=%ld must be a multiple of 4 (pad the flat buffer)", n);
    MD_CHECK(status && step > 0, "adam_ema_step_guarded: needs the status words and a step number > 0");
    const int grid = (int)((n / 4 + 255) / 256 < 4096 ? (n / 4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(adam_ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, ema_a, ema_b, n, (const float*)nullptr, *hyper, beta1, beta2, eps, status, step);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int mapdit_adam_ema_step_ranges(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema_a, float* ema_b,
                                           const mapdit_range_t* ranges_dev, int nranges, long total_blocks, const mapdit_adam_scalars_t* hyper,
                                           float beta1, float beta2, float eps, const int* status, int step, void* stream) {
    MD_CHECK(params && grads && exp_avg && exp_avg_sq && hyper && ranges_dev && nranges > 0 && total_blocks > 0 && total_blocks < (1l << 31),
             "adam_ema_step_ranges: null/empty argument");
    MD_CHECK(!status || step > 0, "adam_ema_step_ranges: the guarded form needs a step number > 0");
    hipLaunchKernelGGL(adam_ema_ranges_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       ema_a, ema_b, ranges_dev, nranges, *hyper, beta1, beta2, eps, status, step);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int mapdit_grad_nonfinite_check_ranges(const float* grads, const mapdit_range_t* ranges_dev, int nranges, long total_blocks,
                                                  int* status, int step, void* stream) {
    MD_CHECK(grads && ranges_dev && status && nranges > 0 && total_blocks > 0 && total_blocks < (1l << 31) && step > 0,
             "grad_nonfinite_check_ranges: null/empty argument (step must be > 0)");
    hipLaunchKernelGGL(grad_nonfinite_ranges_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, grads, ranges_dev, nranges,
                       status, step);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int mapdit_grad_nonfinite_check(const float* grads, long n, int* status, int step, void* stream) {
    MD_CHECK(grads && status && n > 0 && step > 0, "grad_nonfinite_check: null/empty argument (step must be > 0)");
    MD_CHECK(n % 4 == 0 && ((uintptr_t)grads & 15) == 0, "grad_nonfinite_check: n=%ld must be a multiple of 4 and the buffer 16-byte aligned", n);
    const int grid = (int)((n / 4 + 255) / 256 < 2048 ? (n / 4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(grad_nonfinite_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, grads, n, status, step);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
#endif
