// Gaussian-diffusion pointwise math: q_sample, the MSE + variational-bound training loss
// with its gradient, and the fused p_mean_variance + p_sample update.  Schedule tables live on the device as
// fp32 arrays (uploaded once), replacing the per-call numpy uploads of reference gaussian_diffusion.py:861-873.
//
// Table layout `tab` = 8 rows of `nsteps` floats:
//   0 sqrt_alphas_cumprod        1 sqrt_one_minus_alphas_cumprod   2 sqrt_recip_alphas_cumprod
//   3 sqrt_recipm1_alphas_cumprod 4 posterior_log_variance_clipped  5 log(betas)
//   6 posterior_mean_coef1        7 posterior_mean_coef2
#include "common.h"

MAPDIT_DEFINE_DEV_ERROR(diffusion)

namespace {

#define INV_LN2 1.44269504088896341f

// x_t = sqrt(acp[t]) x0 + sqrt(1-acp[t]) noise            (reference gaussian_diffusion.py:215-230)
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const long* __restrict__ t,
                                const float* __restrict__ tab, int nsteps, float* __restrict__ xt, long total, int per) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long tt = MAPDIT_CHECKED_INDEX(diffusion, t[i / per], nsteps, MAPDIT_DEVERR_TIMESTEP);
    xt[i] = tab[tt] * x0[i] + tab[nsteps + tt] * noise[i];
}

__device__ __forceinline__ float cdf_approx(float x, float* dcdf) {
    // diffusion_utils.py:39-44
    const float k = 0.79788456080286536f;   // sqrt(2/pi)
    const float u = k * (x + 0.044715f * x * x * x);
    const float th = tanhf(u);
    *dcdf = 0.5f * (1.f - th * th) * k * (1.f + 3.f * 0.044715f * x * x);
    return 0.5f * (1.f + th);
}

// training_losses, MSE + LEARNED_RANGE branch (reference gaussian_diffusion.py:715-787, 682-713,
// diffusion_utils.py:10-37,62-88).  One block per sample.  Writes mse[n], vb[n], loss[n] and
// G[n, 0:C] = d mse[n] / d eps,  G[n, C:2C] = d vb[n] / d v  (the vb term sees eps detached).
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ mo, const float* __restrict__ x0,
                                                 const float* __restrict__ xt, const float* __restrict__ noise,
                                                 const long* __restrict__ t, const float* __restrict__ tab, int nsteps,
                                                 float* __restrict__ mse, float* __restrict__ vb, float* __restrict__ loss,
                                                 float* __restrict__ G, int per /* C*H*W */) {
    __shared__ float red[2][4];
    const int n = blockIdx.x;
    const long tt = MAPDIT_CHECKED_INDEX(diffusion, t[n], nsteps, MAPDIT_DEVERR_TIMESTEP);
    const float ra = tab[2 * nsteps + tt], rm1 = tab[3 * nsteps + tt], minlog = tab[4 * nsteps + tt],
                maxlog = tab[5 * nsteps + tt], c1 = tab[6 * nsteps + tt], c2 = tab[7 * nsteps + tt];
    const float inv_per = 1.f / (float)per;
    const bool first = tt == 0;
    float a_mse = 0.f, a_vb = 0.f;
    for (int e = threadIdx.x; e < per; e += 256) {
        const size_t ie = (size_t)n * 2 * per + e, iv = ie + per, ix = (size_t)n * per + e;
        const float eps = mo[ie], v = mo[iv], x_0 = x0[ix], x_t = xt[ix], nz = noise[ix];
        const float diff = nz - eps;
        a_mse += diff * diff;
        G[ie] = -2.f * diff * inv_per;
        const float frac = (v + 1.f) * 0.5f;
        const float lv = frac * maxlog + (1.f - frac) * minlog;
        const float xs = ra * x_t - rm1 * eps;
        const float mean = c1 * xs + c2 * x_t;
        const float tmean = c1 * x_0 + c2 * x_t;
        const float dlv_dv = 0.5f * (maxlog - minlog);
        float term, dterm_dlv;
        if (!first) {
            const float e1 = expf(minlog - lv), e2 = expf(-lv), dm = tmean - mean;
            term = 0.5f * (-1.f + lv - minlog + e1 + dm * dm * e2);
            dterm_dlv = 0.5f * (1.f - e1 - dm * dm * e2);
        } else {
            const float cx = x_0 - mean, inv = expf(-0.5f * lv);
            const float pin = inv * (cx + 1.f / 255.f), mnn = inv * (cx - 1.f / 255.f);
            float dcp, dcm;
            const float cp = cdf_approx(pin, &dcp), cm = cdf_approx(mnn, &dcm);
            float lp, dlp;   // log prob and d log prob / d lv   (d inv / d lv = -inv/2 => d pin / d lv = -pin/2)
            if (x_0 < -0.999f) {
                lp = logf(fmaxf(cp, 1e-12f));
                dlp = cp > 1e-12f ? dcp * (-0.5f * pin) / cp : 0.f;
            } else if (x_0 > 0.999f) {
                lp = logf(fmaxf(1.f - cm, 1e-12f));
                dlp = (1.f - cm) > 1e-12f ? -dcm * (-0.5f * mnn) / (1.f - cm) : 0.f;
            } else {
                const float dl = cp - cm;
                lp = logf(fmaxf(dl, 1e-12f));
                dlp = dl > 1e-12f ? (dcp * (-0.5f * pin) - dcm * (-0.5f * mnn)) / dl : 0.f;
            }
            term = -lp;
            dterm_dlv = -dlp;
        }
        a_vb += term;
        G[iv] = dterm_dlv * dlv_dv * inv_per * INV_LN2;
    }
    a_mse = wave_sum(a_mse);
    a_vb = wave_sum(a_vb);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a_mse; red[1][threadIdx.x >> 6] = a_vb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * inv_per;
        const float b = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * inv_per * INV_LN2;
        mse[n] = m; vb[n] = b; loss[n] = m + b;
    }
}

// dout[n, 0:C] = (gl[n]+gm[n]) G[n, 0:C];  dout[n, C:2C] = (gl[n]+gv[n]) G[n, C:2C]
__global__ void loss_bwd_kernel(const float* __restrict__ G, const float* __restrict__ gl, const float* __restrict__ gm,
                                const float* __restrict__ gv, float* __restrict__ dout, long total, int per) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i / (2 * per));
    const bool is_v = (i % (2 * per)) >= per;
    const float g = (gl ? gl[n] : 0.f) + (is_v ? (gv ? gv[n] : 0.f) : (gm ? gm[n] : 0.f));
    dout[i] = g * G[i];
}

// p_mean_variance + p_sample (reference gaussian_diffusion.py:254-332, 376-417) for EPSILON / LEARNED_RANGE.
// `t` holds each sample's index into the (respaced) schedule, on the device, so the launch replays from a hipGraph.
__global__ void p_sample_kernel(const float* __restrict__ mo, const float* __restrict__ x, const float* __restrict__ noise,
                                const long* __restrict__ t, const float* __restrict__ tab, int nsteps, int clip,
                                float* __restrict__ sample, float* __restrict__ xstart, long total, int per) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long n = i / per, e = i % per;
    const long tt = MAPDIT_CHECKED_INDEX(diffusion, t[n], nsteps, MAPDIT_DEVERR_TIMESTEP);
    const float eps = mo[n * 2 * per + e], v = mo[n * 2 * per + per + e], x_t = x[i];
    const float ra = tab[2 * nsteps + tt], rm1 = tab[3 * nsteps + tt], minlog = tab[4 * nsteps + tt],
                maxlog = tab[5 * nsteps + tt], c1 = tab[6 * nsteps + tt], c2 = tab[7 * nsteps + tt];
    const float frac = (v + 1.f) * 0.5f;
    const float lv = frac * maxlog + (1.f - frac) * minlog;
    float xs = ra * x_t - rm1 * eps;
    if (clip) xs = fminf(fmaxf(xs, -1.f), 1.f);
    const float mean = c1 * xs + c2 * x_t;
    sample[i] = mean + (tt != 0 ? expf(0.5f * lv) * noise[i] : 0.f);
    if (xstart) xstart[i] = xs;
}

// DDIM step (reference gaussian_diffusion.py:513-567) and its reverse ODE step (:569-605) for EPSILON / LEARNED_RANGE models:
// x0^ from the predicted noise (clipped on request), the noise re-derived from x0^ (_predict_eps_from_xstart), then Eq. 12 of
// Song et al. 2020.  dtab = [alphas_cumprod | alphas_cumprod_prev | alphas_cumprod_next] (fp32 rows of the respaced schedule).
__global__ void ddim_kernel(const float* __restrict__ mo, const float* __restrict__ x, const float* __restrict__ noise,
                            const long* __restrict__ t, const float* __restrict__ tab, const float* __restrict__ dtab, int nsteps,
                            int clip, float eta, int reverse, float* __restrict__ sample, float* __restrict__ xstart, long total,
                            int per) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long n = i / per, e = i % per;
    const long tt = MAPDIT_CHECKED_INDEX(diffusion, t[n], nsteps, MAPDIT_DEVERR_TIMESTEP);
    const float epsm = mo[n * 2 * per + e], x_t = x[i];
    const float ra = tab[2 * nsteps + tt], rm1 = tab[3 * nsteps + tt];
    float xs = ra * x_t - rm1 * epsm;
    if (clip) xs = fminf(fmaxf(xs, -1.f), 1.f);
    const float eps = (ra * x_t - xs) / rm1;
    float out;
    if (reverse) {
        const float abn = dtab[2 * nsteps + tt];
        out = xs * sqrtf(abn) + sqrtf(1.f - abn) * eps;
    } else {
        const float ab = dtab[tt], abp = dtab[nsteps + tt];
        const float sigma = eta * sqrtf((1.f - abp) / (1.f - ab)) * sqrtf(1.f - ab / abp);
        out = xs * sqrtf(abp) + sqrtf(1.f - abp - sigma * sigma) * eps;
        if (tt != 0) out += sigma * noise[i];
    }
    sample[i] = out;
    if (xstart) xstart[i] = xs;
}

}  // namespace

extern "C" int mapdit_ddim_step(const float* model_out, const float* x, const float* noise, const int64_t* t, const float* tab,
                                const float* dtab, int nsteps, int clip_denoised, float eta, int reverse, float* sample,
                                float* pred_xstart, int N, int per_sample, void* stream) {
    MD_CHECK(model_out && x && t && tab && dtab && sample && N > 0, "ddim_step: null/empty argument");
    MD_CHECK(reverse || noise, "ddim_step: the forward step needs the noise tensor");
    MD_CHECK(!reverse || eta == 0.f, "ddim_step: the reverse ODE is deterministic (eta must be 0)");
    const long total = (long)N * per_sample;
    hipLaunchKernelGGL(ddim_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, model_out, x, noise, (const long*)t, tab,
                       dtab, nsteps, clip_denoised, eta, reverse, sample, pred_xstart, total, per_sample);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int mapdit_q_sample(const float* x0, const float* noise, const int64_t* t, const float* tab, int nsteps, float* xt,
                               int N, int per_sample, void* stream) {
    MD_CHECK(x0 && noise && t && tab && xt && N > 0, "q_sample: null/empty argument");
    const long total = (long)N * per_sample;
    hipLaunchKernelGGL(q_sample_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x0, noise, (const long*)t, tab,
                       nsteps, xt, total, per_sample);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int mapdit_loss_fwd(const float* model_out, const float* x0, const float* xt, const float* noise, const int64_t* t,
                               const float* tab, int nsteps, float* mse, float* vb, float* loss, float* G, int N, int per_sample,
                               void* stream) {
    MD_CHECK(model_out && x0 && xt && noise && t && tab && mse && vb && loss && G && N > 0, "loss_fwd: null/empty argument");
    hipLaunchKernelGGL(loss_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, model_out, x0, xt, noise, (const long*)t, tab,
                       nsteps, mse, vb, loss, G, per_sample);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int mapdit_loss_bwd(const float* G, const float* g_loss, const float* g_mse, const float* g_vb, float* dout, int N,
                               int per_sample, void* stream) {
    MD_CHECK(G && dout && N > 0, "loss_bwd: null/empty argument");
    const long total = (long)N * 2 * per_sample;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, G, g_loss, g_mse, g_vb, dout,
                       total, per_sample);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}

extern "C" int mapdit_psample_step(const float* model_out, const float* x, const float* noise, const int64_t* t, const float* tab,
                                   int nsteps, int clip_denoised, float* sample, float* pred_xstart, int N, int per_sample,
                                   void* stream) {
    MD_CHECK(model_out && x && noise && t && tab && sample && N > 0, "psample_step: null/empty argument");
    const long total = (long)N * per_sample;
    hipLaunchKernelGGL(p_sample_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, model_out, x, noise, (const long*)t, tab,
                       nsteps, clip_denoised, sample, pred_xstart, total, per_sample);
    MD_LAUNCH_CHECK();
    return MAPDIT_OK;
}
