"""``GaussianDiffusion`` on HIP kernels — the surface of reference diffusion/gaussian_diffusion.py that the
training and sampling scripts use: ``num_timesteps``, the float64 numpy schedule tables, ``q_sample``,
``training_losses``, ``p_mean_variance``, ``p_sample``, ``p_sample_loop`` (+ ``_progressive``).

What differs by design: the schedule tables are uploaded to the GPU once (the reference uploads a numpy table on
every ``_extract_into_tensor`` call, :861-873), the loss / posterior / sampling pointwise math is one fused kernel
each (``mapdit_loss_fwd`` / ``mapdit_psample_step``), and there is no host synchronisation inside a step.

Built: the configuration ``create_diffusion`` produces by default and every reference script uses — epsilon
prediction, learned-range variance, MSE (+ vb) loss.  Other enum members exist for API parity and raise
NotImplementedError when exercised.
"""
import enum
import math

import numpy as np
import torch

from .. import _lib as L


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self in (LossType.KL, LossType.RESCALED_KL)


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    """reference gaussian_diffusion.py:98-122."""
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "squaredcos_cap_v2":
        bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        n = num_diffusion_timesteps
        return np.array([min(1 - bar((i + 1) / n) / bar(i / n), 0.999) for i in range(n)])
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def mean_flat(tensor):
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


class _LossFunction(torch.autograd.Function):
    """mse / vb / loss per sample with the gradient wrt the model output (mapdit_loss_fwd / mapdit_loss_bwd)."""

    @staticmethod
    def forward(ctx, model_output, x_start, x_t, noise, t, tab, nsteps):
        n = model_output.shape[0]
        per = x_start[0].numel()
        mo = model_output.contiguous().float()
        mse, vb, loss = (torch.empty(n, device=mo.device) for _ in range(3))
        G = torch.empty_like(mo)
        with torch.cuda.device(mo.device):
            L.lib().loss_fwd(mo.data_ptr(), x_start.data_ptr(), x_t.data_ptr(), noise.data_ptr(), t.data_ptr(), tab.data_ptr(),
                             nsteps, mse.data_ptr(), vb.data_ptr(), loss.data_ptr(), G.data_ptr(), n, per, L.cur_stream())
        ctx.save_for_backward(G)
        ctx.per = per
        return loss, mse, vb

    @staticmethod
    def backward(ctx, g_loss, g_mse, g_vb):
        (G,) = ctx.saved_tensors
        dout = torch.empty_like(G)
        gl, gm, gv = (None if g is None else g.contiguous().float() for g in (g_loss, g_mse, g_vb))
        with torch.cuda.device(G.device):
            L.lib().loss_bwd(G.data_ptr(), L.ptr(gl), L.ptr(gm), L.ptr(gv), dout.data_ptr(), G.shape[0], ctx.per, L.cur_stream())
        return dout, None, None, None, None, None, None


class GaussianDiffusion:
    """reference gaussian_diffusion.py:144-201 (constructor and tables)."""

    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type):
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert len(betas.shape) == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = (np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
                                               if len(self.posterior_variance) > 1 else np.array([]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self._tab_cache = {}

    # ---- device-resident schedule (layout documented in include/mapdit.h) ------------------------------------------
    def _tables(self, device):
        tab = self._tab_cache.get(device)
        if tab is None:
            rows = [self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, self.sqrt_recip_alphas_cumprod,
                    self.sqrt_recipm1_alphas_cumprod, self.posterior_log_variance_clipped, np.log(self.betas),
                    self.posterior_mean_coef1, self.posterior_mean_coef2]
            tab = torch.from_numpy(np.stack(rows)).float().to(device).contiguous()     # fp32, as _extract_into_tensor
            self._tab_cache[device] = tab
        return tab

    def _ddim_tables(self, device):
        tab = self._tab_cache.get(("ddim", device))
        if tab is None:
            rows = [self.alphas_cumprod, self.alphas_cumprod_prev, self.alphas_cumprod_next]
            tab = torch.from_numpy(np.stack(rows)).float().to(device).contiguous()
            self._tab_cache[("ddim", device)] = tab
        return tab

    def _supported(self):
        if (self.model_mean_type != ModelMeanType.EPSILON or self.model_var_type != ModelVarType.LEARNED_RANGE
                or self.loss_type != LossType.MSE):
            raise NotImplementedError("only the create_diffusion() default (EPSILON, LEARNED_RANGE, MSE) is built; "
                                      f"got {self.model_mean_type}, {self.model_var_type}, {self.loss_type}")

    def _wrap_model(self, model):
        return model

    @staticmethod
    def _prep(x):
        assert x.is_cuda, "the diffusion kernels run on the MI355X only (no CPU path)"
        return x.contiguous().float()

    # ---- forward process ------------------------------------------------------------------------------------------------
    def q_sample(self, x_start, t, noise=None):
        """reference gaussian_diffusion.py:215-230."""
        x_start = self._prep(x_start)
        if noise is None:
            noise = torch.randn_like(x_start)
        assert noise.shape == x_start.shape
        noise = self._prep(noise)
        t = t.to(device=x_start.device, dtype=torch.int64).contiguous()
        out = torch.empty_like(x_start)
        with torch.cuda.device(x_start.device):
            L.lib().q_sample(x_start.data_ptr(), noise.data_ptr(), t.data_ptr(), self._tables(x_start.device).data_ptr(),
                             self.num_timesteps, out.data_ptr(), x_start.shape[0], x_start[0].numel(), L.cur_stream())
        return out

    # ---- training ----------------------------------------------------------------------------------------------------------
    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """reference gaussian_diffusion.py:715-787 -> {"loss", "mse", "vb"}, each [N]."""
        self._supported()
        if model_kwargs is None:
            model_kwargs = {}
        x_start = self._prep(x_start)
        if noise is None:
            noise = torch.randn_like(x_start)
        noise = self._prep(noise)
        t = t.to(device=x_start.device, dtype=torch.int64).contiguous()
        x_t = self.q_sample(x_start, t, noise=noise)
        model_output = model(x_t, t, **model_kwargs)
        B, C = x_t.shape[:2]
        assert model_output.shape == (B, C * 2, *x_t.shape[2:])
        loss, mse, vb = _LossFunction.apply(model_output, x_start, x_t, noise, t, self._tables(x_start.device),
                                            self.num_timesteps)
        return {"vb": vb, "mse": mse, "loss": loss}

    # ---- reverse process ---------------------------------------------------------------------------------------------------
    def _step_math(self, model_output, x, t, noise, clip_denoised):
        x = self._prep(x)
        mo = self._prep(model_output)
        sample, xstart = torch.empty_like(x), torch.empty_like(x)
        with torch.cuda.device(x.device):
            L.lib().psample_step(mo.data_ptr(), x.data_ptr(), noise.data_ptr(), t.data_ptr(), self._tables(x.device).data_ptr(),
                                 self.num_timesteps, int(bool(clip_denoised)), sample.data_ptr(), xstart.data_ptr(),
                                 x.shape[0], x[0].numel(), L.cur_stream())
        return sample, xstart

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """reference gaussian_diffusion.py:254-332 (mean = the p_sample kernel with zero noise)."""
        self._supported()
        if denoised_fn is not None:
            raise NotImplementedError("denoised_fn is not built")
        x = self._prep(x)
        t = t.to(device=x.device, dtype=torch.int64).contiguous()
        B, C = x.shape[:2]
        assert t.shape == (B,)
        model_output = model(x, t, **(model_kwargs or {}))
        assert model_output.shape == (B, C * 2, *x.shape[2:])
        mean, xstart = self._step_math(model_output, x, t, torch.zeros_like(x), clip_denoised)
        tab = self._tables(x.device)
        n = self.num_timesteps
        frac = (model_output[:, C:].float() + 1) / 2
        shape = (-1,) + (1,) * (x.dim() - 1)
        log_var = frac * tab[5][t].view(shape) + (1 - frac) * tab[4][t].view(shape)
        return {"mean": mean, "variance": torch.exp(log_var), "log_variance": log_var, "pred_xstart": xstart, "extra": None,
                "model_output": model_output}

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None):
        """reference gaussian_diffusion.py:376-417."""
        self._supported()
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn are not built")
        x = self._prep(x)
        t = t.to(device=x.device, dtype=torch.int64).contiguous()
        model_output = self._wrap_model(model)(x, t, **(model_kwargs or {}))
        noise = torch.randn_like(x)
        sample, xstart = self._step_math(model_output, x, t, noise, clip_denoised)
        return {"sample": sample, "pred_xstart": xstart}

    # ---- DDIM (reference gaussian_diffusion.py:513-680; no reference script uses it) ----------------------------------
    def _ddim_math(self, model_output, x, t, noise, clip_denoised, eta, reverse):
        x = self._prep(x)
        mo = self._prep(model_output)
        sample, xstart = torch.empty_like(x), torch.empty_like(x)
        with torch.cuda.device(x.device):
            L.lib().ddim_step(mo.data_ptr(), x.data_ptr(), L.ptr(noise), t.data_ptr(), self._tables(x.device).data_ptr(),
                              self._ddim_tables(x.device).data_ptr(), self.num_timesteps, int(bool(clip_denoised)), float(eta),
                              int(bool(reverse)), sample.data_ptr(), xstart.data_ptr(), x.shape[0], x[0].numel(), L.cur_stream())
        return sample, xstart

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        """reference gaussian_diffusion.py:513-567."""
        self._supported()
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn are not built")
        x = self._prep(x)
        t = t.to(device=x.device, dtype=torch.int64).contiguous()
        model_output = self._wrap_model(model)(x, t, **(model_kwargs or {}))
        noise = torch.randn_like(x)
        sample, xstart = self._ddim_math(model_output, x, t, noise, clip_denoised, eta, False)
        return {"sample": sample, "pred_xstart": xstart}

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        """reference gaussian_diffusion.py:569-605: x_{t+1} along the deterministic reverse ODE."""
        assert eta == 0.0, "Reverse ODE only for deterministic path"
        self._supported()
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn are not built")
        x = self._prep(x)
        t = t.to(device=x.device, dtype=torch.int64).contiguous()
        model_output = self._wrap_model(model)(x, t, **(model_kwargs or {}))
        sample, xstart = self._ddim_math(model_output, x, t, None, clip_denoised, 0.0, True)
        return {"sample": sample, "pred_xstart": xstart}

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                         device=None, progress=False, eta=0.0):
        """reference gaussian_diffusion.py:607-636."""
        final = None
        for sample in self.ddim_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                        denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs,
                                                        device=device, progress=progress, eta=eta):
            final = sample
        return final["sample"]

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                     model_kwargs=None, device=None, progress=False, eta=0.0):
        """reference gaussian_diffusion.py:638-680."""
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else torch.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        for i in indices:
            t = torch.full((shape[0],), i, device=device, dtype=torch.int64)
            with torch.no_grad():
                out = self.ddim_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                                       model_kwargs=model_kwargs, eta=eta)
                yield out
                img = out["sample"]

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                      device=None, progress=False):
        """reference gaussian_diffusion.py:419-462."""
        final = None
        for sample in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                     denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs,
                                                     device=device, progress=progress):
            final = sample
        with torch.cuda.device(final["sample"].device):
            L.lib().device_error_poll(L.cur_stream())      # out-of-range label / timestep anywhere in the loop -> MapditError
        return final["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False):
        """reference gaussian_diffusion.py:464-511."""
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else torch.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        for i in indices:
            t = torch.full((shape[0],), i, device=device, dtype=torch.int64)
            with torch.no_grad():
                out = self.p_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                                    model_kwargs=model_kwargs)
                yield out
                img = out["sample"]
