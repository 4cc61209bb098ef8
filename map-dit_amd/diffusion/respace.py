"""Timestep respacing — reference diffusion/respace.py:12-129."""
import numpy as np
import torch

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """Pick the retained timesteps (reference respace.py:12-62): "ddimN" = fixed integer stride; otherwise a list (or
    comma-separated string) of per-section counts, each section strided fractionally and rounded."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == desired:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    start, kept = 0, []
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0                      # accumulated, as the reference does (j*stride can round differently)
        for _ in range(count):
            kept.append(start + round(cur))
            cur += stride
        start += size
    return set(kept)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process over a subset of the base process's steps (reference respace.py:65-113)."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = []
        self.original_num_steps = len(kwargs["betas"])
        base = GaussianDiffusion(**kwargs)
        last, new_betas = 1.0, []
        for i, acp in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - acp / last)
                last = acp
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)
        self._map_cache = {}

    def _mapped_t(self, ts):
        """respace.py:124-129: ts -> timestep_map[ts], with the map resident on the device (uploaded once)."""
        key = (ts.device, ts.dtype)
        m = self._map_cache.get(key)
        if m is None:
            m = torch.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
            self._map_cache[key] = m
        return m[ts]

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self)

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def training_losses(self, model, *args, **kwargs):
        return super().training_losses(self._wrap_model(model), *args, **kwargs)

    def _scale_timesteps(self, t):
        return t


class _WrappedModel:
    """reference respace.py:117-129."""

    def __init__(self, model, diffusion):
        self.model = model
        self.diffusion = diffusion
        self.timestep_map = diffusion.timestep_map
        self.original_num_steps = diffusion.original_num_steps

    def __call__(self, x, ts, **kwargs):
        return self.model(x, self.diffusion._mapped_t(ts), **kwargs)
