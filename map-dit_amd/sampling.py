"""hipGraph-captured ancestral sampling (BASELINE config 5; reference sample_fid.py:48-76 + gaussian_diffusion.py:464-511).

One denoise step of ``p_sample_loop`` — timestep map lookup, the classifier-free-guidance DiT forward, the fused
p_mean_variance / p_sample kernel, the N(0,1) draw and the on-device ``t -= 1`` — is captured once into a hipGraph
(torch.cuda.CUDAGraph is hipGraph on ROCm) and replayed ``num_timesteps`` times.  The reference's per-step host work
(``th.tensor([i]*B)``, ~8 numpy table uploads, the timestep_map upload: SURVEY §3.2) disappears: schedule tables,
the map and the step counter live on the device, and a replay is a single graph launch.
"""
from __future__ import annotations

import torch

from . import _lib as L


class GraphedSampler:
    """Captures ``x_{t-1} = p_sample(model_fn(x_t, map[t], **kw), x_t, t)`` for a fixed batch shape.

    model: the DiT module (eval mode); diffusion: a SpacedDiffusion; cfg_scale: None -> model.forward, else
    model.forward_with_cfg (y must then hold [labels, null labels], as sample_fid.py:56-66 builds it)."""

    def __init__(self, model, diffusion, shape, y, cfg_scale=None, clip_denoised=False):
        assert not model.training, "sampling runs the model in eval mode"
        diffusion._supported()
        self.model, self.diffusion = model, diffusion
        dev = next(model.parameters()).device
        self.dev = dev
        self.img = torch.zeros(*shape, device=dev)
        # the warm-up and the capture below each run one real step (t -> t - 1): start high enough to stay inside the schedule
        self.t = torch.full((shape[0],), max(diffusion.num_timesteps - 1, 0), device=dev, dtype=torch.int64)
        self.y = y.to(dev).clone()
        self.cfg_scale, self.clip = cfg_scale, bool(clip_denoised)
        self.tab = diffusion._tables(dev)
        self.tmap = torch.tensor(diffusion.timestep_map, device=dev, dtype=torch.int64)
        self.graph = None
        with torch.no_grad():
            # eager warm-up on a side stream (allocates the engine workspace, builds the cached bf16 weight images)
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(2):
                    self.t.fill_(max(diffusion.num_timesteps - 1, 0))
                    self._step()
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            self.t.fill_(max(diffusion.num_timesteps - 1, 0))
            with torch.cuda.graph(self.graph):
                self._step()

    def _step(self):
        d = self.diffusion
        mapped = self.tmap[self.t.clamp_min(0)]
        if self.cfg_scale is None:
            out = self.model.forward(self.img, mapped, self.y)
        else:
            out = self.model.forward_with_cfg(self.img, mapped, self.y, self.cfg_scale)
        noise = torch.randn_like(self.img)
        nxt = torch.empty_like(self.img)
        L.lib().psample_step(out.data_ptr(), self.img.data_ptr(), noise.data_ptr(), self.t.data_ptr(), self.tab.data_ptr(),
                             d.num_timesteps, int(self.clip), nxt.data_ptr(), None, self.img.shape[0], self.img[0].numel(),
                             L.cur_stream())
        self.img.copy_(nxt)
        self.t.sub_(1)

    def refresh_weights(self):
        """After the model's parameters were replaced in place (``load_state_dict``): rebuild the cached weight images
        the captured graph reads (the eager forward does this lazily; a graph replay runs no Python)."""
        rt = self.model._runtime(self.img.shape[0], train=False)
        with torch.cuda.device(self.dev):
            rt.lib.engine_prepare_weights(rt.handle, 0, L.cur_stream())
        rt.weights_key = self.model._weights_key()

    @torch.no_grad()
    def sample(self, noise=None, steps=None):
        """Run the reverse chain from ``noise`` (or fresh N(0,1)); ``steps`` bounds the prefix (default: all)."""
        n = self.diffusion.num_timesteps
        steps = n if steps is None else steps
        if noise is None:
            self.img.normal_()
        else:
            self.img.copy_(noise)
        self.t.fill_(n - 1)
        for _ in range(steps):
            self.graph.replay()
        out = self.img.clone()
        self.model.check_device_errors()      # a label outside the embedding table (e.g. a null label the model has no row for)
        return out


def p_sample_loop_graphed(diffusion, model, shape, noise=None, clip_denoised=False, model_kwargs=None, device=None):
    """Drop-in for ``diffusion.p_sample_loop(model.forward[_with_cfg], shape, noise, clip_denoised, model_kwargs=...)``
    with the loop body replayed from a hipGraph.  ``model`` is the DiT module itself."""
    kw = dict(model_kwargs or {})
    y = kw.pop("y")
    cfg = kw.pop("cfg_scale", None)
    assert not kw, f"unsupported model_kwargs: {sorted(kw)}"
    return GraphedSampler(model, diffusion, shape, y, cfg, clip_denoised).sample(noise)


# ---- shared pieces of the sampler CLIs (reference sample.py / sample_fid.py / sample_ema.py) ---------------------------

def str2bool(v) -> bool:
    """argparse type for the reference's ``--use-vae`` style switches (the reference uses ``type=bool``, under which any
    non-empty string - including "False" - is true; here "false"/"0"/"no" mean false)."""
    if isinstance(v, bool):
        return v
    return str(v).strip().lower() not in ("false", "0", "no", "off", "")


def load_train_args(result_dir: str) -> dict:
    """config.yaml written by the training harness (reference train.py:35-40; sample.py:20-24)."""
    import os

    import yaml
    with open(os.path.join(result_dir, "config.yaml"), "r") as f:
        return yaml.safe_load(f)


def load_weights(model, result_dir: str, ema_std: float = 0.05, ckpt: str | None = None, verbose: bool = True):
    """EMA weights reconstructed post hoc from ``<result_dir>/ema`` (default), or a raw checkpoint's ``model`` entry
    when ``ckpt`` names one (without .pt) - reference sample.py:29-37."""
    import os

    from .src.ema import calculate_posthoc_ema
    if ckpt is not None:
        sd = torch.load(os.path.join(result_dir, "checkpoints", f"{ckpt}.pt"), map_location="cpu", weights_only=True)["model"]
    else:
        sd = calculate_posthoc_ema(ema_std, os.path.join(result_dir, "ema"), verbose=verbose)
    model.load_state_dict({k: v.float() for k, v in sd.items()})
    model.eval()
    return model


def denormalize(samples: torch.Tensor, train_args: dict) -> torch.Tensor:
    """Undo the per-channel standardisation of the training latents (reference sample.py:66-69)."""
    mean = torch.tensor(train_args["stats_mean"], device=samples.device).reshape(1, -1, 1, 1)
    std = torch.tensor(train_args["stats_std"], device=samples.device).reshape(1, -1, 1, 1)
    return samples * std + mean


def load_vae(vae_path: str | None, device):
    """The reference decodes latents with diffusers' AutoencoderKL("stabilityai/sd-vae-ft-mse") fetched from the hub
    (sample.py:72-74).  No network here: the decoder is out of scope - a local copy can be passed."""
    try:
        from diffusers import AutoencoderKL
    except ImportError as e:
        raise RuntimeError("--use-vae needs the `diffusers` package and a local copy of stabilityai/sd-vae-ft-mse "
                           "(--vae-path); pass `--use-vae false` to write the latents instead") from e
    return AutoencoderKL.from_pretrained(vae_path or "stabilityai/sd-vae-ft-mse").to(device)


def run_sampler(model, diffusion, z, y, cfg_scale, use_graph: bool = True, progress: bool = False):
    """``diffusion.p_sample_loop(model.forward[_with_cfg], z.shape, z, clip_denoised=False, ...)`` - through the captured
    hipGraph (default) or the eager loop."""
    if use_graph:
        return GraphedSampler(model, diffusion, z.shape, y, cfg_scale, clip_denoised=False).sample(z)
    kw = dict(y=y) if cfg_scale is None else dict(y=y, cfg_scale=cfg_scale)
    fn = model.forward if cfg_scale is None else model.forward_with_cfg
    return diffusion.p_sample_loop(fn, z.shape, z, clip_denoised=False, model_kwargs=kw, progress=progress, device=z.device)


def save_image_grid(samples: torch.Tensor, path: str, nrow: int, value_range=(-1.0, 1.0), padding: int = 2):
    """What the reference gets from torchvision.utils.save_image(samples, path, nrow=, normalize=True, value_range=)
    (sample.py:79): min-max scale to [0, 1] over ``value_range``, tile ``nrow`` images per row with a 2-pixel black
    border, write 8-bit.  1/3/4-channel inputs become L / RGB / RGBA."""
    import numpy as np
    from PIL import Image
    x = samples.detach().float().cpu()
    lo, hi = value_range
    x = ((x.clamp(lo, hi) - lo) / max(hi - lo, 1e-5))
    n, c, h, w = x.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    grid = torch.zeros(c, rows * (h + padding) + padding, cols * (w + padding) + padding)
    for i in range(n):
        r, q = divmod(i, cols)
        top, left = padding + r * (h + padding), padding + q * (w + padding)
        grid[:, top:top + h, left:left + w] = x[i]
    arr = (grid * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    mode = {1: "L", 3: "RGB", 4: "RGBA"}.get(c)
    assert mode, f"cannot write a {c}-channel image"
    Image.fromarray(arr[:, :, 0] if c == 1 else np.ascontiguousarray(arr), mode).save(path)


def to_uint8_nhwc(samples: torch.Tensor):
    """[-1,1] float NCHW -> uint8 NHWC as the FID npz stores it (reference sample_fid.py:87-90; .byte() truncates)."""
    s = samples.clamp(-1, 1)
    return (255 * (s + 1) / 2).byte().permute(0, 2, 3, 1).cpu().numpy()
