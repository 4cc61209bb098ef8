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
        self.t = torch.zeros(shape[0], device=dev, dtype=torch.int64)
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
                    self._step()
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
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
        return self.img.clone()


def p_sample_loop_graphed(diffusion, model, shape, noise=None, clip_denoised=False, model_kwargs=None, device=None):
    """Drop-in for ``diffusion.p_sample_loop(model.forward[_with_cfg], shape, noise, clip_denoised, model_kwargs=...)``
    with the loop body replayed from a hipGraph.  ``model`` is the DiT module itself."""
    kw = dict(model_kwargs or {})
    y = kw.pop("y")
    cfg = kw.pop("cfg_scale", None)
    assert not kw, f"unsupported model_kwargs: {sorted(kw)}"
    return GraphedSampler(model, diffusion, shape, y, cfg, clip_denoised).sample(noise)
