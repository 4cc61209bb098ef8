#!/usr/bin/env python3
"""Would a 16-bit residual stream stay inside north_star's 1e-3?  (CPU; the oracle is the arithmetic)

The fp16 engine keeps the residual-stream checkpoints X[0..2L] and the gradient that flows back through them in fp32: 10 of the
RESID epilogue's bytes per element and 18 of resid_mod_bwd's are that stream.  This tool rounds them to fp16 in the oracle -
forward value at every checkpoint, backward gradient under the engine's loss scale (oracle.dit_oracle.EnginePlan(residual16=True);
the modulate fused into the producing epilogue still sees the unrounded value) - and reports, against the reference's own fixture:
eval logits, training losses and pooled / worst-tensor gradient error, next to the shipped fp16 plan (fp32 residual stream).

    python tools/precision_residual16.py [fixtures = b2_n2 s2_n4 xl2_n2]
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dit_oracle as O  # noqa: E402
from oracle.diffusion_oracle import DiffusionOracle  # noqa: E402

torch.set_num_threads(8)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def sub(a, stride, limit=20000):
    f = a.detach().reshape(-1)
    return (f if f.numel() <= limit else f[::stride]).numpy()


def measure(name):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False))
    cfg = O.DiTConfig(**{k[4:]: g[k].item() for k in g if k.startswith("cfg_")})
    gains = g["gains"].item()
    sd0 = O.init_state_dict(cfg, seed=int(g["wseed"]), gains=None if gains < 0 else gains, perturb_reference=float(g["perturb"]))
    x, t, y, y_eff, noise = (torch.from_numpy(g[k]) for k in ("x", "t", "y", "y_eff", "noise"))
    n = x.shape[0]
    scale = 2.0 ** (math.floor(math.log2(n * cfg.in_channels * cfg.input_size ** 2)) - 5)      # the engine's automatic loss scale
    plans = [("fp16 operands, fp32 residual stream (shipped)", O.EnginePlan(O.f16_round)),
             ("fp16 operands, fp16 residual stream + gradient", O.EnginePlan(O.f16_round, residual16=True, grad_scale=scale)),
             ("bf16 operands, fp32 residual stream", O.EnginePlan(O.bf16_round)),
             ("bf16 operands, bf16 residual stream + gradient", O.EnginePlan(O.bf16_round, residual16=True, grad_scale=1.0))]
    print(f"== {name}: depth {cfg.depth}, hidden {cfg.hidden_size}, {n} samples; loss scale 2^{int(math.log2(scale))}")
    stride = 7 if "postw/x_embedder.weight" in g else 4099
    for label, plan in plans:
        with torch.no_grad():
            out = O.dit_forward({k: v.clone() for k, v in sd0.items()}, cfg, x, t, y, train=False, rnd=plan)
        ref = g["eval_out"]
        e_log = rel(sub(out, 7) if ref.shape != tuple(out.shape) else out.numpy(), ref)
        sd = {k: v.clone().requires_grad_(k not in O.BUFFER_KEYS) for k, v in sd0.items()}
        drop = torch.zeros(n, dtype=torch.bool)
        losses = DiffusionOracle("").training_losses(
            lambda xx, tt, **kw: O.dit_forward(sd, cfg, xx, tt, kw["y"], train=True, drop=drop, rnd=plan), x, t, dict(y=y_eff), noise=noise)
        losses["loss"].mean().backward()
        e_loss = rel(losses["loss"].detach().numpy(), g["train_loss"])
        num = den = 0.0
        worst, worst_k = 0.0, ""
        for k, v in sd.items():
            if v.grad is None or v.dim() == 0 or "grad/" + k not in g:
                continue
            gref = g["grad/" + k]
            mine = sub(v.grad, stride)
            if mine.shape != gref.reshape(-1).shape:
                continue
            d2, r2 = float(((mine.astype(np.float64) - gref.reshape(-1)) ** 2).sum()), float((gref.astype(np.float64) ** 2).sum())
            num, den = num + d2, den + r2
            e = rel(mine, gref.reshape(-1))
            if gref.size >= 64 and e > worst and np.linalg.norm(gref) >= 1e-7:
                worst, worst_k = e, k
        print(f"   {label:52s} logits {e_log:.3e}   loss {e_loss:.3e}   gradients pooled {(num / (den + 1e-60)) ** 0.5:.3e}  worst {worst:.3e} ({worst_k})")


for nm in (sys.argv[1:] or ["b2_n2", "s2_n4", "xl2_n2"]):
    measure(nm)
