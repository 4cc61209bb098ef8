"""Power-function EMA profiles and post-hoc EMA reconstruction (Karras et al., arXiv 2312.02696, Sec. 3 / Alg. 2-3),
host-side float64 like the reference (src/ema.py:10-114).

During training the fused optimiser kernel keeps two EMA copies (relative widths 0.05 and 0.1; optim.FusedAdamEMA) and
the harness snapshots them as fp16 ``ema/{std:.3f}_{t:07d}.pt`` files ({"std", "t", "state_dict"}, src/ema.py:143-155).
After training, the average for ANY width is the least-squares combination of the stored snapshots; the samplers call
``calculate_posthoc_ema`` for that (sample.py:34, sample_fid.py:34, sample_ema.py:34).
"""
from __future__ import annotations

import os
import re
from typing import Dict, List, Tuple

import numpy as np
import torch

_SNAPSHOT_RE = re.compile(r"^(?P<std>[0-9]*\.[0-9]+)_(?P<t>\d+)\.pt$")


def std_to_gamma(std) -> np.ndarray:
    """Relative std sigma_rel -> exponent gamma of the profile t^gamma: the largest real root of
    g^3 + 7 g^2 + (16 - s^-2) g + (12 - s^-2) = 0   (reference src/ema.py:10-20)."""
    std = np.asarray(std)
    inv_var = std.astype(np.float64).reshape(-1) ** -2
    roots = [np.roots([1.0, 7.0, 16.0 - v, 12.0 - v]).real.max() for v in inv_var]
    return np.asarray(roots, dtype=np.float64).reshape(std.shape)


def gamma_to_std(gamma) -> np.ndarray:
    """Inverse of std_to_gamma (reference src/ema.py:23-30)."""
    g = np.asarray(gamma).astype(np.float64)
    return np.sqrt((g + 1) / ((g + 2) ** 2 * (g + 3)))


def calc_beta(std, t):
    """Per-step lerp weight of the power-function EMA, (1 - 1/t)^(gamma+1) (reference src/ema.py:33-40)."""
    return (1 - 1 / t) ** (std_to_gamma(np.asarray(std)) + 1)


def p_dot_p(t_a, gamma_a, t_b, gamma_b):
    """Inner product of two power-function profiles (reference src/ema.py:43-53)."""
    later = np.maximum(t_a, t_b)
    expo = np.where(t_a < t_b, gamma_b, -gamma_a)
    return (gamma_a + 1) * (gamma_b + 1) * (t_a / t_b) ** expo / ((gamma_a + gamma_b + 1) * later)


def solve_weights(t_i, gamma_i, t_r, gamma_r) -> np.ndarray:
    """Least-squares weights X [n_snapshots, n_targets] such that sum_i X[i, r] * profile_i ~= profile_r
    (reference src/ema.py:56-66)."""
    col = lambda v: np.float64(v).reshape(-1, 1)
    row = lambda v: np.float64(v).reshape(1, -1)
    gram = p_dot_p(col(t_i), col(gamma_i), row(t_i), row(gamma_i))
    rhs = p_dot_p(col(t_i), col(gamma_i), row(t_r), row(gamma_r))
    return np.linalg.solve(gram, rhs)


def list_snapshots(ema_dir: str) -> List[Tuple[float, int, str]]:
    """(std, t, filename) of every ``{std:.3f}_{t:07d}.pt`` in directory order (the order the weights are applied in)."""
    found = []
    for name in os.listdir(ema_dir):
        m = _SNAPSHOT_RE.match(name)
        if m:
            found.append((float(m.group("std")), int(m.group("t")), name))
    return found


def calculate_posthoc_ema(out_std: float, ema_dir: str, verbose: bool = True) -> Dict[str, torch.Tensor]:
    """state_dict of the EMA of relative width ``out_std`` at the last snapshot time (reference src/ema.py:69-114).
    A width that was tracked during training is returned as stored (fp16); anything else is the fp32 weighted sum of all
    snapshots, accumulated in directory order like the reference."""
    snaps = list_snapshots(ema_dir)
    assert snaps, "No EMA snapshots found in the results directory"
    stds = np.array([s for s, _, _ in snaps])
    ts = np.array([t for _, t, _ in snaps])
    t_out = ts.max()
    load = lambda name: torch.load(os.path.join(ema_dir, name), weights_only=True)["state_dict"]

    if out_std in stds:
        hit = int(np.argmax((stds == out_std) & (ts == t_out)))
        return load(snaps[hit][2])

    w = solve_weights(ts, std_to_gamma(stds), t_out, std_to_gamma(out_std)).reshape(-1)
    acc = None
    for k, (_, _, name) in enumerate(snaps):
        if verbose:
            print(f"computing ema state_dict (std={out_std}): {k + 1}/{len(snaps)}", end="\r", flush=True)
        sd = load(name)
        if acc is None:
            acc = {key: torch.zeros_like(v, dtype=torch.float32) for key, v in sd.items()}
        for key in acc:
            acc[key] += sd[key].float() * w[k]
    if verbose:
        print()
    return acc


class EMA:
    """Counterpart of the reference's ``EMA`` helper (src/ema.py:117-155) for training loops that keep ``torch.optim.Adam``:
    one power-function EMA copy of the model per relative width, updated after every optimiser step and snapshotted as
    fp16 state dicts.  (``mapdit_amd.optim.FusedAdamEMA`` folds the same update into the optimiser kernel.)

    The copies are flat clones of the model's parameter buffer, so an update is ONE ``lerp_`` per width instead of one per
    parameter tensor."""

    @torch.no_grad()
    def __init__(self, net, results_dir, stds=(0.05, 0.1)):
        self.net = net
        self.stds = tuple(stds)
        self.flat = {s: net._pflat.detach().clone() for s in self.stds}
        self.ema_dir = os.path.join(results_dir, "ema")
        os.makedirs(self.ema_dir, exist_ok=True)

    @torch.no_grad()
    def update(self, t, model=None):
        """ema <- lerp(ema, model, (1 - 1/t)^(gamma+1))   (reference src/ema.py:126-140)."""
        model = model or self.net
        for s, flat in self.flat.items():
            flat.lerp_(model._pflat, float(calc_beta(s, t)))

    def state_dict(self, std):
        """The EMA weights of width ``std`` under the model's state-dict keys (buffers are the model's own)."""
        flat = self.flat[std]
        sd = {}
        params = dict(self.net.named_parameters())
        offs = {id(p): o for p, o in zip(self.net.parameters(), self.net._poffs)}
        for k, v in self.net.state_dict().items():
            if k in params:
                p = params[k]
                o = offs[id(p)]
                sd[k] = flat[o:o + p.numel()].view(p.shape).clone()
            else:
                sd[k] = v.clone()
        return sd

    @torch.no_grad()
    def save_snapshot(self, t):
        """``ema/{std:.3f}_{t:07d}.pt`` = {"std", "t", "state_dict" (fp16, CPU)}   (reference src/ema.py:143-155)."""
        for s in self.stds:
            sd = {k: v.cpu().half() for k, v in self.state_dict(s).items()}
            torch.save({"std": s, "t": t, "state_dict": sd}, os.path.join(self.ema_dir, f"{s:.3f}_{t:07d}.pt"))
