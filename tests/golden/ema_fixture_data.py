"""Synthetic EMA snapshot set shared by make_golden.py (reference side) and tests/test_ema.py (build side)."""
import torch


def ema_snapshot_set():
    """Deterministic synthetic EMA snapshot directory contents: [(std, t, {key: fp16 tensor})] - shared by this
    generator and tests/test_ema.py (which rebuilds the same tensors from the same seeds)."""
    g = torch.Generator().manual_seed(77)
    base = {"a.weight": torch.randn(6, 5, generator=g), "b.gain": torch.randn((), generator=g), "c.buf": torch.randn(1, 4, 3, generator=g)}
    snaps = []
    for t in (40, 80, 120, 160):
        for std in (0.05, 0.1):
            sd = {k: (v + 0.1 * torch.randn(v.shape, generator=g) * (t / 160)).half() for k, v in base.items()}
            snaps.append((std, t, sd))
    return snaps
