"""CPU oracle for the post-hoc EMA reconstruction (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates reference src/ema.py:43-114 with explicit loops in float64 (no broadcasting tricks), so that it shares no
code path with the product module map-dit_amd/src/ema.py.  Pinned by tests/golden/ema.npz (values produced by the
reference's own functions, tests/golden/make_golden.py ema).
"""
from __future__ import annotations

import numpy as np


def std_to_gamma(std: float) -> float:
    """src/ema.py:10-20: largest real root of g^3 + 7g^2 + (16 - 1/s^2) g + (12 - 1/s^2)."""
    v = float(std) ** -2
    return float(np.roots([1.0, 7.0, 16.0 - v, 12.0 - v]).real.max())


def profile_dot(t_a: float, g_a: float, t_b: float, g_b: float) -> float:
    """src/ema.py:43-53."""
    expo = g_b if t_a < t_b else -g_a
    return (g_a + 1) * (g_b + 1) * (t_a / t_b) ** expo / ((g_a + g_b + 1) * max(t_a, t_b))


def solve_weights(ts, gammas, t_out: float, gamma_out: float) -> np.ndarray:
    """src/ema.py:56-66 for a single target profile: w with sum_i w_i p_i ~= p_out."""
    n = len(ts)
    A = np.zeros((n, n), dtype=np.float64)
    b = np.zeros(n, dtype=np.float64)
    for i in range(n):
        for j in range(n):
            A[i, j] = profile_dot(float(ts[i]), float(gammas[i]), float(ts[j]), float(gammas[j]))
        b[i] = profile_dot(float(ts[i]), float(gammas[i]), float(t_out), float(gamma_out))
    return np.linalg.solve(A, b)


def posthoc(snapshots, out_std: float) -> dict:
    """snapshots: list of (std, t, {key: float16 ndarray}) in application order -> {key: float32 ndarray}
    (src/ema.py:69-114, the reconstruction branch; accumulation in float32 like the reference)."""
    ts = [t for _, t, _ in snapshots]
    gs = [std_to_gamma(s) for s, _, _ in snapshots]
    w = solve_weights(ts, gs, max(ts), std_to_gamma(out_std))
    acc = {k: np.zeros(v.shape, dtype=np.float32) for k, v in snapshots[0][2].items()}
    for wk, (_, _, sd) in zip(w, snapshots):
        for k in acc:
            acc[k] = acc[k] + (sd[k].astype(np.float32) * np.float32(wk)).astype(np.float32)
    return acc
