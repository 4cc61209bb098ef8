"""Post-hoc EMA (reference src/ema.py:10-114): the product module map-dit_amd/src/ema.py and the
loop-form oracle against values computed by the reference's own functions (tests/golden/ema.npz)."""
import os
import sys

import numpy as np
import torch

from conftest import load_golden

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from ema_fixture_data import ema_snapshot_set  # noqa: E402

from mapdit_amd.src import ema as E  # noqa: E402
from oracle import ema_oracle as EO  # noqa: E402


def test_gamma_beta_and_weights_match_reference():
    g = load_golden("ema")
    assert np.array_equal(E.std_to_gamma(g["stds"]), g["gammas"])
    assert np.array_equal(E.gamma_to_std(g["gammas"]), g["stds_back"])
    np.testing.assert_allclose(g["stds_back"], g["stds"], rtol=1e-12)
    for std in (0.05, 0.1):
        got = np.array([E.calc_beta(std, t) for t in g["beta_t"]], dtype=np.float64)
        assert np.array_equal(got, g[f"betas_{std}"])
    ts, st, targets = g["sw_ts"], g["sw_stds"], g["sw_targets"]
    X = E.solve_weights(ts, E.std_to_gamma(st), np.full(len(targets), ts.max()), E.std_to_gamma(targets))
    assert np.array_equal(X, g["sw_weights"])
    # oracle (explicit loops): same numbers up to the conditioning of the 8x8 Gram system
    for r, target in enumerate(targets):
        w = EO.solve_weights(ts, [EO.std_to_gamma(s) for s in st], ts.max(), EO.std_to_gamma(target))
        np.testing.assert_allclose(w, g["sw_weights"][:, r], rtol=1e-7, atol=1e-10)
    # the optimiser's scalar helpers agree with the module (one definition of the schedule)
    from mapdit_amd import optim
    assert optim.std_to_gamma(0.05) == float(E.std_to_gamma(0.05)) and optim.calc_beta(0.1, 10) == float(E.calc_beta(0.1, 10))


def test_posthoc_reconstruction_matches_reference(tmp_path):
    g = load_golden("ema")
    snaps = ema_snapshot_set()
    for std, t, sd in snaps:
        torch.save({"std": std, "t": t, "state_dict": sd}, tmp_path / f"{std:.3f}_{t:07d}.pt")
    (tmp_path / "notes.txt").write_text("not a snapshot")                   # ignored, like any non-matching name
    found = E.list_snapshots(str(tmp_path))
    assert sorted((s, t) for s, t, _ in found) == sorted((s, t) for s, t, _ in snaps)
    by_name = {f"{std:.3f}_{t:07d}.pt": (std, t, {k: v.numpy() for k, v in sd.items()}) for std, t, sd in snaps}
    for target in (0.075, 0.02):
        res = E.calculate_posthoc_ema(target, str(tmp_path), verbose=False)
        ordered = [by_name[n] for n in g["order"]]                             # the order the reference accumulated in
        ora = EO.posthoc(ordered, target)
        for k, v in res.items():
            want = g[f"posthoc_{target}/{k}"]
            assert v.dtype == torch.float32 and str(g[f"posthoc_{target}_dtype/{k}"]) == "torch.float32"
            # fp32 accumulation order follows os.listdir (as in the reference): allow reordering noise only
            np.testing.assert_allclose(v.numpy(), want, rtol=2e-6, atol=2e-6)
            np.testing.assert_allclose(ora[k], want, rtol=2e-6, atol=2e-6)
    # a width tracked during training comes back as stored (fp16, last snapshot time)
    res = E.calculate_posthoc_ema(0.1, str(tmp_path), verbose=False)
    last = [sd for std, t, sd in snaps if std == 0.1 and t == 160][0]
    for k, v in res.items():
        assert v.dtype == torch.float16 and torch.equal(v, last[k])
        assert np.array_equal(v.float().numpy(), g[f"posthoc_0.1/{k}"])
