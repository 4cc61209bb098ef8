import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def golden_cfg(g):
    from oracle.dit_oracle import DiTConfig
    kw = {k[4:]: g[k].item() for k in g if k.startswith("cfg_")}
    return DiTConfig(**kw)


def golden_state_dict(g, cfg):
    from oracle.dit_oracle import init_state_dict
    gains = g["gains"].item()
    return init_state_dict(cfg, seed=int(g["wseed"]), gains=None if gains < 0 else gains,
                           perturb_reference=float(g["perturb"]))


def sub(a, stride=7, limit=20000):
    f = a.detach().reshape(-1)
    return (f if f.numel() <= limit else f[::stride]).cpu().numpy()


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
