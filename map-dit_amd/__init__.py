"""MI355X-native MaP-DiT hot path behind the reference's Python API.

    from mapdit_amd.src.models import DIT_MODELS        # reference: src/models.py:50-56
    from mapdit_amd.diffusion import create_diffusion   # reference: diffusion/__init__.py:10-46

All compute runs in hand-written HIP kernels for gfx950 loaded from ``libmapdit_hip.so``
(C ABI: ``include/mapdit.h``); PyTorch supplies device memory, streams and torch.distributed.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
__version__ = "0.1.0"
