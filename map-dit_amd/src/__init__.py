"""Mirror of the reference's ``src`` package for the hot path: ``src.models.DIT_MODELS`` and ``src.dit.DiT``."""
