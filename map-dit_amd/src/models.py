"""DiT-{XS,S,B,L,XL}/{2,4,8} constructors — same names and kwargs as reference src/models.py:4-56."""
from .dit import DiT

# family -> (depth, hidden_size, num_heads)                       reference src/models.py:4-47
_FAMILIES = {"XL": (28, 1152, 16), "L": (24, 1024, 16), "B": (12, 768, 12), "S": (12, 384, 6), "XS": (6, 256, 4)}


def _factory(fam: str, patch: int):
    depth, hidden, heads = _FAMILIES[fam]

    def make(**kwargs):
        return DiT(depth=depth, hidden_size=hidden, patch_size=patch, num_heads=heads, **kwargs)

    make.__name__ = f"DiT_{fam}_{patch}"
    return make


DIT_MODELS = {f"DiT-{fam}/{p}": _factory(fam, p) for fam in _FAMILIES for p in (2, 4, 8)}
globals().update({f.__name__: f for f in DIT_MODELS.values()})
