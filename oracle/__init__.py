"""TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the MaP-DiT hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline.  The product path
(``map-dit_amd``) never routes through this package and has no CPU fallback.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md F2), so the oracle is pinned by fixtures generated in the build
container by importing the reference itself (``tests/golden/make_golden.py``),
committed under ``tests/golden/*.npz`` and checked by ``tests/test_oracle_golden.py``.
"""
