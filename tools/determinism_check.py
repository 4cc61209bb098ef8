#!/usr/bin/env python3
"""Run tests/dp_worker.py (single process, 3 steps) in several fresh processes, each with the workspace memory pre-filled with a
different byte pattern, and compare the resulting parameters bit for bit: catches reads of uninitialised workspace memory, which an
in-process reproducibility test cannot see (the caching allocator hands the same stale block back)."""
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
outs = []
for i, prec in [(0, "bf16"), (1, "bf16"), (2, "bf16")]:
    d = tempfile.mkdtemp()
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MAPDIT_TEST_POISON=str(0x11 * (i + 1)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), d, "allreduce", prec, "3"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    outs.append(torch.load(os.path.join(d, "rank0.pt"), weights_only=False))
for k in ("p", "g", "m", "e0"):
    same = [torch.equal(outs[0][k], o[k]) for o in outs[1:]]
    diff = [float((outs[0][k] - o[k]).abs().max()) for o in outs[1:]]
    print(k, "identical across processes:", same, "max abs diff", diff)
