#!/usr/bin/env python3
"""Repeat the two-rank rehearsal (tests/dp_worker.py over gloo, both ranks on this GPU) and compare runs bit for bit."""
import os
import sys
import tempfile
import pathlib

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from test_train_gpu import _run_dp  # noqa: E402

tmp = pathlib.Path(tempfile.mkdtemp())
runs = {}
for mode in ("allreduce", "zero1"):
    runs[mode] = [_run_dp(tmp, f"{mode}{i}", 2, mode, "bf16") for i in range(4)]
for mode, rs in runs.items():
    for k in ("p", "g", "m"):
        print(mode, k, "run-to-run identical:", [torch.equal(rs[0][0][k], r[0][k]) for r in rs[1:]],
              "ranks identical:", [torch.equal(r[0][k], r[1][k]) for r in rs])
print("zero1 == allreduce (p):", [torch.equal(a[0]["p"], z[0]["p"]) for a, z in zip(runs["allreduce"], runs["zero1"])])
