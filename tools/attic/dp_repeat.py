#!/usr/bin/env python3
"""Repeat the two-rank rehearsal (tests/dp_worker.py over gloo, both ranks on this GPU) and compare runs bit for bit; names the
parameters / rows in which two runs differ.   python tools/dp_repeat.py [repeats]"""
import os
import sys
import tempfile
import pathlib

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from test_train_gpu import _run_dp  # noqa: E402
from mapdit_amd.src.models import DIT_MODELS  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = DIT_MODELS["DiT-XS/2"](in_channels=4, input_size=32, num_classes=7)
base = m._pflat.data_ptr()
layout = sorted(((p.data_ptr() - base) // 4, k, tuple(p.shape)) for k, p in m.named_parameters())


def where(idx):
    out = {}
    for i in idx.tolist():
        for off, k, sh in layout:
            n = 1
            for s in sh:
                n *= s
            if off <= i < off + n:
                c = sh[-1] if sh else 1
                out.setdefault((k, (i - off) // c), 0)
                out[(k, (i - off) // c)] += 1
                break
    return ", ".join(f"{k} row {r} ({n} elements)" for (k, r), n in list(out.items())[:6])


tmp = pathlib.Path(tempfile.mkdtemp())
runs = {}
for mode in ("allreduce", "zero1"):
    runs[mode] = [_run_dp(tmp, f"{mode}{i}", 2, mode, "bf16") for i in range(R)]
ref = runs["allreduce"][0][0]
bad = 0
for mode, rs in runs.items():
    for i, r in enumerate(rs):
        for k in ("g", "p"):
            if not torch.equal(r[0][k], r[1][k]):
                print(f"{mode} run {i}: ranks differ in {k}")
            if not torch.equal(r[0][k], ref[k]):
                bad += 1
                idx = torch.nonzero(r[0][k] != ref[k]).flatten()
                print(f"{mode} run {i}: {k} differs from the first run in {idx.numel()} elements: {where(idx)}")
print(f"{2 * R} two-rank runs, {bad} buffers differing from the first run")
