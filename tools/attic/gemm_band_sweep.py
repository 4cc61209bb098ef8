#!/usr/bin/env python3
"""Sweep MAPDIT_GEMM_BAND (column tiles per band of the 256^2 kernel's tile order) on the block's GEMM shapes."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mapdit_amd  # noqa: E402
from tools.gemm_bench import run  # noqa: E402

L = mapdit_amd._lib
D, M = 768, 65536
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
x, h, dy, dh = rnd(M, D), rnd(M, 4 * D), rnd(M, D), rnd(M, 4 * D)
w3, w4, w4t = rnd(3 * D, D), rnd(4 * D, D), rnd(D, 4 * D)
out = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)


def ep(ldo):
    e = L.Epilogue()
    e.kind = L.EPI_STORE_BF16
    e.out = out.data_ptr()
    e.ldo = ldo
    e.alpha = 1.0
    return e


cases = [("NT M x3D x D ", 0, M, 3 * D, D, x, D, w3, D, 3 * D),
         ("NT M x4D x D ", 0, M, 4 * D, D, x, D, w4, D, 4 * D),
         ("NT M x D x4D ", 0, M, D, 4 * D, h, 4 * D, w4t, 4 * D, D),
         ("NN M x D x4D (fc1 dX)", 1, M, D, 4 * D, dh, 4 * D, w4, D, D),
         ("NN M x4D x D (fc2 dX)", 1, M, 4 * D, D, dy, D, w4t, 4 * D, 4 * D),
         ("NN M x D x3D (qkv dX)", 1, M, D, 3 * D, dh, 4 * D, w3, D, D)]
for name, layout, m, n, k, a, lda, b, ldb, ldo in cases:
    row = []
    for band in ("auto", 1, 2, 3, 4, 6, 12):
        L.lib().gemm_tuning(256, 2, 0 if band == "auto" else band)
        ms = sorted(run(layout, m, n, k, a, lda, b, ldb, ep(ldo), 10) for _ in range(3))[1]
        row.append(f"{band}:{2.0 * m * n * k / ms / 1e9:6.0f}")
    print(f"{name:24s} " + "  ".join(row))
