#!/usr/bin/env python3
"""Reference point only (not used by the product): what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the
block's GEMM shapes, bf16 in / bf16 or fp32 out, no fused epilogue."""
import torch

D, M = 768, 65536
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
x, h, dy, dh = rnd(M, D), rnd(M, 4 * D), rnd(M, D), rnd(M, 4 * D)
w3, w4, w4t = rnd(3 * D, D), rnd(4 * D, D), rnd(D, 4 * D)


def bench(fn, flops, name):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = sorted(ts)[2]
    print(f"{name:28s} {ms:8.3f} ms {flops / ms / 1e9:8.1f} TFLOP/s")


bench(lambda: x @ w3.t(), 2.0 * M * 3 * D * D, "NT qkv fwd  [M,3D]=x W^T")
bench(lambda: x @ w4.t(), 2.0 * M * 4 * D * D, "NT fc1 fwd  [M,4D]=x W^T")
bench(lambda: h @ w4t.t(), 2.0 * M * D * 4 * D, "NT fc2 fwd  [M,D]=h W^T")
bench(lambda: dh @ w4, 2.0 * M * D * 4 * D, "NN fc1 dX   [M,D]=dh W")
bench(lambda: dy @ w4t, 2.0 * M * 4 * D * D, "NN fc2 dX   [M,4D]=dy W")
bench(lambda: dh.t() @ x, 2.0 * M * 4 * D * D, "TN fc1 dW   [4D,D]=dh^T x")
bench(lambda: dy.t() @ x, 2.0 * M * D * D, "TN proj dW  [D,D]=dy^T x")
