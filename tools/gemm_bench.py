#!/usr/bin/env python3
"""Micro-benchmark of mapdit_gemm_bf16 on the shapes of one DiT block (random bf16 operands, interleaved A/B of the
128^2 and 256^2 kernels in ONE process, median of rounds — guide §5.4 rules 24/25).

    python tools/gemm_bench.py [--hidden 768] [--tokens 65536]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapdit_amd  # noqa: E402

L = mapdit_amd._lib


def run(layout, M, N, K, a, lda, b, ldb, ep, iters):
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.gemm_bf16(layout, M, N, K, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(ep), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.gemm_bf16(layout, M, N, K, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(ep), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--tokens", type=int, default=65536)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cases", default="", help="comma-separated substrings: only the cases whose name contains one of them")
    ap.add_argument("--four-phase", action="store_true", help="also the four-phase K loop (one output quadrant per phase)")
    args = ap.parse_args()
    D, M = args.hidden, args.tokens
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
    x, w_qkv, w_fc1, w_fc2 = rnd(M, D), rnd(3 * D, D) * 0.03, rnd(4 * D, D) * 0.03, rnd(D, 4 * D) * 0.03
    h, dy, dh = rnd(M, 4 * D), rnd(M, D), rnd(M, 4 * D)
    dqkv = rnd(M, 3 * D)
    out_bf = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
    out_bf2 = torch.empty_like(out_bf)
    out_bf3 = torch.empty_like(out_bf)
    scales = torch.empty(2 * M * (D // 64), device=dev)
    out_f32 = torch.empty(64, 4 * D * D, device=dev)
    xres = torch.randn(M, D, device=dev)
    xout = torch.empty_like(xres)
    gate = torch.randn(M // 256, 6 * D, device=dev)

    def ep(kind, **kw):
        e = L.Epilogue()
        e.kind = kind
        for k, v in kw.items():
            setattr(e, k, v)
        return e

    cases = [
        ("qkv  fwd NT store", 0, M, 3 * D, D, x, D, w_qkv, D, ep(L.EPI_STORE_BF16, out=out_bf.data_ptr(), ldo=3 * D, alpha=1.0)),
        ("qkv  fwd NT heads", 0, M, 3 * D, D, x, D, w_qkv, D,
         ep(L.EPI_QKV_HEADS, out=out_bf.data_ptr(), out2=out_bf2.data_ptr(), out3=out_bf3.data_ptr(), out4=scales.data_ptr(),
            rows_per_sample=256, alpha=1.0)),
        ("fc1  fwd NT store", 0, M, 4 * D, D, x, D, w_fc1, D, ep(L.EPI_STORE_BF16, out=out_bf.data_ptr(), ldo=4 * D, alpha=1.0)),
        ("fc2  fwd NT store", 0, M, D, 4 * D, h, 4 * D, w_fc2, 4 * D, ep(L.EPI_STORE_BF16, out=out_bf.data_ptr(), ldo=D, alpha=1.0)),
        ("proj fwd NT store", 0, M, D, D, x, D, w_qkv, D, ep(L.EPI_STORE_BF16, out=out_bf.data_ptr(), ldo=D, alpha=1.0)),
        ("qkv  dX  NN store", 1, M, D, 3 * D, dqkv, 3 * D, w_qkv, D, ep(L.EPI_STORE_BF16, out=out_bf.data_ptr(), ldo=D, alpha=1.0)),
        ("fc2  dX  NN store", 1, M, 4 * D, D, dy, D, w_fc2, 4 * D, ep(L.EPI_STORE_BF16, out=out_bf.data_ptr(), ldo=4 * D, alpha=1.0)),
        ("fc1  fwd NT silu2", 0, M, 4 * D, D, x, D, w_fc1, D, ep(L.EPI_SILU2, out=out_bf.data_ptr(), out2=out_bf2.data_ptr(), ldo=4 * D)),
        ("fc1  fwd NT silugrad", 0, M, 4 * D, D, x, D, w_fc1, D, ep(L.EPI_SILU2_GRAD, out=out_bf.data_ptr(), out2=out_bf2.data_ptr(), ldo=4 * D)),
        ("fc2  fwd NT resid", 0, M, D, 4 * D, h, 4 * D, w_fc2, 4 * D,
         ep(L.EPI_RESID, out=out_bf.data_ptr(), out2=xout.data_ptr(), aux=xres.data_ptr(), gate=gate.data_ptr(), ldg=6 * D,
            rows_per_sample=256, ldo=D, alpha=0.9, beta=0.4)),
        ("proj fwd NT resid", 0, M, D, D, x, D, w_qkv, D,
         ep(L.EPI_RESID, out=out_bf.data_ptr(), out2=xout.data_ptr(), aux=xres.data_ptr(), gate=gate.data_ptr(), ldg=6 * D,
            rows_per_sample=256, ldo=D, alpha=0.9, beta=0.4)),
        ("fc2  dX  NN dsilu", 1, M, 4 * D, D, dy, D, w_fc2, 4 * D, ep(L.EPI_DSILU, out=out_bf.data_ptr(), aux=h.data_ptr(), ldo=4 * D)),
        ("fc1  dX  NN store", 1, M, D, 4 * D, dh, 4 * D, w_fc1, D, ep(L.EPI_STORE_BF16, out=out_bf.data_ptr(), ldo=D, alpha=1.0)),
        ("fc1  dW  TN split", 2, 4 * D, D, M, dh, 4 * D, x, D, None),
        ("proj dW  TN split", 2, D, D, M, dy, D, x, D, None),
    ]
    variants = [("128", 128, 4), ("256/r3", 256, 7), ("256/w", 256, 2)]
    if args.four_phase:
        variants.append(("256/4ph", 256, 4))
    if args.cases:
        cases = [c for c in cases if any(k in c[0] for k in args.cases.split(","))]
    print(f"{'case':20s} {'kernel':>8s} {'ms':>8s} {'TFLOP/s':>9s}")
    for name, layout, m, n, k, a, lda, b, ldb, e in cases:
        flops = 2.0 * m * n * k
        res = {v[0]: [] for v in variants}
        for _ in range(args.rounds):
            for vname, tile, phases in variants:
                L.lib().gemm_tuning(tile, phases, 0)
                ee = e
                if ee is None:
                    tiles = ((m + tile - 1) // tile) * ((n + tile - 1) // tile)
                    want = (512 if tile == 256 else 1024) // tiles
                    units = k // 64
                    s = max(d for d in range(1, max(1, min(want, units // 4, 64)) + 1) if units % d == 0)
                    ee = ep(L.EPI_STORE_F32, out=out_f32.data_ptr(), ldo=n, alpha=1.0, split_k=s, slab_stride=m * n)
                res[vname].append(run(layout, m, n, k, a, lda, b, ldb, ee, args.iters))
        for vname, _, _ in variants:
            ms = sorted(res[vname])[len(res[vname]) // 2]
            print(f"{name:20s} {vname:>8s} {ms:8.3f} {flops / ms / 1e9:9.1f}")
    L.lib().gemm_tuning(0, 2, 0)


if __name__ == "__main__":
    main()
