#!/usr/bin/env python3
"""Run-to-run and kernel-to-kernel bit comparison of the block's GEMM cases at full size: every case is launched `--reps` times
with the round-4 kernel (gemm_tuning phases 2) and once with the round-3 kernel (phases 7); all outputs must be the same bits.

    python tools/gemm_repro.py [--tokens 65536] [--reps 6]
"""
import argparse, ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapdit_amd
L = mapdit_amd._lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--tokens", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=6)
    args = ap.parse_args()
    D, M = args.hidden, args.tokens
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
    x, w_qkv, w_fc1, w_fc2 = rnd(M, D), rnd(3 * D, D) * 0.03, rnd(4 * D, D) * 0.03, rnd(D, 4 * D) * 0.03
    h, dy, dh = rnd(M, 4 * D), rnd(M, D), rnd(M, 4 * D)
    outs = [torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16) for _ in range(3)]
    scales = torch.empty(2 * M * (D // 64), device=dev)
    out_f32 = torch.empty(64, 4 * D * D, device=dev)
    xres = torch.randn(M, D, device=dev, generator=g)
    xout = torch.empty_like(xres)
    gate = torch.randn(M // 256, 6 * D, device=dev, generator=g)

    def ep(kind, **kw):
        e = L.Epilogue()
        e.kind = kind
        for k, v in kw.items():
            setattr(e, k, v)
        return e

    o0, o1, o2 = outs
    cases = [
        ("qkv NT store", 0, M, 3 * D, D, x, D, w_qkv, D, ep(L.EPI_STORE_BF16, out=o0.data_ptr(), ldo=3 * D, alpha=1.0), [o0]),
        ("qkv NT heads", 0, M, 3 * D, D, x, D, w_qkv, D,
         ep(L.EPI_QKV_HEADS, out=o0.data_ptr(), out2=o1.data_ptr(), out3=o2.data_ptr(), out4=scales.data_ptr(), rows_per_sample=256, alpha=1.0),
         [o0, o1, o2, scales]),
        ("fc1 NT silugrad", 0, M, 4 * D, D, x, D, w_fc1, D, ep(L.EPI_SILU2_GRAD, out=o0.data_ptr(), out2=o1.data_ptr(), ldo=4 * D), [o0, o1]),
        ("fc2 NT resid", 0, M, D, 4 * D, h, 4 * D, w_fc2, 4 * D,
         ep(L.EPI_RESID, out=o0.data_ptr(), out2=xout.data_ptr(), aux=xres.data_ptr(), gate=gate.data_ptr(), ldg=6 * D, rows_per_sample=256,
            ldo=D, alpha=0.9, beta=0.4), [o0, xout]),
        ("proj NT resid", 0, M, D, D, x, D, w_qkv, D,
         ep(L.EPI_RESID, out=o0.data_ptr(), out2=xout.data_ptr(), aux=xres.data_ptr(), gate=gate.data_ptr(), ldg=6 * D, rows_per_sample=256,
            ldo=D, alpha=0.9, beta=0.4), [o0, xout]),
        ("fc2 dX NN mulaux", 1, M, 4 * D, D, dy, D, w_fc2, 4 * D, ep(L.EPI_MUL_AUX, out=o0.data_ptr(), aux=h.data_ptr(), ldo=4 * D), [o0]),
        ("fc1 dX NN store", 1, M, D, 4 * D, dh, 4 * D, w_fc1, D, ep(L.EPI_STORE_BF16, out=o0.data_ptr(), ldo=D, alpha=1.0), [o0]),
        ("qkv dX NN store", 1, M, D, 3 * D, dh, 4 * D, w_qkv, D, ep(L.EPI_STORE_BF16, out=o0.data_ptr(), ldo=D, alpha=1.0), [o0]),
        ("fc1 dW TN split", 2, 4 * D, D, M, dh, 4 * D, x, D,
         ep(L.EPI_STORE_F32, out=out_f32.data_ptr(), ldo=D, alpha=1.0, split_k=16, slab_stride=4 * D * D), [out_f32]),
        ("fc2 dW TN split", 2, D, 4 * D, M, dy, D, dh, 4 * D,
         ep(L.EPI_STORE_F32, out=out_f32.data_ptr(), ldo=4 * D, alpha=1.0, split_k=16, slab_stride=4 * D * D), [out_f32]),
        ("proj dW TN split", 2, D, D, M, dy, D, x, D,
         ep(L.EPI_STORE_F32, out=out_f32.data_ptr(), ldo=D, alpha=1.0, split_k=32, slab_stride=D * D), [out_f32]),
    ]
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    bad = 0
    for name, layout, m, n, k, a, lda, b, ldb, e, results in cases:
        def run(phases):
            lib.gemm_tuning(256, phases, 0)
            for r in results:
                r.zero_()
            lib.gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
            torch.cuda.synchronize()
            return [r.clone() for r in results]
        ref = run(7)
        diffs = []
        for rep in range(args.reps):
            got = run(2)
            nd = sum(int((g_.view(torch.int16 if g_.dtype == torch.bfloat16 else torch.int32) !=
                          r_.view(torch.int16 if r_.dtype == torch.bfloat16 else torch.int32)).sum()) for g_, r_ in zip(got, ref))
            diffs.append(nd)
        print(f"{name:20s} elements differing from the round-3 kernel per repetition: {diffs}")
        bad += sum(diffs)
    lib.gemm_tuning(0, 2, 0)
    print("OK" if bad == 0 else f"MISMATCH ({bad})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
