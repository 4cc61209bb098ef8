#!/usr/bin/env python3
"""Round 5, VERDICT r04 item 2(b): what does the K LOOP cost in the 256x128 geometry that "two tiles in flight per CU" needs (64 accumulator
registers per wave, three-deep ring of 48-KiB K-tile stages, 16 fragment reads + 6 LDS-DMA pieces per 32 MFMAs)?  The experiment kernel
(gemm.hip, gemm_x128_kernel, -DMAPDIT_GEMM_EXPERIMENTS) against the shipped 256^2 kernels on plain 16-bit stores, one process, interleaved.
Its results are checked bit for bit against the shipped kernel (same accumulation order).

    python tools/gemm_x128.py --build     # here (hipcc cross-compiles): tools/_stamps/libgemm_exp.so
    python tools/gemm_x128.py             # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "_stamps", "libgemm_exp.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(ROOT, "map-dit_amd", "csrc", "gemm.hip")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast", "-fno-slp-vectorize",
                           "-fno-vectorize", "-DMAPDIT_GEMM_EXPERIMENTS", "-Wno-unused-function", src, "-o", SO])
    print("built", SO)


def main():
    import torch
    sys.path.insert(0, ROOT)
    import mapdit_amd
    L = mapdit_amd._lib
    lib = C.CDLL(SO)
    lib.mapdit_gemm_bf16.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(L.Epilogue), C.c_void_p]
    lib.mapdit_gemm_tuning.argtypes = [C.c_int, C.c_int, C.c_long]
    lib.mapdit_last_error.restype = C.c_char_p
    D, M = 768, 65536
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
    x, h, dh, dq = rnd(M, D), rnd(M, 4 * D), rnd(M, 4 * D), rnd(M, 3 * D)
    w_qkv, w_fc1, w_fc2 = rnd(3 * D, D) * 0.03, rnd(4 * D, D) * 0.03, rnd(D, 4 * D) * 0.03
    st = torch.cuda.current_stream().cuda_stream
    cases = [("fc2  fwd NT K=3072 N=768", 0, M, D, 4 * D, h, 4 * D, w_fc2, 4 * D),
             ("proj fwd NT K=768  N=768", 0, M, D, D, x, D, w_qkv, D),
             ("fc1  fwd NT K=768  N=3072", 0, M, 4 * D, D, x, D, w_fc1, D),
             ("fc1  dX  NN K=3072 N=768", 1, M, D, 4 * D, dh, 4 * D, w_fc1, D),
             ("qkv  dX  NN K=2304 N=768", 1, M, D, 3 * D, dq, 3 * D, w_qkv, D),
             ("fc2  dX  NN K=768  N=3072", 1, M, 4 * D, D, x, D, w_fc2, 4 * D)]
    variants = [("256/r3", 7), ("256/w", 2), ("256x128", 8)]
    print(f"{'case':28s} {'kernel':>8s} {'us':>8s} {'TFLOP/s':>8s}")
    for name, layout, m, n, k, a, lda, b, ldb in cases:
        outs, res = {}, {v: [] for v, _ in variants}
        for rnd_ in range(3):
            for v, ph in variants:
                lib.mapdit_gemm_tuning(256, ph, 0)
                out = outs.setdefault(v, torch.zeros(m, n, device=dev, dtype=torch.bfloat16))
                e = L.Epilogue()
                e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, out.data_ptr(), n, 1.0
                for _ in range(2):
                    rc = lib.mapdit_gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
                    assert rc == 0, lib.mapdit_last_error()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    lib.mapdit_gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
                e1.record()
                torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / 8)
        same = torch.equal(outs["256x128"], outs["256/w"])
        for v, _ in variants:
            ms = sorted(res[v])[1]
            print(f"{name:28s} {v:>8s} {ms * 1e3:8.1f} {2.0 * m * n * k / ms / 1e9:8.0f}" + (f"   bits == 256/w: {same}" if v == "256x128" else ""))
    lib.mapdit_gemm_tuning(0, 2, 0)


if __name__ == "__main__":
    build() if "--build" in sys.argv else main()
