#!/usr/bin/env python3
"""Ablation timing of the two-phase 256x256 K loop (results are garbage, timing only): builds gemm.hip four times
(-DMAPDIT_GEMM_ABLATE=0..3: bit 0 = no LDS-DMA in the loop, bit 1 = no fragment reads after the first K-tile) with the per-workgroup
time records of the instrumented build, and reports launch time and mean cycles per workgroup.
    python tools/gemm_ablate.py --build   (here)      python tools/gemm_ablate.py   (GPU box)"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_stamps")


def so(n):
    return os.path.join(OUT, f"libgemm_ablate{n}.so")


if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "map-dit_amd", "csrc", "gemm.hip")
    procs = [subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast",
                               "-fno-slp-vectorize", "-DMAPDIT_GEMM_STAMPS", "-DMAPDIT_GEMM_EXPERIMENTS", f"-DMAPDIT_GEMM_ABLATE={n}", "-Wno-unused-function", src,
                               "-o", so(n)]) for n in range(4)]
    assert all(p.wait() == 0 for p in procs)
    print("built")
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, ROOT)
import mapdit_amd  # noqa: E402

L = mapdit_amd._lib
D, M = 768, 65536
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
x, w4, dh = rnd(M, D), rnd(4 * D, D) * 0.03, rnd(M, 4 * D)
out = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
slabs = torch.empty(8, 4 * D * D, device="cuda")
st = torch.cuda.current_stream().cuda_stream
rec = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
names = {0: "full loop", 1: "no LDS-DMA", 2: "no fragment reads", 3: "neither (MFMA + barriers)"}
libs = {}
for n in range(4):
    lib = C.CDLL(so(n))
    lib.mapdit_gemm_bf16.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(L.Epilogue), C.c_void_p]
    lib.mapdit_debug_set_wg_times.argtypes = [C.c_void_p]
    lib.mapdit_debug_set_wg_times(rec.data_ptr())
    libs[n] = lib
cases = [("NT [65536,768]x[3072,768]^T store", 0, M, 4 * D, D, x, D, w4, D, None),
         ("NN [65536,3072]x[3072,768] store", 1, M, D, 4 * D, dh, 4 * D, w4, D, None),
         ("TN [3072,65536]x[65536,768] split 7", 2, 4 * D, D, M, dh, 4 * D, x, D, 7)]
for name, layout, m, n, k, a, lda, b, ldb, split in cases:
    print(f"== {name}")
    for rnd_ in range(2):
        for ab in range(4):
            lib = libs[ab]
            e = L.Epilogue()
            if split:
                e.kind, e.out, e.ldo, e.alpha, e.split_k, e.slab_stride = L.EPI_STORE_F32, slabs.data_ptr(), n, 1.0, split, m * n
            else:
                e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, out.data_ptr(), n, 1.0
            for _ in range(30):
                lib.mapdit_gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(20):
                lib.mapdit_gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / 20
            nwg = ((m + 255) // 256) * ((n + 255) // 256) * (split or 1)
            r = rec.cpu()[:nwg * 4].view(nwg, 4)
            cyc = r[:, 2].float()
            dur = (r[:, 1] - r[:, 0]).float() / 100.0
            if rnd_ == 1:
                print(f"   {names[ab]:28s} {ms * 1e3:7.1f} us  {2.0 * m * n * k / ms / 1e9:7.0f} TFLOP/s   cycles per workgroup {cyc.mean():8.0f}"
                      f"   clock {cyc.mean() / dur.mean() / 1e3:.2f} GHz   fill {((r[:, 3] >> 32) & 0xffff).float().mean():6.0f}"
                      f"  K loop {((r[:, 3] >> 8) & 0xffffff).float().mean():7.0f} ({k // (split or 1) // 64} K-tiles)"
                      f"  epilogue {((r[:, 3] >> 48) & 0xffff).float().mean():6.0f}")
