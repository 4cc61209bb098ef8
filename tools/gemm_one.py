#!/usr/bin/env python3
"""Run ONE GEMM case repeatedly (for rocprofv3 --pmc): python tools/gemm_one.py {nt|nn|tn|fc1} [tile]
(fc1 = the bench's roofline kernel: NT [65536,768]x[3072,768]^T with the SILU2_GRAD dual-store epilogue)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapdit_amd
L = mapdit_amd._lib
kind = sys.argv[1]
if len(sys.argv) > 2: os.environ["MAPDIT_GEMM_TILE"] = sys.argv[2]
D, M = 768, 65536
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
x, w, dh = rnd(M, D), rnd(4 * D, D) * 0.03, rnd(M, 4 * D)
out = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
out2 = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
slabs = torch.empty(16, 4 * D * D, device="cuda")
e = L.Epilogue()
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    if kind == "nt":
        e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, out.data_ptr(), 4 * D, 1.0
        L.lib().gemm_bf16(0, M, 4 * D, D, x.data_ptr(), D, w.data_ptr(), D, C.byref(e), st)
    elif kind == "fc1":
        e.kind, e.out, e.out2, e.ldo = L.EPI_SILU2_GRAD, out.data_ptr(), out2.data_ptr(), 4 * D
        L.lib().gemm_bf16(0, M, 4 * D, D, x.data_ptr(), D, w.data_ptr(), D, C.byref(e), st)
    elif kind == "nn":
        e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, out.data_ptr(), D, 1.0
        L.lib().gemm_bf16(1, M, D, 4 * D, dh.data_ptr(), 4 * D, w.data_ptr(), D, C.byref(e), st)
    else:
        e.kind, e.out, e.ldo, e.alpha, e.split_k, e.slab_stride = L.EPI_STORE_F32, slabs.data_ptr(), D, 1.0, 7, 4 * D * D
        L.lib().gemm_bf16(2, 4 * D, D, M, dh.data_ptr(), 4 * D, x.data_ptr(), D, C.byref(e), st)
torch.cuda.synchronize()
