#!/usr/bin/env python3
"""Stress of the weight-gradient path of a [256, 256] linear (TN split-K GEMM into slabs -> mapdit_weightnorm_bwd) while another
process keeps the GPU busy: every iteration must reproduce the first one bit for bit.   python tools/dw_stress.py [iters]"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapdit_amd  # noqa: E402

L = mapdit_amd._lib

if len(sys.argv) > 1 and sys.argv[1] == "burn":
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    while True:
        for _ in range(50):
            a = (a @ a).clamp(-1, 1)
        torch.cuda.synchronize()

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
if os.environ.get("BURN_STEP"):      # the other process runs training steps (many short kernels) instead of long matmuls
    burners = [subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_stress.py"), str(10 ** 9), "1"],
                                env=dict(os.environ, STEP_STRESS_CHILD="1")) for _ in range(int(os.environ.get("BURNERS", "1")))]
else:
    burners = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "burn"]) for _ in range(int(os.environ.get("BURNERS", "1")))]
try:
    import time
    time.sleep(float(os.environ.get("BURN_WARMUP", "20")))             # let the burners get onto the GPU
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rows = cols = 256
    K, S = 2048, 8
    # two operand sets, alternating: a consumer that reads a stale copy of the slabs sees the OTHER set's values
    dys = [torch.randn(K, rows, device=dev, generator=g).bfloat16() for _ in range(2)]
    xs = [torch.randn(K, cols, device=dev, generator=g).bfloat16() for _ in range(2)]
    W = torch.randn(rows, cols, device=dev, generator=g)
    G = torch.empty(64, rows, cols, device=dev)
    dW = torch.empty(rows, cols, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib = L.lib()
    dbg = None
    if os.environ.get("WN_DEBUG"):          # hipcc ... -DMAPDIT_WN_DEBUG weights.hip -> tools/_stamps/libwn_debug.so
        dlib = C.CDLL(os.path.join(ROOT, "tools", "_stamps", os.environ.get("WN_DEBUG_LIB", "libwn_debug_bperm.so")))
        dlib.mapdit_weightnorm_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        dlib.mapdit_wn_dbg_set.argtypes = [C.c_void_p]
        dbg = torch.zeros(rows, 132, device=dev)
        dlib.mapdit_wn_dbg_set(dbg.data_ptr())
    ref_dbg = [None, None]
    refs = [None, None]
    bad_gemm = bad_sum = bad_dw = 0
    other = torch.randn(1 << 22, device=dev)
    for it in range(iters):
        if os.environ.get("PREFILL"):
            G.copy_(other[: G.numel()].view_as(G) * (it + 1))       # stale contents of the scratch differ every iteration
        e = L.Epilogue()
        e.kind, e.out, e.ldo, e.alpha, e.split_k, e.slab_stride = L.EPI_STORE_F32, G.data_ptr(), cols, 1.0, S, rows * cols
        dy, x = dys[it & 1], xs[it & 1]
        lib.gemm_bf16(2, rows, cols, K, dy.data_ptr(), rows, x.data_ptr(), cols, C.byref(e), st)
        Gc = G[:S].clone() if os.environ.get('CLONE_SLABS') else None      # (a kernel between the GEMM and its consumer)
        if dbg is not None:
            dlib.mapdit_weightnorm_bwd(W.data_ptr(), G.data_ptr(), cols, S, rows * cols, dW.data_ptr(), rows, cols, 1.0, 0, st)
        else:
            lib.weightnorm_bwd(W.data_ptr(), G.data_ptr(), cols, S, rows * cols, dW.data_ptr(), rows, cols, 1.0, 0, st)
        torch.cuda.synchronize()
        if refs[it & 1] is None:
            refs[it & 1] = (Gc, G[0].clone(), dW.clone())
            ref_dbg[it & 1] = dbg.clone() if dbg is not None else None
            continue
        ref_G, ref_sum, ref_dW = refs[it & 1]
        if Gc is not None and not torch.equal(Gc, ref_G):
            bad_gemm += 1
            idx = torch.nonzero((Gc != ref_G).any(dim=2))
            print(f"iter {it}: GEMM slabs differ at (slab,row) {idx[:6].tolist()} ({int((Gc != ref_G).sum())} elements)")
        if not torch.equal(G[0], ref_sum):
            bad_sum += 1
            print(f"iter {it}: parked slab sum differs in rows {torch.nonzero((G[0] != ref_sum).any(dim=1)).flatten()[:8].tolist()}")
        if not torch.equal(dW, ref_dW):
            bad_dw += 1
            rws = torch.nonzero((dW != ref_dW).any(dim=1)).flatten()[:8].tolist()
            print(f"iter {it}: dW differs in rows {rws}")
            if dbg is not None and bad_dw <= 10:
                r = rws[0]
                a, b = dbg[r].cpu(), ref_dbg[it & 1][r].cpu()
                names = ["ss", "gw", "a1", "a2"]
                print("      " + "  ".join(f"{n} {float(a[i]):.9g} vs {float(b[i]):.9g}" for i, n in enumerate(names)))
                ln = torch.nonzero(a[4:68] != b[4:68]).flatten().tolist()
                ls = torch.nonzero(a[68:] != b[68:]).flatten().tolist()
                print(f"      lanes whose partial of gw differs: {ln[:16]} ({len(ln)}); of ss: {ls[:16]} ({len(ls)}): " +
                      ", ".join(f"{float(a[68 + l]):.9g} vs {float(b[68 + l]):.9g}" for l in ls[:4]) +
                      f"; sum of this run's ss partials {float(a[68:].double().sum()):.6f}, of the reference's {float(b[68:].double().sum()):.6f}")
    print(f"{iters} iterations: GEMM output differed {bad_gemm}x, slab sum {bad_sum}x, dW {bad_dw}x")
finally:
    for b in burners:
        b.kill()
