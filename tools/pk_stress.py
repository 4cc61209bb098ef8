#!/usr/bin/env python3
"""Do kernels with packed fp32 VALU math (v_pk_*_f32) reproduce their results bit for bit while another process shares the GPU?
Runs the fc1-shaped GEMM with the SiLU + derivative epilogue (hand-written float2 math), the whole forward of one engine and a few
pointwise kernels repeatedly beside a process that runs training steps.   python tools/pk_stress.py [iters]"""
import ctypes as C
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapdit_amd  # noqa: E402

L = mapdit_amd._lib
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
kids = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "step_stress.py"), str(10 ** 9), "1"],
                         env=dict(os.environ, STEP_STRESS_CHILD="1")) for _ in range(int(os.environ.get("BURNERS", "1")))]
try:
    time.sleep(20)
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    M, D = 8192, 768
    x = torch.randn(M, D, device=dev, generator=g).bfloat16()
    w = (torch.randn(4 * D, D, device=dev, generator=g) * 0.05).bfloat16()
    o1 = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
    o2 = torch.empty_like(o1)
    st = torch.cuda.current_stream().cuda_stream
    lib = L.lib()
    ref = None
    bad = 0
    for it in range(iters):
        e = L.Epilogue()
        e.kind, e.out, e.out2, e.ldo = L.EPI_SILU2_GRAD, o1.data_ptr(), o2.data_ptr(), 4 * D
        lib.gemm_bf16(0, M, 4 * D, D, x.data_ptr(), D, w.data_ptr(), D, C.byref(e), st)
        torch.cuda.synchronize()
        cur = (o1.clone(), o2.clone())
        if ref is None:
            ref = cur
            continue
        for k in range(2):
            if not torch.equal(cur[k], ref[k]):
                bad += 1
                d = torch.nonzero(cur[k] != ref[k])
                if bad <= 8:
                    print(f"iter {it}: output {k} differs in {d.shape[0]} elements, rows {sorted(set(d[:, 0].tolist()))[:6]}, "
                          f"columns {sorted(set(d[:, 1].tolist()))[:10]}")
    print(f"{iters} launches of the fc1 GEMM with the SiLU + derivative epilogue beside a training process: {bad} differing outputs")
finally:
    for k in kids:
        k.kill()
