#!/usr/bin/env python3
"""Sustained-throughput vs. clock for the 256x256 GEMM schedules: each variant runs back to back for SECONDS seconds (the chip's
power management settles within that), then reports TFLOP/s over the last half and the in-kernel clock of the K loop
(s_memtime / s_memrealtime stamps, instrumented library).   python tools/gemm_stamps.py --build; python tools/gemm_clock.py"""
import ctypes as C
import os
import subprocess
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import mapdit_amd  # noqa: E402

L = mapdit_amd._lib
lib = C.CDLL(os.path.join(HERE, "_stamps", "libgemm_stamps.so"))
lib.mapdit_gemm_bf16.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(L.Epilogue), C.c_void_p]
lib.mapdit_gemm_tuning.argtypes = [C.c_int, C.c_int, C.c_long]
lib.mapdit_debug_set_stamps_block.argtypes = [C.c_void_p, C.c_int]
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
TILES, PTS = 12, 11
stamps = torch.zeros(2 * TILES * PTS + 8, dtype=torch.int64, device="cuda")
lib.mapdit_debug_set_stamps_block(stamps.data_ptr(), 1032)
D, M = 768, 65536
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
x, w4, dh = rnd(M, D), rnd(4 * D, D) * 0.03, rnd(M, 4 * D)
zx, zw = torch.zeros_like(x), torch.zeros_like(w4)
out = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream


def power():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
        return r.stdout.strip()[:400]
    except Exception as e:  # noqa: BLE001
        return f"(rocm-smi: {e})"


def run(name, layout, m, n, k, a, lda, b, ldb, phases):
    lib.mapdit_gemm_tuning(256, phases, 0)
    e = L.Epilogue()
    e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, out.data_ptr(), n, 1.0
    t0 = time.time()
    n_launch, t_half, n_half = 0, None, 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pw = ""
    while True:
        for _ in range(50):
            lib.mapdit_gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
        n_launch += 50
        torch.cuda.synchronize()
        now = time.time() - t0
        if t_half is None and now > SECONDS / 2:
            t_half, n_half = now, n_launch
            ev0.record()
            pw = power()
        if now > SECONDS:
            break
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / max(n_launch - n_half, 1)
    tt = stamps.cpu()[2 * TILES * PTS:]
    ghz = (int(tt[2]) - int(tt[1])) / max(int(tt[6]) - int(tt[5]), 1) * 0.1
    print(f"{name:34s} phases={phases}: {2.0 * m * n * k / ms / 1e9:7.1f} TFLOP/s  {ms * 1e3:7.1f} us  K-loop clock {ghz:.2f} GHz  "
          f"K loop {int(tt[2] - tt[1])} cyc  tile {int(tt[4] - tt[0])} cyc")
    if pw:
        print("     ", pw.replace("\n", " ")[:300])


for ph in (2, 1, 2, 1):
    run("NT [65536,768]x[3072,768]^T random", 0, M, 4 * D, D, x, D, w4, D, ph)
for ph in (2, 1):
    run("NN [65536,3072]x[3072,768] random", 1, M, D, 4 * D, dh, 4 * D, w4, D, ph)
for ph in (2, 1):
    run("NT [65536,768]x[3072,768]^T zeros", 0, M, 4 * D, D, zx, D, zw, D, ph)
