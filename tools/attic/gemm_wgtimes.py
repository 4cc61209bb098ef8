#!/usr/bin/env python3
"""Per-workgroup wall time of the 256x256 GEMM (instrumented library): every workgroup records s_memrealtime at entry / exit, its
cycle count and its XCC id.  Prints the distribution per schedule: where a launch's time goes beyond (tiles / CUs) x tile time.
    python tools/gemm_stamps.py --build ; python tools/gemm_wgtimes.py"""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import mapdit_amd  # noqa: E402

L = mapdit_amd._lib
lib = C.CDLL(os.path.join(HERE, "_stamps", "libgemm_stamps.so"))
lib.mapdit_gemm_bf16.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(L.Epilogue), C.c_void_p]
lib.mapdit_gemm_tuning.argtypes = [C.c_int, C.c_int, C.c_long]
lib.mapdit_debug_set_wg_times.argtypes = [C.c_void_p]
D, M = 768, 65536
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
x, w4, dh = rnd(M, D), rnd(4 * D, D) * 0.03, rnd(M, 4 * D)
out = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
rec = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
lib.mapdit_debug_set_wg_times(rec.data_ptr())


def run(name, layout, m, n, k, a, lda, b, ldb, phases):
    lib.mapdit_gemm_tuning(256, phases, 0)
    e = L.Epilogue()
    e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, out.data_ptr(), n, 1.0
    for _ in range(200):
        lib.mapdit_gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        lib.mapdit_gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 20
    nwg = ((m + 255) // 256) * ((n + 255) // 256)
    r = rec.cpu()[:nwg * 4].view(nwg, 4)
    t0, t1, cyc, xcc = r[:, 0], r[:, 1], r[:, 2].float(), r[:, 3] & 0xff
    span = (t1.max() - t0.min()).item() / 100.0          # us (100 MHz)
    dur = (t1 - t0).float() / 100.0
    order = torch.argsort(t0)
    rounds = [dur[order[i * 256:(i + 1) * 256]].mean().item() for i in range(nwg // 256)]
    print(f"== {name} phases={phases}: launch {ms * 1e3:.1f} us ({2.0 * m * n * k / ms / 1e9:.0f} TFLOP/s), first entry -> last exit {span:.1f} us, "
          f"{nwg} workgroups")
    print(f"   per workgroup: mean {dur.mean():.2f} us  median {dur.median():.2f}  p10 {dur.kthvalue(max(1, nwg // 10)).values:.2f}  "
          f"p90 {dur.kthvalue(nwg * 9 // 10).values:.2f}  max {dur.max():.2f};  cycles mean {cyc.mean():.0f} median {cyc.median():.0f} "
          f"-> clock {cyc.mean() / dur.mean() / 1e3:.2f} GHz;  sum / 256 CUs = {dur.sum().item() / 256:.1f} us")
    print("   mean duration by dispatch round (256 workgroups each): " + " ".join(f"{v:.1f}" for v in rounds))
    per_x = [dur[xcc == i].mean().item() for i in range(8)]
    cnt_x = [int((xcc == i).sum()) for i in range(8)]
    print("   by XCC: " + " ".join(f"{v:.1f}({c})" for v, c in zip(per_x, cnt_x)))
    last = (t1.float() - t0.min().float()) / 100.0
    busy_end = torch.sort(last).values
    print(f"   exits: 50% of workgroups done at {busy_end[nwg // 2]:.1f} us, 90% at {busy_end[nwg * 9 // 10]:.1f}, 99% at {busy_end[nwg * 99 // 100]:.1f}, "
          f"all at {busy_end[-1]:.1f}")


for ph in (2, 1):
    run("NT [65536,768]x[3072,768]^T", 0, M, 4 * D, D, x, D, w4, D, ph)
for ph in (2, 1):
    run("NN [65536,3072]x[3072,768]", 1, M, D, 4 * D, dh, 4 * D, w4, D, ph)
