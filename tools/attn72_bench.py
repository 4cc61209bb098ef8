#!/usr/bin/env python3
"""Micro-benchmark of the head_dim-72 attention forward on the DiT-XL/2 sampling shape (256 rows of the CFG batch x 16 heads, 256
tokens): raw-q/k form (inference) and the training form that writes the normalised rows back.
    python tools/attn72_bench.py [--batch 256] [--heads 16] [--iters 20] [--f16]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapdit_amd  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--heads", type=int, default=16)
ap.add_argument("--tokens", type=int, default=256)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--f16", action="store_true")
a = ap.parse_args()
lib = mapdit_amd._lib.lib()
B, H, T, hd = a.batch, a.heads, a.tokens, 72
D = H * hd
dt = torch.float16 if a.f16 else torch.bfloat16
sfx = "_f16" if a.f16 else ""
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
q, k, v = rn(B * H, T, hd).to(dt), rn(B * H, T, hd).to(dt), rn(B * H, T, hd).to(dt)
o = torch.empty(B * T, D, device="cuda", dtype=dt)
lse = torch.empty(B * H, T, device="cuda")
scales = torch.empty(2, B * H, T, device="cuda")
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
cases = [("fwd raw (inference)", lambda: getattr(lib, "attn_cos_fwd_rawqk" + sfx)(p(q), p(k), p(v), p(o), p(lse), B, T, H, hd, st)),
         ("fwd raw + save (training)", lambda: getattr(lib, "attn_cos_fwd_rawqk_save" + sfx)(p(q), p(k), p(v), p(o), p(lse), p(scales), B, T, H, hd, st))]
for name, fn in cases:
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    gb = 4 * B * H * T * hd * 2 / 1e9
    print(f"{name:28s} {us:8.1f} us   {gb / us * 1e3:6.2f} TB/s over q, k, v, o   {4.0 * T * T * hd * B * H / us / 1e6:7.1f} TFLOP/s")
