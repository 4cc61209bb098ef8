#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels on the DiT-B/2 shape (256 samples x 12 heads, 256 tokens, head_dim 64): forward, backward
(one-kernel form; MAPDIT_ATTN_BWD=2 in the environment: the two-pass form), HIP-event timing, random unit-norm-8 rows.
    python tools/attn_bench.py [--batch 256] [--heads 12] [--tokens 256] [--f16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapdit_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--heads", type=int, default=12)
ap.add_argument("--tokens", type=int, default=256)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--f16", action="store_true")
a = ap.parse_args()
lib = mapdit_amd._lib.lib()
B, H, T, D = a.batch, a.heads, a.tokens, a.heads * 64
dt = torch.float16 if a.f16 else torch.bfloat16
sfx = "_f16" if a.f16 else ""
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
unit8 = lambda x: (8 * x / x.norm(dim=-1, keepdim=True)).to(dt)
qn, kn, v = unit8(rn(B * H, T, 64)), unit8(rn(B * H, T, 64)), rn(B * H, T, 64).to(dt)
dO, o = (rn(B * T, D) * 0.01).to(dt), torch.empty(B * T, D, device="cuda", dtype=dt)
lse, delta = torch.empty(B * H, T, device="cuda"), torch.empty(B * H, T, device="cuda")
scales = torch.full((2, B * H, T), 1.0, device="cuda")
dqkv = torch.empty(B * T, 3 * D, device="cuda", dtype=dt)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
fwd = lambda: getattr(lib, "attn_cos_fwd" + sfx)(p(qn), p(kn), p(v), p(o), p(lse), B, T, H, 64, st)
bwd = lambda: getattr(lib, "attn_cos_bwd_fused" + sfx)(p(qn), p(kn), p(v), p(dO), p(o), p(lse), p(delta), p(scales), p(dqkv), B, T, H, 64, st)
for name, fn, flop in (("fwd", fwd, 4.0), ("bwd", bwd, 10.0)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    units = 4 if name == "fwd" else 8
    byts = units * B * H * T * 64 * 2
    print(f"{name}: {us:8.1f} us  {flop * B * H * T * T * 64 / us / 1e6:7.1f} TFLOP/s (5-product count)  {byts / us / 1e6:5.2f} TB/s ({units} head tensors)"
          f"  [{'two-pass' if os.environ.get('MAPDIT_ATTN_BWD', '')[:1] == '2' and name == 'bwd' else 'one kernel' if name == 'bwd' else ''}]")
