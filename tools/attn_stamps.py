#!/usr/bin/env python3
"""In-kernel time stamps of the streaming attention backward (build: tools/build_probe.sh attention SB_STAMP 1; run with
MAPDIT_LIB=tools/_ab/lib_SB_STAMP1.so): cycles between the stamp points of waves 0 and 4 of workgroup 0, intervals 16..23."""
import ctypes
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapdit_amd  # noqa: E402

exec(open(os.path.join(os.path.dirname(__file__), "attn_bench.py")).read().split("for name, fn, flop in")[0].replace("a = ap.parse_args()", "a = ap.parse_args([])"))
for _ in range(3):
    bwd()
torch.cuda.synchronize()
cdll = ctypes.CDLL(os.environ["MAPDIT_LIB"])
buf = (ctypes.c_longlong * 128)()
assert cdll.mapdit_debug_attn_stamps(buf) == 0
names = ["S/dP issue", "softmax", "dV/dK", "dq_tile", "commit", "barrier", "post", ]
for w in range(2):
    print("wave", 4 * w)
    for t in range(8):
        st = [buf[(w * 8 + t) * 8 + i] for i in range(8)]
        nxt = buf[(w * 8 + t + 1) * 8] if t < 7 else None
        d = [st[i + 1] - st[i] for i in range(7)]
        print(f"  t={16 + t}: " + "  ".join(f"{n} {x}" for n, x in zip(names, d)) + (f"  | to next {nxt - st[7]}  total {nxt - st[0]}" if nxt else ""))
