#!/usr/bin/env python3
"""In-kernel stamps of the ONE-phase K loop of the 256x256 GEMM (same instrumented build as tools/gemm_stamps.py):
per K-tile, for a wave of each wave group: LOAD = {24 fragment reads + 8 LDS-DMA pieces issued | group 1's vmcnt(4) | lgkmcnt(0) +
barrier}, MFMA = {64 MFMAs | vmcnt(0) + barrier}.      python tools/gemm_stamps.py --build ; python tools/gemm_stamps1.py [phases]"""
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
lib.mapdit_debug_set_stamps_block.argtypes = [C.c_void_p, C.c_int]
phases = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib.mapdit_gemm_tuning(256, phases, 0)
TILES, PTS = 12, 11
stamps = torch.zeros(2 * TILES * PTS + 8, dtype=torch.int64, device="cuda")
D, M = 768, 65536
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
dh, w4, x, w4s = rnd(M, 4 * D), rnd(4 * D, D), rnd(M, D), rnd(4 * D, D) * 0.03
out = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
slabs = torch.empty(8, 4 * D * D, device="cuda")
st = torch.cuda.current_stream().cuda_stream
names = ["reads+dma issue", "vmcnt(4) [g1]", "lgkm+barrier", "64 mfma", "vmcnt0+barrier"]
for label, blk in (("NN fc1 dX (K=3072)", 8), ("NT fc1 fwd (K=768) store", 1032), ("TN fc1 dW (K=65536, split 7)", 8)):
    lib.mapdit_debug_set_stamps_block(stamps.data_ptr(), blk)
    e = L.Epilogue()
    for _ in range(20):
        if label.startswith("NN"):
            e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, out.data_ptr(), D, 1.0
            rc = lib.mapdit_gemm_bf16(1, M, D, 4 * D, dh.data_ptr(), 4 * D, w4.data_ptr(), D, C.byref(e), st)
        elif label.startswith("NT"):
            e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, out.data_ptr(), 4 * D, 1.0
            rc = lib.mapdit_gemm_bf16(0, M, 4 * D, D, x.data_ptr(), D, w4s.data_ptr(), D, C.byref(e), st)
        else:
            e.kind, e.out, e.ldo, e.alpha, e.split_k, e.slab_stride = L.EPI_STORE_F32, slabs.data_ptr(), D, 1.0, 7, 4 * D * D
            rc = lib.mapdit_gemm_bf16(2, 4 * D, D, M, dh.data_ptr(), 4 * D, x.data_ptr(), D, C.byref(e), st)
        assert rc == 0
    torch.cuda.synchronize()
    s = stamps.cpu()[:2 * TILES * PTS].view(2, TILES, PTS)
    tt = stamps.cpu()[2 * TILES * PTS:]
    print(f"== {label}, phases={phases}: cycles per K-tile section (median over K-tiles 2..10), wave group 0 | 1")
    if phases == 1:
        for i, nm in enumerate(names):
            d = (s[:, 2:11, i + 1] - s[:, 2:11, i]).float()
            print(f"   {nm:16s} {d[0].median().item():7.0f} | {d[1].median().item():7.0f}")
    per = [int(s[0, i + 1, 0] - s[0, i, 0]) for i in range(TILES - 1)]
    print(f"   K-tile durations (group 0): {per}")
    ghz = (int(tt[2]) - int(tt[1])) / max(int(tt[6]) - int(tt[5]), 1) * 0.1
    print(f"   fill {int(tt[1] - tt[0])}  K loop {int(tt[2] - tt[1])}  epilogue {int(tt[4] - tt[2])}  clock {ghz:.2f} GHz")
