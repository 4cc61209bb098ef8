#!/usr/bin/env python3
"""Timeline of the 256x256 GEMM's K-tile loop from in-kernel shader-clock stamps (guide: 'In-kernel stamps').

Builds gemm.hip with -DMAPDIT_GEMM_STAMPS into tools/_stamps/libgemm_stamps.so (a separate library; the product library has
no instrumentation), runs one NN GEMM [65536,3072] x [3072,768] and one TN GEMM, and prints, for a wave of each of the two
wave groups, the cycles spent per K-tile in: LOAD A (fragment reads + DMA issue | DMA wait | barrier), MFMA A, LOAD B, MFMA B.

    python tools/gemm_stamps.py --build      # here (hipcc cross-compiles)
    python tools/gemm_stamps.py              # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "_stamps")
SO = os.path.join(OUT, "libgemm_stamps.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "map-dit_amd", "csrc", "gemm.hip")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast",
                           "-fno-slp-vectorize", "-DMAPDIT_GEMM_STAMPS", "-DMAPDIT_GEMM_EXPERIMENTS", "-Wno-unused-function", src, "-o", SO])
    print("built", SO)


def main():
    import torch
    sys.path.insert(0, ROOT)
    import mapdit_amd
    L = mapdit_amd._lib
    lib = C.CDLL(SO)
    lib.mapdit_gemm_bf16.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(L.Epilogue), C.c_void_p]
    lib.mapdit_gemm_bf16.restype = C.c_int
    lib.mapdit_last_error.restype = C.c_char_p
    TILES, PTS = 12, 11
    stamps = torch.zeros(2 * TILES * PTS + 8, dtype=torch.int64, device="cuda")
    # set the device-side pointer through the tiny setter kernel
    mod_launch = getattr(lib, "mapdit_debug_set_stamps", None)
    assert mod_launch is not None, "stamp build lacks mapdit_debug_set_stamps"
    mod_launch.argtypes = [C.c_void_p]
    mod_launch(stamps.data_ptr())
    torch.cuda.synchronize()
    D, M = 768, 65536
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
    dh, w4, x = rnd(M, 4 * D), rnd(4 * D, D), rnd(M, D)
    out = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    slabs = torch.empty(8, 4 * D * D, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    names = ["A:issue", "A:dma-wait", "A:barrier", "A:mfma", "A:barrier2", "B:issue", "B:dma-wait", "B:barrier", "B:mfma", "B:barrier2"]
    h2, w2 = rnd(M, 4 * D), rnd(D, 4 * D)
    for label in ("NT fc2 fwd (K=3072)", "NN fc1 dX (K=3072)", "TN fc1 dW (K=65536, split 7)"):
        e = L.Epilogue()
        for _ in range(3):
            if label.startswith("NT"):
                e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, out.data_ptr(), D, 1.0
                rc = lib.mapdit_gemm_bf16(0, M, D, 4 * D, h2.data_ptr(), 4 * D, w2.data_ptr(), 4 * D, C.byref(e), st)
            elif label.startswith("NN"):
                e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, out.data_ptr(), D, 1.0
                rc = lib.mapdit_gemm_bf16(1, M, D, 4 * D, dh.data_ptr(), 4 * D, w4.data_ptr(), D, C.byref(e), st)
            else:
                e.kind, e.out, e.ldo, e.alpha, e.split_k, e.slab_stride = L.EPI_STORE_F32, slabs.data_ptr(), D, 1.0, 7, 4 * D * D
                rc = lib.mapdit_gemm_bf16(2, 4 * D, D, M, dh.data_ptr(), 4 * D, x.data_ptr(), D, C.byref(e), st)
            assert rc == 0, lib.mapdit_last_error()
        torch.cuda.synchronize()
        s = stamps.cpu()[:2 * TILES * PTS].view(2, TILES, PTS)
        print(f"== {label}: cycles per K-tile section (median over K-tiles 2..{TILES - 1}), wave group 0 | 1")
        for i, nm in enumerate(names):
            d = (s[:, 2:, i + 1] - s[:, 2:, i]).float()
            print(f"   {nm:12s} {d[0].median().item():7.0f} | {d[1].median().item():7.0f}")
        per = (s[:, 3:, 0] - s[:, 2:-1, 0]).float()
        print(f"   K-tile total {per[0].median().item():7.0f} | {per[1].median().item():7.0f}   (MFMA issue alone: 2 x 512)")
    if "--kloop-only" in sys.argv:
        return
    # whole-tile timeline of the forward shape (NT [65536,768] x [3072,768]^T, 12 K-tiles per output tile, 12 rounds of tiles):
    # where the time of one tile goes outside the K loop, for a first-round workgroup and for later ones
    setb = lib.mapdit_debug_set_stamps_block
    setb.argtypes = [C.c_void_p, C.c_int]
    xa, wb = rnd(M, D), rnd(4 * D, D)
    o2 = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
    o3 = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
    for kind in ("bf16 store", "SiLU dual store"):
        for blk in (8, 1032, 2056):
            setb(stamps.data_ptr(), blk)
            e = L.Epilogue()
            if kind == "bf16 store":
                e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_BF16, o2.data_ptr(), 4 * D, 1.0
            else:
                e.kind, e.out, e.out2, e.ldo = L.EPI_SILU2, o2.data_ptr(), o3.data_ptr(), 4 * D
            import time
            t_end = time.time() + (2.2 if blk == 8 else 0.0)      # >= 2 s of back-to-back launches before the clock is read
            n = 0
            while n < 3 or time.time() < t_end:
                rc = lib.mapdit_gemm_bf16(0, M, 4 * D, D, xa.data_ptr(), D, wb.data_ptr(), D, C.byref(e), st)
                assert rc == 0, lib.mapdit_last_error()
                n += 1
                if n % 64 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            tt = stamps.cpu()[2 * TILES * PTS:]
            kt = stamps.cpu()[:2 * TILES * PTS].view(2, TILES, PTS)
            per_tile = [int(kt[0, i + 1, 0] - kt[0, i, 0]) for i in range(TILES - 1)]
            print(f"   K-tile durations of that workgroup (wave group 0), tiles 0..{TILES - 2}: {per_tile}; last tile "
                  f"{int(kt[0, TILES - 1, PTS - 1] - kt[0, TILES - 1, 0])}; K-loop entry -> first tile {int(kt[0, 0, 0] - tt[1])}; "
                  f"last tile end -> epilogue start {int(tt[2] - kt[0, TILES - 1, PTS - 1])}; wave group 1 ends its last tile "
                  f"{int(kt[1, TILES - 1, PTS - 1] - kt[0, TILES - 1, PTS - 1])} after group 0")
            d = [int(tt[i + 1] - tt[i]) for i in range(4)]
            ghz = (int(tt[2]) - int(tt[1])) / max(int(tt[6]) - int(tt[5]), 1) * 0.1
            print(f"== NT fc1 shape, {kind}, workgroup {blk}: cycles  fill (entry -> K loop) {d[0]}  K loop (12 tiles) {d[1]}  "
                  f"epilogue pass 0 {d[2]}  pass 1 {d[3]}  | total {int(tt[4] - tt[0])}  | K-loop clock {ghz:.2f} GHz (random data)")


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
