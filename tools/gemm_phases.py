#!/usr/bin/env python3
"""Where a tile's time goes in the round-4 256x256 GEMM kernel (instrumented build, tools/gemm_stamps.py --build): per workgroup,
waves 0 and 4 sum the shader cycles they spend in {wait for the prologue + first barrier, K loop, issue of the next tile's
prologue, epilogue}; printed as mean cycles per tile over all workgroups.

    python tools/gemm_stamps.py --build ; python tools/gemm_phases.py [--cases heads,silugrad]
"""
import argparse, ctypes as C, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mapdit_amd
L = mapdit_amd._lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="")
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--tokens", type=int, default=65536)
    ap.add_argument("--cold", action="store_true", help="write 1 GiB elsewhere before every launch: the cache state a launch meets inside the "
                                                        "training step (its operands were produced long ago, L2 / Infinity Cache hold other data)")
    ap.add_argument("--force-w", action="store_true", help="phases = 6: the stamped (wave-private epilogue) kernel for EVERY epilogue - RESID and the "
                                                          "fp32 split-K store (weight gradients, TN) included: their K loops are the shipped kernels' K loop")
    args = ap.parse_args()
    lib = C.CDLL(os.path.join(HERE, "_stamps", "libgemm_stamps.so"))
    lib.mapdit_gemm_bf16.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(L.Epilogue), C.c_void_p]
    lib.mapdit_gemm_tuning.argtypes = [C.c_int, C.c_int, C.c_long]
    lib.mapdit_debug_set_wg_times.argtypes = [C.c_void_p]
    lib.mapdit_last_error.restype = C.c_char_p
    D, M = args.hidden, args.tokens
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
    x, w_qkv, w_fc1, w_fc2 = rnd(M, D), rnd(3 * D, D) * 0.03, rnd(4 * D, D) * 0.03, rnd(D, 4 * D) * 0.03
    h, dy, dh = rnd(M, 4 * D), rnd(M, D), rnd(M, 4 * D)
    o0, o1, o2 = (torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16) for _ in range(3))
    scales = torch.empty(2 * M * (D // 64), device=dev)
    out_f32 = torch.empty(64, 4 * D * D, device=dev)
    xres = torch.randn(M, D, device=dev, generator=g)
    xout = torch.empty_like(xres)
    gate = torch.randn(M // 256, 6 * D, device=dev, generator=g)
    rec = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
    lib.mapdit_debug_set_wg_times(rec.data_ptr())

    def ep(kind, **kw):
        e = L.Epilogue()
        e.kind = kind
        for k, v in kw.items():
            setattr(e, k, v)
        return e

    resid = lambda: ep(L.EPI_RESID, out=o0.data_ptr(), out2=xout.data_ptr(), aux=xres.data_ptr(), gate=gate.data_ptr(), ldg=6 * D,
                       rows_per_sample=256, ldo=D, alpha=0.9, beta=0.4)
    cases = [
        ("qkv NT store", 0, M, 3 * D, D, x, D, w_qkv, D, ep(L.EPI_STORE_BF16, out=o0.data_ptr(), ldo=3 * D, alpha=1.0)),
        ("qkv NT heads", 0, M, 3 * D, D, x, D, w_qkv, D,
         ep(L.EPI_QKV_HEADS, out=o0.data_ptr(), out2=o1.data_ptr(), out3=o2.data_ptr(), out4=scales.data_ptr(), rows_per_sample=256, alpha=1.0)),
        ("fc1 NT silugrad", 0, M, 4 * D, D, x, D, w_fc1, D, ep(L.EPI_SILU2_GRAD, out=o0.data_ptr(), out2=o1.data_ptr(), ldo=4 * D)),
        ("fc2 NT resid", 0, M, D, 4 * D, h, 4 * D, w_fc2, 4 * D, resid()),
        ("proj NT resid", 0, M, D, D, x, D, w_qkv, D, resid()),
        ("fc2 dX NN mulaux", 1, M, 4 * D, D, dy, D, w_fc2, 4 * D, ep(L.EPI_MUL_AUX, out=o0.data_ptr(), aux=h.data_ptr(), ldo=4 * D)),
        ("fc1 dX NN store", 1, M, D, 4 * D, dh, 4 * D, w_fc1, D, ep(L.EPI_STORE_BF16, out=o0.data_ptr(), ldo=D, alpha=1.0)),
        ("fc1 dW TN split16", 2, 4 * D, D, M, dh, 4 * D, x, D,
         ep(L.EPI_STORE_F32, out=out_f32.data_ptr(), ldo=D, alpha=1.0, split_k=16, slab_stride=4 * D * D)),
        # the engine's own launches of the weight gradients (pick_split_k: one round of the chip) and plain long-K launches of the other
        # two layouts beside them (round 5)
        ("fc1 dW TN split7", 2, 4 * D, D, M, dh, 4 * D, x, D,
         ep(L.EPI_STORE_F32, out=out_f32.data_ptr(), ldo=D, alpha=1.0, split_k=7, slab_stride=4 * D * D)),
        ("fc2 dW TN split7", 2, D, 4 * D, M, dy, D, h, 4 * D,
         ep(L.EPI_STORE_F32, out=out_f32.data_ptr(), ldo=4 * D, alpha=1.0, split_k=7, slab_stride=4 * D * D)),
        ("qkv dW TN split9", 2, 3 * D, D, M, h, 4 * D, x, D,
         ep(L.EPI_STORE_F32, out=out_f32.data_ptr(), ldo=D, alpha=1.0, split_k=9, slab_stride=3 * D * D)),
        ("fc2 NT store K3072", 0, M, D, 4 * D, h, 4 * D, w_fc2, 4 * D, ep(L.EPI_STORE_BF16, out=o0.data_ptr(), ldo=D, alpha=1.0)),
        ("fc1 dX NN store K3072", 1, M, D, 4 * D, dh, 4 * D, w_fc1, D, ep(L.EPI_STORE_BF16, out=o0.data_ptr(), ldo=D, alpha=1.0)),
        ("fc1 NT store K768", 0, M, 4 * D, D, x, D, w_fc1, D, ep(L.EPI_STORE_BF16, out=o0.data_ptr(), ldo=4 * D, alpha=1.0)),
    ]
    if args.cases:
        cases = [c for c in cases if any(k in c[0] for k in args.cases.split(","))]
    st = torch.cuda.current_stream().cuda_stream
    lib.mapdit_gemm_tuning(256, 6 if args.force_w else 2, 0)
    print(f"{'case':20s} {'us':>7s} {'TF/s':>6s} | per tile, wave 0: {'wait':>6s} {'kloop':>6s} {'issue':>6s} {'epi':>6s} {'total':>6s} | wave 4: "
          f"{'wait':>6s} {'kloop':>6s} {'issue':>6s} {'epi':>6s}   tiles/wg")
    for name, layout, m, n, k, a, lda, b, ldb, e in cases:
        for _ in range(30):
            rc = lib.mapdit_gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
            assert rc == 0, lib.mapdit_last_error()
        torch.cuda.synchronize()
        rec.zero_()
        if args.cold:
            trash = torch.empty(1 << 28, device=dev)
            ms = 0.0
            for _ in range(10):
                trash.fill_(1.0)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                lib.mapdit_gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
                ev1.record()
                torch.cuda.synchronize()
                ms += ev0.elapsed_time(ev1) / 10
        else:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(10):
                lib.mapdit_gemm_bf16(layout, m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, C.byref(e), st)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / 10
        # (with --cold the records hold the last launch only: each launch overwrites them)
        r = rec.cpu().view(-1, 16).double()
        r = r[r[:, 4] > 0]
        tiles = r[:, 4].mean().item()
        w0 = (r[:, 0:4].sum(0) / r[:, 4].sum()).tolist()
        w4 = (r[:, 8:12].sum(0) / r[:, 12].sum()).tolist()
        print(f"{name:20s} {ms * 1e3:7.1f} {2.0 * m * n * k / ms / 1e9:6.0f} |                   {w0[0]:6.0f} {w0[1]:6.0f} {w0[2]:6.0f} {w0[3]:6.0f} {sum(w0):6.0f} |         "
              f"{w4[0]:6.0f} {w4[1]:6.0f} {w4[2]:6.0f} {w4[3]:6.0f}   {tiles:.1f}"
              f"   K-tiles/tile {k / 64 / max(e.split_k, 1):.1f}: {w0[1] / (k / 64 / max(e.split_k, 1)):.0f} cycles per K-tile (2,048 of MFMA issue)")


if __name__ == "__main__":
    main()
