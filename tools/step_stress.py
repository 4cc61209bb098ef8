#!/usr/bin/env python3
"""One training forward + backward of DiT-XS/2 (8 samples, fixed inputs, weights restored every time) repeated while other
processes run the same loop on the same GPU: every repetition must reproduce the first gradient buffer bit for bit.
    python tools/step_stress.py [reps] [concurrent processes]"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapdit_amd  # noqa: E402,F401
from mapdit_amd.diffusion import create_diffusion  # noqa: E402
from mapdit_amd.src.models import DIT_MODELS  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nproc = int(sys.argv[2]) if len(sys.argv) > 2 else 2
child = os.environ.get("STEP_STRESS_CHILD")
kids = [] if child else [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(10 ** 9), "1"],
                                          env=dict(os.environ, STEP_STRESS_CHILD="1")) for _ in range(nproc - 1)]
try:
    dev = torch.device("cuda", 0)
    torch.manual_seed(21)
    m = DIT_MODELS["DiT-XS/2"](in_channels=4, input_size=32, num_classes=7).to(dev).train()
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "gain_" in k:
                p.fill_(0.2)
    p0 = m._pflat.detach().clone()
    g = torch.Generator().manual_seed(22)
    n = 8
    x, y = torch.randn(n, 4, 32, 32, generator=g).to(dev), torch.randint(0, 7, (n,), generator=g).to(dev)
    t, noise = torch.randint(0, 1000, (n,), generator=g).to(dev), torch.randn(n, 4, 32, 32, generator=g).to(dev)
    diff = create_diffusion("")
    base = m._pflat.data_ptr()
    layout = sorted(((p.data_ptr() - base) // 4, k, tuple(p.shape)) for k, p in m.named_parameters())

    def where(idx):
        out = {}
        for i in idx.tolist():
            for off, k, sh in layout:
                cnt = 1
                for s in sh:
                    cnt *= s
                if off <= i < off + cnt:
                    c = sh[-1] if sh else 1
                    out[(k, (i - off) // c)] = out.get((k, (i - off) // c), 0) + 1
                    break
        return ", ".join(f"{k} row {r} ({c} el)" for (k, r), c in list(out.items())[:5]) + (" ..." if len(out) > 5 else "")

    ref, bad = None, 0
    if not child and kids:
        import time
        time.sleep(20)                                       # let the other processes get onto the GPU
    for it in range(reps):
        with torch.no_grad():
            m._pflat.copy_(p0)
        if m._gflat is not None:
            m._gflat.zero_()
        loss = diff.training_losses(m, x, t, dict(y=y), noise=noise)["loss"].mean()
        loss.backward()
        torch.cuda.synchronize()
        gcur = m._gflat.detach().clone()
        if ref is None:
            ref = gcur
        elif not torch.equal(gcur, ref):
            bad += 1
            idx = torch.nonzero(gcur != ref).flatten()
            if not child and bad <= 12:
                print(f"rep {it}: {idx.numel()} gradient elements differ: {where(idx)}")
                # anatomy of the first differing row: dW = a1 G - a2 W, so a wrong a2 (the row's dot product G.W) shows as a
                # difference proportional to the row of W, a wrong piece of G as a localised difference
                i0 = int(idx[0])
                for off, k, sh in layout:
                    cnt = 1
                    for s_ in sh:
                        cnt *= s_
                    if off <= i0 < off + cnt and len(sh) == 2:
                        r = (i0 - off) // sh[1]
                        lo = off + r * sh[1]
                        d = (gcur[lo:lo + sh[1]] - ref[lo:lo + sh[1]]).double()
                        w = m._pflat[lo:lo + sh[1]].double()
                        coef = float((d @ w) / (w @ w))
                        res = d - coef * w
                        nz = torch.nonzero(res.abs() > 1e-3 * d.abs().max()).flatten()
                        print(f"      {k} row {r}: |diff| max {float(d.abs().max()):.3e} (|grad| max {float(ref[lo:lo + sh[1]].abs().max()):.3e}); "
                              f"diff = {coef:+.3e} x W row + residual of norm {float(res.norm() / d.norm()):.3f} x |diff|; residual lives in "
                              f"columns {nz[:1].tolist()}..{nz[-1:].tolist()} ({nz.numel()} of {sh[1]})")
                        break
    if not child:
        print(f"{reps} repetitions with {nproc} processes on the GPU: {bad} differing gradient buffers")
finally:
    for k in kids:
        k.kill()
