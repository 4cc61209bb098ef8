#!/usr/bin/env python3
"""Training trajectories of the engine precisions on identical seeds: the bf16 and fp16-operand fast paths against bf16x3 (which
follows the reference's fp32 run to ~5e-6 over optimiser steps, tests/test_train_gpu.py).  Same model init, data, timesteps,
label drops and schedule as map-dit_amd/train.py; prints the per-step losses of both runs and their relative gap.

    python tools/precision_trajectory.py --model DiT-S/2 --batch 64 --steps 200
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args, precision):
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.optim import FusedAdamEMA, create_lr_lambda
    from mapdit_amd.src.models import DIT_MODELS
    dev = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    model = DIT_MODELS[args.model](in_channels=4, input_size=32, num_classes=1000).to(dev).train()
    model.gemm_precision = precision
    diffusion = create_diffusion(timestep_respacing="")
    opt = FusedAdamEMA(model, lr=args.lr, betas=(0.9, 0.99), ema_stds=(0.05, 0.1),
                       lr_lambda=create_lr_lambda(max(args.steps // 150, 1), max(args.steps // 10, 1)))
    g = torch.Generator(device=dev).manual_seed(args.seed + 1)
    losses = []
    for _ in range(args.steps):
        x = torch.randn(args.batch, 4, 32, 32, device=dev, generator=g)
        y = torch.randint(0, 1000, (args.batch,), device=dev, generator=g)
        t = torch.randint(0, diffusion.num_timesteps, (args.batch,), device=dev)
        loss = diffusion.training_losses(model, x, t, dict(y=y))["loss"].mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    return torch.stack(losses).double().cpu(), model._pflat.detach().double().cpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="DiT-S/2")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    fast, w_fast = run(args, "bf16")
    again, _ = run(args, "bf16")
    half, w_half = run(args, "f16")
    exact, w_exact = run(args, "bf16x3")
    print(f"{args.model}, batch {args.batch}, {args.steps} steps, lr {args.lr}: per-step mean loss, bf16 | f16 | bf16x3 | relative gaps to bf16x3")
    for i in list(range(0, min(10, args.steps))) + list(range(10, args.steps, max(args.steps // 20, 1))):
        print(f"  step {i + 1:5d}  {fast[i]:.6f}  {half[i]:.6f}  {exact[i]:.6f}  {abs(fast[i] - exact[i]) / exact[i]:.2e}  {abs(half[i] - exact[i]) / exact[i]:.2e}")
    k = max(args.steps // 10, 1)
    for name, run_, w in (("bf16", fast, w_fast), ("f16", half, w_half)):
        rel = ((run_ - exact).abs() / exact)
        print(f"{name}: max gap first 10 steps {rel[:10].max():.2e}; all steps {rel.max():.2e}; mean of last {k}: {run_[-k:].mean():.5f} "
              f"(bf16x3 {exact[-k:].mean():.5f}); all losses finite = {bool(torch.isfinite(run_).all())}; "
              f"weights after {args.steps} steps: |w - w_bf16x3| / |w_bf16x3| = {(w - w_exact).norm() / w_exact.norm():.3e}")
    print(f"bf16 run repeated: bit-identical losses = {bool(torch.equal(fast, again))}")


if __name__ == "__main__":
    main()
