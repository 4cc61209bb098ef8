#!/usr/bin/env python3
"""Sampling throughput (BASELINE config 5 shape): p_sample_loop with classifier-free guidance, eager vs hipGraph replay.

    python tools/sample_bench.py --model DiT-XL/2 --n 128 --steps 50
Prints one JSON line: denoise steps/s, ms per step, latent images/s for a full 250-step chain.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapdit_amd  # noqa: E402
from mapdit_amd.diffusion import create_diffusion  # noqa: E402
from mapdit_amd.sampling import GraphedSampler  # noqa: E402
from mapdit_amd.src.models import DIT_MODELS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="DiT-XL/2")
    ap.add_argument("--n", type=int, default=128, help="images per batch (the CFG batch is 2n rows)")
    ap.add_argument("--steps", type=int, default=50, help="timed denoise steps (of the 250-step schedule)")
    ap.add_argument("--cfg-scale", type=float, default=1.5)
    ap.add_argument("--precision", choices=["bf16", "f16"], default="bf16")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = DIT_MODELS[args.model](in_channels=4, input_size=32, num_classes=1000).to(dev).eval().requires_grad_(False)
    model.gemm_precision = args.precision
    d = create_diffusion("250")
    n = args.n
    z = torch.randn(n, 4, 32, 32, device=dev)
    z = torch.cat([z, z], 0)
    y = torch.cat([torch.randint(0, 1000, (n,), device=dev), torch.full((n,), 1000, device=dev)])
    kw = dict(y=y, cfg_scale=args.cfg_scale)

    # eager reference-API loop (p_sample per step)
    img = z
    with torch.no_grad():
        for i in (249, 248):
            img = d.p_sample(model.forward_with_cfg, img, torch.full((2 * n,), i, device=dev), clip_denoised=False, model_kwargs=kw)["sample"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img = z
        for k in range(args.steps):
            img = d.p_sample(model.forward_with_cfg, img, torch.full((2 * n,), 249 - k, device=dev), clip_denoised=False, model_kwargs=kw)["sample"]
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / args.steps

    s = GraphedSampler(model, d, z.shape, y, cfg_scale=args.cfg_scale)
    s.sample(z, steps=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.sample(z, steps=args.steps)
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / args.steps
    T = (model.input_size // model.patch_size) ** 2
    D, L = model.hidden_size, model.depth
    P = model.patch_size ** 2 * 4
    f_fwd = L * (24 * T * D * D + 4 * T * T * D + 12 * D * D) + 2 * T * (P + 1) * D + 2 * T * D * 2 * P + 2 * (256 * D + D * D) + 4 * D * D + 32 * D
    print(json.dumps({"metric": f"p_sample_loop denoise step, {args.model}, cfg {args.cfg_scale}, batch 2x{n}",
                      "ms_per_step_eager": 1e3 * eager, "ms_per_step_hipgraph": 1e3 * graphed,
                      "steps_per_s_hipgraph": 1 / graphed, "images_per_s_250_steps": n / (250 * graphed),
                      "fwd_tflops_per_s": 2 * n * f_fwd / graphed / 1e12, "dtype": args.precision}))


if __name__ == "__main__":
    main()
