"""MI355X counterpart of the reference's ``sample_ema.py``: the same 8 noise draws sampled under post-hoc EMA profiles of
relative width 0.0075 / 0.01 / 0.05 / 0.1 / 0.15, one column per width (reference sample_ema.py:24-80; flags :83-92)."""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from . import sampling as S
from .diffusion import create_diffusion
from .src.ema import calculate_posthoc_ema
from .train import get_model

EMA_STDS = [0.0075, 0.01, 0.05, 0.1, 0.15]       # sample_ema.py:25


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--result-dir", type=str, required=True)
    p.add_argument("--use-vae", type=S.str2bool, default=True)
    p.add_argument("--output-file", type=str, default="sample.png")
    p.add_argument("--class-label", type=int, default=88)
    p.add_argument("--cfg-scale", type=float, default=4.0)
    p.add_argument("--num-sampling-steps", type=int, default=250)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--vae-path", type=str, default=None, help="local copy of stabilityai/sd-vae-ft-mse (no network here)")
    p.add_argument("--no-graph", action="store_true", help="eager p_sample_loop instead of the captured hipGraph")
    p.add_argument("--precision", choices=["bf16", "f16", "bf16x3"], default="f16")
    return p


@torch.no_grad()          # the reference switches autograd off globally (torch.set_grad_enabled(False)); scoped here
def main(argv=None):
    args = build_parser().parse_args(argv)
    device = torch.device("cuda")
    train_args = S.load_train_args(args.result_dir)
    model = get_model(train_args).to(device).eval()
    model.gemm_precision = args.precision
    vae = S.load_vae(args.vae_path, device) if args.use_vae else None
    diffusion = create_diffusion(str(args.num_sampling_steps))
    n = 8
    shape = (2 * n, train_args["in_channels"], train_args["input_size"], train_args["input_size"])
    if not 0 <= args.class_label < int(train_args["num_classes"]):       # the reference fails inside F.embedding (IndexError)
        raise ValueError(f"--class-label {args.class_label} is outside the trained model's {train_args['num_classes']} classes")
    y = torch.cat([torch.tensor([args.class_label] * n), torch.tensor([train_args["num_classes"]] * n)]).to(device)
    graphed = None
    res = []
    for std in EMA_STDS:
        if args.seed is not None:
            torch.manual_seed(args.seed)                                   # same noise for every profile
        sd = calculate_posthoc_ema(std, os.path.join(args.result_dir, "ema"), verbose=False)
        model.load_state_dict({k: v.float() for k, v in sd.items()})
        z = torch.randn(n, *shape[1:], device=device)
        z = torch.cat([z, z], 0)
        if args.no_graph:
            samples = S.run_sampler(model, diffusion, z, y, args.cfg_scale, use_graph=False, progress=True)
        else:
            if graphed is None:              # weights are re-imaged per replay set: the graph reads the cached bf16 images
                graphed = S.GraphedSampler(model, diffusion, shape, y, args.cfg_scale)
            else:
                graphed.refresh_weights()
            samples = graphed.sample(z)
        samples, _ = samples.chunk(2, dim=0)
        res.append(samples)
    samples = torch.stack(res, dim=1)
    samples = samples.view(-1, *samples.shape[2:])
    samples = S.denormalize(samples, train_args)
    if vae is not None:
        samples = vae.decode(samples).sample.cpu()
    else:
        np.save(args.output_file + ".npy", samples.cpu().numpy())
    samples = samples.clamp(-1, 1)
    S.save_image_grid(samples, args.output_file, nrow=len(EMA_STDS), value_range=(-1, 1))
    print(f"output class: {args.class_label}")
    return samples


if __name__ == "__main__":
    main()
