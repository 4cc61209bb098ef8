"""MI355X counterpart of the reference's ``sample.py``: four class-conditional samples with classifier-free guidance
from the post-hoc EMA weights of a results directory, written as an image grid.  Same flags and defaults
(reference sample.py:83-96) plus ``--vae-path`` / ``--no-graph`` / ``--precision``.  The VAE decoder is not part of this
engine: with ``--use-vae false`` the de-normalised latents are written (PNG grid of the 4 latent channels
as RGBA + ``<output>.npy``)."""
from __future__ import annotations

import argparse

import numpy as np
import torch

from . import sampling as S
from .diffusion import create_diffusion
from .train import get_model


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--result-dir", type=str, required=True)
    p.add_argument("--use-vae", type=S.str2bool, default=True)
    p.add_argument("--output-file", type=str, default="sample.png")
    p.add_argument("--class-label", type=int, default=88)
    p.add_argument("--cfg-scale", type=float, default=4.0)
    p.add_argument("--num-sampling-steps", type=int, default=250)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--ema-std", type=float, default=0.05)
    p.add_argument("--ckpt", type=str, default=None, help="Checkpoint to load instead of EMA (should not include .pt extension).")
    p.add_argument("--vae-path", type=str, default=None, help="local copy of stabilityai/sd-vae-ft-mse (no network here)")
    p.add_argument("--no-graph", action="store_true", help="eager p_sample_loop instead of the captured hipGraph")
    p.add_argument("--precision", choices=["bf16", "f16", "bf16x3"], default="f16")
    return p


@torch.no_grad()          # the reference switches autograd off globally (torch.set_grad_enabled(False)); scoped here
def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    device = torch.device("cuda")
    train_args = S.load_train_args(args.result_dir)
    model = get_model(train_args).to(device)
    S.load_weights(model, args.result_dir, args.ema_std, args.ckpt, verbose=True)
    model.gemm_precision = args.precision
    vae = S.load_vae(args.vae_path, device) if args.use_vae else None

    n = 4                                                                  # sample.py:40
    z = torch.randn(n, train_args["in_channels"], train_args["input_size"], train_args["input_size"], device=device)
    if not 0 <= args.class_label < int(train_args["num_classes"]):       # the reference fails inside F.embedding (IndexError)
        raise ValueError(f"--class-label {args.class_label} is outside the trained model's {train_args['num_classes']} classes")
    y = torch.tensor([args.class_label] * n, device=device)
    z = torch.cat([z, z], dim=0)                                          # CFG batch: conditional | null class
    y = torch.cat([y, torch.tensor([train_args["num_classes"]] * n, device=device)], dim=0)
    diffusion = create_diffusion(str(args.num_sampling_steps))
    samples = S.run_sampler(model, diffusion, z, y, args.cfg_scale, use_graph=not args.no_graph, progress=True)
    samples, _ = samples.chunk(2, dim=0)
    samples = S.denormalize(samples, train_args)
    if vae is not None:
        samples = vae.decode(samples).sample.cpu()
    else:
        np.save(args.output_file + ".npy", samples.cpu().numpy())
    samples = samples.clamp(-1, 1)
    S.save_image_grid(samples, args.output_file, nrow=2, value_range=(-1, 1))
    print(f"output class: {args.class_label}")
    return samples


if __name__ == "__main__":
    main()
