"""MI355X counterpart of the reference's ``sample_fid.py``: ``--num-samples`` images in batches of ``--batch-size`` with
random labels (classifier-free guidance when ``--cfg-scale`` > 1), stored as one uint8 NHWC array ``arr_0`` in
``<result-dir>/fid_samples/<output-file>`` (reference sample_fid.py:48-97; flags :100-114).  One hipGraph is captured for
the batch shape and replayed for every batch.  Without a VAE (``--use-vae false``) the array holds the de-normalised
latents clamped to [-1, 1] and quantised the same way."""
from __future__ import annotations

import argparse
import math
import os

import numpy as np
import torch

from . import sampling as S
from .diffusion import create_diffusion
from .train import get_model


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--result-dir", type=str, required=True)
    p.add_argument("--use-vae", type=S.str2bool, default=True)
    p.add_argument("--cfg-scale", type=float, default=1.5)
    p.add_argument("--num-classes", type=int, default=None,
                   help="default: the trained model's num_classes from config.yaml (the reference defaults to 1000 whatever "
                        "the model was trained with; a mismatch would index past its label table)")
    p.add_argument("--num-samples", type=int, default=10_000)
    p.add_argument("--batch-size", type=int, default=128)
    p.add_argument("--num-sampling-steps", type=int, default=250)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--output-file", type=str, default="samples.npz", help="Filename in which to store samples.")
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
    if args.num_classes is None:
        args.num_classes = int(train_args["num_classes"])
    elif args.num_classes != int(train_args["num_classes"]):
        raise ValueError(f"--num-classes {args.num_classes} != the trained model's num_classes {train_args['num_classes']} "
                         "(config.yaml): labels and the null label would fall outside its embedding table")
    model = get_model(train_args).to(device)
    S.load_weights(model, args.result_dir, args.ema_std, args.ckpt, verbose=False)
    model.gemm_precision = args.precision
    vae = S.load_vae(args.vae_path, device) if args.use_vae else None
    diffusion = create_diffusion(str(args.num_sampling_steps))

    n = args.batch_size
    use_cfg = args.cfg_scale > 1.0
    shape = (2 * n if use_cfg else n, train_args["in_channels"], train_args["input_size"], train_args["input_size"])
    graphed = None
    gathered = []
    for _ in range(math.ceil(args.num_samples / n)):
        z = torch.randn(n, *shape[1:], device=device)
        y = torch.randint(0, args.num_classes, (n,), device=device)
        if use_cfg:                                                       # sample_fid.py:56-62
            z = torch.cat([z, z], dim=0)
            y = torch.cat([y, torch.tensor([args.num_classes] * n, device=device)], dim=0)
        if args.no_graph:
            samples = S.run_sampler(model, diffusion, z, y, args.cfg_scale if use_cfg else None, use_graph=False)
        else:
            if graphed is None:
                graphed = S.GraphedSampler(model, diffusion, shape, y, args.cfg_scale if use_cfg else None)
            graphed.y.copy_(y)
            samples = graphed.sample(z)
        if use_cfg:
            samples, _ = samples.chunk(2, dim=0)
        samples = S.denormalize(samples, train_args)
        if vae is not None:
            samples = vae.decode(samples).sample
        gathered.append(S.to_uint8_nhwc(samples))
    out = np.concatenate(gathered, axis=0)[: args.num_samples]
    os.makedirs(os.path.join(args.result_dir, "fid_samples"), exist_ok=True)
    path = os.path.join(args.result_dir, "fid_samples", args.output_file)
    np.savez(path, arr_0=out)
    return path


if __name__ == "__main__":
    main()
