#!/usr/bin/env python3
"""Training harness with the reference's CLI (train.py:19-141, 225-246) on the MI355X engine.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m mapdit_amd.train \
        --data-path /data/latents --results-dir /results --model DiT-B/2
    python -m mapdit_amd.train --synthetic --results-dir /tmp/res --model DiT-S/2 --num-steps 100

Same arguments, defaults, schedule derivations, experiment directory layout (NNN-DiT-X-p/, config.yaml, log.txt,
checkpoints/NNNNNNN.pt with {"model", "opt"}, ema/{std:.3f}_{t:07d}.pt fp16 snapshots) and log line as the reference.
Added: --synthetic (N(0,1) latents, no dataset), the README's --use-* flags (accepted; the snapshot hard-wires all of
them on, SURVEY F5 — turning one off is refused), and data parallelism when launched under torchrun (global batch =
--batch-size, sharded evenly; gradients summed over RCCL and averaged).
"""
import argparse
import math
import os
from glob import glob
from time import time

import torch
import yaml

from . import parallel
from .diffusion import create_diffusion
from .optim import FusedAdamEMA, create_lr_lambda
from .src.models import DIT_MODELS

MP_FLAGS = ["cosine-attention", "weight-normalization", "forced-weight-normalization", "mp-residual", "mp-silu",
            "no-layernorm", "mp-pos-enc", "mp-embedding"]          # reference README.md:59-66


# Off forms this engine builds besides --no-use-forced-weight-normalization (DiT(..., mp_silu=False) etc.; src/dit.py).  The other
# three (--no-use-cosine-attention, --no-use-weight-normalization, --no-use-no-layernorm) name layers the snapshot does not contain
# (plain SDPA with learnt scale, biased nn.Linear, LayerNorm) and are refused.
BUILT_OFF_FORMS = ["mp-residual", "mp-silu", "mp-pos-enc", "mp-embedding", "weight-normalization", "cosine-attention", "no-layernorm"]


def get_model(args):
    """reference utils.py:9-17."""
    a = vars(args) if isinstance(args, argparse.Namespace) else args
    kw = dict(rotation_modulation=True) if a.get("use_rotation_modulation") else {}
    if not a.get("use_forced_weight_normalization", True):
        kw["forced_weight_normalization"] = False
    for flag in BUILT_OFF_FORMS:                    # off forms that are built (parity unpinned: README lines, no reference code)
        if not a.get("use_" + flag.replace("-", "_"), True):
            kw[flag.replace("-", "_")] = False
    return DIT_MODELS[a["model"]](in_channels=a["in_channels"], input_size=a["input_size"], num_classes=a["num_classes"], **kw)


def setup_experiment(model_name, results_dir):
    """reference train.py:200-214."""
    os.makedirs(results_dir, exist_ok=True)
    idx = len(glob(os.path.join(results_dir, "*")))
    exp = os.path.join(results_dir, f"{idx:03d}-{model_name.replace('/', '-')}")
    os.makedirs(os.path.join(exp, "checkpoints"), exist_ok=True)
    return exp


class LatentDataset(torch.utils.data.Dataset):
    """reference train.py:144-176: posterior_means.pt / posterior_stds.pt / labels.pt / stats.pt; a sample is
    (mean + randn * std) standardised per channel."""

    def __init__(self, path):
        ld = lambda n: torch.load(os.path.join(path, n), weights_only=True)
        self.means, self.stds, self.labels, self.stats = ld("posterior_means.pt"), ld("posterior_stds.pt"), ld("labels.pt"), ld("stats.pt")
        self.m = torch.as_tensor(self.stats["mean"]).view(-1, 1, 1)
        self.s = torch.as_tensor(self.stats["std"]).view(-1, 1, 1)
        assert self.means.shape[0] == self.labels.shape[0] == self.stds.shape[0]

    channels = property(lambda self: self.means.shape[1])
    data_size = property(lambda self: self.means.shape[2])

    def __len__(self):
        return self.means.shape[0]

    def __getitem__(self, i):
        f = self.means[i] + torch.randn_like(self.means[i]) * self.stds[i]
        return (f - self.m) / self.s, self.labels[i]


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--data-path", type=str, default=None)
    p.add_argument("--synthetic", action="store_true", help="N(0,1) 4x32x32 latents and uniform labels instead of a dataset")
    p.add_argument("--results-dir", type=str, required=True)
    p.add_argument("--model", type=str, choices=list(DIT_MODELS.keys()), default="DiT-XS/2")
    p.add_argument("--num-classes", type=int, default=1000)
    p.add_argument("--num-steps", type=int, default=400_000)
    p.add_argument("--batch-size", type=int, default=256)
    p.add_argument("--lr", type=float, default=1e-2)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--verbose", type=int, choices=[0, 1, 2], default=1)
    p.add_argument("--num-workers", type=int, default=4)
    p.add_argument("--log-every", type=int, default=100)
    p.add_argument("--ckpt-every", type=int, default=50_000)
    p.add_argument("--num-lin-warmup", type=int, default=None)
    p.add_argument("--start-decay", type=int, default=None)
    p.add_argument("--ema-snapshot-every", type=int, default=None)
    p.add_argument("--precision", choices=["bf16", "f16", "bf16x3"], default="f16",
                   help="GEMM operand precision: f16 (default: IEEE fp16 operands, logits within 1e-3 of the fp32 reference; power-of-two "
                        "loss scale, steps with non-finite gradients are refused and the scale halved), bf16 (the same engine with bf16 "
                        "operands: 2 %% faster, ~6e-3) or bf16x3 (fp32-accurate forward and backward, the reference's numerics to ~1e-5, "
                        "several times slower)")
    p.add_argument("--grad-comm", choices=["allreduce", "zero1", "zero1w", "zero1w-bf16"], default=None,
                   help="data-parallel gradient exchange: per-stage all-reduce overlapped with backward (default), or "
                        "reduce-scatter + sharded optimiser + all-gather (zero1), or sharded weight passes - forced weight norm, imaging, "
                        "Jacobian, Adam / EMA on 1/world of the rows, 16-bit images all-gathered (zero1w; -bf16: 16-bit gradient exchange)")
    for f in MP_FLAGS:
        p.add_argument(f"--use-{f}", dest="use_" + f.replace("-", "_"), action=argparse.BooleanOptionalAction, default=True)
    p.add_argument("--use-rotation-modulation", action=argparse.BooleanOptionalAction, default=False,
                   help="block conditioning by rotation modulation (reference README.md:1-3; not in its code snapshot: this build's "
                        "own restatement, parity unpinned): ~5.4 %% fewer parameters; bf16 and f16 precisions")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    # --no-use-forced-weight-normalization is the one off-path the snapshot's code defines (skip the in-place rewrite); the other seven are
    # built as restatements of their README lines (BUILT_OFF_FORMS, parity unpinned: the snapshot hard-wires every feature on)
    off = [f for f in MP_FLAGS if not getattr(args, "use_" + f.replace("-", "_"))
           and f != "forced-weight-normalization" and f not in BUILT_OFF_FORMS]
    if off:
        raise NotImplementedError(f"--no-use-{off[0]}: no off form built; built: --no-use-forced-weight-normalization, "
                                  + ", ".join("--no-use-" + f for f in BUILT_OFF_FORMS))
    if args.precision == "bf16x3" and any(not getattr(args, "use_" + f.replace("-", "_")) for f in BUILT_OFF_FORMS):
        raise NotImplementedError("the --no-use-* off forms are built for the f16 / bf16 engines, not for --precision bf16x3")
    if not args.use_no_layernorm and args.use_rotation_modulation:
        raise NotImplementedError("--no-use-no-layernorm (the LayerNorm form) is built for the AdaLN modulation, not with --use-rotation-modulation")
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(args.seed)
    lo, hi = parallel.shard_batch(int(args.batch_size), rank, world)
    per_rank = hi - lo

    if args.synthetic:
        args.in_channels, args.input_size = 4, 32
        args.stats_mean, args.stats_std = [0.0] * 4, [1.0] * 4
        loader = None
    else:
        assert args.data_path, "--data-path or --synthetic"
        ds = LatentDataset(args.data_path)
        sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=True, drop_last=True) if world > 1 else None
        loader = torch.utils.data.DataLoader(ds, batch_size=per_rank, num_workers=args.num_workers, shuffle=sampler is None,
                                             sampler=sampler, pin_memory=True, drop_last=True)
        args.in_channels, args.input_size = ds.channels, ds.data_size
        args.stats_std = [float(v) for v in ds.stats["std"]]
        args.stats_mean = [float(v) for v in ds.stats["mean"]]

    exp = None
    if rank == 0:
        exp = setup_experiment(args.model, args.results_dir)
        with open(os.path.join(exp, "config.yaml"), "w") as f:
            yaml.dump(vars(args), f)
        os.makedirs(os.path.join(exp, "ema"), exist_ok=True)
        logf = open(os.path.join(exp, "log.txt"), "a")

    def log(msg):
        if rank == 0 and args.verbose:
            print(msg, flush=True)
            logf.write(msg + "\n")
            logf.flush()

    diffusion = create_diffusion(timestep_respacing="")
    model = get_model(args).to(dev).train()
    model.gemm_precision = args.precision
    log(f"model parameters: {sum(p.numel() for p in model.parameters() if p.requires_grad):,}")
    if args.ema_snapshot_every is None:
        args.ema_snapshot_every = args.num_steps // 250
    if args.num_lin_warmup is None:
        args.num_lin_warmup = args.num_steps // 150
    if args.start_decay is None:
        args.start_decay = args.num_steps // 10
    # Identical parameters on every rank come from the shared seed above (no broadcast needed).  From here on each rank must draw
    # its OWN timesteps, noise and label drops - the reference's single process draws them independently for all 256 samples
    # (train.py:86; gaussian_diffusion.py:735; label_embedder.py:23) - so the default generators are re-seeded per rank.
    torch.manual_seed(args.seed + 1000003 * (rank + 1))
    reducer = parallel.make_reducer(model, args.grad_comm)
    opt = FusedAdamEMA(model, lr=args.lr, betas=(0.9, 0.99), ema_stds=(0.05, 0.1),
                       lr_lambda=create_lr_lambda(args.num_lin_warmup, args.start_decay), grad_scale=reducer.grad_scale)
    reducer.attach(opt)

    def batches():
        if loader is None:
            g = torch.Generator(device=dev).manual_seed(args.seed + 1 + rank)
            while True:
                yield (torch.randn(per_rank, 4, 32, 32, device=dev, generator=g),
                       torch.randint(0, args.num_classes, (per_rank,), device=dev, generator=g))
        epoch = 0
        while True:
            if world > 1:
                loader.sampler.set_epoch(epoch)
            for x, y in loader:
                yield x.to(dev, non_blocking=True), y.to(dev, non_blocking=True)
            epoch += 1

    train_steps, log_steps, running, start = 0, 0, torch.zeros((), device=dev), time()
    log(f"training for {args.num_steps} steps...")
    for x, y in batches():
        t = torch.randint(0, diffusion.num_timesteps, (x.shape[0],), device=dev)
        loss = diffusion.training_losses(model, x, t, dict(y=y))["loss"].mean()
        opt.zero_grad()
        loss.backward()
        reducer.finish()
        opt.step()
        running += loss.detach()
        log_steps += 1
        train_steps += 1
        if train_steps % args.log_every == 0:
            model.check_device_errors()          # an out-of-range label / timestep raises here (the reference: IndexError)
            refused = opt.poll_overflow()        # fp16: steps refused for non-finite gradients since the last log line
            if refused:
                log(f"(step={train_steps:07d}) {refused} optimiser step(s) refused: non-finite gradients; fp16 loss scale now "
                    f"{model.loss_scale or model.effective_loss_scale():g}")
            avg = running / log_steps
            if world > 1:
                torch.distributed.all_reduce(avg)
                avg /= world
            sps = log_steps / (time() - start)
            log(f"(step={train_steps:07d}) train loss: {avg.item():.4f}, train steps/sec: {sps:.2f}")
            running.zero_()
            log_steps, start = 0, time()
        if train_steps % args.ckpt_every == 0 or (args.ema_snapshot_every and train_steps % args.ema_snapshot_every == 0):
            reducer.gather_state()               # ZeRO-1: Adam moments / EMA copies are sharded; complete them (collective)
        if rank == 0 and train_steps % args.ckpt_every == 0:
            torch.save({"model": model.state_dict(), "opt": opt.state_dict()},                 # reference train.py:125-132
                       os.path.join(exp, "checkpoints", f"{train_steps:07d}.pt"))
        if rank == 0 and args.ema_snapshot_every and train_steps % args.ema_snapshot_every == 0:
            for std in opt.ema_stds:                                   # reference src/ema.py:143-155
                sd = {k: v.cpu().half() for k, v in opt.ema_state_dict(std).items()}
                torch.save({"std": std, "t": train_steps, "state_dict": sd}, os.path.join(exp, "ema", f"{std:.3f}_{train_steps:07d}.pt"))
        if train_steps >= args.num_steps:
            break
    log("done!")
    return exp


if __name__ == "__main__":
    main()
