#!/usr/bin/env python3
"""Headline benchmark: latent-images/sec of one full MaP-DiT training step (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = label drop + q_sample + DiT-B/2 forward (incl. the forced-weight-norm pass that re-images the bf16
weights) + MSE/vb loss + full backward + gradient sum over ranks (RCCL) + Adam + LR schedule + 2 EMA copies,
on a synthetic batch of 32x32x4 latents resident in HBM.  Weak scaling: the per-GPU batch is fixed (256).
Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the block-MLP fc1 GEMM), timed with HIP
events on its launch stream inside the timed region; `cpu_baseline` times the CPU oracle (oracle/) on a bounded
sample of the same model on the host cores — a reported baseline, not the target.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0          # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA


def fwd_flops_per_sample(depth, D, T, P):
    """Matmul-only forward FLOPs per sample (BASELINE.md §3 / SURVEY.md §8d)."""
    return depth * (24 * T * D * D + 4 * T * T * D + 12 * D * D) + 2 * T * (P + 1) * D + 2 * T * D * 2 * P \
        + 2 * (256 * D + D * D) + 4 * D * D + 32 * D


def cpu_baseline(model_name: str, batch: int, steps: int):
    """The CPU oracle (own fp32 PyTorch-eager restatement, pinned to the reference by tests/golden) timed on the host."""
    from oracle import dit_oracle as O
    from oracle.diffusion_oracle import DiffusionOracle
    cfg = O.model_config(model_name, in_channels=4, input_size=32, num_classes=1000)
    sd = O.init_state_dict(cfg, seed=0)
    params = [k for k in sd if k not in O.BUFFER_KEYS]
    m = {k: torch.zeros_like(sd[k]) for k in params}
    v = {k: torch.zeros_like(sd[k]) for k in params}
    emas = [{k: sd[k].clone() for k in params} for _ in range(2)]
    d = DiffusionOracle("")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(batch, 4, 32, 32, generator=g)
    y = torch.randint(0, 1000, (batch,), generator=g)

    def step(i):
        t = torch.randint(0, 1000, (batch,), generator=g)
        drop = torch.rand(batch, generator=g) < 0.1
        leaf = {k: (sd[k].requires_grad_(True) if k in m else sd[k]) for k in sd}
        loss = d.training_losses(lambda xx, tt, **kw: O.dit_forward(leaf, cfg, xx, tt, kw["y"], train=True, drop=drop),
                                 x, t, dict(y=y))["loss"].mean()
        loss.backward()
        with torch.no_grad():
            for k in params:
                p = sd[k].detach()
                O.adam_step(p, sd[k].grad, m[k], v[k], i, lr=1e-2)
                for e, s in zip(emas, (0.05, 0.1)):
                    e[k].lerp_(p, O.ema_beta(s, i))
                sd[k] = p
    step(1)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i + 2)
    dt = time.perf_counter() - t0
    return {"value": batch * steps / dt, "unit": "latent-img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{model_name}, batch {batch}, {steps} full steps (fwd+loss+bwd+Adam+2 EMA), fp32 eager oracle, "
                      f"torch {torch.__version__}, {dt / steps:.2f} s/step"}


def pmc_traffic(model, batch):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled per the
    guide's gfx950 correction + WRITE_SIZE; profiles/r01_fc1_pmc_traffic.json).  The counters cannot be collected from
    inside this process; they hold for the kernel and shape they were taken on (DiT-B/2, 256 samples) and are null otherwise."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_fc1_pmc_traffic.json")
    if model != "DiT-B/2" or batch != 256 or not os.path.exists(path):
        return {"traffic": None}
    with open(path) as f:
        d = json.load(f)
    return {"traffic": d["hbm_bytes_per_launch"], "traffic_unit": "bytes/launch",
            "traffic_algorithmic": d["algorithmic_bytes_per_launch"], "traffic_source": "profiles/r01_fc1_pmc_traffic.json"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="DiT-B/2")
    ap.add_argument("--batch-per-gpu", type=int, default=256)
    ap.add_argument("--precision", choices=["bf16", "bf16x3"], default="bf16",
                    help="bf16x3: the fp32-accurate parity engine (informational; the headline metric is bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=2)
    args = ap.parse_args()

    import torch.distributed as dist
    import mapdit_amd
    from mapdit_amd import _lib as L
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.optim import FusedAdamEMA, create_lr_lambda
    from mapdit_amd.parallel import OverlappedGradReducer, init_from_env
    from mapdit_amd.src.models import DIT_MODELS

    rank, world, local = init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product has no CPU path)"
    if os.environ.get("MAPDIT_FORCE_DEVICE"):              # rehearsal of the N > 1 path on a one-GPU box (gloo backend)
        local = int(os.environ["MAPDIT_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(0)                                   # model seed 0
    model = DIT_MODELS[args.model](in_channels=4, input_size=32, num_classes=1000).to(dev).train()
    model.gemm_precision = args.precision
    diffusion = create_diffusion(timestep_respacing="")
    num_steps = 400_000                                    # train.py defaults -> warm-up / decay points
    reducer = OverlappedGradReducer(model)                 # all-reduce of block i overlaps backward of blocks i-1..0
    opt = FusedAdamEMA(model, lr=1e-2, betas=(0.9, 0.99), ema_stds=(0.05, 0.1),
                       lr_lambda=create_lr_lambda(num_steps // 150, num_steps // 10), grad_scale=reducer.grad_scale)
    B = args.batch_per_gpu
    g = torch.Generator(device=dev).manual_seed(1 + rank)  # data seed 1 + rank
    x = torch.randn(B, 4, 32, 32, device=dev, generator=g)
    y = torch.randint(0, 1000, (B,), device=dev, generator=g)

    def step():
        t = torch.randint(0, diffusion.num_timesteps, (B,), device=dev)
        loss = diffusion.training_losses(model, x, t, dict(y=y))["loss"].mean()
        opt.zero_grad()
        loss.backward()
        reducer.finish()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    rt = model._rt[True if args.precision == "bf16" else (args.precision, True)]
    L.lib().engine_profile_begin(rt.handle, L.PROF_FC1_FWD, model.depth * args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    cnt, tot_ms = C.c_int(0), C.c_double(0.0)
    L.lib().engine_profile_end(rt.handle, C.byref(cnt), C.byref(tot_ms))
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    final_loss = float(loss.item())

    T = (model.input_size // model.patch_size) ** 2
    D, Hm = model.hidden_size, model.blocks[0].mlp.hidden_dim
    P = model.patch_size ** 2 * model.in_channels
    f_fwd = fwd_flops_per_sample(model.depth, D, T, P)
    value = world * B * args.steps / elapsed
    fc1_flops = 2.0 * (B * T) * Hm * D
    fc1_ms = tot_ms.value / max(cnt.value, 1)
    achieved = fc1_flops / (fc1_ms * 1e-3) / 1e12 if cnt.value else None
    out = {
        "metric": f"latent-images/sec training step, {args.model} {args.precision} @256",
        "value": value, "unit": "latent-img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.precision == "bf16" else "bf16x3 (two-term split bf16 operands, fp32 accumulate and storage)", "data": "synthetic",
        "config": {"workload": f"{args.model} full training step on 32x32x4 latents (fwd+loss+bwd+allreduce+Adam+2xEMA), "
                               "all magnitude-preserving features on, bf16 GEMM operands / fp32 accumulate, master and "
                               "residual fp32",
                   "global_batch": world * B, "per_gpu_batch": B, "tokens_per_sample": T, "parallelism": f"dp{world}",
                   "seeds": {"model": 0, "data": "1+rank"}, "final_loss": final_loss},
        "step_mfma_frac": value * 3 * f_fwd / (world * PEAK_BF16_DENSE_TFLOPS * 1e12),
        "roofline": {"bound": "mfma", "kernel": "gemm_mfma256_kernel<0, 0, EpiSilu2Grad> (NT, block-MLP fc1: "
                                                  f"[{B * T},{D}]x[{Hm},{D}]^T)",
                     "achieved": achieved, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                     "frac": (achieved / PEAK_BF16_DENSE_TFLOPS) if achieved else None, **pmc_traffic(args.model, B),
                     "launches_timed": cnt.value, "avg_launch_ms": fc1_ms, "flops_per_launch": fc1_flops},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, args.cpu_batch, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
