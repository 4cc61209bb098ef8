#!/usr/bin/env python3
"""Headline benchmark: latent-images/sec of one full MaP-DiT training step (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = fresh synthetic batch (latents, labels, timesteps: drawn on the device) + label drop + q_sample + DiT-B/2 forward
(incl. the forced-weight-norm pass that re-images the bf16 weights) + MSE/vb loss + full backward + gradient reduction over
ranks (RCCL) + Adam + LR schedule + 2 EMA copies.

Scaling.  BASELINE.json's configuration is a GLOBAL batch of 256 per node, sharded over the GPUs (per-GPU batch 256/N;
SURVEY.md §8e): that is the default (`"scaling": "strong"`).  `--scaling weak` keeps 256 samples per GPU instead.

Prints ONE JSON line on rank 0.  `value` = samples of all ranks / wall time of the K timed steps (barrier + synchronize on both
sides, max over ranks); `ms_per_step_median` is the median over per-step HIP-event intervals of the same K steps.  `roofline` is
for the dominant kernel (the block-MLP fc1 GEMM), timed with HIP events on its launch stream inside the timed region.  `parity`
is the error of THIS engine (the timed precision) against the reference's own outputs for the same model (tests/golden fixture).
`cpu_baseline` times the CPU oracle (oracle/) on a bounded sample of the same model on the host cores — a baseline, not a target.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0          # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
FIXTURE_FOR = {"DiT-B/2": "b2_n2", "DiT-S/2": "s2_n4", "DiT-S/4": "s4_n8", "DiT-XL/2": "xl2_n2"}


def fwd_flops_per_sample(depth, D, T, P):
    """Matmul-only forward FLOPs per sample (BASELINE.md §3 / SURVEY.md §8d)."""
    return depth * (24 * T * D * D + 4 * T * T * D + 12 * D * D) + 2 * T * (P + 1) * D + 2 * T * D * 2 * P \
        + 2 * (256 * D + D * D) + 4 * D * D + 32 * D


def physical_cores():
    """(physical cores, logical CPUs) of the host, from /proc/cpuinfo."""
    seen = set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return (len(seen) or None), os.cpu_count()


def cpu_steps(model_name: str, batch: int, steps: int):
    """`steps` full training steps of the CPU oracle (own fp32 PyTorch-eager restatement, pinned to the reference by
    tests/golden) on the host: the checker, timed as the reference's CPU path.  Returns latent-img/s and seconds per step."""
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

    def step(i):
        x = torch.randn(batch, 4, 32, 32, generator=g)
        y = torch.randint(0, 1000, (batch,), generator=g)
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
    return batch * steps / dt, dt / steps


def pick_cpu_threads(logical):
    """torch's default (one thread per logical CPU: 128 on the GPU box) is ~5x slower than the best setting for a batch of 8 -
    the eager oracle's small ops drown in synchronisation.  One short sweep on DiT-S/4 (batch 8, 2 steps each) picks the count
    the baseline then runs with; the sweep is reported."""
    sweep = {}
    for n in (8, 16, 32, 64):
        if logical and n > logical:
            break
        torch.set_num_threads(n)
        sweep[n] = cpu_steps("DiT-S/4", 8, 2)[1]
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    return best, {str(k): round(v, 3) for k, v in sweep.items()}


def cpu_baseline(model_name: str, batch: int, steps: int, c1_steps: int):
    phys, logical = physical_cores()
    default_threads = torch.get_num_threads()
    threads, sweep = pick_cpu_threads(logical)
    val, sps = cpu_steps(model_name, batch, steps)
    out = {"value": val, "unit": "latent-img/s", "cores": threads, "kind": "port",
           "host": {"physical_cores": phys, "logical_cpus": logical, "torch_threads": threads, "torch_default_threads": default_threads,
                    "thread_sweep_s_per_step_S4_batch8": sweep, "torch": torch.__version__},
           "sample": f"{model_name}, batch {batch}, {steps} full steps (fwd+loss+bwd+Adam+2 EMA), fp32 eager oracle, "
                     f"{sps:.2f} s/step"}
    if c1_steps > 0:        # BASELINE.json configs[0]: the reference's own CPU-runnable case (SURVEY.md §8d "C1")
        v1, s1 = cpu_steps("DiT-S/4", 8, c1_steps)
        out["configs0"] = {"value": v1, "unit": "latent-img/s",
                           "sample": f"DiT-S/4, batch 8, {c1_steps} full steps, fp32 eager oracle, {s1:.3f} s/step"}
    return out


def pmc_traffic(model, batch):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled per the guide's
    gfx950 correction + WRITE_SIZE).  The counters cannot be collected from inside this process; the committed figure holds for
    the kernel SOURCE it was taken on (sha256 of csrc/gemm.hip recorded next to it), the model and the per-GPU batch: anything
    else reports null rather than a stale number."""
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, "map-dit_amd", "csrc", "gemm.hip")
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None
    for name in ("r05_fc1_pmc_traffic.json", "r04_fc1_pmc_traffic.json", "r03_fc1_pmc_traffic.json", "r02_fc1_pmc_traffic.json", "r01_fc1_pmc_traffic.json"):
        path = os.path.join(here, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            d = json.load(f)
        if d.get("model", "DiT-B/2") != model or d.get("per_gpu_batch", 256) != batch:
            continue
        if d.get("gemm_hip_sha256") != sha:
            return {"traffic": None, "traffic_note": f"profiles/{name} was collected on another build of csrc/gemm.hip"}
        return {"traffic": d["hbm_bytes_per_launch"], "traffic_unit": "bytes/launch",
                "traffic_algorithmic": d["algorithmic_bytes_per_launch"], "traffic_source": f"profiles/{name}"}
    return {"traffic": None}


def parity_leg(model_name, precision, dev):
    """Error of the timed engine against the REFERENCE's outputs on the committed fixture of this model (tests/golden/*.npz,
    generated by running /root/reference in the build container): eval logits, per-sample training loss and parameter gradients
    (norm-wise relative error over the sub-sampled gradient entries the fixture keeps, all tensors pooled; and the worst tensor).
    The oracle is used only to regenerate the fixture's seeded weights."""
    import numpy as np
    fx = FIXTURE_FOR.get(model_name)
    path = os.path.join(ROOT, "tests", "golden", f"{fx}.npz") if fx else None
    if not path or not os.path.exists(path):
        return None
    from oracle import dit_oracle as O                     # checker: seeded weights of the fixture
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.src.models import DIT_MODELS
    g = dict(np.load(path, allow_pickle=False))
    cfg = O.DiTConfig(**{k[4:]: g[k].item() for k in g if k.startswith("cfg_")})
    gains = g["gains"].item()
    sd = O.init_state_dict(cfg, seed=int(g["wseed"]), gains=None if gains < 0 else gains, perturb_reference=float(g["perturb"]))
    m = DIT_MODELS[model_name](in_channels=4, input_size=32, num_classes=1000)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    m.gemm_precision = precision
    x, t, y, y_eff, noise = (torch.from_numpy(g[k]).to(dev) for k in ("x", "t", "y", "y_eff", "noise"))

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

    with torch.no_grad():
        out = m(x, t, y)
    ref = g["eval_out"]
    got = out.cpu().numpy()
    if got.shape != ref.shape:
        got = got.reshape(-1)[::7] if got.size > 20000 else got.reshape(-1)
    res = {"fixture": f"tests/golden/{fx}.npz", "logits_rel": rel(got, ref)}
    m.train()
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels        # the fixture's labels already carry the drop
    losses = create_diffusion("").training_losses(m, x, t, dict(y=y_eff), noise=noise)
    losses["loss"].mean().backward()
    torch.cuda.synchronize()
    res["loss_rel"] = rel(losses["loss"].detach().cpu().numpy(), g["train_loss"])
    num = den = 0.0
    worst, worst_k = 0.0, ""
    stride = 7 if "postw/x_embedder.weight" in g else 4099      # tests/golden/make_golden.py: STRIDE / BIG_STRIDE
    for k, p in m.named_parameters():
        gref = g["grad/" + k].astype(np.float64)
        if p.dim() == 0 or gref.size < 64:
            continue
        f = p.grad.detach().reshape(-1)
        mine = (f if f.numel() <= 20000 else f[::stride]).double().cpu().numpy()
        if mine.shape != gref.shape:
            continue
        d2, r2 = float(((mine - gref) ** 2).sum()), float((gref ** 2).sum())
        num, den = num + d2, den + r2
        e = (d2 / (r2 + 1e-60)) ** 0.5
        if e > worst and r2 > 1e-14:
            worst, worst_k = e, k
    res["grad_rel"] = (num / (den + 1e-60)) ** 0.5
    res["grad_rel_worst"] = worst
    res["grad_rel_worst_tensor"] = worst_k
    del m
    torch.cuda.empty_cache()
    lim = PARITY_LIMITS.get(precision)
    if lim:
        res["limits"] = lim
        res["within_limits"] = all(res[k] <= v for k, v in lim.items())
    return res


# Stated limits of the timed engines against the reference's outputs (the same numbers tests/test_model_gpu.py::
# test_named_models_match_reference asserts per precision; f16's logits limit is north_star's 1e-3).  A timed engine outside its limits
# makes bench.py exit non-zero AFTER printing its line: a fast step with wrong results is not a result.
PARITY_LIMITS = {
    "bf16": {"logits_rel": 1.6e-2, "loss_rel": 6e-3, "grad_rel": 8e-3, "grad_rel_worst": 4e-2},
    "f16": {"logits_rel": 1e-3, "loss_rel": 5e-4, "grad_rel": 1.5e-3, "grad_rel_worst": 3e-3},
    "bf16x3": {"logits_rel": 1e-3, "loss_rel": 1e-4, "grad_rel": 2e-4, "grad_rel_worst": 2e-4},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="DiT-B/2")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: --global-batch per node, sharded over the GPUs (BASELINE.json); weak: --global-batch per GPU")
    ap.add_argument("--global-batch", type=int, default=256)
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="override the per-GPU batch directly")
    ap.add_argument("--precision", choices=["bf16", "f16", "bf16x3"], default="bf16",
                    help="bf16: BASELINE.json's metric (the headline).  f16: the same engine with IEEE fp16 operands (also timed as the "
                         "`f16` object of the default run).  bf16x3: the fp32-accurate parity engine (informational)")
    ap.add_argument("--no-f16-leg", action="store_true", help="default bf16 run: skip the extra timed fp16 leg")
    ap.add_argument("--grad-comm", choices=["allreduce", "zero1", "zero1w", "zero1w-bf16"], default=None,
                    help="gradient exchange under data parallelism (default: $MAPDIT_GRAD_COMM or allreduce); zero1w: sharded weight "
                         "passes (forced WN + imaging, Jacobian, Adam/EMA on 1/world of the rows), -bf16: with a 16-bit gradient exchange")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="ONE process does what rank 0 of a W-rank zero1w job would do, WITHOUT the collectives: the one-GPU measurement of "
                         "what sharding the batch-independent weight passes saves (use with --batch-per-gpu 256/W).  Not a training run.")
    ap.add_argument("--rotation-modulation", action="store_true",
                    help="BASELINE config 3's block conditioning (README.md:1-3; not in the reference snapshot: parity unpinned)")
    ap.add_argument("--mp-off", default="", help="comma-separated off forms of the README's --use-* flags: mp_silu, mp_residual, mp_pos_enc, "
                                                 "mp_embedding, weight_normalization, cosine_attention, no_layernorm (README.md:57-66; not in the "
                                                 "reference snapshot: parity unpinned)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=10)          # ~10 s of host work at ~1 s per DiT-B/2 batch-8 step with the swept thread count
    ap.add_argument("--cpu-c1-steps", type=int, default=10)
    args = ap.parse_args()

    import torch.distributed as dist
    import mapdit_amd
    from mapdit_amd import _lib as L
    from mapdit_amd import parallel
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.optim import FusedAdamEMA, create_lr_lambda
    from mapdit_amd.src.models import DIT_MODELS

    rank, world, local = parallel.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product has no CPU path)"
    if os.environ.get("MAPDIT_FORCE_DEVICE"):              # rehearsal of the N > 1 path on a one-GPU box (gloo backend)
        local = int(os.environ["MAPDIT_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.batch_per_gpu is not None:
        B = args.batch_per_gpu
    elif args.scaling == "strong":
        lo, hi = parallel.shard_batch(args.global_batch, rank, world)
        B = hi - lo
    else:
        B = args.global_batch
    global_batch = B * world

    mp_off = [f.strip() for f in args.mp_off.split(",") if f.strip()]
    assert all(f in ("mp_silu", "mp_residual", "mp_pos_enc", "mp_embedding", "weight_normalization", "cosine_attention", "no_layernorm")
               for f in mp_off), f"--mp-off: unknown flag in {mp_off}"
    unpinned = bool(args.rotation_modulation or mp_off)      # configurations with no reference code behind them

    def timed_run(precision, steps, warmup):
        """Builds the model in `precision`, runs `warmup` untimed + `steps` timed training steps; returns the measurements."""
        parity = None
        if rank == 0 and not args.no_parity and not unpinned:
            parity = parity_leg(args.model, precision, dev)
        torch.manual_seed(0)                               # model seed 0 on every rank: identical replicas, no broadcast needed
        mkw = dict(rotation_modulation=True) if args.rotation_modulation else {}
        mkw.update({f: False for f in mp_off})
        model = DIT_MODELS[args.model](in_channels=4, input_size=32, num_classes=1000, **mkw).to(dev).train()
        model.gemm_precision = precision
        torch.manual_seed(1000 + rank)                     # from here on every rank draws its OWN timesteps, noise and label drops
        diffusion = create_diffusion(timestep_respacing="")
        num_steps = 400_000                                # train.py defaults -> warm-up / decay points
        if args.emulate_world > 1:
            assert world == 1, "--emulate-world is a one-process measurement"
            reducer = parallel.ShardedPassReducer(model, emulate=(0, args.emulate_world))
        else:
            reducer = parallel.make_reducer(model, args.grad_comm)
        opt = FusedAdamEMA(model, lr=1e-2, betas=(0.9, 0.99), ema_stds=(0.05, 0.1),
                           lr_lambda=create_lr_lambda(num_steps // 150, num_steps // 10), grad_scale=reducer.grad_scale)
        reducer.attach(opt)
        g = torch.Generator(device=dev).manual_seed(1 + rank)  # data seed 1 + rank

        def step():
            x = torch.randn(B, 4, 32, 32, device=dev, generator=g)              # a fresh batch every step
            y = torch.randint(0, 1000, (B,), device=dev, generator=g)
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

        for _ in range(warmup):
            step()
        fence()
        rt = model._rt[True if precision == "bf16" else (precision, True)]
        # every FOURTH fc1 launch is bracketed by HIP events on the launch stream (an event costs the queue a ~6 us bubble: all 12 per step
        # were 0.14 ms of measurement inside the measured step); 3 launches per step x steps is still hundreds of samples
        L.lib().engine_profile_begin_strided(rt.handle, L.PROF_FC1_FWD, model.depth * steps // 4 + 8, 4)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            loss = step()
            marks[i + 1].record()                          # on the compute stream; no host synchronisation inside the region
        fence()
        elapsed = time.perf_counter() - t0
        cnt, tot_ms = C.c_int(0), C.c_double(0.0)
        L.lib().engine_profile_end(rt.handle, C.byref(cnt), C.byref(tot_ms))
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        if world > 1:
            tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        final_loss = float(loss.item())
        model.check_device_errors()
        geo = dict(T=(model.input_size // model.patch_size) ** 2, D=model.hidden_size, Hm=model.blocks[0].mlp.hidden_dim,
                   P=model.patch_size ** 2 * model.in_channels, depth=model.depth)
        res = dict(parity=parity, elapsed=elapsed, per_step=per_step, fc1_count=cnt.value, fc1_ms=tot_ms.value, final_loss=final_loss,
                   grad_comm=reducer.name, geo=geo)
        del model, opt, reducer, rt
        torch.cuda.empty_cache()
        return res

    r = timed_run(args.precision, args.steps, args.warmup)
    parity, elapsed, per_step, final_loss = r["parity"], r["elapsed"], r["per_step"], r["final_loss"]
    geo = r["geo"]

    T, D, Hm, P = geo["T"], geo["D"], geo["Hm"], geo["P"]
    f_fwd = fwd_flops_per_sample(geo["depth"], D, T, P)
    value = world * B * args.steps / elapsed
    fc1_flops = 2.0 * (B * T) * Hm * D
    fc1_ms = r["fc1_ms"] / max(r["fc1_count"], 1)
    achieved = fc1_flops / (fc1_ms * 1e-3) / 1e12 if r["fc1_count"] else None
    at = f"@{global_batch}" if args.scaling == "strong" else f"@{B}/GPU"
    out = {
        "metric": f"latent-images/sec training step, {args.model} {args.precision}{' rotation-modulation' if args.rotation_modulation else ''}"
                  f"{' off:' + '+'.join(mp_off) if mp_off else ''} {at}",
        "value": value, "unit": "latent-img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_step_median": statistics.median(per_step),
        "ms_per_step_min": min(per_step), "ms_per_step_max": max(per_step),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": {"bf16": "bf16", "f16": "f16", "bf16x3": "bf16x3 (two-term split bf16 operands, fp32 accumulate and storage)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": f"{args.model} full training step on 32x32x4 latents (fresh batch+fwd+loss+bwd+grad reduction+Adam+"
                               f"2xEMA), {'magnitude-preserving features OFF: ' + ', '.join(mp_off) + ' (README off forms, parity unpinned)' if mp_off else 'all magnitude-preserving features on'}, "
                               f"{'fp16' if args.precision == 'f16' else 'bf16'} GEMM operands / "
                               "fp32 accumulate, master and residual fp32",
                   "global_batch": global_batch, "per_gpu_batch": B, "tokens_per_sample": T, "parallelism": f"dp{world}",
                   "grad_comm": r["grad_comm"],
                   **({"emulated_world": args.emulate_world,
                       "note": f"ONE process doing rank 0's share of a {args.emulate_world}-rank zero1w job without the collectives: the compute side of "
                               "sharded weight passes, not a training run and not a multi-GPU measurement"} if args.emulate_world > 1 else {}),
                   "seeds": {"model": 0, "data": "1+rank", "t/noise/drop": "1000+rank"}, "final_loss": final_loss},
        "step_mfma_frac": value * 3 * f_fwd / (world * PEAK_BF16_DENSE_TFLOPS * 1e12),
        "parity": parity if not unpinned else {
            "pinned": False,
            "note": "off forms of the README's --use-* flags (README.md:57-66) are not in the reference snapshot, which hard-wires every flag "
                    "on (SURVEY F5): there is no reference output to compare with.  The engine is held to this repo's own restatement "
                    "(oracle.dit_oracle.DiTConfig; tests/test_mp_flags_gpu.py): parity unpinned"} if mp_off else {
            "pinned": False,
            "note": "rotation modulation is described in the reference's README but absent from its code snapshot (SURVEY F6): there "
                    "is no reference output to compare with.  The engine is held to this repo's own restatement of the README "
                    "(oracle.dit_oracle.modulate_rot; tests/test_rotation.py: logits 4.7e-3 bf16 / 4.8e-4 f16 at DiT-B/2 size): parity unpinned"},
        "roofline": {"bound": "mfma", "kernel": "block-MLP fc1 GEMM with the SiLU + derivative epilogue (NT: "
                                                  f"[{B * T},{D}]x[{Hm},{D}]^T)",
                     "achieved": achieved, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                     "frac": (achieved / PEAK_BF16_DENSE_TFLOPS) if achieved else None, **pmc_traffic(args.model, B),
                     "launches_timed": r["fc1_count"], "avg_launch_ms": fc1_ms, "flops_per_launch": fc1_flops},
    }
    # The same step with IEEE fp16 operands (gemm_precision = "f16": same kernels, same MFMA rate, 10 mantissa bits): the path
    # whose forward logits are inside north_star's 1e-3 of the reference.  The headline above stays BASELINE.json's bf16.
    if args.precision == "bf16" and world == 1 and not args.no_f16_leg and not unpinned:
        h = timed_run("f16", args.steps, max(3, args.warmup // 2))
        h_ms = h["fc1_ms"] / max(h["fc1_count"], 1)
        out["f16"] = {"value": world * B * args.steps / h["elapsed"], "unit": "latent-img/s", "ms_per_step": 1e3 * h["elapsed"] / args.steps,
                      "ms_per_step_median": statistics.median(h["per_step"]), "steps": args.steps, "parity": h["parity"],
                      "final_loss": h["final_loss"], "fc1_avg_launch_ms": h_ms,
                      "fc1_tflops": fc1_flops / (h_ms * 1e-3) / 1e12 if h["fc1_count"] else None,
                      "note": "same engine, IEEE fp16 GEMM / attention operands, static power-of-two loss scale"}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not unpinned:
            out["cpu_baseline"] = cpu_baseline(args.model, args.cpu_batch, args.cpu_steps, args.cpu_c1_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    bad = [name for name, pr in ((args.precision, out.get("parity")), ("f16", (out.get("f16") or {}).get("parity")))
           if isinstance(pr, dict) and pr.get("within_limits") is False]
    if rank == 0 and bad:
        print(f"bench.py: parity of the timed engine(s) {bad} is outside the stated limits (see the `parity` objects)", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
