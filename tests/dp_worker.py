"""Worker of tests/test_train_gpu.py::test_two_rank_data_parallel_*: one data-parallel rank of a short training run.

    python -m torch.distributed.run --nproc-per-node W ... tests/dp_worker.py OUT_DIR MODE PRECISION STEPS

Every rank builds the same seeded model, takes ITS shard of a fixed global batch (same tensors on every rank, generated from a
seed), runs STEPS optimiser steps through the reducer MODE ("allreduce" | "zero1") and writes its final flat parameter buffer,
the reduced gradient buffer of the last step and (after gather_state) the Adam / EMA state to OUT_DIR/rank{r}.pt.
With WORLD_SIZE=1 the same script is the single-process run on the whole batch.  Collectives run on gloo when
MAPDIT_DIST_BACKEND=gloo (ranks share one GPU on a one-GPU box); with MAPDIT_DP_DEVICE_PER_RANK=1 every rank takes GPU LOCAL_RANK
(RCCL over xGMI: tests/test_train_gpu.py::test_two_gpu_rccl_data_parallel, on boxes with more than one GPU)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, mode, precision, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    import mapdit_amd  # noqa: F401
    from mapdit_amd import parallel
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.optim import FusedAdamEMA
    from mapdit_amd.src.models import DIT_MODELS
    rank, world, _ = parallel.init_from_env()
    local = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("MAPDIT_DP_DEVICE_PER_RANK") else 0     # RCCL: one GPU per rank
    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    if os.environ.get("MAPDIT_TEST_POISON"):               # tools/determinism_check.py: stale allocator memory of a known pattern
        junk = torch.full((1 << 28,), int(os.environ["MAPDIT_TEST_POISON"]), dtype=torch.uint8, device=dev)
        del junk
    torch.manual_seed(21)
    import json
    kw = json.loads(os.environ.get("MAPDIT_TEST_MODEL_KW", "{}"))        # e.g. off forms of the README flags: {"weight_normalization": false}
    dit_kw = json.loads(os.environ.get("MAPDIT_TEST_DIT_KW", "{}"))      # a model outside the named ones, e.g. {"depth": 2, "hidden_size": 768, "num_heads": 12}
    if dit_kw:
        from mapdit_amd.src.dit import DiT
        m = DiT(patch_size=2, in_channels=4, input_size=32, num_classes=7, **dit_kw, **kw).to(dev).train()
    else:
        m = DIT_MODELS["DiT-XS/2"](in_channels=4, input_size=32, num_classes=7, **kw).to(dev).train()
    m.gemm_precision = precision
    if precision == "f16":
        m.loss_scale = 1024.0                               # (the automatic choice depends on the per-rank batch: fixed, a sample's bits do not)
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels
    with torch.no_grad():                                   # non-trivial gains so that every gradient path is alive
        for k, p in m.named_parameters():
            if "gain_" in k:
                p.fill_(0.2)
    red = parallel.make_reducer(m, mode)
    opt = FusedAdamEMA(m, lr=1e-2, betas=(0.9, 0.99), ema_stds=(0.05, 0.1), grad_scale=red.grad_scale)
    red.attach(opt)
    diff = create_diffusion("")
    g = torch.Generator().manual_seed(22)
    n = int(os.environ.get("MAPDIT_TEST_BATCH", "16"))
    lo, hi = parallel.shard_batch(n, rank, world)
    # MAPDIT_TEST_OVERFLOW="rank:step": that rank's loss is blown up in that step (fp16: its gradients overflow) - the non-finite
    # guard must refuse the step on EVERY rank
    bad_rank, bad_step = (int(v) for v in os.environ.get("MAPDIT_TEST_OVERFLOW", "-1:-1").split(":"))
    for it in range(steps):
        x, y = torch.randn(n, 4, 32, 32, generator=g), torch.randint(0, 7, (n,), generator=g)
        t, noise = torch.randint(0, 1000, (n,), generator=g), torch.randn(n, 4, 32, 32, generator=g)
        x, y, t, noise = (v[lo:hi].to(dev) for v in (x, y, t, noise))
        loss = diff.training_losses(m, x, t, dict(y=y), noise=noise)["loss"].mean()
        if rank == bad_rank and it == bad_step:
            loss = loss * 1e12
        opt.zero_grad()
        loss.backward()
        red.finish()
        opt.step()
    if mode.startswith("zero1w") and world > 1:              # stale foreign rows: an inference forward must be refused until the state is gathered
        from mapdit_amd._lib import MapditError
        m.eval()                                            # (eval: no forced rewrite of the masters - the saved state stays comparable)
        try:
            with torch.no_grad():
                m(x, t, y)
            raise AssertionError("an inference forward on stale sharded masters was not refused")
        except MapditError:
            pass
    red.gather_state()
    if mode.startswith("zero1w") and world > 1:
        with torch.no_grad():
            assert torch.isfinite(m(x, t, y)).all()
        m.train()
    torch.cuda.synchronize()
    parts = getattr(opt, "shards", None)
    torch.save({"p": m._pflat.cpu(), "g": m._gflat.cpu() * red.grad_scale, "m": opt.exp_avg.cpu(), "v": opt.exp_avg_sq.cpu(),
                "e0": opt.ema[0].cpu(), "e1": opt.ema[1].cpu(), "shards": parts, "loss": float(loss), "refused": opt.overflow_steps()},
               os.path.join(out_dir, f"rank{rank}.pt"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
