"""Data parallelism for the training step: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference is single-process; its semantics under DP are those of one large batch: the loss is
the mean over the global batch (train.py:91), so parameter gradients are averaged over ranks.  Samples are
independent through forward/backward — the only exchange is the gradient sum.  Parameters, Adam state and EMA
copies are replicated; forced weight normalisation is a deterministic function of the weights, so replicas stay
bit-identical as long as every rank applies the same reduced gradient (RCCL all-reduce returns identical bits
on every rank).

Gradients live in ONE flat fp32 buffer (``model._gflat``), so the reduction is a handful of large collectives
instead of one per tensor.  The buffer is reduced in ``n_buckets`` contiguous slices launched back to back on a
side stream so the tail of one bucket's ring overlaps the head of the next; the optimiser waits on the stream.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Rendezvous from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def bucket_slices(numel: int, n_buckets: int, align: int = 1024):
    """Contiguous, aligned slices covering [0, numel)."""
    n_buckets = max(1, n_buckets)
    per = (numel + n_buckets - 1) // n_buckets
    per = (per + align - 1) // align * align
    out, lo = [], 0
    while lo < numel:
        hi = min(numel, lo + per)
        out.append((lo, hi))
        lo = hi
    return out


class GradReducer:
    """Sum-all-reduce of a flat gradient buffer across the data-parallel group (mean is applied by the optimiser's
    gradient scale = 1/world).  Works with any torch.distributed backend: RCCL on GPUs, gloo on CPU for tests."""

    def __init__(self, group=None, n_buckets: int = 4):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_buckets = n_buckets
        self._stream = None

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def reduce(self, flat: torch.Tensor):
        if self.world == 1:
            return
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                for lo, hi in bucket_slices(flat.numel(), self.n_buckets):
                    dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)
        else:
            for lo, hi in bucket_slices(flat.numel(), self.n_buckets):
                dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)


def shard_batch(global_batch: int, rank: int, world: int):
    """[lo, hi) of this rank's samples when a global batch is split evenly (strong scaling); raises on ragged splits."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per
