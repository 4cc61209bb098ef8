"""Data parallelism for the training step: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference is single-process; its semantics under DP are those of one large batch: the loss is
the mean over the global batch (train.py:91), so parameter gradients are averaged over ranks.  Samples are
independent through forward/backward — the only exchange is the gradient sum.  Parameters, Adam state and EMA
copies are replicated; forced weight normalisation is a deterministic function of the weights, so replicas stay
bit-identical as long as every rank applies the same reduced gradient (an all-reduce returns identical bits on
every rank).

Gradients live in ONE flat fp32 buffer (``model._gflat``), laid out in module registration order, so each DiT
block owns one contiguous slice.  ``OverlappedGradReducer`` hooks the engine's staged backward
(``mapdit_engine_backward_stages``): as soon as a stage's kernels are enqueued, the slice it finalised is handed to an
asynchronous all-reduce (RCCL runs it on its own stream behind the compute already queued), so the reduction of block
i overlaps the backward of blocks i-1 ... 0 — the usual DDP bucketing, with the engine's stages as buckets.
``GradReducer`` is the plain variant (reduce everything after backward).  The mean is applied by the optimiser's
gradient scale 1/world.
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
            backend = os.environ.get("MAPDIT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


def _world(group):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class GradReducer:
    """Sum-all-reduce of a flat gradient buffer after backward, in a few large buckets.  Works with any
    torch.distributed backend: RCCL on GPUs, gloo on CPU for tests."""

    def __init__(self, group=None, n_buckets: int = 4):
        self.group = group
        self.world = _world(group)
        self.n_buckets = n_buckets

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def reduce(self, flat: torch.Tensor):
        if self.world == 1:
            return
        works = [dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for lo, hi in bucket_slices(flat.numel(), self.n_buckets)]
        for w in works:
            w.wait()


def stage_slices(model):
    """[lo, hi) of the flat parameter/gradient buffer owned by each backward stage of the engine:
    stage 0 = final_layer, stage k = blocks[depth-k], stage depth+1 = the embedders (x, t, y)."""
    L = model.depth
    ranges = {}
    params = list(model.named_parameters())
    for (name, p), off in zip(params, model._poffs):
        if name.startswith("final_layer."):
            st = 0
        elif name.startswith("blocks."):
            st = L - int(name.split(".")[1])
        else:
            st = L + 1
        size = (p.numel() + 31) // 32 * 32
        lo, hi = ranges.get(st, (off, off))
        assert off == hi or st not in ranges, f"parameters of stage {st} are not contiguous in the flat buffer"
        ranges[st] = (min(lo, off), off + size)
    total = model._pflat.numel()
    out = [ranges[s] for s in range(L + 2)]
    assert sorted(out)[0][0] == 0 and sorted(out)[-1][1] == total
    assert sum(hi - lo for lo, hi in out) == total
    return out


class OverlappedGradReducer:
    """All-reduce each backward stage's gradient slice as soon as that stage is enqueued (see module docstring).
    Usage: r = OverlappedGradReducer(model); ...; loss.backward(); r.finish(); opt.step()."""

    def __init__(self, model, group=None, force_collective: bool = False):
        self.model, self.group = model, group
        self.world = _world(group)
        self.slices = stage_slices(model)
        self.works = []
        # force_collective: issue the all-reduces in a one-rank group too (an identity through RCCL: rehearses the stream hand-over
        # between the engine's kernels and the collective on a one-GPU box)
        self.force_collective = force_collective
        staged = self.world > 1 or force_collective or os.environ.get("MAPDIT_FORCE_STAGED_BACKWARD")
        model._stage_hook = self._on_stage if staged else None

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def _on_stage(self, stage: int):
        if self.world == 1 and not self.force_collective:
            return
        lo, hi = self.slices[stage]
        self.works.append(dist.all_reduce(self.model._gflat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Make the current stream wait for every outstanding reduction (call before the optimiser step)."""
        for w in self.works:
            w.wait()
        self.works.clear()

    def reduce(self, flat=None):          # same call site as GradReducer
        self.finish()


def shard_batch(global_batch: int, rank: int, world: int):
    """[lo, hi) of this rank's samples when a global batch is split evenly (strong scaling); raises on ragged splits."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per
