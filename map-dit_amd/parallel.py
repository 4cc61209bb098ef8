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


def _host_staged(group, t) -> bool:
    """gloo with device tensors (the one-GPU rehearsals of the N > 1 path, MAPDIT_DIST_BACKEND=gloo): the collective is run on a
    host copy, synchronously - RCCL collectives are ordered on the compute stream and asynchronous, gloo's own CUDA staging (pool
    streams, pinned buffers, worker threads) is not something the rehearsal should depend on.  (The run-to-run differences once
    blamed on it were a compiler matter: SLP-packed fp32 math going wrong in lanes 48-63 while two processes share a GPU, see
    csrc/Makefile.)"""
    return t.is_cuda and _world(group) > 1 and dist.get_backend(group) == "gloo"


class _Done:
    def wait(self):
        return True


def _all_reduce_sum(t, group):
    """Sum-all-reduce of `t` in place; returns a work handle."""
    if _host_staged(group, t):
        h = t.detach().cpu()                    # synchronises with the stream that produced t
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        torch.cuda.synchronize(t.device)
        return _Done()
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


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
        self.works.append(_all_reduce_sum(self.model._gflat[lo:hi], self.group))

    def finish(self):
        """Make the current stream wait for every outstanding reduction (call before the optimiser step)."""
        for w in self.works:
            w.wait()
        self.works.clear()

    def reduce(self, flat=None):          # same call site as GradReducer
        self.finish()

    # The parameters and Adam moments are replicated (every rank steps all of them from the same reduced gradients); the two EMA
    # copies, which no step reads, are kept current by one rank each: rank r owns elements [r * per, (r + 1) * per) (the < 4 * world
    # elements behind the last range stay replicated).  16 of the optimiser kernel's 44 bytes per parameter on (world - 1) / world
    # of the model drop out of every step; gather_state() completes the copies before a snapshot or checkpoint reads them.
    def attach(self, opt):
        self.opt = opt
        if self.world > 1 and opt.ema:
            n = self.model._pflat.numel()
            self.per = (n // (4 * self.world)) * 4
            rank = dist.get_rank(self.group)
            opt.ema_ranges = [(rank * self.per, (rank + 1) * self.per)]
            if self.world * self.per < n:                  # the tail behind the last equal range: every rank keeps it current
                opt.ema_ranges.append((self.world * self.per, n))
            self.name = type(self).name + ", EMA copies sharded"

    def gather_state(self):
        opt = getattr(self, "opt", None)
        if opt is None or self.world == 1 or not opt.ema or opt.ema_ranges is None:
            return
        rank = dist.get_rank(self.group)
        for buf in opt.ema:
            mine = buf[rank * self.per:(rank + 1) * self.per]
            whole = buf[:self.world * self.per]
            before = _range_checksum(mine)
            if _host_staged(self.group, buf):
                parts = [torch.empty(self.per, dtype=buf.dtype) for _ in range(self.world)]
                dist.all_gather(parts, mine.detach().cpu(), group=self.group)
                whole.copy_(torch.cat(parts))
                torch.cuda.synchronize(buf.device)
            elif dist.get_backend(self.group) == "nccl":
                dist.all_gather_into_tensor(whole, mine, group=self.group)          # in place: `mine` is part `rank` of `whole`
            else:
                dist.all_gather(list(whole.chunk(self.world)), mine.clone(), group=self.group)
            _verify_gather(whole, self.per, self.world, before, self.group)


def _range_checksum(x):
    """(sum, sum of squares) of a range in float64: what its owner publishes before a gather."""
    d = x.detach().double()
    return torch.stack([d.sum(), (d * d).sum()])


def _verify_gather(whole, per, world, mine_before, group):
    """After an in-place all-gather of `world` ranges of `per` elements: every range of the gathered buffer must carry the checksum
    its owner computed BEFORE the gather.  The in-place RCCL forms (all_gather_into_tensor with the input a slice of the output)
    only ever ran with one rank before a multi-GPU node was available; a wrong offset or an aliasing problem there would corrupt EMA
    snapshots and checkpoints silently.  Cost: one read of the buffer, at snapshot / checkpoint cadence."""
    if dist.get_backend(group) == "nccl" and whole.is_cuda:
        pub = torch.empty(world, 2, dtype=torch.float64, device=whole.device)
        dist.all_gather_into_tensor(pub, mine_before.view(1, 2), group=group)
    else:
        parts = [torch.empty(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, mine_before.cpu(), group=group)
        pub = torch.stack(parts).to(whole.device)
    got = torch.stack([_range_checksum(whole[r * per:(r + 1) * per]) for r in range(world)])
    # The verdict is collective: a corruption seen by one receiver only must stop EVERY rank here - a rank that raised alone would
    # leave the others waiting in their next collective until the RCCL timeout, with the real error lost.  bad[r] = 1 where this
    # rank's copy of range r disagrees with its owner's checksum; MAX over ranks.
    bad = torch.tensor([0.0 if torch.allclose(got[r], pub[r], rtol=1e-12, atol=0.0) else 1.0 for r in range(world)])
    if dist.get_backend(group) == "nccl" and whole.is_cuda:
        bad = bad.to(whole.device)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
    if float(bad.max()) > 0:
        ranges = [r for r in range(world) if float(bad[r]) > 0]
        raise RuntimeError(f"gather_state: ranges of ranks {ranges} arrived on some rank with a different checksum than their owners "
                           f"computed (raised on every rank; this is rank {dist.get_rank(group)}): the gathered state is not to be trusted")


class _ReducerBase:
    name = "none"

    def attach(self, opt):
        """Give the reducer the optimiser whose step it partitions (ZeRO-1); a no-op for the replicated reducers."""

    def gather_state(self):
        """Make optimiser / EMA state complete on every rank before a checkpoint (collective; a no-op when replicated)."""


OverlappedGradReducer.name = "allreduce-fp32-per-stage-overlapped"
GradReducer.gather_state = _ReducerBase.gather_state
GradReducer.name = "allreduce-fp32-after-backward"
GradReducer.attach = _ReducerBase.attach


class Zero1Reducer(_ReducerBase):
    """Reduce-scatter + sharded optimiser + all-gather (ZeRO stage 1; SURVEY.md §8e "optional").

    Every backward stage's gradient slice [lo, hi) is cut into `world` equal parts; rank r receives the SUM of part r
    (``reduce_scatter``, issued as soon as the stage is enqueued, so it overlaps the backward of the earlier blocks), runs the
    fused Adam + EMA kernel on its parts only (1/world of the optimiser's 44 B/parameter HBM traffic and of its state), and the
    updated parameter parts are all-gathered back into every rank's flat parameter buffer before the next forward.  Bytes on the
    links equal those of the ring all-reduce it replaces (reduce-scatter + all-gather IS the all-reduce, with the optimiser in the
    middle).  Replicas stay bit-identical: every parameter element is computed by exactly one rank and copied.

    Adam moments and EMA copies of the parts a rank does not own are never touched: ``FusedAdamEMA.state_dict`` /
    ``ema_state_dict`` of a sharded optimiser are partial; gather with ``gather_state()`` before checkpointing."""

    name = "zero1: reduce-scatter fp32 per stage + sharded Adam/EMA + all-gather"

    def __init__(self, model, group=None, force_collective: bool = False):
        self.model, self.group = model, group
        self.world = _world(group)
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.slices = stage_slices(model)
        for lo, hi in self.slices:
            assert (hi - lo) % (4 * self.world) == 0, "stage slices must split into 4-element aligned parts"
        self.works = []
        self.force_collective = force_collective
        self.opt = None
        staged = self.world > 1 or force_collective
        model._stage_hook = self._on_stage if staged else None
        self._native_rs = self.world > 1 and dist.get_backend(group) == "nccl"

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def parts(self):
        """[(lo, hi)] element ranges of the flat buffers this rank owns, one per backward stage."""
        out = []
        for lo, hi in self.slices:
            n = (hi - lo) // self.world
            out.append((lo + self.rank * n, lo + (self.rank + 1) * n))
        return out

    def attach(self, opt):
        self.opt = opt
        if self.world > 1 or self.force_collective:
            opt.shards = self.parts()
            opt.after_step = self._gather_params
            if self.world > 1:
                # non-finite guard: a rank checks the gradients of its own parts only - every rank must refuse the step together
                # (status[0] = last bad step: MAX makes it the current step everywhere if any rank flagged it)
                opt.status_sync = self._sync_status

    def _sync_status(self, status):
        # RCCL: one 8-byte device-side all-reduce per step, nothing read back.  A host-staged group (gloo: the CPU rehearsal of the
        # N > 1 path and the two-ranks-on-one-GPU tests, never a product configuration) cannot take device pointers: there - and only
        # there - the verdict words pass through the host, which synchronises the stream once per optimiser step.  "Nothing is read
        # back in the training loop" (mapdit.h, optim.py) is a statement about the RCCL path.
        if _host_staged(self.group, status):
            h = status.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            status.copy_(h)
        else:
            dist.all_reduce(status, op=dist.ReduceOp.MAX, group=self.group)

    def _on_stage(self, stage: int):
        if self.world == 1 and not self.force_collective:
            return
        lo, hi = self.slices[stage]
        g = self.model._gflat
        n = (hi - lo) // self.world
        mine = g[lo + self.rank * n: lo + (self.rank + 1) * n]
        if self._native_rs or (self.world == 1 and torch.cuda.is_available() and dist.is_initialized() and dist.get_backend(self.group) == "nccl"):
            # in place: the output is the rank-th part of the input (RCCL's in-place reduce-scatter layout)
            self.works.append(dist.reduce_scatter_tensor(mine, g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:                                   # gloo has no reduce-scatter: all-reduce the slice, every rank then reads its own part
            self.works.append(_all_reduce_sum(g[lo:hi], self.group))

    def finish(self):
        for w in self.works:
            w.wait()
        self.works.clear()

    reduce = lambda self, flat=None: self.finish()

    def gather_state(self):
        if self.opt is not None:
            for buf in [self.opt.exp_avg, self.opt.exp_avg_sq] + list(self.opt.ema):
                self._gather(buf, verify=True)            # (checkpoint cadence; the per-step parameter gather is not checked)

    def _gather_params(self):
        """All-gather of the parameter parts each rank has just updated (in place in the flat parameter buffer)."""
        self._gather(self.model._pflat)

    def _gather(self, p, verify: bool = False):
        if self.world == 1 and not self.force_collective:
            return
        works, sums = [], []
        for lo, hi in self.slices:
            n = (hi - lo) // self.world
            mine = p[lo + self.rank * n: lo + (self.rank + 1) * n]
            if verify and self.world > 1:
                sums.append(_range_checksum(mine))
            if self._native_rs or self.world == 1:
                works.append(dist.all_gather_into_tensor(p[lo:hi], mine, group=self.group, async_op=True))
            elif _host_staged(self.group, p):
                parts = [torch.empty(n, dtype=p.dtype) for _ in range(self.world)]
                dist.all_gather(parts, mine.detach().cpu(), group=self.group)
                p[lo:hi].copy_(torch.cat(parts))
                torch.cuda.synchronize(p.device)
            else:
                works.append(dist.all_gather(list(p[lo:hi].chunk(self.world)), mine.clone(), group=self.group, async_op=True))
        for w in works:
            w.wait()
        if verify and self.world > 1:
            for (lo, hi), before in zip(self.slices, sums):
                _verify_gather(p[lo:hi], (hi - lo) // self.world, self.world, before, self.group)


_SHARDED_SUFFIXES = ("attn.qkv_proj.weight", "attn.out_proj.weight", "mlp.net.0.weight", "mlp.net.2.weight", "modulation.1.weight")
_B_OF = {"attn.qkv_proj.weight": 0, "attn.out_proj.weight": 1, "mlp.net.0.weight": 2, "mlp.net.2.weight": 3, "modulation.1.weight": 4}   # mapdit.h MAPDIT_B_*


class ShardedPassReducer(_ReducerBase):
    """Data parallelism with SHARDED WEIGHT PASSES (round 5; ``--grad-comm zero1w`` / ``zero1w-bf16``).

    The step of a small per-GPU batch is dominated by passes that do not depend on the batch: forced weight normalisation + 16-bit
    imaging of every weight (reference mp_linear.py:38-44 once per forward), the weight-norm Jacobian of every weight gradient, Adam and
    the two EMA copies (train.py:57, src/ema.py:135-140) - 2.1 ms of a 10.2 ms DiT-B/2 step at 32 samples per GPU, replicated on every
    rank by the all-reduce scheme.  Here the ROWS of every block linear (QKV, proj, fc1, fc2, modulation: 99 % of the parameters) are
    split over the ranks, and each rank runs all of those passes on its rows only:

    * backward leaves those weights' gradients as RAW split-K sums (``mapdit_engine_set_shard``); per stage, as soon as it is enqueued,
      each weight's raw gradient is reduce-scattered (rank r receives the sum of its rows) - fp32, or with ``grad_dtype="bf16"`` as a
      16-bit exchange: every rank's bf16 copy of the owner's rows travels (all-to-all) and is summed in fp32 on receipt
      (``mapdit_sum_bf16_chunks``): half the bytes, one rounding of each rank's contribution to 8 mantissa bits;
    * the weight-norm Jacobian - linear in G, so it commutes with the sum over ranks - runs on the owned rows
      (``mapdit_engine_jacobian_shard``), then the fused Adam + EMA kernel (``FusedAdamEMA.shards``);
    * the next forward's weight pass rewrites / images the owned rows, and the 16-BIT IMAGES (260 MB for DiT-B/2, not the 521 MB of fp32
      parameters ZeRO-1 gathers) are all-gathered before the network runs.
    The small parameters (embedders, final layer, gains: < 1 %) stay replicated and are all-reduced.  Bytes on the links: 521 + 260 MB
    (fp32 exchange) or 260 + 260 MB (bf16) per rank and step against 2 x 521 MB for the all-reduce.

    fp32 master weights, Adam moments and EMA copies of rows a rank does not own go stale on that rank: ``gather_state()`` completes them
    (checkpoints, EMA snapshots, switching to eval).  ``emulate=(rank, world)``: one process does what rank ``rank`` of ``world`` would do
    WITHOUT the collectives - the one-GPU measurement of the saving (bench.py --emulate-world); its training results are not meaningful.
    """

    def __init__(self, model, group=None, force_collective: bool = False, grad_dtype: str = "fp32", emulate=None):
        assert grad_dtype in ("fp32", "bf16")
        assert getattr(model, "gemm_precision", "f16") != "bf16x3", "sharded weight passes: bf16 / f16 engines only"
        self.model, self.group, self.grad_dtype = model, group, grad_dtype
        self.emulate = emulate
        self.world = emulate[1] if emulate else _world(group)
        self.rank = emulate[0] if emulate else (dist.get_rank(group) if self.world > 1 else 0)
        self.name = (f"zero1w: sharded weight passes (forced WN + imaging, Jacobian, Adam/EMA on 1/{self.world} of the rows), "
                     f"{grad_dtype} reduce-scatter of raw weight gradients + all-gather of 16-bit images" + (" [EMULATED, no collectives]" if emulate else ""))
        self.slices = stage_slices(model)
        L = model.depth
        self.weights = []                     # sharded weights: dict(stage, off, rows, cols, pidx)
        from . import _lib as Lib
        for (name, p), off in zip(model.named_parameters(), model._poffs):
            if not name.startswith("blocks."):
                continue
            _, bi, suffix = name.split(".", 2)
            if suffix in _SHARDED_SUFFIXES and p.dim() == 2 and p.shape[0] % (4 * self.world) == 0 and self.world > 1:
                self.weights.append(dict(stage=L - int(bi), off=off, rows=p.shape[0], cols=p.shape[1],
                                         pidx=Lib.NUM_GLOBAL + int(bi) * Lib.NUM_BLOCK + _B_OF[suffix]))
        # replicated remainder of every stage slice: what lies between / around the sharded weights
        self.rest = []
        for st, (lo, hi) in enumerate(self.slices):
            pos, out = lo, []
            for w in sorted((w for w in self.weights if w["stage"] == st), key=lambda w: w["off"]):
                if w["off"] > pos:
                    out.append((pos, w["off"]))
                pos = w["off"] + w["rows"] * w["cols"]
                assert pos <= hi
            if pos < hi:
                out.append((pos, hi))
            self.rest.append(out)
        self.works = []
        self.opt = None
        self.force_collective = force_collective
        self._send16 = self._recv16 = None
        self._pending16 = []                  # (weight, offset in the 16-bit buffers) of exchanges whose fp32 sum is still to run
        self._off16, tot = {}, 0
        for w in self.weights:
            self._off16[w["off"]] = tot
            tot += w["rows"] * w["cols"]
        self._tot16 = tot
        self._emu_ready = False
        model._shard = (self.rank, self.world) if self.world > 1 else None
        model._stage_hook = self._on_stage if self.world > 1 else None
        model._after_prepare_hook = self._gather_images if self.world > 1 else None
        self._native = self.world > 1 and not emulate and dist.get_backend(group) == "nccl"
        # MAPDIT_ZERO1W_COALESCE=0: one collective per weight (the form the two-rank rehearsals run); default on RCCL: one grouped launch
        # per backward stage, one grouped all-gather per block, the forward fenced per block (see _gather_images)
        self._coalesce = os.environ.get("MAPDIT_ZERO1W_COALESCE", "1") != "0"
        self._fence_stream = None
        self._fence_events = None

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def own(self, w):
        """[lo, hi) of the flat buffers: the rows of sharded weight `w` this rank owns."""
        n = (w["rows"] // self.world) * w["cols"]
        return w["off"] + self.rank * n, w["off"] + (self.rank + 1) * n

    def parts(self):
        """Element ranges of the flat buffers this rank's optimiser steps: its rows of every sharded weight + everything replicated."""
        out = [self.own(w) for w in self.weights] + [r for rs in self.rest for r in rs]
        return sorted(out)

    def attach(self, opt):
        self.opt = opt
        if self.world > 1:
            opt.shards = self.parts()
            opt.ema_ranges = None
            # after a step this rank's copy of the OTHER ranks' master rows is stale: anything but the next training forward (which uses
            # the gathered images) must call gather_state() first - the model refuses an inference forward until then
            opt.after_step = lambda: setattr(self.model, "_shard_stale", not self.emulate)
            if not self.emulate:
                opt.status_sync = self._sync_status

    def _sync_status(self, status):
        if _host_staged(self.group, status):               # (rehearsal groups only: see Zero1Reducer._sync_status)
            h = status.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            status.copy_(h)
        else:
            dist.all_reduce(status, op=dist.ReduceOp.MAX, group=self.group)

    # ---- gradient exchange, per backward stage -------------------------------------------------------------------------------------
    def _on_stage(self, stage: int):
        if self.emulate:
            return
        g = self.model._gflat
        for lo, hi in self.rest[stage]:
            self.works.append(_all_reduce_sum(g[lo:hi], self.group))
        mine = [w for w in self.weights if w["stage"] == stage]
        if self._native and self.grad_dtype == "fp32" and self._coalesce and len(mine) > 1:
            # RCCL: the stage's reduce-scatters as ONE grouped launch (torch's coalescing manager: ncclGroupStart / End around them) -
            # five collectives of 2 ... 9 MB each pay their launch latency once
            from torch.distributed.distributed_c10d import _coalescing_manager
            with _coalescing_manager(group=self.group, device=g.device, async_ops=True) as cm:
                for w in mine:
                    mlo, mhi = self.own(w)
                    dist.reduce_scatter_tensor(g[mlo:mhi], g[w["off"]:w["off"] + w["rows"] * w["cols"]], op=dist.ReduceOp.SUM, group=self.group)
            self.works.append(cm)
            return
        for w in mine:
            lo, hi = w["off"], w["off"] + w["rows"] * w["cols"]
            mlo, mhi = self.own(w)
            if self.grad_dtype == "bf16":
                self._exchange16(g, w, lo, hi)
            elif self._native:
                # in place: the output is the rank-th part of the input (RCCL's in-place reduce-scatter layout)
                self.works.append(dist.reduce_scatter_tensor(g[mlo:mhi], g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:                               # gloo has no reduce-scatter: all-reduce the weight, the rank then uses its own rows
                self.works.append(_all_reduce_sum(g[lo:hi], self.group))

    def _exchange16(self, g, w, lo, hi):
        """16-bit exchange of one weight's raw gradient: bf16 copies of every owner's rows travel (asynchronously, under the backward of
        the earlier blocks); the owner sums them in fp32 in finish()."""
        from . import _lib as Lib
        n = hi - lo
        per = n // self.world
        if self._send16 is None:
            self._send16 = torch.empty(self._tot16, dtype=torch.bfloat16, device=g.device)
            self._recv16 = torch.empty(self._tot16, dtype=torch.bfloat16, device=g.device)
        o = self._off16[w["off"]]
        send, recv = self._send16[o:o + n], self._recv16[o:o + n]
        with torch.cuda.device(g.device):
            Lib.lib().f32_to_bf16(g[lo:hi].data_ptr(), send.data_ptr(), n, 1.0, Lib.cur_stream())
        if _host_staged(self.group, g):
            parts = [torch.empty(2 * n, dtype=torch.uint8) for _ in range(self.world)]          # (as bytes: any backend)
            dist.all_gather(parts, send.cpu().view(torch.uint8), group=self.group)
            recv.copy_(torch.cat([p_.view(torch.bfloat16)[self.rank * per:(self.rank + 1) * per] for p_ in parts]))
        else:       # chunk r of every rank's copy -> rank r, received in rank order
            self.works.append(dist.all_to_all_single(recv, send, group=self.group, async_op=True))
        self._pending16.append((w, o))

    def finish(self):
        for w in self.works:
            w.wait()
        self.works.clear()
        if self.world > 1:
            from . import _lib as Lib
            rt = self._train_rt()
            g = self.model._gflat
            with torch.cuda.device(self.model._pflat.device):
                for w, o in self._pending16:  # 16-bit exchange: the owner's fp32 sum of every rank's bf16 copy of its rows
                    mlo, mhi = self.own(w)
                    per = mhi - mlo
                    Lib.lib().sum_bf16_chunks(g[mlo:mhi].data_ptr(), self._recv16[o:o + per * self.world].data_ptr(), self.world, per, per,
                                              Lib.cur_stream())
                self._pending16.clear()
                # the Jacobian on the owned rows (in place in the gradient buffer)
                Lib.lib().engine_jacobian_shard(rt.handle, Lib.cur_stream())

    reduce = lambda self, flat=None: self.finish()

    def _train_rt(self):
        m = self.model
        precision = getattr(m, "gemm_precision", "f16")
        rt = m._rt.get(True if precision == "bf16" else (precision, True))
        assert rt is not None, "no training engine yet: run a training forward first"
        return rt

    # ---- all-gather of the 16-bit weight images, after the (sharded) weight pass of a training forward --------------------------------
    def _image_views(self, rt):
        """Flat byte views of every sharded weight's 16-bit image (and split image) inside the engine's workspace."""
        import ctypes as C
        if getattr(self, "_views_of", None) is rt:
            return self._views
        ws, base, out = rt.workspace, rt.workspace.data_ptr(), []
        self._views_per_weight = []
        from . import _lib as Lib
        self._num_global, self._num_block = Lib.NUM_GLOBAL, Lib.NUM_BLOCK
        for w in self.weights:
            n_before = len(out)
            img, img3, rows, cols, sh = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
            rt.lib.engine_weight_image(rt.handle, w["pidx"], C.byref(img), C.byref(img3), C.byref(rows), C.byref(cols), C.byref(sh))
            assert sh.value == 1 and rows.value == w["rows"] and cols.value == w["cols"], f"engine and reducer disagree on the sharding of weight {w}"
            for ptr, width in ((img.value, cols.value), (img3.value, 3 * cols.value)):
                if ptr:
                    o = ptr - base
                    assert 0 <= o and o + rows.value * width * 2 <= ws.numel()
                    out.append(ws[o:o + rows.value * width * 2])          # (bytes: every backend moves uint8)
            self._views_per_weight.append(len(out) - n_before)
        self._views_of, self._views = rt, out
        return out

    def _gather_images_fenced(self, rt):
        """RCCL: one grouped all-gather per block (its four linears' images, the modulation image and split image), issued in block order
        right after the weight pass; the engine's forward waits for block i's gather just before block i (mapdit_engine_set_block_fences),
        so every gather but the first runs under the blocks before it.  Modulation images feed the ONE batched modulation GEMM at the
        start of the forward: they travel with block 0's group."""
        import ctypes as C
        from torch.distributed.distributed_c10d import _coalescing_manager
        views = self._image_views(rt)                      # per sharded weight: [image] or [image, split image], in self.weights order
        L = self.model.depth
        per_block = [[] for _ in range(L)]
        k = 0
        for w, nv in zip(self.weights, self._views_per_weight):
            blk = L - w["stage"]
            is_mod = (w["pidx"] - self._num_global) % self._num_block == 4
            for _ in range(nv):
                per_block[0 if is_mod else blk].append(views[k])
                k += 1
        dev = self.model._pflat.device
        if self._fence_stream is None:
            self._fence_stream = torch.cuda.Stream(device=dev)
            self._fence_events = [torch.cuda.Event() for _ in range(L)]
        handles = (C.c_void_p * L)()
        for b in range(L):
            if per_block[b]:
                with _coalescing_manager(group=self.group, device=dev, async_ops=True) as cm:
                    for whole in per_block[b]:
                        n = whole.numel() // self.world
                        dist.all_gather_into_tensor(whole, whole[self.rank * n:(self.rank + 1) * n], group=self.group)
                with torch.cuda.stream(self._fence_stream):
                    cm.wait()                              # the fence stream waits for the collective; the compute stream does not
                    self._fence_events[b].record(self._fence_stream)
            else:
                self._fence_events[b].record(torch.cuda.current_stream(dev))
            handles[b] = self._fence_events[b].cuda_event
        rt.lib.engine_set_block_fences(rt.handle, handles, L)

    def _gather_images(self, rt):
        if self._native and self._coalesce:
            return self._gather_images_fenced(rt)
        if self.emulate:
            # one-GPU measurement: nothing is gathered, so the other ranks' rows of the images would be whatever the workspace held.  Once,
            # image EVERY row (unsharded pass, no rewrite) so that the network computes on finite numbers; they go stale, the timing does not care.
            if not self._emu_ready:
                from . import _lib as Lib
                with torch.cuda.device(self.model._pflat.device):
                    rt.lib.engine_set_shard(rt.handle, 0, 1)
                    rt.lib.engine_prepare_weights(rt.handle, 0, Lib.cur_stream())
                    rt.lib.engine_set_shard(rt.handle, self.rank, self.world)
                self._emu_ready = True
            return
        works = []
        for whole in self._image_views(rt):
            n = whole.numel() // self.world
            mine = whole[self.rank * n:(self.rank + 1) * n]
            if self._native:
                works.append(dist.all_gather_into_tensor(whole, mine, group=self.group, async_op=True))      # in place
            elif _host_staged(self.group, whole):
                parts = [torch.empty(n, dtype=whole.dtype) for _ in range(self.world)]
                dist.all_gather(parts, mine.detach().cpu(), group=self.group)
                whole.copy_(torch.cat(parts))
            else:
                works.append(dist.all_gather(list(whole.chunk(self.world)), mine.clone(), group=self.group, async_op=True))
        for w in works:
            w.wait()

    # ---- completing the sharded state (checkpoints, EMA snapshots, evaluation) ------------------------------------------------------
    def gather_state(self):
        if self.world == 1 or self.emulate:
            return
        bufs = [self.model._pflat]
        if self.opt is not None:
            bufs += [self.opt.exp_avg, self.opt.exp_avg_sq] + list(self.opt.ema)
        for buf in bufs:
            for w in self.weights:
                lo, hi = w["off"], w["off"] + w["rows"] * w["cols"]
                mlo, mhi = self.own(w)
                whole, mine = buf[lo:hi], buf[mlo:mhi]
                before = _range_checksum(mine)
                if self._native:
                    dist.all_gather_into_tensor(whole, mine, group=self.group)
                elif _host_staged(self.group, buf):
                    parts = [torch.empty(mhi - mlo, dtype=buf.dtype) for _ in range(self.world)]
                    dist.all_gather(parts, mine.detach().cpu(), group=self.group)
                    whole.copy_(torch.cat(parts))
                    torch.cuda.synchronize(buf.device)
                else:
                    dist.all_gather(list(whole.chunk(self.world)), mine.clone(), group=self.group)
                _verify_gather(whole, mhi - mlo, self.world, before, self.group)
        self.model._shard_stale = False
        self.model.mark_weights_changed()


def make_reducer(model, mode: str | None = None, group=None):
    """``allreduce`` (default; per-stage all-reduce overlapped with backward, replicated optimiser), ``zero1``, or ``zero1w`` /
    ``zero1w-bf16`` (sharded weight passes with an fp32 / 16-bit gradient exchange; round 5)."""
    mode = mode or os.environ.get("MAPDIT_GRAD_COMM") or "allreduce"
    if mode == "allreduce":
        return OverlappedGradReducer(model, group)
    if mode == "zero1":
        return Zero1Reducer(model, group)
    if mode in ("zero1w", "zero1w-bf16"):
        return ShardedPassReducer(model, group, grad_dtype="bf16" if mode.endswith("bf16") else "fp32")
    raise ValueError(f"unknown gradient exchange {mode!r} (allreduce | zero1 | zero1w | zero1w-bf16)")


def shard_batch(global_batch: int, rank: int, world: int):
    """[lo, hi) of this rank's samples when a global batch is split evenly (strong scaling); raises on ragged splits."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per
