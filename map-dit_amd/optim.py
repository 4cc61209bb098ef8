"""Optimiser step of the reference training loop as one fused HIP pass over flat buffers:
``torch.optim.Adam(lr, betas=(0.9, 0.99))`` (reference train.py:57), the LambdaLR schedule (train.py:179-197) and
the two power-function EMA copies (reference src/ema.py:10-40, 117-140; std 0.05 and 0.1).

With data parallelism the mean over ranks is folded in as the gradient scale (``1 / world_size``) after the
sum-all-reduce of the flat gradient buffer (see parallel.py).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L


def create_lr_lambda(num_lin_warmup: int, start_decay: int):
    """reference train.py:179-197."""

    def lr_lambda(step):
        if step + 1 < num_lin_warmup:
            return (step + 1) / num_lin_warmup
        if step >= start_decay:
            return 1.0 / math.sqrt(max(step / start_decay, 1))
        return 1.0

    return lr_lambda


def std_to_gamma(std: float) -> float:
    """reference src/ema.py:10-20."""
    t = float(std) ** -2
    return float(np.roots([1, 7, 16 - t, 12 - t]).real.max())


def calc_beta(std: float, t: int) -> float:
    """reference src/ema.py:33-40."""
    return (1 - 1 / t) ** (std_to_gamma(std) + 1)


class FusedAdamEMA:
    """Adam + LR schedule + two EMA copies over the model's flat parameter / gradient buffers (one kernel launch)."""

    def __init__(self, model, lr: float = 1e-2, betas=(0.9, 0.99), eps: float = 1e-8, ema_stds=(0.05, 0.1),
                 lr_lambda=None, grad_scale: float = 1.0, nonfinite_guard=None):
        assert len(ema_stds) in (0, 2), "the fused kernel carries exactly two EMA copies (or none)"
        self.model = model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.ema_stds = tuple(ema_stds)
        self.lr_lambda = lr_lambda or (lambda step: 1.0)
        self.grad_scale = grad_scale
        self.step_count = 0
        flat = model._pflat
        assert flat is not None and flat.is_cuda and flat.dtype == torch.float32
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.ema = [flat.clone() for _ in self.ema_stds]
        self._gammas = [std_to_gamma(s) for s in self.ema_stds]
        self.shards = [(0, flat.numel())]            # element ranges of the flat buffers this rank updates
        self.after_step = None
        # Data parallel with replicated parameters: every rank must run Adam on every parameter, but nothing in a step READS the EMA
        # copies - each rank keeps them current for its own range only (16 of the kernel's 44 bytes per parameter) and the ranges
        # are gathered when a snapshot or checkpoint needs them (parallel.OverlappedGradReducer.attach / gather_state).
        self.ema_ranges = None                       # [(lo, hi), ...] ascending: update the EMA copies only there; None = everywhere
        # Non-finite gradient guard (mapdit.h, mapdit_grad_nonfinite_check): the reference trains in fp32 (train.py:222-223) and
        # cannot overflow; the fp16 engine's activation gradients can (static loss scale).  With the guard a step whose gradients
        # hold an inf / NaN is REFUSED ON THE DEVICE - parameters, moments and EMA copies stay untouched - without a read-back in
        # the loop; poll_overflow() (logging cadence) reports refused steps and halves the model's loss scale.
        # None = on while the model computes in "f16", off otherwise; True / False force it.
        self.nonfinite_guard = nonfinite_guard
        self._status = torch.zeros(2, dtype=torch.int32, device=flat.device)     # {last bad launch, number of bad launches}
        self._overflows_seen = 0
        # `step_count` counts APPLIED optimiser steps (bias corrections, LR schedule, EMA beta(t), the saved Adam "step": a refused step
        # advances none of them, as with torch's GradScaler); `_launches` numbers the step() calls and is what the device-side guard
        # compares (never rewinds, so a rolled-back run cannot meet a stale verdict).  A refusal becomes known to the host at the next
        # poll: until then step_count runs ahead by the refused steps of that window and is rolled back there.
        self._launches = 0
        self._launches_polled = 0
        self.growth_interval = 2000                  # applied steps without a refusal after which a halved loss scale doubles again
        self.max_refused_at_unit_scale = 8           # consecutive refused steps with the loss scale at 1 (or no scale to lower): raise
        self._scale_ceiling = None                   # the scale the run started from; growth never passes it
        self._clean_since = 0                        # step_count at the last refusal / scale change
        self._stuck = 0
        self._poll_every_step = False                # set while refusals are occurring: poll after every step until one is applied
        self.status_sync = None                      # ZeRO-1: all-reduce(MAX) of the status words so that every rank refuses together

    def zero_grad(self, set_to_none: bool = True):
        for p in self.model.parameters():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def current_lr(self) -> float:
        return self.lr * self.lr_lambda(self.step_count)

    def step(self):
        m = self.model
        assert m._gflat is not None, "no gradients: call backward() first"
        assert m._pflat.numel() == self.exp_avg.numel(), "the model was re-flattened after the optimiser was built"
        lr = self.current_lr()                       # LambdaLR: lr for optimiser step k uses lambda(k), k from 0
        self.step_count += 1
        self._launches += 1
        t = self.step_count
        launch = self._launches
        b1, b2 = self.betas
        betas = [(1 - 1 / t) ** (g + 1) for g in self._gammas] if self.ema_stds else [0.0, 0.0]
        # the per-step values travel as kernel arguments: nothing is uploaded in the training loop
        hyper = L.AdamScalars(lr / (1 - b1 ** t), 1.0 / math.sqrt(1 - b2 ** t), betas[0], betas[1], self.grad_scale)
        guard = self.nonfinite_guard
        if guard is None:
            guard = getattr(m, "gemm_precision", "bf16") == "f16"
        if guard:
            with torch.cuda.device(m._pflat.device):
                live = [(lo, hi) for lo, hi in self.shards if hi > lo]
                if len(live) > 8:
                    tab, nblk = self._range_table(live)
                    L.lib().grad_nonfinite_check_ranges(m._gflat.data_ptr(), tab.data_ptr(), len(live), nblk, self._status.data_ptr(), launch,
                                                        L.cur_stream())
                else:
                    for lo, hi in live:
                        L.lib().grad_nonfinite_check(m._gflat.data_ptr() + lo * 4, hi - lo, self._status.data_ptr(), launch, L.cur_stream())
            if self.status_sync is not None:
                self.status_sync(self._status)
        segs = []                                    # (lo, hi, with_ema)
        for lo, hi in self.shards:                   # the whole flat buffer unless a ZeRO-1 reducer handed over its partition
            if self.ema_ranges is None or not self.ema:
                segs.append((lo, hi, True))
                continue
            pos = lo
            for a, b in self.ema_ranges:
                a, b = max(a, lo), min(b, hi)
                if a >= b:
                    continue
                segs += [(pos, a, False), (a, b, True)]
                pos = b
            segs.append((pos, hi, False))
        segs = [sg for sg in segs if sg[1] > sg[0]]
        if len(segs) > 8 and all(w for _, _, w in segs):
            # many ranges (sharded weight passes: a rank's rows of every weight + the replicated rest): ONE launch over a device table
            tab, nblk = self._range_table([(lo, hi) for lo, hi, _ in segs])
            with torch.cuda.device(m._pflat.device):
                L.lib().adam_ema_step_ranges(m._pflat.data_ptr(), m._gflat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                             self.ema[0].data_ptr() if self.ema else None, self.ema[1].data_ptr() if self.ema else None,
                                             tab.data_ptr(), len(segs), nblk, C.byref(hyper), b1, b2, self.eps,
                                             self._status.data_ptr() if guard else None, launch if guard else 0, L.cur_stream())
            segs = []
        for lo, hi, with_ema in segs:
            if hi <= lo:
                continue
            off = lo * 4
            ema = self.ema if with_ema else None
            with torch.cuda.device(m._pflat.device):
                args = (m._pflat.data_ptr() + off, m._gflat.data_ptr() + off, self.exp_avg.data_ptr() + off,
                        self.exp_avg_sq.data_ptr() + off, ema[0].data_ptr() + off if ema else None,
                        ema[1].data_ptr() + off if ema else None, hi - lo, C.byref(hyper), b1, b2, self.eps)
                if guard:
                    L.lib().adam_ema_step_guarded(*args, self._status.data_ptr(), launch, L.cur_stream())
                else:
                    L.lib().adam_ema_step_scalars(*args, L.cur_stream())
        if self.after_step is not None:
            self.after_step()                        # ZeRO-1: all-gather of the updated parameter shards
        m.mark_weights_changed()
        if guard and self._poll_every_step:
            self.poll_overflow()

    def _range_table(self, ranges):
        """Device table of mapdit_range_t {lo, hi, first_block} for the multi-range kernels (built once per set of ranges)."""
        key = tuple(ranges)
        cached = getattr(self, "_range_cache", None)
        if cached is None or cached[0] != key:
            rows, blk = [], 0
            for lo, hi in ranges:
                assert lo % 4 == 0 and hi % 4 == 0
                rows.append((lo, hi, blk))
                blk += (hi - lo + 4095) // 4096
            self._range_cache = (key, torch.tensor(rows, dtype=torch.int64, device=self.model._pflat.device), blk)
        return self._range_cache[1], self._range_cache[2]

    # ---- non-finite gradient guard ------------------------------------------------------------------------------
    def overflow_steps(self) -> int:
        """Number of optimiser steps the guard has refused so far (synchronises: call at logging cadence)."""
        return int(self._status[1].item())

    def poll_overflow(self, halve_loss_scale: bool = True) -> int:
        """Steps refused since the previous poll (synchronises the stream; train.py calls it at logging cadence).

        * `step_count` is rolled back by that number: a refused step is not an optimiser step (bias corrections, LR schedule, EMA
          beta(t) and the saved Adam "step" do not advance for updates that never happened).
        * With `halve_loss_scale` the model's fp16 loss scale is halved ONCE PER REFUSED STEP (not once per poll), floor 1.
        * While refusals are occurring the optimiser polls after every step by itself, so a scale 2^k too high costs k batches and
          not k * log_every; it returns to the caller's cadence after the first applied step.
        * Refusals that a lower scale cannot cure (scale already 1, or a NaN that comes from the data / a bf16 run with the guard
          forced on) raise FloatingPointError after `max_refused_at_unit_scale` consecutive refused steps instead of refusing forever.
        * After `growth_interval` applied steps without a refusal a halved scale doubles again, up to the scale the run started from.
        """
        total = self.overflow_steps()
        new = total - self._overflows_seen
        self._overflows_seen = total
        launched = self._launches - self._launches_polled        # step() calls since the previous poll; `new` of them were refused
        self._launches_polled = self._launches
        f16 = getattr(self.model, "gemm_precision", "") == "f16"
        if new > 0:
            self.step_count = max(self.step_count - new, 0)
            self._clean_since = self.step_count
            self._poll_every_step = True
            cur = self.model.effective_loss_scale() if f16 else 1.0
            if self._scale_ceiling is None and f16:
                self._scale_ceiling = cur
            if halve_loss_scale and f16 and cur > 1.0:
                self.model.loss_scale = max(cur / 2.0 ** new, 1.0)
                self._stuck = 0
            else:
                self._stuck += new
                if self._stuck >= self.max_refused_at_unit_scale:
                    raise FloatingPointError(
                        f"{self._stuck} consecutive optimiser steps refused for non-finite gradients with the loss scale at "
                        f"{cur:g}: lowering the scale cannot cure this (NaN / inf in the data or in the weights?)")
        elif launched > 0:                                       # every step of the window was applied
            self._poll_every_step = False
            self._stuck = 0
            if (f16 and halve_loss_scale and self._scale_ceiling is not None
                    and self.step_count - self._clean_since >= self.growth_interval):
                cur = self.model.effective_loss_scale()
                if cur < self._scale_ceiling:
                    self.model.loss_scale = min(cur * 2.0, self._scale_ceiling)
                self._clean_since = self.step_count
        return new

    # ---- checkpoint interchange with the reference (train.py:125-132 stores torch.optim.Adam.state_dict()) -------
    def _slots(self):
        return [(o, p.numel(), p.shape) for p, o in zip(self.model.parameters(), self.model._poffs)]

    def state_dict(self):
        """Same layout as ``torch.optim.Adam(model.parameters()).state_dict()``: per-parameter ``step`` / ``exp_avg`` /
        ``exp_avg_sq`` keyed by the parameter's index (the parameter order equals the reference's), one param group."""
        state = {}
        if self.step_count > 0:
            for i, (o, n, shape) in enumerate(self._slots()):
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + n].view(shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(shape).clone()}
        group = {"lr": self.current_lr(), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "initial_lr": self.lr, "params": list(range(len(self.model._poffs)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts what ``state_dict`` above or the reference's ``opt.state_dict()`` holds."""
        slots = self._slots()
        group = sd["param_groups"][0]
        assert len(sd["param_groups"]) == 1 and len(group["params"]) == len(slots), "optimizer state does not match the model"
        self.betas, self.eps = tuple(group["betas"]), group["eps"]
        self.lr = group.get("initial_lr", group["lr"])
        steps = {int(v["step"]) for v in sd["state"].values()}
        assert len(steps) <= 1, "per-parameter step counts differ"
        self.step_count = steps.pop() if steps else 0
        # the guard's verdict words belong to the run that was loaded over: a refusal recorded for a step this optimiser is about to
        # repeat must not be met again (launch numbers never rewind either: `_launches` is left alone)
        self._status.zero_()
        self._overflows_seen = 0
        self._clean_since = self.step_count
        self._stuck = 0
        self._poll_every_step = False
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, (o, n, shape) in enumerate(slots):
            st = sd["state"].get(i) or sd["state"].get(str(i))
            if st is None:
                continue
            self.exp_avg[o:o + n].view(shape).copy_(st["exp_avg"])
            self.exp_avg_sq[o:o + n].view(shape).copy_(st["exp_avg_sq"])

    def load_ema_state_dict(self, std: float, sd):
        """Restore one EMA copy from a snapshot's ``state_dict`` (resuming a run)."""
        flat = self.ema[self.ema_stds.index(std)]
        for (name, p), o in zip(self.model.named_parameters(), self.model._poffs):
            flat[o:o + p.numel()].view(p.shape).copy_(sd[name].to(flat.dtype))

    def ema_state_dict(self, std: float):
        """EMA weights as a state_dict with the reference's keys (what src/ema.py:143-155 snapshots), fp32."""
        i = self.ema_stds.index(std)
        flat = self.ema[i]
        sd = {}
        params = dict(self.model.named_parameters())
        offs = dict(zip((id(p) for p in self.model.parameters()), self.model._poffs))
        for k, v in self.model.state_dict().items():
            if k in params:
                p = params[k]
                o = offs[id(p)]
                sd[k] = flat[o:o + p.numel()].view(p.shape).clone()
            else:
                sd[k] = v.clone()
        return sd
