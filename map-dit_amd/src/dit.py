"""``DiT`` — the reference's nn.Module surface (src/dit.py:12-118) on top of the HIP engine.

Same constructor arguments, same ``state_dict`` keys, same ``forward(x, t, y)`` /
``forward_with_cfg(x, t, y, cfg_scale)``, train/eval semantics (training-mode forward rewrites the
weights in place = forced weight normalisation, mp_linear.py:38-40), autograd through
``loss.backward()``.  The sub-modules exist to carry the parameters under the reference's names;
the arithmetic of the whole network runs in ``libmapdit_hip.so`` (``mapdit_engine_*``), there is no
PyTorch fallback.

Snapshot semantics: every magnitude-preserving feature is on, conditioning is the
MP-AdaLN shift/scale/gate form — exactly what the reference snapshot implements.
"""
from __future__ import annotations

import copy
import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L

FOURIER_DIM = 256


# ----------------------------------------------------------------------------------------------------
# parameter containers named like the reference's modules
# ----------------------------------------------------------------------------------------------------
class _Fused(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} is a parameter container: its arithmetic runs inside the fused "
                           "MaP-DiT engine (call the DiT module)")


class MPLinear(_Fused):
    """reference src/basic/mp_linear.py:9-46 (gain is the constant 1)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.weight = nn.Parameter(torch.empty(out_dim, in_dim))
        nn.init.normal_(self.weight)


class MPLinearChunk(_Fused):
    """reference src/basic/mp_linear.py:48-75."""

    def __init__(self, in_dim, out_dim, n_chunks):
        super().__init__()
        self.in_dim, self.n_chunks = in_dim, n_chunks
        self.weight = nn.Parameter(torch.empty(n_chunks * out_dim, in_dim))
        nn.init.normal_(self.weight)


class MPEmbedding(_Fused):
    """reference src/basic/mp_embedding.py:8-24."""

    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.normal_(self.weight)


class MPSiLU(_Fused):
    """reference src/basic/mp_silu.py:5-7 (no parameters)."""


class Attention(_Fused):
    """reference src/layers/attention.py:9-27."""

    def __init__(self, in_dim, num_heads):
        super().__init__()
        assert in_dim % num_heads == 0
        self.num_heads, self.head_dim = num_heads, in_dim // num_heads
        self.qkv_proj = MPLinearChunk(in_dim, in_dim, 3)
        self.out_proj = MPLinear(in_dim, in_dim)
        self.scale = 1.0 / math.sqrt(self.head_dim)


class MLP(_Fused):
    """reference src/layers/mlp.py:7-22."""

    def __init__(self, in_dim, out_dim, mlp_ratio=4.0, hidden_dim=None):
        super().__init__()
        self.hidden_dim = int(in_dim * mlp_ratio) if hidden_dim is None else hidden_dim
        self.net = nn.Sequential(MPLinear(in_dim, self.hidden_dim), MPSiLU(), MPLinear(self.hidden_dim, out_dim))


class DiTBlock(_Fused):
    """reference src/blocks/dit_block.py:10-29.  ``rotation_modulation`` (not in the reference snapshot, README.md:1-3 only; parity
    unpinned, semantics in oracle.dit_oracle.modulate_rot): the modulation linear yields (theta [D/2], scale, gate) x 2 = 5 D rows
    instead of (shift, scale, gate) x 2 = 6 D."""

    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0, rotation_modulation=False):
        super().__init__()
        self.attn = Attention(hidden_size, num_heads)
        self.mlp = MLP(hidden_size, hidden_size, mlp_ratio=mlp_ratio)
        if rotation_modulation:
            assert hidden_size % 2 == 0
            self.modulation = nn.Sequential(MPSiLU(), MPLinearChunk(hidden_size, hidden_size // 2, 10))      # 5 D rows
        else:
            self.modulation = nn.Sequential(MPSiLU(), MPLinearChunk(hidden_size, hidden_size, 6))
        self.gain_msa = nn.Parameter(torch.tensor(0.0))
        self.gain_mlp = nn.Parameter(torch.tensor(0.0))


class MPScale(_Fused):
    """reference src/blocks/final_layer.py:12-18."""

    def __init__(self, in_dim, angle_dim=8, zero_init=True):
        super().__init__()
        self.angle_dim = angle_dim
        self.linear = MPLinear(in_dim, angle_dim)
        self.reference = nn.Parameter(torch.zeros(angle_dim) if zero_init else torch.ones(angle_dim))


class FinalLayer(_Fused):
    """reference src/blocks/final_layer.py:24-51."""

    def __init__(self, hidden_size, patch_size, out_channels, learn_sigma=True):
        super().__init__()
        self.learn_sigma = learn_sigma
        self.linear = MPLinearChunk(hidden_size, patch_size * patch_size * out_channels, 2 if learn_sigma else 1)
        self.modulation = nn.Sequential(MPSiLU(), MPLinearChunk(hidden_size, hidden_size, 2))
        self.gain_mod = nn.Parameter(torch.tensor(0.0))
        self.mean_scale = MPScale(hidden_size, zero_init=False)
        if learn_sigma:
            self.sigma_scale = MPScale(hidden_size, zero_init=True)


class MPFourier(_Fused):
    """reference src/blocks/timestep_embedder.py:8-16."""

    def __init__(self, num_channels):
        super().__init__()
        self.register_buffer("scale", (2 * torch.pi * torch.randn(num_channels)).to(torch.float32))
        self.register_buffer("shift", (2 * torch.pi * torch.rand(num_channels)).to(torch.float32))


class TimestepEmbedder(_Fused):
    """reference src/blocks/timestep_embedder.py:24-40."""

    def __init__(self, hidden_size, frequency_embedding_size=FOURIER_DIM):
        super().__init__()
        self.mlp = MLP(frequency_embedding_size, hidden_size, hidden_dim=hidden_size)
        self.embedding = MPFourier(frequency_embedding_size)


class LabelEmbedder(_Fused):
    """reference src/blocks/label_embedder.py:6-34 (token_drop is real: it is the path's only RNG consumer)."""

    def __init__(self, num_classes, hidden_size, dropout_prob):
        super().__init__()
        use_cfg_embedding = dropout_prob > 0
        self.embedding = MPEmbedding(num_classes + use_cfg_embedding, hidden_size)
        self.num_classes, self.dropout_prob = num_classes, dropout_prob

    def token_drop(self, labels, force_drop_ids=None):
        if force_drop_ids is None:
            drop_ids = torch.rand(labels.shape[0], device=labels.device) < self.dropout_prob
        else:
            drop_ids = force_drop_ids == 1
        return torch.where(drop_ids, self.num_classes, labels)


def get_2d_sincos_pos_embed(embed_dim: int, grid_size: int) -> np.ndarray:
    """MAE sin-cos table as the reference builds it (src/pos_embed.py:4-60): first half encodes the w index."""
    gw, gh = np.meshgrid(np.arange(grid_size, dtype=np.float32), np.arange(grid_size, dtype=np.float32))

    def one_d(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    return np.concatenate([one_d(embed_dim // 2, gw), one_d(embed_dim // 2, gh)], axis=1)


# ----------------------------------------------------------------------------------------------------
# engine runtime (one per (module, mode))
# ----------------------------------------------------------------------------------------------------
class _Runtime:
    """Owns a mapdit engine handle and its workspace tensor."""

    def __init__(self, model: "DiT", max_batch: int, train: bool, precision: str = "bf16"):
        lib = L.lib()
        self.lib, self.train, self.max_batch, self.precision = lib, train, max_batch, precision
        self.device = model.pos_embed.device
        rows = model.y_embedder.embedding.weight.shape[0]
        self.cfg = L.Config(depth=len(model.blocks), hidden=model.hidden_size, patch=model.patch_size,
                            input_size=model.input_size, in_channels=model.in_channels, num_heads=model.num_heads,
                            mlp_hidden=model.blocks[0].mlp.hidden_dim, table_rows=rows, max_batch=max_batch,
                            precision=L.PRECISIONS[precision], rotation=int(getattr(model, "rotation_modulation", False)),
                            mp_off=sum(bit for name, bit in L.MP_OFF.items() if not getattr(model, name, True)),
                            loss_scale=float(getattr(model, "loss_scale", 0.0)) if precision == "f16" else 0.0)
        need = lib.engine_workspace_bytes(C.byref(self.cfg), int(train))
        if need == 0:
            raise L.MapditError(f"unsupported DiT configuration: {lib.last_error().decode()}")
        self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            lib.engine_create(C.byref(self.cfg), int(train), self.workspace.data_ptr(), need, L.cur_stream(), C.byref(h))
        self.handle = h
        self.bound_key = None
        self.weights_key = None
        self.generation = 0           # bumped by every forward that saves activations: the engine keeps ONE saved forward

    def bind(self, model: "DiT"):
        params = model._param_table()
        grads = model._grad_table() if self.train else None
        key = tuple(t.data_ptr() for t in params) + (tuple(g.data_ptr() for g in grads) if grads else ())
        if key == self.bound_key:
            return
        n = len(params)
        pa = (C.c_void_p * n)(*[t.data_ptr() for t in params])
        ga = (C.c_void_p * n)(*[g.data_ptr() for g in grads]) if grads else None
        self.lib.engine_bind(self.handle, pa, ga)
        self.bound_key = key
        self.weights_key = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.engine_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _DiTFunction(torch.autograd.Function):
    """Autograd node for the whole network.  Parameter gradients are written by the engine straight into the
    module's flat gradient buffer (``p.grad`` are views of it), so they are not returned to autograd."""

    @staticmethod
    def forward(ctx, model, rt, x, t, y_eff, anchor):
        out = torch.empty(x.shape[0], 2 * model.in_channels, model.input_size, model.input_size, device=x.device)
        rt.lib.engine_forward(rt.handle, x.data_ptr(), t.data_ptr(), y_eff.data_ptr(), x.shape[0], 1, out.data_ptr(),
                              L.cur_stream())
        rt.generation += 1
        ctx.model, ctx.rt, ctx.generation = model, rt, rt.generation
        return out

    @staticmethod
    def backward(ctx, dout):
        model, rt = ctx.model, ctx.rt
        # The engine holds the activations of exactly one forward.  forward(A); forward(B); lossA.backward() would silently
        # differentiate A's loss through B's activations: refuse it (so does a runtime rebuilt for a larger batch in between).
        if ctx.generation != rt.generation or not any(r is rt for r in model._rt.values()):
            raise L.MapditError("backward through a stale forward: the engine keeps the saved activations of the most recent "
                                "training forward only (one outstanding forward per model; run backward before the next forward)")
        dout = dout.contiguous().float()
        accumulate = model._attach_grads()
        keep = model._gflat.clone() if accumulate else None
        rt.bind(model)
        hook = getattr(model, "_stage_hook", None)
        if hook is None:
            rt.lib.engine_backward(rt.handle, dout.data_ptr(), L.cur_stream())
        else:
            # staged backward: after each stage the data-parallel reducer all-reduces the slice that stage finalised
            assert keep is None, "gradient accumulation is not supported together with the overlapped DP reducer"
            for stage in range(model.depth + 2):
                rt.lib.engine_backward_stages(rt.handle, dout.data_ptr(), stage, stage, L.cur_stream())
                hook(stage)
        if keep is not None:
            model._gflat.add_(keep)
        return None, None, None, None, None, None


def _strip_compile_prefix(state_dict, prefix, *args):
    pre = prefix + "_orig_mod."
    for k in [k for k in state_dict if k.startswith(pre)]:
        state_dict[prefix + k[len(pre):]] = state_dict.pop(k)


class DiT(nn.Module):
    """Diffusion model with a Transformer backbone (reference src/dit.py:12-62)."""

    def __init__(self, depth: int, hidden_size: int, patch_size: int, input_size: int = 32, in_channels: int = 3,
                 num_heads: int = 16, mlp_ratio: float = 4.0, class_dropout_prob: float = 0.1, num_classes: int = 1000,
                 learn_sigma: bool = True, rotation_modulation: bool = False, forced_weight_normalization: bool = True,
                 mp_silu: bool = True, mp_residual: bool = True, mp_pos_enc: bool = True, mp_embedding: bool = True,
                 weight_normalization: bool = True, cosine_attention: bool = True, no_layernorm: bool = True):
        super().__init__()
        if not learn_sigma:
            raise NotImplementedError("learn_sigma=False is not built (every reference script uses the default True)")
        self.learn_sigma = learn_sigma
        self.in_channels = in_channels
        self.out_channels = in_channels
        self.input_size = input_size
        self.patch_size = patch_size
        self.num_heads = num_heads
        self.hidden_size = hidden_size
        self.depth = depth
        self.mlp_ratio = mlp_ratio
        self.class_dropout_prob = class_dropout_prob
        self.num_classes = num_classes
        self.rotation_modulation = bool(rotation_modulation)      # README.md:1-3; not in the snapshot: parity unpinned
        # README.md:61 --use-forced-weight-normalization.  Off = the training forward skips the in-place rewrite of the weights
        # (mp_linear.py:38-40, 68-70, mp_embedding.py:17-19) and keeps everything else: the one flag whose off-path can be read
        # off the snapshot's code; the other seven name layers the snapshot does not contain.
        self.forced_weight_normalization = bool(forced_weight_normalization)
        # Off forms of four more README flags (README.md:62-66: --use-mp-residual, --use-mp-silu, --use-mp-pos-enc, --use-mp-embedding).
        # The snapshot hard-wires them on and holds no code for the off forms: PARITY UNPINNED - each is this build's
        # restatement of one README line with upstream DiT's form of the operation (oracle.dit_oracle.DiTConfig), held to that
        # restatement by tests/test_mp_flags_gpu.py.  True = the snapshot's arithmetic.  The remaining three (below) needed kernels of their
        # own: weight passes without normalize(), a softmax with its maximum taken out, a LayerNorm in front of modulate().
        self.mp_silu, self.mp_residual = bool(mp_silu), bool(mp_residual)
        self.mp_pos_enc, self.mp_embedding = bool(mp_pos_enc), bool(mp_embedding)
        # README.md:60 --use-weight-normalization off (unpinned like the four above): MPLinear / MPLinearChunk multiply by
        # W * gain / sqrt(in_dim), i.e. mp_linear.py:44,74 without their normalize(); the weight passes run with MAPDIT_WN_PLAIN
        self.weight_normalization = bool(weight_normalization)
        # README.md:58 --use-cosine-attention off (unpinned): attention.py:42-43 dropped - q, k enter the scaled-dot-product attention unnormalised
        self.cosine_attention = bool(cosine_attention)
        # README.md:64 --use-no-layernorm off (unpinned) = WITH the transformer layer normalisation: upstream DiT's LayerNorm (no affine, eps 1e-6)
        # in front of every modulate() - both branches of every block and the final layer
        self.no_layernorm = bool(no_layernorm)

        self.x_embedder = MPLinear(patch_size * patch_size * in_channels + 1, hidden_size)
        self.t_embedder = TimestepEmbedder(hidden_size)
        self.y_embedder = LabelEmbedder(num_classes, hidden_size, class_dropout_prob)
        pe = torch.from_numpy(get_2d_sincos_pos_embed(hidden_size, input_size // patch_size)).float().unsqueeze(0)
        if self.mp_pos_enc:           # (off form: the raw table, added plainly, as upstream DiT does)
            pe = pe * math.sqrt(pe.shape[-1]) / (torch.linalg.vector_norm(pe, dim=-1, keepdim=True) + 1e-4)   # dit.py:46-48
        self.register_buffer("pos_embed", pe)
        self.blocks = nn.ModuleList([DiTBlock(hidden_size, num_heads, mlp_ratio=mlp_ratio, rotation_modulation=self.rotation_modulation)
                                     for _ in range(depth)])
        self.final_layer = FinalLayer(hidden_size, patch_size, self.out_channels, learn_sigma=learn_sigma)

        self._rt = {}                 # {train (bool) | (precision, train): _Runtime}
        # "f16" (default): IEEE fp16 GEMM / attention operands, fp32 accumulation - forward logits within 1e-3 of the fp32 reference
        # (BASELINE.json's tolerance) | "bf16": the same engine with bf16 operands - 2 % faster, ~6e-3 | "bf16x3": fp32-accurate
        # forward and backward (several times slower).  See mapdit.h.
        self.gemm_precision = "f16"
        self._loss_scale = 0.0        # "f16" only: power-of-two loss scale of the backward (0 = chosen from the batch size)
        self._pflat = None            # flat fp32 storage behind every parameter (views)
        self._gflat = None            # flat gradient buffer, p.grad are views of it
        self._gviews = None
        self._w_epoch = 0             # bumped by anything that rewrites parameters through raw pointers
        self._flatten_parameters()
        # checkpoints / EMA snapshots written by the reference come from a torch.compile'd module (train.py:46, src/ema.py:121):
        # every key carries an "_orig_mod." prefix
        self._register_load_state_dict_pre_hook(_strip_compile_prefix)

    # ---- flat parameter / gradient storage ----------------------------------------------------------------
    def _flatten_parameters(self):
        """Re-home every parameter as a view of one flat buffer (128-byte aligned slots): lets the optimiser, EMA
        and the data-parallel gradient reduction work on single contiguous buffers."""
        params = list(self.parameters())
        if not params:
            return
        dev, dt = params[0].device, params[0].dtype
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 31) // 32 * 32
        flat = torch.zeros(total, device=dev, dtype=dt)
        for p, o in zip(params, offs):
            flat[o:o + p.numel()].view(p.shape).copy_(p.data)
            p.data = flat[o:o + p.numel()].view(p.shape)
        self._pflat, self._poffs = flat, offs
        self._gflat, self._gviews = None, None
        self._rt = {}

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten_parameters()
        return out

    def _attach_grads(self) -> bool:
        """Make every p.grad a view of the flat gradient buffer.  Returns True when gradients that are already
        there must be accumulated into (p.grad was not reset to None since the previous backward)."""
        self._attach_grads_storage()
        fresh, accumulate = [], False
        for p, g in zip(self.parameters(), self._gviews):
            if p.grad is None:
                p.grad = g
                fresh.append(g)
            elif p.grad.data_ptr() != g.data_ptr():
                g.copy_(p.grad)
                p.grad = g
                accumulate = True
            else:
                accumulate = True
        if accumulate:
            for g in fresh:
                g.zero_()
        return accumulate

    def _param_table(self):
        """Tensors in the order of the engine's parameter table (include/mapdit.h: MAPDIT_P_*, MAPDIT_B_*)."""
        f = self.final_layer
        tab = [self.x_embedder.weight, self.t_embedder.mlp.net[0].weight, self.t_embedder.mlp.net[2].weight,
               self.y_embedder.embedding.weight, f.linear.weight, f.modulation[1].weight, f.mean_scale.linear.weight,
               f.mean_scale.reference, f.sigma_scale.linear.weight, f.sigma_scale.reference, f.gain_mod,
               self.t_embedder.embedding.scale, self.t_embedder.embedding.shift, self.pos_embed]
        for b in self.blocks:
            tab += [b.attn.qkv_proj.weight, b.attn.out_proj.weight, b.mlp.net[0].weight, b.mlp.net[2].weight,
                    b.modulation[1].weight, b.gain_msa, b.gain_mlp]
        return tab

    def _grad_table(self):
        self._attach_grads_storage()
        by_id = {id(p): g for p, g in zip(self.parameters(), self._gviews)}
        scratch = self._buffer_grad_scratch
        return [by_id.get(id(t), scratch) for t in self._param_table()]

    def _attach_grads_storage(self):
        if self._gflat is None or self._gflat.device != self._pflat.device:
            params = list(self.parameters())
            self._gflat = torch.zeros_like(self._pflat, dtype=torch.float32)
            self._gviews = [self._gflat[o:o + p.numel()].view(p.shape) for p, o in zip(params, self._poffs)]
        if getattr(self, "_buffer_grad_scratch", None) is None or self._buffer_grad_scratch.device != self._pflat.device:
            self._buffer_grad_scratch = torch.zeros(8, device=self._pflat.device)   # buffers have no gradient

    # ---- runtime ----------------------------------------------------------------------------------------------
    def _runtime(self, batch: int, train: bool) -> _Runtime:
        if self._pflat is None or not self._pflat.is_cuda:
            raise L.MapditError("MaP-DiT runs on the MI355X only: move the module to a cuda device (there is no CPU path)")
        if self._pflat.dtype != torch.float32:
            raise L.MapditError("master parameters must be fp32 (bf16 / fp16 are the engine's internal GEMM operand types)")
        precision = getattr(self, "gemm_precision", "f16")
        if precision not in L.PRECISIONS:
            raise L.MapditError(f"gemm_precision must be one of {sorted(L.PRECISIONS)}, got {precision!r}")
        slot = train if precision == "bf16" else (precision, train)
        rt = self._rt.get(slot)
        if rt is None or rt.max_batch < batch or rt.device != self._pflat.device:
            if rt is not None:
                del self._rt[slot]
            rt = _Runtime(self, max(batch, 1), train, precision)
            self._rt[slot] = rt
        rt.bind(self)
        # data parallelism with sharded weight passes (parallel.ShardedPassReducer): the training engine works on this rank's rows
        shard = getattr(self, "_shard", None) if train and precision != "bf16x3" else None
        if (shard or (0, 1)) != getattr(rt, "shard", (0, 1)):
            rt.lib.engine_set_shard(rt.handle, *(shard or (0, 1)))
            rt.shard = shard or (0, 1)
        return rt

    # "f16" only: the power of two the backward multiplies the incoming gradient by (mapdit_config_t.loss_scale); 0 = chosen per
    # backward from the batch size.  Assigning it reaches the engines that already exist (mapdit_engine_set_loss_scale).
    @property
    def loss_scale(self) -> float:
        return self.__dict__.get("_loss_scale", self.__dict__.get("loss_scale", 0.0))      # (objects pickled before round 4 carry the plain attribute)

    @loss_scale.setter
    def loss_scale(self, value):
        value = float(value)
        m, e = math.frexp(value) if value > 0 and math.isfinite(value) else (0.0, 0)
        if not (value == 0.0 or m == 0.5):
            raise L.MapditError(f"loss_scale must be 0 (automatic) or a finite power of two, got {value!r}")
        self._loss_scale = value
        for slot, rt in getattr(self, "_rt", {}).items():
            if rt.precision == "f16" and rt.train:
                rt.lib.engine_set_loss_scale(rt.handle, value)

    def effective_loss_scale(self) -> float:
        """The loss scale the most recent fp16 backward ran with (the automatic choice resolved); 1.0 before any backward."""
        rt = self._rt.get(("f16", True))
        if rt is None:
            return self.loss_scale or 1.0
        out = C.c_float()
        rt.lib.engine_loss_scale(rt.handle, C.byref(out))
        return float(out.value)

    def _peek(self, what: str, block: int = 0) -> torch.Tensor:
        """Diagnostics (mapdit_engine_peek): a copy of an intermediate of the last training-mode forward, as a 2-d
        [rows, ld] tensor (fp32 or bf16).  Names: _lib.PEEK_IDS."""
        precision = getattr(self, "gemm_precision", "f16")
        rt = self._rt.get(True if precision == "bf16" else (precision, True))
        if rt is None:
            raise L.MapditError("_peek needs a training-mode forward first")
        ptr, n, ld, dt = C.c_void_p(), C.c_long(), C.c_int(), C.c_int()
        rt.lib.engine_peek(rt.handle, L.PEEK_IDS[what], block, C.byref(ptr), C.byref(n), C.byref(ld), C.byref(dt))
        off = ptr.value - rt.workspace.data_ptr()
        size = 2 if dt.value else 4
        assert 0 <= off and off + n.value * size <= rt.workspace.numel()
        flat = rt.workspace[off: off + n.value * size].view({0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[dt.value])
        return flat.view(-1, ld.value).clone()

    def _weights_key(self):
        return (self._w_epoch, sum(p._version for p in self.parameters()))

    def mark_weights_changed(self):
        """Call after rewriting parameters through raw pointers (the fused optimiser does)."""
        self._w_epoch += 1

    def check_device_errors(self):
        """Raise MapditError if a kernel since the last call saw a class label outside the embedding table or a timestep outside
        the schedule (the reference raises IndexError at that point; a kernel can only clamp the index and record it).
        Synchronises the stream: call it at logging / checkpoint cadence, not per step."""
        with torch.cuda.device(self._pflat.device):
            L.lib().device_error_poll(L.cur_stream())

    def _check_inputs(self, x, t, y):
        if x.requires_grad and torch.is_grad_enabled():
            raise L.MapditError("gradients with respect to the input latents are not produced by the engine (no reference script "
                                "uses them); detach x, or differentiate through a copy of the model in the reference framework")
        assert x.dim() == 4 and x.shape[1] == self.in_channels and x.shape[2] == x.shape[3] == self.input_size, \
            f"x must be [N,{self.in_channels},{self.input_size},{self.input_size}], got {tuple(x.shape)}"
        assert t.shape == (x.shape[0],) and y.shape == (x.shape[0],)
        x = x.contiguous().float()
        t = t.to(device=x.device, dtype=torch.int64).contiguous()
        y = y.to(device=x.device, dtype=torch.int64).contiguous()
        return x, t, y

    # ---- reference API ----------------------------------------------------------------------------------------------
    # The reference wraps the model in torch.compile (train.py:46, sample.py:25).  The whole network is one opaque engine call,
    # there is nothing for a tracing compiler to fuse: dynamo is told not to trace into forward, so torch.compile(model) is a
    # thin wrapper around the eager call (and its "_orig_mod."-prefixed state dicts load, see _strip_compile_prefix).
    @torch.compiler.disable
    def forward(self, x, t, y):
        """x [N,C,H,W], t [N], y [N] -> [N,2C,H,W]   (reference src/dit.py:70-105)."""
        x, t, y = self._check_inputs(x, t, y)
        if self.training and self.class_dropout_prob > 0:
            y = self.y_embedder.token_drop(y)                                   # label_embedder.py:19-34
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if not need_grad and getattr(self, "_shard_stale", False):
            raise L.MapditError("sharded weight passes (--grad-comm zero1w): this rank holds stale master rows of the other ranks after an "
                                "optimiser step - call reducer.gather_state() before an inference forward, a checkpoint or an EMA snapshot")
        rt = self._runtime(x.shape[0], train=need_grad)
        with torch.cuda.device(x.device):
            if self.training:
                # forced weight norm (1): rewrites the fp32 weights in place, then images them
                rt.lib.engine_prepare_weights(rt.handle, 1 if self.forced_weight_normalization else 0, L.cur_stream())
                rt.weights_key = None
                for other in self._rt.values():
                    other.weights_key = None
                hook = getattr(self, "_after_prepare_hook", None)          # sharded weight passes: all-gather of the 16-bit images
                if hook is not None and need_grad:
                    hook(rt)
            else:
                key = self._weights_key()
                if rt.weights_key != key:
                    rt.lib.engine_prepare_weights(rt.handle, 0, L.cur_stream())
                    rt.weights_key = key
            if need_grad:
                return _DiTFunction.apply(self, rt, x, t, y, self._anchor())
            out = torch.empty(x.shape[0], 2 * self.in_channels, self.input_size, self.input_size, device=x.device)
            rt.lib.engine_forward(rt.handle, x.data_ptr(), t.data_ptr(), y.data_ptr(), x.shape[0], 0, out.data_ptr(),
                                  L.cur_stream())
            return out

    def _anchor(self):
        a = getattr(self, "_anchor_t", None)
        if a is None or a.device != self._pflat.device:
            a = torch.zeros(1, device=self._pflat.device, requires_grad=True)
            self._anchor_t = a
        return a

    @torch.compiler.disable
    def forward_with_cfg(self, x, t, y, cfg_scale):
        """Classifier-free guidance batch (reference src/dit.py:107-118)."""
        half = x[: len(x) // 2]
        combined = torch.cat([half, half], dim=0)
        model_out = self.forward(combined, t, y)
        out = torch.empty_like(model_out)
        hw = self.input_size * self.input_size
        with torch.cuda.device(model_out.device):
            L.lib().cfg_combine(model_out.data_ptr(), out.data_ptr(), model_out.shape[0], self.in_channels, hw,
                                float(cfg_scale), L.cur_stream())
        return out

    # ---- copying (EMA does copy.deepcopy(model), reference src/ema.py:121) ---------------------------------------------
    def __deepcopy__(self, memo):
        new = DiT(depth=self.depth, hidden_size=self.hidden_size, patch_size=self.patch_size, input_size=self.input_size,
                  in_channels=self.in_channels, num_heads=self.num_heads, mlp_ratio=self.mlp_ratio,
                  class_dropout_prob=self.class_dropout_prob, num_classes=self.num_classes, learn_sigma=self.learn_sigma,
                  rotation_modulation=self.rotation_modulation, forced_weight_normalization=self.forced_weight_normalization,
                  mp_silu=self.mp_silu, mp_residual=self.mp_residual, mp_pos_enc=self.mp_pos_enc, mp_embedding=self.mp_embedding,
                  weight_normalization=self.weight_normalization, cosine_attention=self.cosine_attention,
                  no_layernorm=self.no_layernorm)
        new.to(device=self._pflat.device, dtype=self._pflat.dtype)
        new.load_state_dict(copy.deepcopy(self.state_dict()))
        for p_new, p_old in zip(new.parameters(), self.parameters()):
            p_new.requires_grad_(p_old.requires_grad)
        new.train(self.training)
        new.gemm_precision, new._loss_scale = self.gemm_precision, self.loss_scale
        return new

    def __getstate__(self):
        st = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        st = dict(st)
        for k in ("_rt", "_gflat", "_gviews", "_anchor_t", "_buffer_grad_scratch"):
            st[k] = {} if k == "_rt" else None
        return st
