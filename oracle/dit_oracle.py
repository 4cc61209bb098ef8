"""CPU oracle for the MaP-DiT network (TEST INFRASTRUCTURE — see oracle/__init__.py).

A functional, state-dict-driven restatement of the reference network in plain
PyTorch (any float dtype, CPU).  Gradients come from torch autograd over this
restatement.  Every function cites the reference file:line it follows
(paths relative to /root/reference).

The state-dict key names are the reference's so that one seeded
state dict can be loaded into the reference (golden generation), this oracle,
and the HIP product alike.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Dict, Optional

import numpy as np
import torch

Tensor = torch.Tensor

MP_SILU_DIV = 0.596          # src/basic/mp_silu.py:7
NORM_EPS = 1e-4              # src/utils.py:19
RESIDUAL_T = 0.3             # src/blocks/dit_block.py:35-36
EMBED_T = 0.5                # src/dit.py:84,88
FOURIER_DIM = 256            # src/blocks/timestep_embedder.py:30
SCALE_DIM = 8                # src/blocks/final_layer.py:13


@dataclass(frozen=True)
class DiTConfig:
    """Constructor arguments of the reference DiT (src/dit.py:15-27)."""
    depth: int
    hidden_size: int
    patch_size: int
    input_size: int = 32
    in_channels: int = 3
    num_heads: int = 16
    mlp_ratio: float = 4.0
    class_dropout_prob: float = 0.1
    num_classes: int = 1000
    learn_sigma: bool = True
    # NOT in the reference snapshot (its README.md:1-3 describes it): the block conditioning by rotation modulation.
    # PARITY UNPINNED - see modulate_rot.
    rotation_modulation: bool = False
    # Off forms of four of the README's --use-* flags (reference README.md:57-66).  NOT in the reference snapshot, which hard-wires every
    # magnitude-preserving feature on and contains none of the layers the off forms name: PARITY UNPINNED - each off form
    # below is this build's restatement of one README line together with upstream DiT's form of the same operation, and changes exactly
    # the named operation.  True (the default) is the snapshot's arithmetic.
    mp_silu: bool = True          # False: plain SiLU (no division by 0.596) wherever the snapshot has MPSiLU
    mp_residual: bool = True      # False: x + gate * branch  instead of  mp_sum(x, gate * branch, 0.3)
    mp_pos_enc: bool = True       # False: x_embedder(x) + pos_embed with the raw sin-cos table instead of mp_sum(.., normalize(table), 0.5)
    mp_embedding: bool = True     # False: nn.Embedding (plain row gather, no row normalisation) for the class labels
    # README.md:60 --use-weight-normalization off (same status: unpinned): MPLinear / MPLinearChunk multiply by W * gain / sqrt(in_dim) -
    # mp_linear.py:44,74 without their normalize(); the training forward's in-place rewrite (mp_linear.py:38-40) is its own flag and stays
    weight_normalization: bool = True
    # README.md:58 --use-cosine-attention off (unpinned): attention.py:42-43 (normalize(q), normalize(k)) dropped, the SDPA scale unchanged
    cosine_attention: bool = True
    # README.md:64 --use-no-layernorm off (unpinned) = WITH a LayerNorm: torch.nn.functional.layer_norm(x, (D,), eps=1e-6), no affine (upstream
    # DiT's norm1 / norm2 / norm_final), in front of every modulate() (dit_block.py:35-36, final_layer.py:55)
    no_layernorm: bool = True

    @property
    def grid(self) -> int:
        return self.input_size // self.patch_size

    @property
    def tokens(self) -> int:
        return self.grid * self.grid

    @property
    def patch_dim(self) -> int:
        return self.patch_size * self.patch_size * self.in_channels

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def mlp_hidden(self) -> int:
        return int(self.hidden_size * self.mlp_ratio)

    def to_dict(self):
        """Constructor kwargs; the rotation switch only appears when set (the reference's constructor does not know it)."""
        d = asdict(self)
        if not d["rotation_modulation"]:
            del d["rotation_modulation"]
        for k in ("mp_silu", "mp_residual", "mp_pos_enc", "mp_embedding", "weight_normalization", "cosine_attention", "no_layernorm"):      # (likewise: only when switched off)
            if d[k]:
                del d[k]
        return d


# name -> (depth, hidden, heads); src/models.py:4-47
_FAMILIES = {"XL": (28, 1152, 16), "L": (24, 1024, 16), "B": (12, 768, 12),
             "S": (12, 384, 6), "XS": (6, 256, 4)}


def model_config(name: str, **kwargs) -> DiTConfig:
    """'DiT-B/2' -> DiTConfig, mirroring src/models.py:50-56."""
    fam, patch = name[len("DiT-"):].split("/")
    depth, hidden, heads = _FAMILIES[fam]
    return DiTConfig(depth=depth, hidden_size=hidden, patch_size=int(patch), num_heads=heads, **kwargs)


# ----------------------------------------------------------------------------------------------
# MP primitives
# ----------------------------------------------------------------------------------------------

def normalize(x: Tensor, eps: float = NORM_EPS) -> Tensor:
    """src/utils.py:19-23 (and chunk_normalize :26-34, which is identical per row)."""
    norm = torch.linalg.vector_norm(x, dim=-1, keepdim=True)
    return x * math.sqrt(x.shape[-1]) / (norm + eps)


def mp_sum(a: Tensor, b: Tensor, t) -> Tensor:
    """src/utils.py:15-16.  For a tensor ``t`` the denominator goes through math.sqrt and is
    therefore a detached Python float."""
    if isinstance(t, Tensor):
        den = math.sqrt(float(((1 - t) ** 2 + t ** 2).detach()))
    else:
        den = math.sqrt((1 - t) ** 2 + t ** 2)
    return a.lerp(b, t) / den


def modulate(x: Tensor, shift: Tensor, scale: Tensor, t) -> Tensor:
    """src/utils.py:11-12."""
    return mp_sum(x * scale.unsqueeze(1), shift.unsqueeze(1), t)


def modulate_rot(x: Tensor, theta: Tensor, scale: Tensor, gain) -> Tensor:
    """Rotation modulation - PARITY UNPINNED: the reference snapshot does not contain it; this is the build's own
    restatement of what its README.md:1-3 announces ("rotation modulation ... ~5.4 % fewer parameters", arXiv 2505.19122).
    The parameter saving pins the shape: the two D-wide shift chunks of a block's modulation linear become two D/2-wide angle
    chunks (6 D^2 -> 5 D^2 of a block's 18 D^2: -5.6 %).  Semantics chosen here: the scaled features are rotated pairwise,
        (y[2i], y[2i+1]) = R(gain * theta[i]) (scale[2i] x[2i], scale[2i+1] x[2i+1]),
    with the block's learnable gain (initialised to 0, as in the snapshot) scaling the angle.  A rotation preserves the pair's
    magnitude, so no mp_sum renormalisation is needed, and at gain = 0 (or theta = 0) the result is x * scale - exactly what the
    snapshot's modulate() gives at gain = 0.  x [N,T,D], theta [N,D/2], scale [N,D]."""
    a = x * scale.unsqueeze(1)
    ang = (gain * theta).unsqueeze(1)
    c, s = torch.cos(ang), torch.sin(ang)
    a0, a1 = a[..., 0::2], a[..., 1::2]
    return torch.stack([c * a0 - s * a1, s * a0 + c * a1], dim=-1).reshape(x.shape)


def mp_silu(x: Tensor) -> Tensor:
    """src/basic/mp_silu.py:5-7."""
    return torch.nn.functional.silu(x) / MP_SILU_DIV


def act_fn(cfg):
    """The block / conditioning nonlinearity: MPSiLU (snapshot) or, with ``mp_silu=False`` (README.md:63; unpinned), plain SiLU."""
    return mp_silu if getattr(cfg, "mp_silu", True) else torch.nn.functional.silu


def residual_sum(cfg, x: Tensor, branch: Tensor) -> Tensor:
    """dit_block.py:35-36: mp_sum(x, gate * branch, 0.3); with ``mp_residual=False`` (README.md:62; unpinned) the plain residual."""
    return mp_sum(x, branch, RESIDUAL_T) if getattr(cfg, "mp_residual", True) else x + branch


def patchify(x: Tensor, p: int) -> Tensor:
    """src/utils.py:37-46: b c (h p1) (w p2) -> b (h w) (p1 p2 c)."""
    b, c, hh, ww = x.shape
    h, w = hh // p, ww // p
    return x.reshape(b, c, h, p, w, p).permute(0, 2, 4, 3, 5, 1).reshape(b, h * w, p * p * c)


def unpatchify(x: Tensor, input_size: int, p: int) -> Tensor:
    """src/utils.py:49-59: b (h w) (p1 p2 c) -> b c (h p1) (w p2)."""
    b = x.shape[0]
    h = w = input_size // p
    c = x.shape[-1] // (p * p)
    return x.reshape(b, h, w, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(b, c, h * p, w * p)


def _ident(x: Tensor) -> Tensor:
    return x


def bf16_round(x: Tensor) -> Tensor:
    """Round-to-nearest-even to bfloat16 and back: the storage rounding of the engine's GEMM / attention operands.
    Passing it as ``rnd`` makes the oracle emulate the engine's precision plan (bf16 operands, fp32 accumulation,
    fp32 residual stream and conditioning vectors) so that what is left between the two is accumulation order only."""
    return x.bfloat16().to(x.dtype)


def f16_round(x: Tensor) -> Tensor:
    """Round-to-nearest-even to IEEE fp16 and back: the storage rounding of the fp16 engine (gemm_precision = "f16")."""
    return x.half().to(x.dtype)


class _RoundFwdBwd(torch.autograd.Function):
    """y = rnd(x) in the forward, dx = rnd(dy * scale) / scale in the backward: a tensor that the engine would STORE in 16 bits in
    both directions (a residual-stream checkpoint and the gradient that flows back through it, the latter under the loss scale)."""

    @staticmethod
    def forward(ctx, x, rnd, scale):
        ctx.rnd, ctx.scale = rnd, scale
        return rnd(x)

    @staticmethod
    def backward(ctx, dy):
        return ctx.rnd(dy * ctx.scale) / ctx.scale, None, None


class EnginePlan:
    """The bf16 engine's precision plan as a per-site rounding policy: bf16 GEMM / attention operands on the token path, an
    fp32-accurate conditioning path (timestep MLP, modulation linears, MPScale linears: the engine runs those [samples, D]
    products on two-term split operands).  ``EnginePlan(f16_round)``: the same plan for the fp16 engine.
    ``residual16`` (round 4, a measurement: tools/precision_residual16.py): the residual-stream checkpoints X[0..2L] - fp32 in the
    engine - rounded like the operands, forward and (under ``grad_scale``) backward: site "res"."""
    COND = ("t0", "t2", "mod", "fmod", "scale")

    def __init__(self, rnd=bf16_round, residual16: bool = False, grad_scale: float = 1.0):
        self.rnd = rnd
        self.residual16, self.grad_scale = residual16, grad_scale

    def at(self, site: str):
        if site == "res":
            if not self.residual16:
                return _ident
            return lambda v: _RoundFwdBwd.apply(v, self.rnd, self.grad_scale)
        return _ident if site[:2] in ("x:", "w:") and site[2:] in self.COND else self.rnd


engine_plan = EnginePlan()
engine_plan_f16 = EnginePlan(f16_round)


def _at(rnd, site: str):
    """A rounding policy may differ per site (tools/precision_rank.py ranks the engine's bf16 roundings with one): an object with
    ``at(site) -> callable``; a plain callable applies everywhere.  Sites: "w:<layer>", "x:<layer>" (GEMM operands; layer in qkv,
    proj, fc1, fc2, mod, t0, t2, flin, fmod, scale), "v", "qk" (the normalised q, k), "p" (exp(logits))."""
    if hasattr(rnd, "at"):
        return rnd.at(site)
    return _ident if site == "res" else rnd          # (a plain callable rounds GEMM / attention operands only)


_LAYER_OF = {"t_embedder.mlp.net.0": "t0", "t_embedder.mlp.net.2": "t2", "attn.qkv_proj": "qkv", "attn.out_proj": "proj",
             "mlp.net.0": "fc1", "mlp.net.2": "fc2", "final_layer.modulation.1": "fmod", "modulation.1": "mod",
             "final_layer.linear": "flin", "scale.linear": "scale"}


def _layer_of(key: str) -> str:
    for frag, name in _LAYER_OF.items():                   # (insertion order: the more specific fragments first)
        if frag in key:
            return name
    return "other"


def mp_linear(x: Tensor, sd: Dict[str, Tensor], key: str, train: bool, rnd=_ident, wn: bool = True) -> Tensor:
    """MPLinear / MPLinearChunk forward (src/basic/mp_linear.py:31-46, 67-75), gain == 1.
    Training forward first overwrites the stored weight with its normalised value (F9).
    ``wn=False`` (README.md:60 off form; unpinned): line 44 / 74 without normalize() - the stored weight over sqrt(in_dim)."""
    w = sd[key]
    if train:
        with torch.no_grad():
            w.copy_(normalize(w))
    w_eff = (normalize(w) if wn else w) / math.sqrt(w.shape[1])
    layer = _layer_of(key)
    return torch.nn.functional.linear(_at(rnd, "x:" + layer)(x), _at(rnd, "w:" + layer)(w_eff))


def mp_embedding(idx: Tensor, sd: Dict[str, Tensor], key: str, train: bool, mp: bool = True) -> Tensor:
    """src/basic/mp_embedding.py:15-24.  ``mp=False`` (README.md:66 off form; unpinned): nn.Embedding's plain gather."""
    w = sd[key]
    if not mp:
        return w[idx]
    if train:
        with torch.no_grad():
            w.copy_(normalize(w))
    return normalize(w)[idx]


def sincos_pos_embed(dim: int, grid: int) -> np.ndarray:
    """src/pos_embed.py:4-60 (MAE table, float64).  First half of ``dim`` encodes the w index
    ("w goes first", pos_embed.py:15), second half the h index; each half is [sin | cos]."""
    gw, gh = np.meshgrid(np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32))

    def one_d(d, pos):
        omega = np.arange(d // 2, dtype=np.float64) / (d / 2.0)
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    return np.concatenate([one_d(dim // 2, gw), one_d(dim // 2, gh)], axis=1)


# ----------------------------------------------------------------------------------------------
# State dict
# ----------------------------------------------------------------------------------------------

def param_shapes(cfg: DiTConfig) -> Dict[str, tuple]:
    """Every state_dict entry of the reference model and its shape, in the
    order nn.Module.state_dict() yields them."""
    D, P, Hm = cfg.hidden_size, cfg.patch_dim, cfg.mlp_hidden
    nch = 2 if cfg.learn_sigma else 1
    use_cfg_emb = 1 if cfg.class_dropout_prob > 0 else 0
    s: Dict[str, tuple] = {}
    s["pos_embed"] = (1, cfg.tokens, D)
    s["x_embedder.weight"] = (D, P + 1)
    s["t_embedder.mlp.net.0.weight"] = (D, FOURIER_DIM)
    s["t_embedder.mlp.net.2.weight"] = (D, D)
    s["t_embedder.embedding.scale"] = (FOURIER_DIM,)
    s["t_embedder.embedding.shift"] = (FOURIER_DIM,)
    s["y_embedder.embedding.weight"] = (cfg.num_classes + use_cfg_emb, D)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        s[b + "gain_msa"] = ()
        s[b + "gain_mlp"] = ()
        s[b + "attn.qkv_proj.weight"] = (3 * D, D)
        s[b + "attn.out_proj.weight"] = (D, D)
        s[b + "mlp.net.0.weight"] = (Hm, D)
        s[b + "mlp.net.2.weight"] = (D, Hm)
        s[b + "modulation.1.weight"] = ((5 if cfg.rotation_modulation else 6) * D, D)
    s["final_layer.gain_mod"] = ()
    s["final_layer.linear.weight"] = (nch * cfg.patch_size ** 2 * cfg.in_channels, D)
    s["final_layer.modulation.1.weight"] = (2 * D, D)
    s["final_layer.mean_scale.linear.weight"] = (SCALE_DIM, D)
    s["final_layer.mean_scale.reference"] = (SCALE_DIM,)
    if cfg.learn_sigma:
        s["final_layer.sigma_scale.linear.weight"] = (SCALE_DIM, D)
        s["final_layer.sigma_scale.reference"] = (SCALE_DIM,)
    return s


BUFFER_KEYS = ("pos_embed", "t_embedder.embedding.scale", "t_embedder.embedding.shift")


def init_state_dict(cfg: DiTConfig, seed: int = 0, gains: Optional[float] = None,
                    perturb_reference: float = 0.0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Seeded state dict with the reference's init distributions (SURVEY §8c recipe step 1):
    weights N(0,1) (mp_linear.py:23,63; mp_embedding.py:13), gains 0 (dit_block.py:28-29,
    final_layer.py:47), mean_scale.reference 1 / sigma_scale.reference 0 (final_layer.py:18,50-51),
    Fourier scale 2pi*N(0,1), shift 2pi*U(0,1) (timestep_embedder.py:12-16), pos_embed the
    normalised sin-cos table (dit.py:46-48).  ``gains`` / ``perturb_reference`` give the
    non-default fixtures that exercise the shift path and F8."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for k, shp in param_shapes(cfg).items():
        if k == "pos_embed":
            pe = torch.from_numpy(sincos_pos_embed(cfg.hidden_size, cfg.grid)).float().unsqueeze(0)
            sd[k] = normalize(pe) if cfg.mp_pos_enc else pe          # (off form: the raw table, as upstream DiT adds it)
        elif k.endswith("embedding.scale"):
            sd[k] = (2 * math.pi * torch.randn(shp, generator=g)).float()
        elif k.endswith("embedding.shift"):
            sd[k] = (2 * math.pi * torch.rand(shp, generator=g)).float()
        elif k.endswith("mean_scale.reference"):
            sd[k] = torch.ones(shp) + perturb_reference * torch.randn(shp, generator=g)
        elif k.endswith("sigma_scale.reference"):
            sd[k] = torch.zeros(shp) + perturb_reference * torch.randn(shp, generator=g)
        elif "gain_" in k:
            if gains is None:
                sd[k] = torch.zeros(shp)
            else:
                sd[k] = gains * (0.5 + torch.rand(shp, generator=g))
        else:
            sd[k] = torch.randn(shp, generator=g)
    return {k: v.to(dtype).clone() for k, v in sd.items()}


# ----------------------------------------------------------------------------------------------
# Network
# ----------------------------------------------------------------------------------------------

def _rec(trace, key, value):
    if trace is not None:
        trace[key] = value.detach().clone()


def attention(x: Tensor, sd, prefix: str, cfg: DiTConfig, train: bool, rnd=_ident, trace=None) -> Tensor:
    """src/layers/attention.py:29-51: cosine attention, logits = sqrt(hd)*cos(q,k)."""
    B, T, D = x.shape
    H, hd = cfg.num_heads, cfg.head_dim
    qkv = mp_linear(x, sd, prefix + "qkv_proj.weight", train, rnd, wn=getattr(cfg, "weight_normalization", True))
    # engine: head_dim 64 with 64/128/256 tokens normalises q, k from the fp32 accumulators inside the QKV GEMM's epilogue
    # (one rounding, after the normalisation); the generic attention path stores qkv in bf16 first
    if not (hd == 64 and T in (64, 128, 256)):
        qkv = _at(rnd, "qkv")(qkv)
    _rec(trace, prefix + "qkv", qkv)
    q, k, v = qkv.chunk(3, dim=-1)
    v = _at(rnd, "v")(v)
    q = q.view(B, T, H, hd).transpose(1, 2)
    k = k.view(B, T, H, hd).transpose(1, 2)
    v = v.view(B, T, H, hd).transpose(1, 2)
    cosine = getattr(cfg, "cosine_attention", True)
    if cosine:                                        # attention.py:42-43; the off form (README.md:58, unpinned) feeds q, k as they are
        q, k = normalize(q), normalize(k)
    q, k = _at(rnd, "qk")(q), _at(rnd, "qk")(k)
    _rec(trace, prefix + "qn", q); _rec(trace, prefix + "kn", k); _rec(trace, prefix + "v", v)
    logits = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if rnd is _ident:
        out = torch.softmax(logits, dim=-1) @ v
    else:   # the kernels keep exp(logit) un-normalised (cosine logits are bounded; off form: the row maximum taken out), round it for the PV product
        p = torch.exp(logits if cosine else logits - logits.amax(-1, keepdim=True).detach())
        out = (_at(rnd, "p")(p) @ v) / p.sum(-1, keepdim=True)
    out = out.transpose(1, 2).reshape(B, T, D)
    _rec(trace, prefix + "o", _at(rnd, "x:proj")(out))
    return mp_linear(out, sd, prefix + "out_proj.weight", train, rnd, wn=getattr(cfg, "weight_normalization", True))


def mlp(x: Tensor, sd, prefix: str, train: bool, rnd=_ident, trace=None, act=mp_silu, wn: bool = True) -> Tensor:
    """src/layers/mlp.py:16-25."""
    h = act(mp_linear(x, sd, prefix + "net.0.weight", train, rnd, wn=wn))
    _rec(trace, prefix + "hact", _at(rnd, "x:fc2")(h))
    return mp_linear(h, sd, prefix + "net.2.weight", train, rnd, wn=wn)


class _WithFull:
    """A residual checkpoint as stored (``stored``) together with the unrounded value its producer held (``full``)."""

    def __init__(self, stored, full):
        self.stored, self.full = stored, full


def pre_norm(cfg: DiTConfig, x: Tensor) -> Tensor:
    """What stands in front of modulate(): nothing in the snapshot ("no layernorm", dit_block.py:35-36, final_layer.py:55); with
    ``no_layernorm=False`` (README.md:64 off form; unpinned) upstream DiT's LayerNorm without affine parameters, eps 1e-6."""
    if getattr(cfg, "no_layernorm", True):
        return x
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), eps=1e-6)


def dit_block(x: Tensor, c: Tensor, sd, i: int, cfg: DiTConfig, train: bool, rnd=_ident, trace=None) -> Tensor:
    """src/blocks/dit_block.py:32-37."""
    p = f"blocks.{i}."
    mod = mp_linear(act_fn(cfg)(c), sd, p + "modulation.1.weight", train, rnd, wn=getattr(cfg, "weight_normalization", True))
    _rec(trace, p + "mod", mod)
    if cfg.rotation_modulation:             # (theta, scale, gate) x 2 with D/2-wide angle chunks; see modulate_rot
        D = x.shape[-1]
        th_a, sc_a, g_a, th_m, sc_m, g_m = mod.split([D // 2, D, D, D // 2, D, D], dim=-1)
        mod_a = lambda v: modulate_rot(v, th_a, sc_a, sd[p + "gain_msa"])
        mod_m = lambda v: modulate_rot(v, th_m, sc_m, sd[p + "gain_mlp"])
    else:
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=-1)
        mod_a = lambda v: modulate(pre_norm(cfg, v), sh_a, sc_a, sd[p + "gain_msa"])
        mod_m = lambda v: modulate(pre_norm(cfg, v), sh_m, sc_m, sd[p + "gain_mlp"])
    # x: the stored checkpoint (site "res": identity in every shipped plan); x_full: the value the epilogue that produced it still
    # holds in fp32 - the next branch's modulate is fused into that epilogue and sees the unrounded value
    x_full = x.full if isinstance(x, _WithFull) else x
    x = x.stored if isinstance(x, _WithFull) else x
    xm = mod_a(x_full)
    _rec(trace, p + "xm", _at(rnd, "x:qkv")(xm))
    x_full = residual_sum(cfg, x, g_a.unsqueeze(1) * attention(xm, sd, p + "attn.", cfg, train, rnd, trace))
    x = _at(rnd, "res")(x_full)
    _rec(trace, p + "xmid", x)
    xm2 = mod_m(x_full)
    _rec(trace, p + "xm2", _at(rnd, "x:fc1")(xm2))
    x_full = residual_sum(cfg, x, g_m.unsqueeze(1) * mlp(xm2, sd, p + "mlp.", train, rnd, trace, act=act_fn(cfg), wn=getattr(cfg, "weight_normalization", True)))
    x = _at(rnd, "res")(x_full)
    _rec(trace, p + "xout", x)
    return _WithFull(x, x_full) if x is not x_full else x


def mp_scale(c: Tensor, sd, prefix: str, train: bool, rnd=_ident, wn: bool = True) -> Tensor:
    """src/blocks/final_layer.py:20-22."""
    angle = torch.matmul(mp_linear(c, sd, prefix + "linear.weight", train, rnd, wn=wn), sd[prefix + "reference"]) / math.sqrt(SCALE_DIM)
    return torch.sigmoid(angle)


def final_layer(x: Tensor, c: Tensor, sd, cfg: DiTConfig, train: bool, rnd=_ident, trace=None):
    """src/blocks/final_layer.py:53-61.  Call order of the MPLinears follows the reference
    (modulation, linear, mean_scale, sigma_scale) — it matters only for forced-WN side effects,
    which are per-weight and order independent."""
    p = "final_layer."
    shift, scale = mp_linear(act_fn(cfg)(c), sd, p + "modulation.1.weight", train, rnd, wn=getattr(cfg, "weight_normalization", True)).chunk(2, dim=-1)
    x_mod = modulate(pre_norm(cfg, x), shift, scale, sd[p + "gain_mod"])
    _rec(trace, p + "xmod", _at(rnd, "x:flin")(x_mod))
    out = mp_linear(x_mod, sd, p + "linear.weight", train, rnd, wn=getattr(cfg, "weight_normalization", True))
    _rec(trace, p + "lin", out)
    if cfg.learn_sigma:
        mean, sigma = out.chunk(2, dim=-1)
        return (mean * mp_scale(c, sd, p + "mean_scale.", train, rnd, wn=getattr(cfg, "weight_normalization", True)).view(-1, 1, 1),
                sigma * mp_scale(c, sd, p + "sigma_scale.", train, rnd, wn=getattr(cfg, "weight_normalization", True)).view(-1, 1, 1))
    return out * mp_scale(c, sd, p + "mean_scale.", train, rnd, wn=getattr(cfg, "weight_normalization", True))


def effective_labels(y: Tensor, cfg: DiTConfig, train: bool, drop: Optional[Tensor]) -> Tensor:
    """src/blocks/label_embedder.py:19-34.  ``drop`` is the recorded mask ``rand(N) < p``
    (RNG streams cannot be matched across back ends, so it is an explicit input)."""
    if train and cfg.class_dropout_prob > 0:
        assert drop is not None, "training forward needs the label-drop mask"
        return torch.where(drop, torch.full_like(y, cfg.num_classes), y)
    return y


def dit_forward(sd: Dict[str, Tensor], cfg: DiTConfig, x: Tensor, t: Tensor, y: Tensor,
                train: bool = False, drop: Optional[Tensor] = None, rnd=_ident, trace: Optional[dict] = None) -> Tensor:
    """src/dit.py:70-105.  ``rnd`` (default: identity = the reference's fp32 arithmetic) is applied wherever the HIP
    engine stores a GEMM / attention operand in bf16; see bf16_round().  ``trace``: a dict that receives named
    intermediates (for the stage-by-stage parity test against mapdit_engine_peek)."""
    dt = sd["x_embedder.weight"].dtype
    h = patchify(x.to(dt), cfg.patch_size)
    h = torch.cat([h, torch.ones_like(h[:, :, :1])], dim=-1)
    if cfg.mp_pos_enc:
        h = mp_sum(mp_linear(h, sd, "x_embedder.weight", train, wn=getattr(cfg, "weight_normalization", True)), sd["pos_embed"], EMBED_T)      # fp32 kernel in the engine
    else:                                                        # README.md:65 off form (unpinned): upstream DiT's plain addition
        h = mp_linear(h, sd, "x_embedder.weight", train, wn=getattr(cfg, "weight_normalization", True)) + sd["pos_embed"]

    four = torch.cos(torch.outer(t.to(dt), sd["t_embedder.embedding.scale"]) + sd["t_embedder.embedding.shift"])
    four = math.sqrt(2) * four                                   # timestep_embedder.py:18-21
    _rec(trace, "x0", h); _rec(trace, "four", _at(rnd, "x:t0")(four))
    temb = mp_linear(four, sd, "t_embedder.mlp.net.0.weight", train, rnd, wn=getattr(cfg, "weight_normalization", True))
    temb = mp_linear(act_fn(cfg)(temb), sd, "t_embedder.mlp.net.2.weight", train, rnd, wn=getattr(cfg, "weight_normalization", True))
    yemb = mp_embedding(effective_labels(y, cfg, train, drop), sd, "y_embedder.embedding.weight", train, mp=cfg.mp_embedding)
    c = mp_sum(temb, yemb, EMBED_T)
    _rec(trace, "temb", temb); _rec(trace, "c", c)

    h0 = h
    h = _at(rnd, "res")(h0)
    if h is not h0:
        h = _WithFull(h, h0)
    for i in range(cfg.depth):
        h = dit_block(h, c, sd, i, cfg, train, rnd, trace)
    if isinstance(h, _WithFull):
        h = h.full                                               # the final modulate is fused into the last block's epilogue too

    if cfg.learn_sigma:
        mean, sigma = final_layer(h, c, sd, cfg, train, rnd, trace)
        return torch.cat([unpatchify(mean, cfg.input_size, cfg.patch_size),
                          unpatchify(sigma, cfg.input_size, cfg.patch_size)], dim=1)
    return unpatchify(final_layer(h, c, sd, cfg, train, rnd, trace), cfg.input_size, cfg.patch_size)


def dit_forward_with_cfg(sd, cfg: DiTConfig, x: Tensor, t: Tensor, y: Tensor, cfg_scale: float,
                         train: bool = False, drop: Optional[Tensor] = None) -> Tensor:
    """src/dit.py:107-118."""
    half = x[: len(x) // 2]
    out = dit_forward(sd, cfg, torch.cat([half, half], dim=0), t, y, train, drop)
    C = cfg.in_channels
    eps, rest = out[:, :C], out[:, C:]
    cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
    half_eps = uncond + cfg_scale * (cond - uncond)
    return torch.cat([torch.cat([half_eps, half_eps], dim=0), rest], dim=1)


# ----------------------------------------------------------------------------------------------
# Optimiser / LR schedule / EMA  
# ----------------------------------------------------------------------------------------------

def lr_lambda(step: int, num_lin_warmup: int, start_decay: int) -> float:
    """train.py:179-197."""
    if step + 1 < num_lin_warmup:
        return (step + 1) / num_lin_warmup
    if step >= start_decay:
        return 1.0 / math.sqrt(max(step / start_decay, 1))
    return 1.0


def std_to_gamma(std: float) -> float:
    """src/ema.py:10-20: largest real root of g^3 + 7g^2 + (16 - s^-2) g + (12 - s^-2)."""
    t = float(std) ** -2
    return float(np.roots([1, 7, 16 - t, 12 - t]).real.max())


def ema_beta(std: float, t: int) -> float:
    """src/ema.py:33-40."""
    return (1 - 1 / t) ** (std_to_gamma(std) + 1)


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              beta1: float = 0.9, beta2: float = 0.99, eps: float = 1e-8) -> None:
    """torch.optim.Adam single-tensor update as configured at train.py:57 (no weight decay,
    no amsgrad); ``step`` counts from 1."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
