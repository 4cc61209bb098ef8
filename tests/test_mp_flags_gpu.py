"""Off forms of the reference README's magnitude-preservation flags (README.md:57-66): --no-use-mp-silu, --no-use-mp-residual, --no-use-mp-pos-enc,
--no-use-mp-embedding, --no-use-weight-normalization, --no-use-cosine-attention, --no-use-no-layernorm (the eighth, --no-use-forced-weight-
normalization, is the one the snapshot's own code defines: tests/test_model_gpu.py).

PARITY UNPINNED.  The reference snapshot hard-wires every flag on (train.py's argparse has none of them) and contains no code
for the off forms, so there is no reference output to compare with.  Each off form is this build's restatement of its README line together
with upstream DiT's form of the same operation (oracle.dit_oracle.DiTConfig: plain SiLU; x + gate * branch; x_embedder(x) + raw sin-cos
table; nn.Embedding; MPLinear without its normalize(); scaled-dot-product attention on unnormalised q, k; a LayerNorm without affine parameters
in front of every modulate()) and the engine is held to THAT restatement here: eval logits, training losses, every parameter gradient, and
what the training forward does to the weights, in both engine precisions.  The on forms are the pinned path of every other test file.
"""
import math

import numpy as np
import pytest
import torch

from conftest import rel_err, sub
from oracle import dit_oracle as O

TINY = dict(depth=2, hidden_size=128, patch_size=2, input_size=16, in_channels=4, num_heads=2, num_classes=10)
FLAGS = ["mp_silu", "mp_residual", "mp_pos_enc", "mp_embedding", "weight_normalization", "cosine_attention", "no_layernorm"]


# ---- CPU: what each off form IS (the restatement's own invariants) -----------------------------------------------------------------------
def _fwd(cfg, sd, seed=3, n=3):
    g = torch.Generator().manual_seed(seed)
    x, t, y = torch.randn(n, 4, 16, 16, generator=g), torch.randint(0, 1000, (n,), generator=g), torch.randint(0, 10, (n,), generator=g)
    return O.dit_forward({k: v.clone() for k, v in sd.items()}, cfg, x, t, y, train=False)


def test_off_forms_of_the_restatement():
    on = O.DiTConfig(**TINY)
    sd = O.init_state_dict(on, seed=5, gains=0.3, perturb_reference=0.2)
    base = _fwd(on, sd)
    # every flag changes the network, and only when switched off does it appear among the constructor arguments
    assert "mp_silu" not in on.to_dict()
    for f in FLAGS:
        cfg = O.DiTConfig(**TINY, **{f: False})
        assert cfg.to_dict()[f] is False
        sd_f = O.init_state_dict(cfg, seed=5, gains=0.3, perturb_reference=0.2)
        assert not torch.allclose(_fwd(cfg, sd_f), base), f
    # plain SiLU is 0.596 x MPSiLU (mp_silu.py:7)
    v = torch.randn(100)
    assert torch.allclose(O.act_fn(O.DiTConfig(**TINY, mp_silu=False))(v), 0.596 * O.mp_silu(v), rtol=1e-6, atol=1e-7)
    # plain residual / plain embedding / raw positional table
    a, b = torch.randn(4, 8), torch.randn(4, 8)
    assert torch.equal(O.residual_sum(O.DiTConfig(**TINY, mp_residual=False), a, b), a + b)
    assert torch.allclose(O.residual_sum(on, a, b), (0.7 * a + 0.3 * b) / math.sqrt(0.58), rtol=1e-6, atol=1e-6)
    tab = {"w": torch.randn(11, 16)}
    assert torch.equal(O.mp_embedding(torch.tensor([3, 3, 10]), tab, "w", train=True, mp=False), tab["w"][[3, 3, 10]])
    # weight normalisation off: the stored weight over sqrt(in_dim); equal to the on form exactly when the rows are already normalised
    lin = {"w": torch.randn(6, 16) * 1.7}
    xin = torch.randn(5, 16)
    assert torch.allclose(O.mp_linear(xin, lin, "w", train=False, wn=False), xin @ lin["w"].T / 4.0, rtol=1e-6, atol=1e-6)
    nrm = {"w": O.normalize(lin["w"])}
    assert torch.allclose(O.mp_linear(xin, nrm, "w", False, wn=False), O.mp_linear(xin, lin, "w", False), rtol=1e-4, atol=1e-5)
    raw = O.init_state_dict(O.DiTConfig(**TINY, mp_pos_enc=False), seed=5)["pos_embed"]
    assert float(raw.abs().max()) <= 1.0 and not torch.allclose(raw, sd["pos_embed"])        # sines and cosines, not normalised rows


def test_train_cli_accepts_the_built_off_forms_and_refuses_the_rest():
    from mapdit_amd import train
    p = train.build_parser()
    a = p.parse_args(["--synthetic", "--results-dir", "/tmp/x", "--no-use-mp-silu", "--no-use-mp-residual", "--no-use-mp-pos-enc", "--no-use-mp-embedding",
                      "--no-use-weight-normalization", "--no-use-cosine-attention", "--no-use-no-layernorm"])
    assert (a.use_mp_silu, a.use_mp_residual, a.use_mp_pos_enc, a.use_mp_embedding, a.use_weight_normalization, a.use_cosine_attention,
            a.use_no_layernorm) == (False,) * 7 and a.use_forced_weight_normalization
    assert set(train.BUILT_OFF_FORMS) == {"mp-silu", "mp-residual", "mp-pos-enc", "mp-embedding", "weight-normalization", "cosine-attention", "no-layernorm"}
    a.in_channels, a.input_size = 4, 32                # (main() fills these in from the data set / --synthetic)
    m = train.get_model(a)
    assert (m.weight_normalization, m.cosine_attention, m.no_layernorm) == (False,) * 3
    # what stays refused: the off forms under the fp32-accurate engine, and the LayerNorm form under rotation modulation
    with pytest.raises(NotImplementedError):
        train.main(["--synthetic", "--results-dir", "/tmp/x", "--no-use-mp-silu", "--precision", "bf16x3", "--num-steps", "1"])
    with pytest.raises(NotImplementedError):
        train.main(["--synthetic", "--results-dir", "/tmp/x", "--no-use-no-layernorm", "--use-rotation-modulation", "--num-steps", "1"])


def test_facade_builds_the_off_forms():
    from mapdit_amd.src.dit import DiT
    m = DiT(**TINY, mp_pos_enc=False, mp_embedding=False)
    assert m.mp_silu and m.mp_residual and not m.mp_pos_enc and not m.mp_embedding
    cfg = O.DiTConfig(**TINY, mp_pos_enc=False)
    assert torch.allclose(m.pos_embed, O.init_state_dict(cfg, seed=0)["pos_embed"], atol=1e-6)
    import copy
    m2 = copy.deepcopy(DiT(**TINY, mp_silu=False)) if torch.cuda.is_available() else None     # (deepcopy re-homes parameters on the device)
    assert m2 is None or not m2.mp_silu


# ---- GPU: the engine against the restatement ------------------------------------------------------------------------------------------
LIMITS = {  # logits, loss, gradient tensor (>= 64 entries), scalar gains (of the largest gain gradient)
    # measured (tiny model): logits 2.4e-3 ... 5.8e-3 (plain residual: no renormalisation, operand rounding grows with the stream), gradient
    # tensors <= 5.8e-3 except the MPScale linears under the plain residual (2.2e-2: 8 x 128 sums with heavy cancellation, norm 2e-4)
    "bf16": (1e-2, 2e-2, 3e-2, 1e-2),
    # measured: logits 3.0e-4 ... 7.2e-4, gradient tensors <= 1.2e-3
    "f16": (1e-3, 2e-3, 3e-3, 5e-3),
}


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("off", [("mp_silu",), ("mp_residual",), ("mp_pos_enc",), ("mp_embedding",), ("weight_normalization",), ("cosine_attention",), ("no_layernorm",), tuple(FLAGS)])
def test_engine_off_forms_match_the_restatement(off, precision):
    _engine_against_the_restatement(TINY, off, precision)


XLW = dict(depth=1, hidden_size=1152, patch_size=2, input_size=32, in_channels=4, num_heads=16, num_classes=10)     # head_dim 72, 256 tokens
BW = dict(depth=2, hidden_size=768, patch_size=2, input_size=32, in_channels=4, num_heads=12, num_classes=10)       # head_dim 64, 256 tokens


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("shape,off", [("XLW", ("cosine_attention",)), ("XLW", ("no_layernorm",)), ("XLW", tuple(FLAGS)), ("BW", tuple(FLAGS)),
                                       ("BW", ("weight_normalization", "cosine_attention", "no_layernorm"))])
def test_engine_off_forms_at_model_widths(shape, off, precision):
    """The same comparison at DiT-XL's and DiT-B's width and token count (one / two blocks): head_dim 72 and 64 MFMA attention with the maximum
    taken out, the LayerNorm rows of 1152 and 768 features, the 256-wide GEMM tiles behind them."""
    _engine_against_the_restatement({"XLW": XLW, "BW": BW}[shape], off, precision, size=32)


def _engine_against_the_restatement(shape, off, precision, size=16):
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.src.dit import DiT
    from oracle.diffusion_oracle import DiffusionOracle
    ltol, losstol, gtol, gaintol = LIMITS[precision]
    dev = "cuda"
    cfg = O.DiTConfig(**shape, **{f: False for f in off})
    sd = O.init_state_dict(cfg, seed=11, gains=0.35, perturb_reference=0.3)
    if "weight_normalization" in off:                 # rows of other lengths than sqrt(in_dim): the off form is then another network (eval reads them as stored)
        gs = torch.Generator().manual_seed(13)
        for k in sd:
            if k.endswith(".weight") and "y_embedder" not in k:
                sd[k] = sd[k] * (0.6 + 0.8 * torch.rand(sd[k].shape[0], 1, generator=gs))
    g = torch.Generator().manual_seed(12)
    n = 4
    x, t, y = torch.randn(n, 4, size, size, generator=g), torch.randint(0, 1000, (n,), generator=g), torch.randint(0, 10, (n,), generator=g)
    noise = torch.randn(n, 4, size, size, generator=g)
    y[1] = 10                                         # the null label's row takes part
    m = DiT(**cfg.to_dict())
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    m.gemm_precision = precision
    with torch.no_grad():
        out = m(x.to(dev), t.to(dev), y.to(dev)).cpu()
        again = m(x.to(dev), t.to(dev), y.to(dev)).cpu()
        want = O.dit_forward({k: v.clone() for k, v in sd.items()}, cfg, x, t, y, train=False)
    assert torch.equal(out, again)
    e = rel_err(out.numpy(), want.numpy())
    print(f"off {'+'.join(off)} [{precision}]: eval logits vs the restatement {e:.3e}")
    assert e < ltol
    # one training step: forced weight normalisation rewrites the linears' weights, and - with mp_embedding off - NOT the label table
    m.train()
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels
    losses = create_diffusion("").training_losses(m, x.to(dev), t.to(dev), dict(y=y.to(dev)), noise=noise.to(dev))
    losses["loss"].mean().backward()
    torch.cuda.synchronize()
    osd = {k: v.clone().requires_grad_(k not in O.BUFFER_KEYS) for k, v in sd.items()}
    drop = torch.zeros(n, dtype=torch.bool)
    ref = DiffusionOracle("").training_losses(lambda xx, tt, **kw: O.dit_forward(osd, cfg, xx, tt, kw["y"], train=True, drop=drop),
                                              x, t, dict(y=y), noise=noise)
    ref["loss"].mean().backward()
    assert rel_err(losses["loss"].detach().cpu().numpy(), ref["loss"].detach().numpy()) < losstol
    gain_scale = max(float(osd[k].grad.abs().max()) for k in osd if "gain_" in k)
    worst = worst_gain = 0.0
    worst_name = ""
    for k, p in m.named_parameters():
        gref = osd[k].grad
        assert rel_err(sub(p.detach()), sub(osd[k].detach())) < 2e-6, f"{k}: weights after the training forward"
        if p.dim() == 0:
            worst_gain = max(worst_gain, abs(float(p.grad) - float(gref)) / (gain_scale + 1e-30))
            assert abs(float(p.grad) - float(gref)) < gaintol * gain_scale + 1e-7, (k, float(p.grad), float(gref))
            continue
        e = rel_err(sub(p.grad), sub(gref))
        if gref.numel() >= 64 and float(gref.norm()) >= 1e-7 and e > worst:
            worst, worst_name = e, f"{k} (norm {float(gref.norm()):.1e})"
        assert e < (gtol if gref.numel() >= 64 else 4 * gtol) or float(gref.norm()) < 1e-7, (k, e)
    print(f"off {'+'.join(off)} [{precision}]: worst gradient tensor vs the restatement {worst:.3e} [{worst_name}], worst gain deviation {worst_gain:.3e}")
    if "mp_embedding" in off:
        k = "y_embedder.embedding.weight"
        assert torch.equal(dict(m.named_parameters())[k].detach().cpu(), sd[k]), "nn.Embedding: the training forward must not touch the table"
        gy = dict(m.named_parameters())[k].grad.cpu()
        used = sorted(set(y.tolist()))
        assert float(gy[[r for r in range(11) if r not in used]].abs().max()) == 0.0, "rows of labels that did not occur get no gradient"


@pytest.mark.gpu
def test_off_forms_are_refused_by_the_fp32_accurate_engine():
    from mapdit_amd import _lib as L
    from mapdit_amd.src.dit import DiT
    m = DiT(**TINY, mp_silu=False).to("cuda").eval()
    m.gemm_precision = "bf16x3"
    with pytest.raises(L.MapditError):
        m(torch.randn(2, 4, 16, 16, device="cuda"), torch.zeros(2, dtype=torch.int64, device="cuda"), torch.zeros(2, dtype=torch.int64, device="cuda"))


@pytest.mark.gpu
def test_harness_trains_and_samples_with_the_off_forms(tmp_path):
    """train.py counterpart with all seven built off forms: two optimiser steps, the flags land in config.yaml, and the sampler CLI
    rebuilds the same network from it (reference sample_fid.py:20-37 reads config.yaml -> get_model)."""
    import os
    import yaml
    from mapdit_amd import sample_fid, train
    exp = train.main(["--synthetic", "--results-dir", str(tmp_path), "--model", "DiT-XS/2", "--num-steps", "2", "--batch-size", "8",
                      "--log-every", "1", "--ckpt-every", "2", "--ema-snapshot-every", "2", "--num-classes", "10", "--verbose", "0",
                      "--num-lin-warmup", "2", "--start-decay", "3", "--no-use-mp-silu", "--no-use-mp-residual", "--no-use-mp-pos-enc", "--no-use-mp-embedding",
                      "--no-use-weight-normalization", "--no-use-cosine-attention", "--no-use-no-layernorm"])
    cfg = yaml.safe_load(open(os.path.join(exp, "config.yaml")))
    assert cfg["use_mp_silu"] is False and cfg["use_mp_embedding"] is False and cfg["use_cosine_attention"] is False and cfg["use_forced_weight_normalization"] is True
    m = train.get_model(cfg)
    assert (m.mp_silu, m.mp_residual, m.mp_pos_enc, m.mp_embedding, m.weight_normalization, m.cosine_attention, m.no_layernorm) == (False,) * 7
    ck = torch.load(os.path.join(exp, "checkpoints", "0000002.pt"), weights_only=True)
    assert all(torch.isfinite(v).all() for v in ck["model"].values())
    path = sample_fid.main(["--result-dir", exp, "--use-vae", "false", "--num-classes", "10", "--num-sampling-steps", "2", "--batch-size", "4",
                            "--num-samples", "4", "--output-file", "off.npz"])
    arr = np.load(path)["arr_0"]
    assert arr.shape == (4, 32, 32, 4) and arr.dtype == np.uint8
