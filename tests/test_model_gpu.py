"""End-to-end parity of the HIP engine behind the reference API (DIT_MODELS / create_diffusion) against
the golden vectors captured from the reference (tests/golden/*.npz) and against the CPU oracle.

Tolerances.  The reference computes in fp32; the engine's GEMM / attention operands are bf16 (fp32 accumulate,
fp32 residual stream), so the comparison with the fp32 goldens carries bf16 operand rounding (2^-9 per element,
accumulating over depth).  Every limit below is at most twice what the engine measures on the MI355X (round 3:
profiles/r03_parity_measured.log), so that a regression which doubles an error fails: tiny fixtures logits 2.7e-3 ... 3.5e-3
(limit 7e-3), losses 4e-5 ... 7e-5 (limit 1.5e-4 there, 6e-3 on the named models: s2_n4 measures 2.9e-3), gradients 4.4e-3 ...
5.0e-3 per tensor (limit 1e-2).  Integer/bit-exact items (label drop, timestep maps) are checked exactly, and
forced-weight-norm rewritten weights (fp32 path) at 2e-6.  The fp16 engine's (8x tighter) limits: tests/test_f16_gpu.py.
"""
import numpy as np
import pytest
import torch

from conftest import golden_cfg, golden_state_dict, load_golden, rel_err, sub

pytestmark = pytest.mark.gpu
DEV = "cuda"

LOGIT_TOL = 7e-3          # tiny fixtures (depth 2-3): measured <= 3.5e-3
LOSS_TOL = 1.5e-4         # measured <= 7.2e-5
GRAD_TOL = 1e-2           # per tensor of >= 64 entries: measured <= 5.0e-3
GAIN_TOL = 8e-3           # scalar gain gradients, deviation relative to the largest gain gradient of the model: measured <= 4.1e-3 (s2_n4)
SMALL_GRAD_TOL = 2e-2     # tensors of < 64 entries (MPScale references: sums with heavy cancellation): measured <= 4.9e-3


def build(g, train=False):
    from mapdit_amd.src.dit import DiT
    cfg = golden_cfg(g)
    sd = golden_state_dict(g, cfg)
    m = DiT(**cfg.to_dict())
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    m.train(train)
    m.gemm_precision = "bf16"        # this file's limits are the bf16 engine's (the default engine, "f16", is held to 1e-3 in test_f16_gpu.py)
    return m, cfg, sd


def dev(g, *names):
    return [torch.from_numpy(g[n]).to(DEV) for n in names]


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c"])
def test_eval_forward_matches_reference(name):
    g = load_golden(name)
    m, cfg, _ = build(g)
    x, t, y = dev(g, "x", "t", "y")
    with torch.no_grad():
        out = m(x, t, y)
    e = rel_err(out.cpu().numpy(), g["eval_out"])
    print(f"{name}: eval logits rel err {e:.3e}")
    assert e < LOGIT_TOL
    # calling again must give the same bits (cached weight images, no hidden state)
    with torch.no_grad():
        out2 = m(x, t, y)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("name", ["tiny_a", "tiny_c", "s2_n2"])
def test_eval_forward_matches_bf16_emulating_oracle(name):
    """The gap to the fp32 goldens above is operand rounding and nothing else: when the oracle rounds to bf16 at the
    points where the engine stores bf16 (GEMM / attention operands of the token path; oracle.engine_plan), the engine is 2-10x closer to
    it than to the fp32 reference.  It cannot be much closer than ~2^-9: two implementations of the same precision plan
    that differ only in fp32 accumulation order (relative difference d) disagree after ONE more bf16 rounding by about
    sqrt(d * 2^-8) (a fraction d/2^-8 of the elements lands on the other side of a rounding boundary), so 1e-7 becomes
    2e-5, then 3e-4, ... and saturates near the bf16 rounding noise itself.  test_forward_stage_by_stage_* shows exactly
    that growth, starting from bit-identical Fourier features.  Tolerance 4e-3 norm-wise."""
    from oracle import dit_oracle as O
    g = load_golden(name)
    m, cfg, sd = build(g)
    x, t, y = dev(g, "x", "t", "y")
    with torch.no_grad():
        out = m(x, t, y).cpu()
        ref_emul = O.dit_forward(sd, cfg, x.cpu(), t.cpu(), y.cpu(), train=False, rnd=O.engine_plan)
        ref_fp32 = O.dit_forward(sd, cfg, x.cpu(), t.cpu(), y.cpu(), train=False)
    e_emul = rel_err(out.numpy(), ref_emul.numpy())
    e_fp32 = rel_err(out.numpy(), ref_fp32.numpy())
    print(f"{name}: vs bf16-emulating oracle {e_emul:.3e}, vs fp32 oracle {e_fp32:.3e}")
    assert e_emul < 4e-3
    assert e_emul < 0.85 * e_fp32


PRECISE_TOL = 1e-3      # BASELINE.json north_star: "forward logits within 1e-3 rel of reference"


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c", "s4_n8", "s2_n2", "s2_n4", "b2_n2", "xl_d1", "xl2_n2", "tiny_p8"])
def test_precise_forward_within_1e3_of_reference(name):
    """gemm_precision="bf16x3": fp32 activations and every linear on the MFMA GEMM kernel with hi+lo split operands.
    The logits must be within 1e-3 (norm-wise relative) of the reference's fp32 forward - the fixtures' ``eval_out`` were
    produced by the reference itself (tests/golden/make_golden.py)."""
    g = load_golden(name)
    m, cfg, _ = build(g)
    m.gemm_precision = "bf16x3"
    x, t, y = dev(g, "x", "t", "y")
    with torch.no_grad():
        out = m(x, t, y)
        again = m(x, t, y)
    ref = g["eval_out"]
    got = sub(out) if ref.shape != tuple(out.shape) else out.cpu().numpy()
    e = rel_err(got, ref)
    m.gemm_precision = "bf16"
    with torch.no_grad():
        fast = m(x, t, y)
    e_fast = rel_err(sub(fast) if ref.shape != tuple(fast.shape) else fast.cpu().numpy(), ref)
    print(f"{name}: bf16x3 logits rel err {e:.3e} (bf16: {e_fast:.3e})")
    assert torch.equal(out, again)
    assert e < PRECISE_TOL


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c", "s4_n8", "s2_n2", "s2_n4", "b2_n2", "xl_d1", "xl2_n2", "tiny_p8"])
def test_precise_training_gradients_match_reference(name):
    """Full training step in bf16x3 precision (fp32 activations, every product of the forward AND the backward on the MFMA
    GEMM kernel with two-term split operands, fp32 pointwise / attention backward kernels): per-sample losses and EVERY
    parameter gradient against the reference's own autograd output (tests/golden/make_golden.py).  Tolerance 2e-4 relative
    per tensor (measured ~1e-5); the scalar gain gradients, sums of ~1e5 cancelling terms, to 1e-3 of the largest of them."""
    from mapdit_amd.diffusion import create_diffusion
    g = load_golden(name)
    m, cfg, _ = build(g, train=True)
    m.gemm_precision = "bf16x3"
    x, t, y_eff, noise = dev(g, "x", "t", "y_eff", "noise")
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels      # golden labels already carry the drop
    diff = create_diffusion(timestep_respacing="")
    losses = diff.training_losses(m, x, t, dict(y=y_eff), noise=noise)
    losses["loss"].mean().backward()
    torch.cuda.synchronize()
    for k in ("loss", "mse", "vb"):
        if "train_" + k in g:
            assert rel_err(losses[k].detach().cpu().numpy(), g["train_" + k]) < 1e-4, k
    stride = 7 if "postw/x_embedder.weight" in g else 4099      # the named-model fixtures keep every 4099th gradient entry
    gain_scale = max(float(np.abs(g["grad/" + k]).max()) for k, p in m.named_parameters() if p.dim() == 0)
    worst, worst_k = 0.0, ""
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        gref = g["grad/" + k]
        if p.dim() == 0:
            assert abs(float(p.grad.item()) - float(gref.item())) < 1e-3 * gain_scale + 1e-7, (k, float(p.grad), gref)
            continue
        if "gradnorm/" + k in g and float(g["gradnorm/" + k]) > 1e-7:
            assert abs(float(p.grad.double().norm()) / float(g["gradnorm/" + k]) - 1) < 2e-4, k
        e = rel_err(sub(p.grad, stride=stride), gref)
        if e > worst:
            worst, worst_k = e, k
        assert e < (2e-4 if gref.size >= 64 else 2e-3) or np.linalg.norm(gref) < 1e-7, (k, e)
    print(f"{name}: bf16x3 worst gradient rel err {worst:.3e} ({worst_k})")


@pytest.mark.parametrize("name", ["tiny_a", "tiny_c", "s2_n2", "b2_n2"])
def test_precise_training_mode_losses_match_reference(name):
    """Training-mode forward (forced weight normalisation rewrites the weights, recorded label drop) + the loss
    kernels in bf16x3 precision against the reference's per-sample loss / mse / vb: 1e-4."""
    from mapdit_amd.diffusion import create_diffusion
    g = load_golden(name)
    m, cfg, _ = build(g, train=True)
    m.gemm_precision = "bf16x3"
    x, t, y_eff, noise = dev(g, "x", "t", "y_eff", "noise")
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels
    diff = create_diffusion(timestep_respacing="")
    with torch.no_grad():
        losses = diff.training_losses(m, x, t, dict(y=y_eff), noise=noise)
    for k in ("loss", "mse", "vb"):
        e = rel_err(losses[k].cpu().numpy(), g["train_" + k])
        print(f"{name}: bf16x3 {k} rel err {e:.3e}")
        assert e < 1e-4, k
    for k, p in m.named_parameters():
        if "postw/" + k in g:
            assert rel_err(sub(p.detach()), g["postw/" + k]) < 2e-6, k


@pytest.mark.parametrize("name", ["tiny_a", "s2_n2"])
def test_forward_stage_by_stage_against_emulating_oracle(name):
    """Every intermediate the engine keeps (mapdit_engine_peek) against the same quantity in the bf16-emulating oracle,
    from the Fourier features down to the final linear: localises any disagreement to one kernel."""
    from oracle import dit_oracle as O
    g = load_golden(name)
    m, cfg, sd = build(g, train=True)
    x, t, y_eff = dev(g, "x", "t", "y_eff")
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels
    out = m(x, t, y_eff)
    torch.cuda.synchronize()
    trace = {}
    sd_o = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        ref = O.dit_forward(sd_o, cfg, x.cpu(), t.cpu(), torch.from_numpy(g["y_eff"]), train=True,
                            drop=torch.zeros(x.shape[0], dtype=torch.bool), rnd=O.engine_plan, trace=trace)
    D = cfg.hidden_size
    rows = []

    def cmp(label, got, want, tol):
        e = rel_err(got.float().cpu().numpy().reshape(-1), want.float().numpy().reshape(-1))
        rows.append((label, e, tol))

    # tolerances: the conditioning path is fp32-accurate in the engine (split operands) and unrounded in the oracle's engine
    # plan: 2e-5 (the Fourier features' bf16 copy for the backward must be EXACT); the embedded tokens see one bf16 rounding
    # of bit-identical inputs; block 0 is 1-3 roundings deep; from block 1 on the rounding-flip noise has saturated (see
    # test_eval_forward_matches_bf16_emulating_oracle) and stays flat with depth.
    cmp("four", m._peek("four"), O.bf16_round(trace["four"]), 1e-9)
    cmp("temb", m._peek("temb"), trace["temb"], 2e-5)
    cmp("c", m._peek("c"), trace["c"], 2e-5)
    cmp("x0", m._peek("x0"), trace["x0"], 1e-6)
    mod_all = m._peek("mod_all")
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        cmp(p + "mod", mod_all[:, i * 6 * D:(i + 1) * 6 * D], trace[p + "mod"], 2e-5)
        for nm, key in (("xm", "xm"), ("qkv", "attn.qkv"), ("qn", "attn.qn"), ("kn", "attn.kn"), ("v", "attn.v"), ("o", "attn.o"),
                        ("xmid", "xmid"), ("xm2", "xm2"), ("hact", "mlp.hact"), ("xout", "xout")):
            tol = (3e-4 if nm == "xm" else 2.5e-3) if i == 0 else 6e-3
            if nm == "qkv" and cfg.head_dim in (64, 72):
                continue            # the fused QKV epilogues write q^, k^, v (head_dim 72: raw q, k, v) directly; there is no qkv tensor
            cmp(p + nm, m._peek(nm, i), trace[p + key], tol)
    cmp("final.xmod", m._peek("xmodf"), trace["final_layer.xmod"], 5e-3)
    cmp("final.lin", m._peek("lin"), trace["final_layer.lin"], 5e-3)
    cmp("logits", out.detach(), ref, 5e-3)
    for label, e, tol in rows:
        print(f"{name}: {label:22s} {e:.3e}")
    bad = [(l, e) for l, e, tol in rows if not e < tol]
    assert not bad, bad


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c"])
def test_training_losses_and_gradients(name):
    from mapdit_amd.diffusion import create_diffusion
    g = load_golden(name)
    m, cfg, _ = build(g, train=True)
    x, t, y_eff, noise = dev(g, "x", "t", "y_eff", "noise")
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels      # golden labels already carry the drop
    diff = create_diffusion(timestep_respacing="")
    assert diff.num_timesteps == 1000
    losses = diff.training_losses(m, x, t, dict(y=y_eff), noise=noise)
    losses["loss"].mean().backward()
    torch.cuda.synchronize()
    for k in ("loss", "mse", "vb"):
        e = rel_err(losses[k].detach().cpu().numpy(), g["train_" + k])
        print(f"{name}: {k} rel err {e:.3e}")
        assert e < LOSS_TOL, k
    worst, worst_k = 0.0, ""
    # the scalar gain gradients are sums of ~1e5 signed terms (heavy cancellation): judge them on the scale of the
    # largest gain gradient of the model, not on their own (possibly tiny) magnitude
    gain_scale = max(float(np.abs(g["grad/" + k]).max()) for k, p in m.named_parameters() if p.dim() == 0)
    worst_gain = 0.0
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        gref = g["grad/" + k]
        if p.dim() == 0:
            gain_dev = abs(float(p.grad) - float(gref)) / (gain_scale + 1e-30)
            worst_gain = max(worst_gain, gain_dev)
            assert gain_dev < GAIN_TOL, (k, float(p.grad), float(gref))
            continue
        e = rel_err(sub(p.grad), gref)
        if e > worst:
            worst, worst_k = e, k
        tol = GRAD_TOL if gref.size >= 64 else SMALL_GRAD_TOL
        assert e < tol or np.linalg.norm(gref) < 1e-7, (k, e)
        # forced weight normalisation rewrote the weights in place, on the fp32 path
        if "postw/" + k in g:
            assert rel_err(sub(p.detach()), g["postw/" + k]) < 2e-6, k
    print(f"{name}: worst gradient rel err {worst:.3e} ({worst_k}); worst gain deviation {worst_gain:.3e}")


def test_gradient_accumulation_and_zero_grad():
    from mapdit_amd.diffusion import create_diffusion
    g = load_golden("tiny_c")
    m, cfg, _ = build(g, train=False)            # eval mode: weights are not rewritten, so two backward passes are identical
    x, t, y, noise = dev(g, "x", "t", "y", "noise")
    diff = create_diffusion("")
    diff.training_losses(m, x, t, dict(y=y), noise=noise)["loss"].mean().backward()
    g1 = m.blocks[0].attn.qkv_proj.weight.grad.clone()
    diff.training_losses(m, x, t, dict(y=y), noise=noise)["loss"].mean().backward()
    g2 = m.blocks[0].attn.qkv_proj.weight.grad.clone()
    assert rel_err(g2.cpu().numpy(), (2 * g1).cpu().numpy()) < 1e-6
    for p in m.parameters():
        p.grad = None
    diff.training_losses(m, x, t, dict(y=y), noise=noise)["loss"].mean().backward()
    assert rel_err(m.blocks[0].attn.qkv_proj.weight.grad.cpu().numpy(), g1.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("precision,tol", [("bf16", LOGIT_TOL), ("f16", 2e-3), ("bf16x3", 1e-4)])
def test_sampler_matches_reference(precision, tol):
    """One p_sample step with classifier-free guidance and a 3-step loop prefix with injected noise against the
    reference's outputs; in bf16x3 precision the agreement is 1e-4 or better."""
    from mapdit_amd.diffusion import create_diffusion
    LOGIT_TOL = tol
    g = load_golden("tiny_b")
    m, cfg, _ = build(g)
    m.gemm_precision = precision
    d = create_diffusion("250")
    assert list(d.timestep_map) == g["timestep_map_250"].tolist()
    assert list(create_diffusion("5").timestep_map) == g["timestep_map_5"].tolist()
    z, yy, ts, nz = dev(g, "ps_z", "ps_y", "ps_t", "ps_noise")
    kw = dict(y=yy, cfg_scale=1.5)
    with torch.no_grad():
        mo = d._wrap_model(m.forward_with_cfg)(z, ts, **kw)
        sample, xstart = d._step_math(mo, z, ts, nz, False)
    e1, e2 = rel_err(sample.cpu().numpy(), g["ps_sample"]), rel_err(xstart.cpu().numpy(), g["ps_xstart"])
    print(f"[{precision}] p_sample rel err {e1:.3e}, pred_xstart {e2:.3e}")
    assert e1 < LOGIT_TOL and e2 < 2 * LOGIT_TOL
    # loop prefix with injected per-step noise
    img = z
    noises = torch.from_numpy(g["loop_noise"]).to(DEV)
    with torch.no_grad():
        for k, i in enumerate(list(range(d.num_timesteps))[::-1][:3]):
            tt = torch.full((z.shape[0],), i, device=DEV, dtype=torch.int64)
            mo = d._wrap_model(m.forward_with_cfg)(img, tt, **kw)
            img, _ = d._step_math(mo, img, tt, noises[k], False)
            e = rel_err(img.cpu().numpy(), g["loop_traj"][k])
            print(f"[{precision}] loop step {k}: rel err {e:.3e}")
            assert e < 3 * LOGIT_TOL
    # public p_sample_loop API runs end to end (own RNG): shape / finiteness on a 2-step schedule
    d2 = create_diffusion("2")
    out = d2.p_sample_loop(m.forward_with_cfg, z.shape, z, clip_denoised=False, model_kwargs=kw, progress=False, device=DEV)
    assert out.shape == z.shape and torch.isfinite(out).all()


NAMED_GRAD_TOL = 4e-2     # sub-sampled tensors of >= 64 kept entries: measured <= 2.6e-2 (s2_n4; b2_n2 9.6e-3, xl2_n2 1.5e-2; pooled 3.8e-3)


# Limits of the named (full-depth) models per engine precision.  bf16 is the dtype BENCH is quoted in, f16 the default engine: BOTH are
# asserted here against the reference's own outputs (round 5: when the default flipped to f16 this test silently ran the f16 engine
# against bf16's limits and left bf16 unasserted on every full-depth model).
#   logits, loss, gradient norm per tensor, sub-sampled gradient tensor (>= 64 kept entries), scalar gains (of the largest gain gradient)
NAMED_LIMITS = {
    # measured bf16: logits s4_n8 6.0e-3, s2_n2 3.8e-3, s2_n4 8.1e-3, b2_n2 6.0e-3, xl2_n2 6.2e-3; loss <= 2.9e-3 (s2_n4); norms <= 1.09e-2
    # (s4_n8); tensors <= 2.6e-2 (s2_n4; b2_n2 9.6e-3, xl2_n2 1.5e-2; pooled 3.8e-3)
    "bf16": dict(logits=1.6e-2, loss=6e-3, gnorm=2.2e-2, gtensor=NAMED_GRAD_TOL, gsmall=0.12, gain=GAIN_TOL),
    # f16: north_star's 1e-3 on the logits (measured worst s2_n4 9.3e-4), tests/test_f16_gpu.py's limits on the rest (worst tensor 1.6e-3)
    "f16": dict(logits=1e-3, loss=5e-4, gnorm=3e-3, gtensor=3e-3, gsmall=1e-2, gain=5e-3),
}


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("name", ["s4_n8", "s2_n2", "s2_n4", "b2_n2", "xl2_n2"])
def test_named_models_match_reference(name, precision):
    """DiT-S/4 (BASELINE configs[0]), DiT-S/2 (configs[1], full depth at n = 2 and n = 4), DiT-B/2 (the metric's model) and
    DiT-XL/2 (configs[3] / [4]: patch 2, depth 28, head_dim 72, 256 tokens) built through DIT_MODELS[...] against the reference's own
    outputs (src/dit.py:70-105, diffusion/gaussian_diffusion.py:715-787): eval logits, training losses and every parameter gradient of
    one training step, in BOTH engine precisions with the precision set explicitly."""
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.src.models import DIT_MODELS
    lim = NAMED_LIMITS[precision]
    g = load_golden(name)
    cfg = golden_cfg(g)
    sd = golden_state_dict(g, cfg)
    fam = {384: "S", 768: "B", 1152: "XL"}[cfg.hidden_size]
    m = DIT_MODELS[f"DiT-{fam}/{cfg.patch_size}"](in_channels=4, input_size=32, num_classes=1000)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.gemm_precision = precision
    x, t, y, y_eff, noise = dev(g, "x", "t", "y", "y_eff", "noise")
    with torch.no_grad():
        out = m(x, t, y)
    e = rel_err(out.cpu().numpy(), g["eval_out"])
    print(f"{name} [{precision}]: eval logits rel err {e:.3e}")
    assert e < lim["logits"]
    m.train()
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels
    losses = create_diffusion("").training_losses(m, x, t, dict(y=y_eff), noise=noise)
    losses["loss"].mean().backward()
    e = rel_err(losses["loss"].detach().cpu().numpy(), g["train_loss"])
    print(f"{name} [{precision}]: loss rel err {e:.3e}")
    assert e < lim["loss"]
    worst = worst_e = 0.0
    gain_scale = max(float(g["gradnorm/" + k]) for k, p in m.named_parameters() if p.dim() == 0)
    worst_gain = 0.0
    for k, p in m.named_parameters():
        gn = float(g["gradnorm/" + k])
        if p.dim() == 0:     # cancellation-heavy scalar sums: see test_training_losses_and_gradients
            gain_dev = abs(float(p.grad) - float(g["grad/" + k])) / (gain_scale + 1e-30)
            worst_gain = max(worst_gain, gain_dev)
            assert gain_dev < lim["gain"], (k, float(p.grad), float(g["grad/" + k]))
            continue
        if gn < 1e-7:
            continue
        got = float(p.grad.double().norm())
        worst = max(worst, abs(got / gn - 1))
        # (tensors of < 64 entries - the MPScale references, 8 sums with heavy cancellation - are held to the small-tensor limit)
        assert abs(got / gn - 1) < (lim["gnorm"] if p.numel() >= 64 else lim["gsmall"]), (k, got, gn)
        gref = g["grad/" + k]
        e = rel_err(sub(p.grad, stride=4099), gref)
        if gref.size >= 64:
            worst_e = max(worst_e, e)
        assert e < (lim["gtensor"] if gref.size >= 64 else lim["gsmall"]) or p.numel() < 64, (k, e)
    print(f"{name} [{precision}]: worst gradient-norm deviation {worst:.3e}, worst sub-sampled gradient tensor (>= 64 kept entries) {worst_e:.3e}, "
          f"worst gain deviation {worst_gain:.3e}")


@pytest.mark.parametrize("name", ["xl_d1", "tiny_p8"])
def test_generic_attention_models_match_reference(name):
    """head_dim 72 (the DiT-XL family: hidden 1152, 16 heads) and 16-token patch-8 models run on the generic
    attention path; eval logits, losses and gradients against reference-generated fixtures."""
    from mapdit_amd.diffusion import create_diffusion
    g = load_golden(name)
    m, cfg, _ = build(g)
    x, t, y, y_eff, noise = dev(g, "x", "t", "y", "y_eff", "noise")
    with torch.no_grad():
        out = m(x, t, y)
    e = rel_err(out.cpu().numpy(), g["eval_out"])
    print(f"{name}: eval logits rel err {e:.3e}")
    assert e < LOGIT_TOL
    m.train()
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels
    losses = create_diffusion("").training_losses(m, x, t, dict(y=y_eff), noise=noise)
    losses["loss"].mean().backward()
    assert rel_err(losses["loss"].detach().cpu().numpy(), g["train_loss"]) < LOSS_TOL
    stride = 7 if "postw/x_embedder.weight" in g else 4099
    gain_scale = max(float(g["gradnorm/" + k]) for k, p in m.named_parameters() if p.dim() == 0)
    worst_gain = 0.0
    worst = 0.0
    for k, p in m.named_parameters():
        if p.dim() == 0:
            gain_dev = abs(float(p.grad) - float(g["grad/" + k])) / (gain_scale + 1e-30)
            worst_gain = max(worst_gain, gain_dev)
            assert gain_dev < GAIN_TOL, k
            continue
        gn = float(g["gradnorm/" + k])
        if gn < 1e-7:
            continue
        e = rel_err(sub(p.grad, stride=stride), g["grad/" + k])
        worst = max(worst, e)
        assert e < (GRAD_TOL if p.numel() >= 64 else SMALL_GRAD_TOL), (k, e)
    print(f"{name}: worst gradient rel err {worst:.3e}; worst gain deviation {worst_gain:.3e}")


@pytest.mark.parametrize("precision,ltol,gtol", [("bf16", 7e-3, 1.5e-2), ("f16", 1e-3, 3e-3)])
def test_more_than_256_tokens_matches_reference(precision, ltol, gtol):
    """DiT(input_size=64, patch 2): 1,024 tokens per sample (reference src/dit.py:15-27; its attention is plain SDPA over all
    tokens, src/layers/attention.py:47).  The MFMA attention kernels loop over 256-token key / query tiles - no rescaling between
    tiles, cosine logits are bounded.  Eval logits, training losses and every parameter gradient against the reference's own
    outputs (fixture t1024_d2: depth 2, hidden 128, head_dim 64)."""
    from mapdit_amd.diffusion import create_diffusion
    g = load_golden("t1024_d2")
    m, cfg, _ = build(g)
    assert (cfg.input_size // cfg.patch_size) ** 2 == 1024
    m.gemm_precision = precision
    x, t, y, y_eff, noise = dev(g, "x", "t", "y", "y_eff", "noise")
    with torch.no_grad():
        out = m(x, t, y)
    e = rel_err(out.cpu().numpy(), g["eval_out"])
    print(f"t1024_d2 [{precision}]: eval logits rel err {e:.3e}")
    assert e < ltol
    m.train()
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels
    losses = create_diffusion("").training_losses(m, x, t, dict(y=y_eff), noise=noise)
    losses["loss"].mean().backward()
    el = rel_err(losses["loss"].detach().cpu().numpy(), g["train_loss"])
    worst = 0.0
    for k, p in m.named_parameters():
        gref = g["grad/" + k]
        if p.dim() == 0 or float(g["gradnorm/" + k]) < 1e-7:
            continue
        e = rel_err(sub(p.grad, stride=4099), gref)
        if gref.size >= 64:
            worst = max(worst, e)
            assert e < gtol, (k, e)
        assert abs(float(p.grad.double().norm()) / float(g["gradnorm/" + k]) - 1) < gtol, k
    print(f"t1024_d2 [{precision}]: loss rel err {el:.3e}, worst gradient rel err {worst:.3e}")
    assert el < ltol


def test_deepcopy_and_state_dict_roundtrip():
    import copy
    g = load_golden("tiny_a")
    m, cfg, sd = build(g)
    assert set(m.state_dict().keys()) == set(sd.keys())
    e = copy.deepcopy(m).eval().requires_grad_(False)            # what src/ema.py:121 does
    x, t, y = dev(g, "x", "t", "y")
    with torch.no_grad():
        a, b = m(x, t, y), e(x, t, y)
    assert torch.equal(a, b)
    half = copy.deepcopy(e).cpu().half().state_dict()            # src/ema.py:153
    assert half["x_embedder.weight"].dtype == torch.float16
    with torch.no_grad():
        for p_e, p_m in zip(e.parameters(), m.parameters()):
            p_e.lerp_(p_m, 0.5)


def test_rejects_unsupported():
    from mapdit_amd import _lib as L
    from mapdit_amd.src.dit import DiT
    xl = DiT(depth=1, hidden_size=200, patch_size=2, input_size=16, in_channels=4, num_heads=2, num_classes=10).to(DEV)   # hidden % 128
    with pytest.raises(L.MapditError):
        xl(torch.zeros(2, 4, 16, 16, device=DEV), torch.zeros(2, dtype=torch.long, device=DEV),
           torch.zeros(2, dtype=torch.long, device=DEV))


def test_ddim_matches_reference():
    """ddim_sample / ddim_reverse_sample / ddim_sample_loop (reference gaussian_diffusion.py:513-680) on recorded model outputs:
    fp32 step mathematics, 2e-6."""
    from mapdit_amd.diffusion import create_diffusion
    g = load_golden("ddim")
    d = create_diffusion("250")
    x, mo, t = dev(g, "x", "mo", "t")
    for tag, eta, clip in (("eta0", 0.0, False), ("eta0_clip", 0.0, True), ("eta07", 0.7, False)):
        noise = torch.from_numpy(g[f"{tag}/noise"]).to(DEV)
        sample, xstart = d._ddim_math(mo, x, t, noise, clip, eta, False)
        assert rel_err(sample.cpu().numpy(), g[f"{tag}/sample"]) < 2e-6, tag
        assert rel_err(xstart.cpu().numpy(), g[f"{tag}/xstart"]) < 2e-6, tag
    stub = lambda xx, tt, **kw: mo
    r = d.ddim_reverse_sample(stub, x, t, clip_denoised=False)
    assert rel_err(r["sample"].cpu().numpy(), g["rev/sample"]) < 2e-6
    assert rel_err(r["pred_xstart"].cpu().numpy(), g["rev/xstart"]) < 2e-6
    # public API with its own noise draw: eta = 0 is deterministic and must equal the recorded result
    r = d.ddim_sample(stub, x, t, clip_denoised=False, eta=0.0)
    assert rel_err(r["sample"].cpu().numpy(), g["eta0/sample"]) < 2e-6
    with pytest.raises(AssertionError):
        d.ddim_reverse_sample(stub, x, t, eta=0.5)
    d5 = create_diffusion("5")
    stub5 = lambda xx, tt, **kw: torch.cat([0.3 * xx + 0.01 * tt.float().view(-1, 1, 1, 1), 0.1 * xx], dim=1)
    z = torch.from_numpy(g["loop_noise"]).to(DEV)
    out = d5.ddim_sample_loop(stub5, tuple(z.shape), noise=z, clip_denoised=False, device=DEV, eta=0.0)
    assert rel_err(out.cpu().numpy(), g["loop_final"]) < 1e-5


@pytest.mark.parametrize("name,precision", [("tiny_a", "bf16"), ("tiny_b", "bf16"), ("tiny_p8", "bf16"), ("xl_d1", "bf16"),
                                            ("tiny_a", "bf16x3")])
def test_ragged_batches_and_single_samples(name, precision):
    """Batches of 1, 3, 5 and 7 samples (token-row counts that are not multiples of any tile size, engines re-created as the
    batch grows): every sample's logits are the same bits whatever batch it sits in, and the gradient of a 7-sample batch
    equals the sum of the gradients of its 3 + 4 split (fp32 summation order only: 2e-5 of the tensor)."""
    from mapdit_amd.diffusion import create_diffusion
    g = load_golden(name)
    m, cfg, _ = build(g)
    m.gemm_precision = precision
    gen = torch.Generator().manual_seed(21)
    n = 7
    x = torch.randn(n, cfg.in_channels, cfg.input_size, cfg.input_size, generator=gen).to(DEV)
    t = torch.randint(0, 1000, (n,), generator=gen).to(DEV)
    y = torch.randint(0, cfg.num_classes, (n,), generator=gen).to(DEV)
    noise = torch.randn(x.shape, generator=gen).to(DEV)
    outs = {}
    with torch.no_grad():
        for b in (1, 3, 5, 7):                       # growing batch: the runtime re-creates its engine each time
            outs[b] = m(x[:b].contiguous(), t[:b].contiguous(), y[:b].contiguous())
            assert torch.isfinite(outs[b]).all()
        single_last = m(x[6:7].contiguous(), t[6:7].contiguous(), y[6:7].contiguous())
    for b in (1, 3, 5):
        assert torch.equal(outs[b], outs[7][:b]), b
    assert torch.equal(single_last, outs[7][6:7])
    diff = create_diffusion("")

    def grads(lo, hi):
        for p in m.parameters():
            p.grad = None
        sl = slice(lo, hi)
        loss = diff.training_losses(m, x[sl].contiguous(), t[sl].contiguous(), dict(y=y[sl].contiguous()), noise=noise[sl].contiguous())["loss"]
        loss.sum().backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), m._gflat.clone()

    l7, g7 = grads(0, 7)
    la, ga = grads(0, 3)
    lb, gb = grads(3, 7)
    assert torch.equal(torch.cat([la, lb]), l7)
    # scalar gains are sums of ~1e5 cancelling terms: judged on the scale of the largest gain gradient (as everywhere above)
    gain_scale = max(float(g7[o].abs()) for (k, p), o in zip(m.named_parameters(), m._poffs) if p.dim() == 0)
    for (k, p), o in zip(m.named_parameters(), m._poffs):
        whole = g7[o:o + p.numel()].double()
        parts = ga[o:o + p.numel()].double() + gb[o:o + p.numel()].double()
        if p.dim() == 0:
            assert abs(float(whole) - float(parts)) < 1e-3 * gain_scale + 1e-9, k
        elif float(whole.norm()) > 1e-9:
            assert float((whole - parts).norm() / whole.norm()) < 2e-5, k


def test_out_of_range_label_and_stale_forward_are_refused():
    """What the reference answers with an IndexError / what its autograd tracks per graph: a label outside the embedding
    table never indexes memory (clamped + reported by check_device_errors), a backward through anything but the most recent
    training forward raises, and so does asking for input gradients."""
    from mapdit_amd import _lib as L
    from mapdit_amd.diffusion import create_diffusion
    g = load_golden("tiny_a")
    m, cfg, _ = build(g, train=True)
    x, t, y = dev(g, "x", "t", "y")
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels
    m.check_device_errors()                                   # clean
    bad = y.clone()
    bad[1] = cfg.num_classes + 5                              # one past the null row and beyond
    out = m(x, t, bad)
    out.sum().backward()                                      # backward must not write outside the table gradient either
    with pytest.raises(L.MapditError, match="label"):
        m.check_device_errors()
    m.check_device_errors()                                   # the record was cleared
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    tbad = t.clone()
    tbad[0] = 1000
    with torch.no_grad():
        create_diffusion("").q_sample(x, tbad, torch.zeros_like(x))
    with pytest.raises(L.MapditError, match="timestep"):
        m.check_device_errors()
    # one outstanding forward per model
    for p in m.parameters():
        p.grad = None
    la = m(x, t, y).sum()
    lb = m(x, t, y).sum()
    with pytest.raises(L.MapditError, match="stale"):
        la.backward()
    lb.backward()
    with pytest.raises(L.MapditError, match="input latents"):
        m(x.clone().requires_grad_(True), t, y)


def test_torch_compile_wrapper_runs():
    """torch.compile(model) (reference train.py:46, sample.py:25) must run: same bits as the bare module, training included."""
    g = load_golden("tiny_a")
    m, cfg, _ = build(g)
    x, t, y = dev(g, "x", "t", "y")
    cm = torch.compile(m)
    with torch.no_grad():
        a, b = m(x, t, y), cm(x, t, y)
        c = cm.forward_with_cfg(x, t, y, 1.5)        # sample.py:25,54 passes the compiled module's bound method to the sampler
    assert torch.equal(a, b) and c.shape == a.shape
    m.train()
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels
    cm(x, t, y).sum().backward()
    assert m.blocks[0].mlp.net[0].weight.grad is not None


@pytest.mark.parametrize("precision,ltol,gtol", [("bf16", LOGIT_TOL, GRAD_TOL), ("bf16x3", 1e-4, 2e-4)])
def test_forced_weight_normalization_off(precision, ltol, gtol):
    """README.md:61 `--no-use-forced-weight-normalization`: the training forward leaves the fp32 weights alone (the in-place
    rewrite of mp_linear.py:38-40 is the only thing the flag removes) - i.e. the snapshot's eval-mode arithmetic on weights of
    any norm, with training-mode label drop.  Outputs equal the eval forward bit for bit, weights are unchanged, and the
    gradients are those of the oracle's autograd through the same arithmetic (weight-norm Jacobian at |w| != 1)."""
    import oracle.dit_oracle as O
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.src.dit import DiT
    from oracle.diffusion_oracle import DiffusionOracle
    g = load_golden("tiny_a")
    cfg = golden_cfg(g)
    sd = golden_state_dict(g, cfg)
    m = DiT(**cfg.to_dict(), forced_weight_normalization=False)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    m.gemm_precision = precision
    x, t, y_eff, noise = dev(g, "x", "t", "y_eff", "noise")
    m.eval()
    with torch.no_grad():
        ref_out = m(x, t, y_eff)
    m.train()
    m.y_embedder.token_drop = lambda labels, force_drop_ids=None: labels      # golden labels already carry the drop
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    losses = create_diffusion(timestep_respacing="").training_losses(m, x, t, dict(y=y_eff), noise=noise)
    losses["loss"].mean().backward()
    torch.cuda.synchronize()
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), f"{k} was rewritten although forced weight normalisation is off"
    with torch.no_grad():
        assert torch.equal(m(x, t, y_eff), ref_out)                           # training forward == eval arithmetic
    osd = {k: v.clone().requires_grad_(k not in O.BUFFER_KEYS) for k, v in sd.items()}
    xc, tc, yc, nc = (torch.from_numpy(g[k]) for k in ("x", "t", "y_eff", "noise"))
    ref = DiffusionOracle("").training_losses(lambda xx, tt, **kw: O.dit_forward(osd, cfg, xx, tt, kw["y"], train=False),
                                              xc, tc, dict(y=yc), noise=nc)
    ref["loss"].mean().backward()
    assert rel_err(losses["loss"].detach().cpu().numpy(), ref["loss"].detach().numpy()) < ltol
    gain_scale = max(float(osd[k].grad.abs().max()) for k in osd if osd[k].dim() == 0 and osd[k].grad is not None)
    for k, p in m.named_parameters():
        gref = osd[k].grad
        if p.dim() == 0:
            assert abs(float(p.grad) - float(gref)) < (GAIN_TOL if precision == "bf16" else 1e-3) * gain_scale + 1e-7, k
            continue
        e = rel_err(p.grad.cpu().numpy(), gref.numpy())
        assert e < (gtol if gref.numel() >= 64 else 10 * gtol) or float(gref.norm()) < 1e-7, (k, e)
    # the harness accepts this flag's off form (and, round 5, the seven of tests/test_mp_flags_gpu.py); a combination that is not built
    # (the LayerNorm form under rotation modulation) still refuses
    from mapdit_amd import train
    args = train.build_parser().parse_args(["--synthetic", "--results-dir", "unused", "--no-use-forced-weight-normalization",
                                            "--model", "DiT-XS/8", "--num-classes", "10"])
    args.in_channels, args.input_size = 4, 32
    assert train.get_model(args).forced_weight_normalization is False
    with pytest.raises(NotImplementedError):
        train.main(["--synthetic", "--results-dir", "unused", "--no-use-no-layernorm", "--use-rotation-modulation", "--num-steps", "2"])
