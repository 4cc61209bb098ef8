#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

Imports the reference's own ``src.dit.DiT`` and ``diffusion.create_diffusion`` (CPU, fp32),
loads a state dict produced by *our* seeded initialiser, runs the recipe of SURVEY.md §8(c)
and writes small ``.npz`` fixtures next to this file.  While doing so it compares the full
reference outputs / gradients with the CPU oracle and aborts if they disagree, so the
committed (sub-sampled) fixtures are a travel-safe subset of a full-tensor check.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Nothing of the reference is copied: the fixtures are inputs and expected outputs only.
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)

import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
torch.set_num_threads(8)

from src.dit import DiT as RefDiT                      # noqa: E402  (reference)
from diffusion import create_diffusion as ref_create   # noqa: E402  (reference)

from oracle import dit_oracle as O                     # noqa: E402
from oracle.diffusion_oracle import DiffusionOracle    # noqa: E402

FULL_LIMIT = 20000      # tensors up to this many elements are stored whole
STRIDE = 7              # larger ones: every 7th element of the flattened tensor (+ norm)


BIG_STRIDE = 4099      # output-only fixtures of the named model sizes


def sub(a: torch.Tensor, stride: int = STRIDE) -> np.ndarray:
    f = a.detach().reshape(-1)
    return (f if f.numel() <= FULL_LIMIT else f[::stride]).numpy().copy()


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def build_ref(cfg: O.DiTConfig, sd):
    m = RefDiT(**cfg.to_dict())            # (to_dict leaves the build-only rotation switch out unless it is set)
    missing = m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    return m


def inputs(cfg, n, seed, force_t0=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cfg.in_channels, cfg.input_size, cfg.input_size, generator=g)
    y = torch.randint(0, cfg.num_classes, (n,), generator=g)
    t = torch.randint(0, 1000, (n,), generator=g)
    noise = torch.randn(n, cfg.in_channels, cfg.input_size, cfg.input_size, generator=g)
    if force_t0:
        t[0] = 0
    return x, y, t, noise


def fixture(name, cfg, n, wseed, dseed, gains=None, perturb=0.0, force_t0=False, rng_seed=2,
            sampler=False, full=True, check_tol=2e-5):
    print(f"== {name}: {cfg}")
    sd = O.init_state_dict(cfg, seed=wseed, gains=gains, perturb_reference=perturb)
    x, y, t, noise = inputs(cfg, n, dseed, force_t0)
    out = {"cfg_" + k: np.array(v) for k, v in cfg.to_dict().items()}
    out.update(n=np.array(n), wseed=np.array(wseed), dseed=np.array(dseed),
               gains=np.array(-1.0 if gains is None else gains), perturb=np.array(perturb),
               x=x.numpy(), y=y.numpy(), t=t.numpy(), noise=noise.numpy())

    # ---- eval forward -------------------------------------------------------------------
    ref = build_ref(cfg, sd).eval()
    with torch.no_grad():
        ref_eval = ref(x, t, y)
        orc_eval = O.dit_forward({k: v.clone() for k, v in sd.items()}, cfg, x, t, y, train=False)
    e = rel(orc_eval, ref_eval)
    print(f"   eval  oracle-vs-ref rel {e:.2e}")
    assert e < check_tol
    out["eval_out"] = ref_eval.numpy()
    out["eval_sum"] = np.array(ref_eval.double().sum().item())
    out["eval_absmean"] = np.array(ref_eval.abs().double().mean().item())

    # ---- train: losses + grads + forced-WN-mutated weights ---------------------------------
    ref = build_ref(cfg, sd).train()
    diff = ref_create(timestep_respacing="")
    torch.manual_seed(rng_seed)
    drop = torch.rand(n) < cfg.class_dropout_prob          # label_embedder.py:23 is the first draw
    torch.manual_seed(rng_seed)
    losses = diff.training_losses(ref, x, t, dict(y=y), noise=noise)
    losses["loss"].mean().backward()
    out["drop"] = drop.numpy()
    out["y_eff"] = torch.where(drop, torch.full_like(y, cfg.num_classes), y).numpy()
    for k in ("loss", "mse", "vb"):
        out["train_" + k] = losses[k].detach().numpy()

    osd = {k: v.clone().requires_grad_(k not in O.BUFFER_KEYS) for k, v in sd.items()}
    od = DiffusionOracle("")
    ol = od.training_losses(lambda xx, tt, **kw: O.dit_forward(osd, cfg, xx, tt, kw["y"], train=True, drop=drop),
                            x, t, dict(y=y), noise=noise)
    ol["loss"].mean().backward()
    for k in ("loss", "mse", "vb"):
        e = rel(ol[k].detach(), losses[k].detach())
        print(f"   train {k:4s} oracle-vs-ref rel {e:.2e}")
        assert e < check_tol
    worst = 0.0
    for k, p in ref.named_parameters():
        g_ref = p.grad
        g_or = osd[k].grad
        e = rel(g_or, g_ref)
        worst = max(worst, e)
        assert e < 50 * check_tol, (k, e)
        out["grad/" + k] = sub(g_ref, STRIDE if full else BIG_STRIDE)
        out["gradnorm/" + k] = np.array(g_ref.double().norm().item())
        ew = rel(osd[k].detach(), p.detach())
        assert ew < 1e-6, (k, ew)
        if full:
            out["postw/" + k] = sub(p.detach())
    print(f"   worst grad oracle-vs-ref rel {worst:.2e}")

    # ---- sampler ------------------------------------------------------------------------
    if sampler:
        ref = build_ref(cfg, sd).eval()
        d250 = ref_create("250")
        odx = DiffusionOracle("250")
        assert list(d250.timestep_map) == list(odx.timestep_map)
        nn = n // 2
        z = torch.cat([x[:nn], x[:nn]], 0)
        yy = torch.cat([y[:nn], torch.full((nn,), cfg.num_classes)], 0)
        kw = dict(y=yy, cfg_scale=1.5)
        # single p_sample at a mid step, noise injected by seeding (eval forward draws nothing)
        ts = torch.tensor([137] * n)
        torch.manual_seed(11)
        nz = torch.randn_like(z)
        torch.manual_seed(11)
        with torch.no_grad():
            r = d250.p_sample(ref.forward_with_cfg, z, ts, clip_denoised=False, model_kwargs=kw)
        osd2 = {k: v.clone() for k, v in sd.items()}
        fn = lambda xx, tt, **k2: O.dit_forward_with_cfg(osd2, cfg, xx, tt, k2["y"], k2["cfg_scale"])
        with torch.no_grad():
            ro = odx.p_sample(fn, z, ts, nz, clip_denoised=False, model_kwargs=kw)
        e = rel(ro["sample"], r["sample"])
        print(f"   p_sample oracle-vs-ref rel {e:.2e}")
        assert e < check_tol
        out.update(ps_z=z.numpy(), ps_y=yy.numpy(), ps_t=ts.numpy(), ps_noise=nz.numpy(),
                   ps_sample=r["sample"].numpy(), ps_xstart=r["pred_xstart"].numpy())
        # loop prefix: 3 steps of the 250-schedule (F11: random nets diverge later)
        K = 3
        torch.manual_seed(13)
        step_noise = [torch.randn_like(z) for _ in range(K)]
        torch.manual_seed(13)
        traj = []
        for k, o in enumerate(d250.p_sample_loop_progressive(ref.forward_with_cfg, z.shape, noise=z,
                                                             clip_denoised=False, model_kwargs=kw, device="cpu")):
            traj.append(o["sample"])
            if k + 1 == K:
                break
        otraj = odx.p_sample_loop(fn, z.shape, z, step_noise, clip_denoised=False, model_kwargs=kw, max_steps=K)
        for k in range(K):
            e = rel(otraj[k], traj[k])
            print(f"   loop step {k} oracle-vs-ref rel {e:.2e}")
            assert e < 20 * check_tol
        out["loop_noise"] = torch.stack(step_noise).numpy()
        out["loop_traj"] = torch.stack(traj).numpy()
        out["timestep_map_250"] = np.array(d250.timestep_map)
        out["timestep_map_5"] = np.array(ref_create("5").timestep_map)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"   wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


def tables():
    """Schedule tables of the reference for '' and '250' (pure numpy, float64)."""
    out = {}
    for tag, rs in (("1000", ""), ("250", "250")):
        d = ref_create(rs)
        o = DiffusionOracle(rs)
        for k in ("betas", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "posterior_log_variance_clipped", "posterior_mean_coef1",
                  "posterior_mean_coef2"):
            a = getattr(d, k)
            assert np.allclose(a, getattr(o, k), rtol=1e-13, atol=0), k
            out[f"{tag}/{k}"] = a
    from src.ema import std_to_gamma, calc_beta          # reference
    out["ema_gamma"] = np.array([float(std_to_gamma(np.array(s))) for s in (0.05, 0.1)])
    out["ema_beta_t100"] = np.array([float(calc_beta(s, 100)) for s in (0.05, 0.1)])
    assert np.allclose(out["ema_gamma"], [O.std_to_gamma(0.05), O.std_to_gamma(0.1)])
    assert np.allclose(out["ema_beta_t100"], [O.ema_beta(0.05, 100), O.ema_beta(0.1, 100)])
    out.update(constructor_pins())
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **out)
    print("== tables.npz written")


def constructor_pins():
    """What the reference's OWN constructor produces (src/dit.py:15-62), nothing loaded over it: the normalised sin-cos
    position table (src/pos_embed.py:4-60, dit.py:46-48) sampled at 16 fixed places per width, the Fourier buffers' and every
    parameter's init statistics.  Pins SURVEY A16 and the init distributions (the other fixtures overwrite both with the
    oracle's state dict)."""
    out = {}
    idx = np.array([0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987])
    for name in ("DiT-S/2", "DiT-S/4", "DiT-B/2", "DiT-XL/2", "DiT-S/8"):
        cfg = O.model_config(name, in_channels=4, input_size=32, num_classes=1000)
        torch.manual_seed(0)
        m = RefDiT(**cfg.to_dict())
        pe = m.pos_embed.detach().reshape(-1)
        pos = (idx * 7919) % pe.numel()
        out[f"ctor/{name}/pos_idx"] = pos
        out[f"ctor/{name}/pos_val"] = pe[pos].numpy()
        out[f"ctor/{name}/pos_shape"] = np.array(m.pos_embed.shape)
        out[f"ctor/{name}/pos_rownorm"] = m.pos_embed.detach()[0].norm(dim=-1)[:4].numpy()
        stats = {}
        for k, p in list(m.named_parameters()) + list(m.named_buffers()):
            if k == "pos_embed":
                continue
            f = p.detach().double().reshape(-1)
            stats[k] = (f.numel(), f.mean().item(), f.std().item() if f.numel() > 1 else 0.0, f.min().item(), f.max().item())
        if name in ("DiT-S/2", "DiT-XL/2"):
            out[f"ctor/{name}/param_names"] = np.array(list(stats))
            out[f"ctor/{name}/param_stats"] = np.array([stats[k] for k in stats])
        del m
    return out


def optimizer_fixture():
    """Three Adam + EMA steps of the reference harness pieces (train.py:57,94-105; src/ema.py)
    on a tiny model; pins SURVEY §8(f) N1."""
    import copy
    cfg = O.DiTConfig(depth=1, hidden_size=128, patch_size=4, input_size=32, in_channels=4, num_heads=2, num_classes=10)
    sd = O.init_state_dict(cfg, seed=5, gains=0.2, perturb_reference=0.3)
    ref = build_ref(cfg, sd).train()
    diff = ref_create("")
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2, betas=(0.9, 0.99))
    from src.ema import calc_beta
    emas = {s: copy.deepcopy(ref).eval().requires_grad_(False) for s in (0.05, 0.1)}
    out = {"cfg_" + k: np.array(v) for k, v in cfg.to_dict().items()}
    n = 4
    for step in range(1, 4):
        x, y, t, noise = inputs(cfg, n, 100 + step)
        torch.manual_seed(step)
        drop = torch.rand(n) < cfg.class_dropout_prob
        torch.manual_seed(step)
        loss = diff.training_losses(ref, x, t, dict(y=y), noise=noise)["loss"].mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        with torch.no_grad():
            for s, e in emas.items():
                b = calc_beta(s, step)
                for k, p in e.named_parameters():
                    p.lerp_(ref.get_parameter(k), b)
        out[f"s{step}/x"], out[f"s{step}/y"], out[f"s{step}/t"] = x.numpy(), y.numpy(), t.numpy()
        out[f"s{step}/noise"], out[f"s{step}/drop"] = noise.numpy(), drop.numpy()
        out[f"s{step}/loss"] = np.array(loss.item())
        for k in ("blocks.0.attn.qkv_proj.weight", "blocks.0.gain_msa", "final_layer.mean_scale.reference",
                  "y_embedder.embedding.weight", "x_embedder.weight", "final_layer.linear.weight"):
            out[f"s{step}/w/{k}"] = sub(ref.get_parameter(k))
            out[f"s{step}/ema0.05/{k}"] = sub(emas[0.05].get_parameter(k))
            out[f"s{step}/ema0.1/{k}"] = sub(emas[0.1].get_parameter(k))
    np.savez_compressed(os.path.join(HERE, "optim3.npz"), **out)
    print("== optim3.npz written")


from ema_fixture_data import ema_snapshot_set           # noqa: E402  (tests/golden/ema_fixture_data.py)


def ema_fixture():
    """Post-hoc EMA (reference src/ema.py:10-114): gammas, betas, solve_weights and full reconstructions; pins §8(f) N3."""
    import tempfile
    import src.ema as R
    out = {}
    stds = np.array([0.0075, 0.01, 0.05, 0.075, 0.1, 0.15, 0.2])
    out["stds"] = stds
    out["gammas"] = R.std_to_gamma(stds)
    out["stds_back"] = R.gamma_to_std(out["gammas"])
    out["beta_t"] = np.array([1, 2, 10, 1000, 400000])
    out["betas_0.05"] = np.array([R.calc_beta(0.05, t) for t in out["beta_t"]], dtype=np.float64)
    out["betas_0.1"] = np.array([R.calc_beta(0.1, t) for t in out["beta_t"]], dtype=np.float64)
    ts = np.array([1000, 1000, 2000, 2000, 3000, 3000, 4000, 4000])
    st = np.array([0.05, 0.1] * 4)
    out["sw_ts"], out["sw_stds"] = ts, st
    out["sw_targets"] = np.array([0.0075, 0.03, 0.075, 0.15])
    out["sw_weights"] = R.solve_weights(ts, R.std_to_gamma(st), np.full(4, 4000), R.std_to_gamma(out["sw_targets"]))
    snaps = ema_snapshot_set()
    with tempfile.TemporaryDirectory() as d:
        for std, t, sd in snaps:
            torch.save({"std": std, "t": t, "state_dict": sd}, os.path.join(d, f"{std:.3f}_{t:07d}.pt"))
        out["order"] = np.array(os.listdir(d))            # the reference accumulates in os.listdir order
        for target in (0.075, 0.02, 0.1):
            res = R.calculate_posthoc_ema(target, d, verbose=False)
            for k, v in res.items():
                out[f"posthoc_{target}/{k}"] = v.float().numpy()
                out[f"posthoc_{target}_dtype/{k}"] = np.array(str(v.dtype))
    np.savez_compressed(os.path.join(HERE, "ema.npz"), **out)
    print("== ema.npz written")


def ddim_fixture():
    """DDIM step, reverse-ODE step and a short loop of the reference (gaussian_diffusion.py:513-680) on fixed model outputs:
    the 'model' is a stub that returns a recorded tensor, so the fixture pins the step mathematics alone."""
    out = {}
    d = ref_create("250")
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 4, 8, 8, generator=g)
    mo = torch.randn(4, 8, 8, 8, generator=g)
    t = torch.tensor([0, 1, 120, 249])
    out["x"], out["mo"], out["t"] = x.numpy(), mo.numpy(), t.numpy()
    stub = lambda xx, tt, **kw: mo
    for tag, kw in (("eta0", dict(eta=0.0, clip_denoised=False)), ("eta0_clip", dict(eta=0.0, clip_denoised=True)),
                    ("eta07", dict(eta=0.7, clip_denoised=False))):
        torch.manual_seed(5)
        r = d.ddim_sample(stub, x, t, **kw)
        torch.manual_seed(5)
        out[f"{tag}/noise"] = torch.randn_like(x).numpy()            # the draw ddim_sample made
        out[f"{tag}/sample"], out[f"{tag}/xstart"] = r["sample"].numpy(), r["pred_xstart"].numpy()
    r = d.ddim_reverse_sample(stub, x, t, clip_denoised=False)
    out["rev/sample"], out["rev/xstart"] = r["sample"].numpy(), r["pred_xstart"].numpy()
    # deterministic loop on a 5-step schedule; the stub's output depends on (x, t) so every step matters
    d5 = ref_create("5")
    stub5 = lambda xx, tt, **kw: torch.cat([0.3 * xx + 0.01 * tt.float().view(-1, 1, 1, 1), 0.1 * xx], dim=1)
    out["loop_noise"] = torch.randn(2, 4, 8, 8, generator=g).numpy()
    out["loop_final"] = d5.ddim_sample_loop(stub5, (2, 4, 8, 8), noise=torch.from_numpy(out["loop_noise"]), clip_denoised=False,
                                            device="cpu", eta=0.0).numpy()
    np.savez_compressed(os.path.join(HERE, "ddim.npz"), **out)
    print("== ddim.npz written")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ddim":
        ddim_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ema":
        ema_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "optim3":
        optimizer_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        # shapes that take the generic attention path: head_dim 72 (the XL family) and 16 tokens (patch 8)
        fixture("xl_d1", O.DiTConfig(depth=1, hidden_size=1152, patch_size=4, input_size=32, in_channels=4, num_heads=16,
                                     num_classes=10), n=2, wseed=7, dseed=8, gains=0.3, perturb=0.3, full=False, check_tol=5e-5)
        fixture("tiny_p8", O.DiTConfig(depth=1, hidden_size=128, patch_size=8, input_size=32, in_channels=4, num_heads=2,
                                       num_classes=10), n=3, wseed=9, dseed=10, gains=0.3, perturb=0.3)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round3":
        # more than 256 tokens per sample: 64x64 latents at patch 2 = 1,024 tokens (DiT(input_size=...) is free in the reference,
        # src/dit.py:15-27; attention is plain SDPA over all tokens, src/layers/attention.py:47): depth 2, head_dim 64
        fixture("t1024_d2", O.DiTConfig(depth=2, hidden_size=128, patch_size=2, input_size=64, in_channels=4, num_heads=2,
                                        num_classes=10), n=2, wseed=11, dseed=12, gains=0.3, perturb=0.3, full=False, check_tol=5e-5)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tables":
        tables()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        # BASELINE configs 2 and 4 at full depth: DiT-S/2 at n = 4 and DiT-XL/2 (patch 2, depth 28, head_dim 72) at n = 2
        fixture("s2_n4", O.model_config("DiT-S/2", in_channels=4, input_size=32, num_classes=1000), n=4, wseed=3, dseed=4,
                gains=0.2, perturb=0.3, full=False, check_tol=5e-5)
        fixture("xl2_n2", O.model_config("DiT-XL/2", in_channels=4, input_size=32, num_classes=1000), n=2, wseed=0, dseed=1,
                gains=0.15, perturb=0.3, full=False, check_tol=5e-5)
        sys.exit(0)
    tiny = dict(in_channels=4, num_heads=2, num_classes=10)
    fixture("tiny_a", O.DiTConfig(depth=2, hidden_size=128, patch_size=2, input_size=16, **tiny), n=4, wseed=1, dseed=2)
    fixture("tiny_b", O.DiTConfig(depth=2, hidden_size=128, patch_size=2, input_size=16, **tiny), n=4, wseed=3, dseed=4,
            gains=0.3, perturb=0.5, force_t0=True, sampler=True)
    fixture("tiny_c", O.DiTConfig(depth=1, hidden_size=128, patch_size=4, input_size=32, **tiny), n=2, wseed=5, dseed=6,
            gains=0.2, perturb=0.2)
    tables()
    optimizer_fixture()
    # output-only fixtures at named sizes; weights are regenerated from the seed by the tests
    fixture("s4_n8", O.model_config("DiT-S/4", in_channels=4, input_size=32, num_classes=1000), n=8, wseed=0, dseed=1,
            full=False, check_tol=5e-5)
    fixture("s2_n2", O.model_config("DiT-S/2", in_channels=4, input_size=32, num_classes=1000), n=2, wseed=0, dseed=1,
            gains=0.25, perturb=0.3, full=False, check_tol=5e-5)
    fixture("b2_n2", O.model_config("DiT-B/2", in_channels=4, input_size=32, num_classes=1000), n=2, wseed=0, dseed=1,
            full=False, check_tol=5e-5)
