"""Host-side logic of the drop-in boundary (no GPU): constructor API, state_dict layout, schedule tables,
respacing and error behaviour — checked against the oracle and the reference-generated fixtures."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import dit_oracle as O
from oracle.diffusion_oracle import DiffusionOracle


def test_model_zoo_names_and_shapes():
    from mapdit_amd.src.models import DIT_MODELS
    assert sorted(DIT_MODELS) == sorted(f"DiT-{f}/{p}" for f in ("XL", "L", "B", "S", "XS") for p in (2, 4, 8))
    m = DIT_MODELS["DiT-S/4"](in_channels=4, input_size=32, num_classes=1000)
    cfg = O.model_config("DiT-S/4", in_channels=4, input_size=32, num_classes=1000)
    sd = m.state_dict()
    shapes = O.param_shapes(cfg)
    assert set(sd.keys()) == set(shapes.keys())      # same entries (the oracle keeps its own generation order)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    n_params = sum(p.numel() for p in m.parameters())
    assert abs(n_params - 32.86e6) < 0.05e6          # DiT-S/4 = 32.86 M parameters
    # reference init: gains 0, mean reference 1, sigma reference 0, pos_embed = normalised sin-cos table
    assert float(m.blocks[0].gain_msa) == 0.0 and float(m.final_layer.gain_mod) == 0.0
    assert torch.equal(m.final_layer.mean_scale.reference.data, torch.ones(8))
    assert torch.equal(m.final_layer.sigma_scale.reference.data, torch.zeros(8))
    ref_pe = O.init_state_dict(O.DiTConfig(depth=1, hidden_size=384, patch_size=4, input_size=32, in_channels=4, num_heads=6),
                               seed=0)["pos_embed"]
    assert torch.allclose(m.pos_embed, ref_pe, atol=1e-6)


def test_parameters_are_views_of_one_flat_buffer():
    from mapdit_amd.src.dit import DiT
    m = DiT(depth=1, hidden_size=128, patch_size=2, input_size=16, in_channels=4, num_heads=2, num_classes=10)
    base = m._pflat.data_ptr()
    end = base + m._pflat.numel() * 4
    for p in m.parameters():
        assert base <= p.data_ptr() < end and (p.data_ptr() - base) % 128 == 0     # 128-B slots (base is 256-B aligned on the GPU)
    sd = O.init_state_dict(O.DiTConfig(depth=1, hidden_size=128, patch_size=2, input_size=16, in_channels=4, num_heads=2,
                                       num_classes=10), seed=3)
    m.load_state_dict(sd)
    for p in m.parameters():
        assert base <= p.data_ptr() < end            # load_state_dict copies in place
    m2 = m.half()
    assert m2._pflat.dtype == torch.float16


def test_cpu_call_fails_loudly():
    from mapdit_amd.src.dit import DiT
    from mapdit_amd import _lib as L
    m = DiT(depth=1, hidden_size=128, patch_size=2, input_size=16, in_channels=4, num_heads=2, num_classes=10)
    with pytest.raises(L.MapditError):
        m(torch.zeros(2, 4, 16, 16), torch.zeros(2, dtype=torch.long), torch.zeros(2, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        DiT(depth=1, hidden_size=128, patch_size=2, learn_sigma=False)


def test_create_diffusion_tables_and_respacing():
    from mapdit_amd.diffusion import create_diffusion, space_timesteps
    g = load_golden("tables")
    for tag, rs in (("1000", ""), ("250", "250")):
        d = create_diffusion(rs)
        for k in ("betas", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                  "sqrt_recipm1_alphas_cumprod", "posterior_log_variance_clipped", "posterior_mean_coef1",
                  "posterior_mean_coef2"):
            np.testing.assert_allclose(getattr(d, k), g[f"{tag}/{k}"], rtol=1e-13, atol=0)
    assert create_diffusion("5").timestep_map == [0, 250, 500, 749, 999]
    assert create_diffusion("250").timestep_map == DiffusionOracle("250").timestep_map
    assert create_diffusion("ddim50").num_timesteps == 50
    with pytest.raises(ValueError):
        space_timesteps(10, "20")
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")
    with pytest.raises(NotImplementedError):
        create_diffusion("", noise_schedule="nope")
    d = create_diffusion("", use_kl=True)
    with pytest.raises(NotImplementedError):
        d.training_losses(None, None, None)


def test_compiled_module_checkpoints_load():
    """The reference saves state dicts of a torch.compile'd module (train.py:46,125-132; src/ema.py:121): every key carries
    an "_orig_mod." prefix.  Such checkpoints load unchanged, torch.compile(model) wraps the module without tracing into the
    engine call, and the wrapper's own state dict round-trips."""
    from mapdit_amd.src.dit import DiT
    cfg = O.DiTConfig(depth=1, hidden_size=128, patch_size=2, input_size=16, in_channels=4, num_heads=2, num_classes=10)
    sd = O.init_state_dict(cfg, seed=3, gains=0.2)
    m = DiT(**cfg.to_dict())
    missing = m.load_state_dict({"_orig_mod." + k: v for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    cm = torch.compile(m)                                  # what reference train.py:46 does
    csd = cm.state_dict()
    assert all(k.startswith("_orig_mod.") for k in csd) and len(csd) == len(sd)
    m2 = DiT(**cfg.to_dict())
    m2.load_state_dict(csd)                                # a checkpoint written from the compiled wrapper
    assert torch.equal(m2.blocks[0].attn.qkv_proj.weight, sd["blocks.0.attn.qkv_proj.weight"])
    with pytest.raises(RuntimeError):                      # real mismatches still fail loudly
        m2.load_state_dict({"_orig_mod." + k: v for k, v in sd.items() if "gain_msa" not in k})


def test_flat_buffer_splits_into_stage_parts():
    """ZeRO-1 layout: every backward stage's slice of the flat buffers splits into 8 four-element-aligned parts."""
    from mapdit_amd.parallel import stage_slices
    from mapdit_amd.src.models import DIT_MODELS
    m = DIT_MODELS["DiT-XS/2"](in_channels=4, input_size=32, num_classes=1000)
    sl = stage_slices(m)
    assert len(sl) == m.depth + 2
    for lo, hi in sl:
        assert (hi - lo) % 32 == 0 and lo % 32 == 0
