"""SURVEY §8(f) N3 + N4 on the GPU: checkpoints / EMA snapshots written by the harness feed the three sampler CLIs
(counterparts of the reference's sample.py, sample_fid.py, sample_ema.py), and the optimiser state round-trips through
torch.optim.Adam's own format."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def exp_dir(tmp_path_factory):
    from mapdit_amd import train
    root = tmp_path_factory.mktemp("results")
    return train.main(["--synthetic", "--results-dir", str(root), "--model", "DiT-XS/2", "--num-steps", "8", "--batch-size", "8",
                       "--log-every", "4", "--ckpt-every", "8", "--ema-snapshot-every", "2", "--num-classes", "10",
                       "--num-lin-warmup", "2", "--start-decay", "3", "--verbose", "0"])


def test_checkpoint_is_interchangeable_with_torch_adam(exp_dir):
    from mapdit_amd.optim import FusedAdamEMA
    from mapdit_amd.src.models import DIT_MODELS
    ck = torch.load(os.path.join(exp_dir, "checkpoints", "0000008.pt"), weights_only=True)
    m = DIT_MODELS["DiT-XS/2"](in_channels=4, input_size=32, num_classes=10).to(DEV)
    assert m.gemm_precision == "f16"          # the harness and the samplers above ran in the default precision; nothing here depends on it
    m.load_state_dict(ck["model"])
    # the reference resumes with torch.optim.Adam(...).load_state_dict(ck["opt"]) (train.py:57,125-132)
    adam = torch.optim.Adam(m.parameters(), lr=1e-2, betas=(0.9, 0.99))
    adam.load_state_dict(ck["opt"])
    st = adam.state_dict()["state"]
    assert len(st) == len(list(m.parameters())) and float(st[0]["step"]) == 8.0
    assert ck["opt"]["param_groups"][0]["betas"] == (0.9, 0.99) and ck["opt"]["param_groups"][0]["initial_lr"] == 1e-2
    # and the other way round: torch Adam's state dict into the fused optimiser
    opt = FusedAdamEMA(m)
    opt.load_state_dict(adam.state_dict())
    assert opt.step_count == 8
    again = opt.state_dict()
    for i, (name, p) in enumerate(m.named_parameters()):
        assert again["state"][i]["exp_avg"].shape == p.shape
        assert torch.equal(again["state"][i]["exp_avg"].cpu(), ck["opt"]["state"][i]["exp_avg"].cpu()), name
        assert torch.equal(again["state"][i]["exp_avg_sq"].cpu(), ck["opt"]["state"][i]["exp_avg_sq"].cpu()), name
    assert float(ck["opt"]["state"][3]["exp_avg_sq"].abs().sum()) > 0
    # EMA copies can be restored from a snapshot
    snap = torch.load(os.path.join(exp_dir, "ema", "0.100_0000008.pt"), weights_only=True)
    opt.load_ema_state_dict(0.1, snap["state_dict"])
    back = opt.ema_state_dict(0.1)
    for k, v in snap["state_dict"].items():
        assert torch.equal(back[k].cpu().half(), v), k


def test_sample_fid_cli(exp_dir):
    from mapdit_amd import sample_fid
    base = ["--result-dir", exp_dir, "--use-vae", "false", "--num-classes", "10", "--num-sampling-steps", "2", "--batch-size", "4"]
    # (two steps: an 8-step-old network has no business denoising - with more steps its chain blows up, in the reference too)
    path = sample_fid.main(base + ["--num-samples", "6", "--output-file", "a.npz"])
    arr = np.load(path)["arr_0"]
    assert path.endswith(os.path.join("fid_samples", "a.npz"))
    assert arr.shape == (6, 32, 32, 4) and arr.dtype == np.uint8 and arr.std() > 0
    # no guidance (cfg <= 1 -> model.forward), raw checkpoint instead of EMA, eager loop, post-hoc width not tracked in training
    path = sample_fid.main(base + ["--num-samples", "4", "--cfg-scale", "1.0", "--ckpt", "0000008", "--no-graph", "--output-file", "b.npz"])
    assert np.load(path)["arr_0"].shape == (4, 32, 32, 4)
    path = sample_fid.main(base + ["--num-samples", "4", "--ema-std", "0.075", "--output-file", "c.npz"])
    assert np.load(path)["arr_0"].shape == (4, 32, 32, 4)
    with pytest.raises(RuntimeError, match="diffusers"):                     # the VAE is out of scope and says so
        sample_fid.main(["--result-dir", exp_dir, "--num-samples", "1"])


def test_sample_and_sample_ema_cli(exp_dir, tmp_path):
    from PIL import Image
    from mapdit_amd import sample, sample_ema
    out = str(tmp_path / "grid.png")
    s = sample.main(["--result-dir", exp_dir, "--use-vae", "false", "--num-sampling-steps", "2", "--class-label", "3",
                     "--output-file", out, "--seed", "1"])
    assert s.shape == (4, 4, 32, 32) and float(s.abs().max()) <= 1.0
    lat = np.load(out + ".npy")
    assert lat.shape == (4, 4, 32, 32) and np.isfinite(lat).all()
    img = Image.open(out)
    assert img.size == (2 * 34 + 2, 2 * 34 + 2) and img.mode == "RGBA"     # 2x2 grid, 2-pixel padding like torchvision
    # bf16x3 precision and the eager loop run through the same CLI
    c = sample.main(["--result-dir", exp_dir, "--use-vae", "false", "--num-sampling-steps", "2", "--output-file", out, "--seed", "5",
                     "--precision", "bf16x3", "--no-graph", "--ckpt", "0000008", "--class-label", "9"])
    assert c.shape == (4, 4, 32, 32) and torch.isfinite(c).all()
    out2 = str(tmp_path / "ema.png")
    with pytest.raises(ValueError, match="class-label"):      # default label 88, model trained on 10 classes (reference: IndexError)
        sample_ema.main(["--result-dir", exp_dir, "--use-vae", "false", "--num-sampling-steps", "2", "--output-file", out2])
    e = sample_ema.main(["--result-dir", exp_dir, "--use-vae", "false", "--num-sampling-steps", "2", "--output-file", out2,
                         "--class-label", "0"])
    assert e.shape == (8 * 5, 4, 32, 32)
    assert Image.open(out2).size == (5 * 34 + 2, 8 * 34 + 2)
    assert np.load(out2 + ".npy").shape == (40, 4, 32, 32)


def test_graph_replays_see_reloaded_weights(exp_dir):
    """sample_ema.py swaps EMA profiles under one captured graph: GraphedSampler.refresh_weights() must rebuild the weight
    images the graph reads.  The final step (t = 0) adds no noise, so a replay there is comparable bit for bit."""
    from mapdit_amd import sampling as S
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.src.ema import calculate_posthoc_ema
    from mapdit_amd.train import get_model
    ta = S.load_train_args(exp_dir)
    m = get_model(ta).to(DEV)
    S.load_weights(m, exp_dir, 0.05, verbose=False)
    d = create_diffusion("250")
    n = 4
    z = torch.randn(2 * n, 4, 32, 32, device=DEV)
    y = torch.cat([torch.randint(0, 10, (n,)), torch.full((n,), 10)]).to(DEV)
    gs = S.GraphedSampler(m, d, z.shape, y, cfg_scale=1.5)

    def replay_t0():
        gs.img.copy_(z)
        gs.t.fill_(0)
        gs.graph.replay()
        torch.cuda.synchronize()
        return gs.img.clone()

    def eager_t0():
        with torch.no_grad():
            t0 = torch.zeros(2 * n, dtype=torch.int64, device=DEV)
            mo = d._wrap_model(m.forward_with_cfg)(z, t0, y=y, cfg_scale=1.5)
            return d._step_math(mo, z, t0, torch.zeros_like(z), False)[0]

    first = replay_t0()
    assert torch.equal(first, eager_t0())
    sd = calculate_posthoc_ema(0.15, os.path.join(exp_dir, "ema"), verbose=False)       # a width that needs the solve
    m.load_state_dict({k: v.float() for k, v in sd.items()})
    gs.refresh_weights()
    second = replay_t0()
    assert torch.equal(second, eager_t0())
    assert not torch.equal(first, second)
