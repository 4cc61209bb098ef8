"""Data-parallel plumbing on CPU (gloo, world_size 2): the flat-gradient reducer, the bucket slicing, the batch
sharding, and the invariant the multi-GPU path relies on — mean-of-rank-gradients == gradient of the global batch
(SURVEY.md §8e), checked with the CPU oracle as the per-rank compute."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import mapdit_amd  # noqa: F401
    from mapdit_amd.parallel import GradReducer, init_from_env, shard_batch
    from oracle import dit_oracle as O
    from oracle.diffusion_oracle import DiffusionOracle
    r, w, _ = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # 1. reducer on a flat buffer, several buckets, ragged tail
    flat = torch.arange(10_007, dtype=torch.float32) * (rank + 1)
    red = GradReducer(n_buckets=3)
    red.reduce(flat)
    assert torch.equal(flat, torch.arange(10_007, dtype=torch.float32) * 3)
    assert red.grad_scale == 0.5
    # 2. DP invariant with the oracle as the model: each rank differentiates the mean loss of ITS half of the batch
    cfg = O.DiTConfig(depth=1, hidden_size=64, patch_size=4, input_size=16, in_channels=4, num_heads=1, num_classes=10)
    sd = O.init_state_dict(cfg, seed=3, gains=0.2, perturb_reference=0.2)
    g = torch.Generator().manual_seed(4)
    n = 4
    x, y = torch.randn(n, 4, 16, 16, generator=g), torch.randint(0, 10, (n,), generator=g)
    t, noise = torch.randint(0, 1000, (n,), generator=g), torch.randn(n, 4, 16, 16, generator=g)
    drop = torch.tensor([False, True, False, False])
    d = DiffusionOracle("")

    def grads(lo, hi):
        leaf = {k: v.clone().requires_grad_(k not in O.BUFFER_KEYS) for k, v in sd.items()}
        loss = d.training_losses(lambda xx, tt, **kw: O.dit_forward(leaf, cfg, xx, tt, kw["y"], train=True, drop=drop[lo:hi]),
                                 x[lo:hi], t[lo:hi], dict(y=y[lo:hi]), noise=noise[lo:hi])["loss"].mean()
        loss.backward()
        return torch.cat([leaf[k].grad.reshape(-1) for k in leaf if k not in O.BUFFER_KEYS])

    lo, hi = shard_batch(n, rank, world)
    local = grads(lo, hi)
    red.reduce(local)
    local *= red.grad_scale
    if rank == 0:
        full = grads(0, n)
        ret["dp_err"] = float((local - full).norm() / full.norm())
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_reducer_and_dp_invariant():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["dp_err"] < 1e-5


def _verify_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import mapdit_amd  # noqa: F401
    from mapdit_amd.parallel import _range_checksum, _verify_gather, init_from_env
    init_from_env(backend="gloo")
    per = 1000
    whole = torch.arange(world * per, dtype=torch.float32) * 0.5
    mine = whole[rank * per:(rank + 1) * per]
    before = _range_checksum(mine)
    _verify_gather(whole, per, world, before, None)                # an intact gather passes on every rank
    if rank == 1:
        whole[3] += 1.0                                            # rank 1's COPY of rank 0's range arrives corrupted; rank 0's copy is fine
    try:
        _verify_gather(whole, per, world, before, None)
        ret[rank] = "passed"
    except RuntimeError as e:
        ret[rank] = str(e)
    dist.barrier()                                                 # nobody is left hanging in a collective
    dist.destroy_process_group()


def test_gather_verification_raises_on_every_rank_together():
    """parallel._verify_gather: a corruption that only ONE receiver sees must raise on EVERY rank - the verdict is
    all-reduced - naming the bad range; otherwise the healthy ranks would run into their next collective and hang there."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_verify_worker, args=(2, port, ret), nprocs=2, join=True)
    assert "ranges of ranks [0]" in ret[0] and "ranges of ranks [0]" in ret[1], dict(ret)


def test_bucket_slices_and_shards():
    import mapdit_amd  # noqa: F401
    from mapdit_amd.parallel import bucket_slices, shard_batch
    for numel, nb in ((10, 3), (1 << 20, 4), (130_190_000, 8), (5, 1)):
        sl = bucket_slices(numel, nb)
        assert sl[0][0] == 0 and sl[-1][1] == numel
        assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        assert all(lo % 1024 == 0 for lo, _ in sl)
    assert shard_batch(256, 3, 8) == (96, 128)
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


def test_lr_schedule_and_ema_betas_match_reference_fixtures():
    import numpy as np
    import mapdit_amd  # noqa: F401
    from conftest import load_golden
    from mapdit_amd.optim import calc_beta, create_lr_lambda, std_to_gamma
    g = load_golden("tables")
    np.testing.assert_allclose([std_to_gamma(0.05), std_to_gamma(0.1)], g["ema_gamma"], rtol=1e-12)
    np.testing.assert_allclose([calc_beta(0.05, 100), calc_beta(0.1, 100)], g["ema_beta_t100"], rtol=1e-12)
    lam = create_lr_lambda(400_000 // 150, 400_000 // 10)          # train.py:60-66 defaults
    assert lam(0) == 1 / 2666 and lam(2664) == 2665 / 2666 and lam(2665) == 1.0 and lam(39_999) == 1.0
    assert abs(lam(160_000) - 0.5) < 1e-12


def test_sharded_weight_pass_partition():
    """ShardedPassReducer (round 5, --grad-comm zero1w): the rows of every block linear are split over the ranks, everything else is
    replicated.  Over all ranks the optimiser ranges must cover the flat buffer: every element of a sharded weight on exactly ONE rank,
    every replicated element on EVERY rank; the weight-norm Jacobian is linear in G, so applying it to the owner's rows of the summed raw
    gradient equals the sum of the per-rank Jacobians (what the all-reduce scheme computes) - checked with the oracle's normalize()."""
    import mapdit_amd  # noqa: F401
    from mapdit_amd.parallel import ShardedPassReducer
    from mapdit_amd.src.models import DIT_MODELS
    from oracle.dit_oracle import normalize
    m = DIT_MODELS["DiT-XS/2"](in_channels=4, input_size=32, num_classes=10)
    n = m._pflat.numel()
    for world in (2, 4, 8):
        cover = torch.zeros(n, dtype=torch.int32)
        sharded = torch.zeros(n, dtype=torch.bool)
        for rank in range(world):
            r = ShardedPassReducer(m, emulate=(rank, world))
            assert len(r.weights) == 5 * m.depth and r.grad_scale == 1.0 / world
            for lo, hi in r.parts():
                assert lo % 4 == 0 and hi % 4 == 0 and hi > lo
                cover[lo:hi] += 1
            for w in r.weights:
                sharded[w["off"]:w["off"] + w["rows"] * w["cols"]] = True
                lo, hi = r.own(w)
                assert (hi - lo) * world == w["rows"] * w["cols"] and (lo - w["off"]) % w["cols"] == 0        # whole rows
        assert bool((cover[sharded] == 1).all()) and bool((cover[~sharded] == world).all())
        assert float(sharded.float().mean()) > 0.9                     # the bulk of the parameters is sharded
    m._shard = m._stage_hook = m._after_prepare_hook = None
    # linearity of the Jacobian: J_W(G1 + G2) == J_W(G1) + J_W(G2)
    g = torch.Generator().manual_seed(0)
    W = torch.randn(32, 64, generator=g, dtype=torch.float64)
    G1, G2 = torch.randn(32, 64, generator=g, dtype=torch.float64), torch.randn(32, 64, generator=g, dtype=torch.float64)

    def jac(G):
        w = W.clone().requires_grad_(True)
        (normalize(w) / 8.0).backward(G)
        return w.grad

    assert torch.allclose(jac(G1 + G2), jac(G1) + jac(G2), rtol=1e-12, atol=1e-12)
