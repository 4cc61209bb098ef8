"""Properties at the sizes the benchmark runs (BASELINE.json: DiT-B/2, 256 samples per GPU = 65,536 token rows), where the
oracle cannot follow in seconds: exact-integer GEMMs over every layout, exact power-of-two linearity of the backward
kernels, batch independence of a sample's logits (bit-identical in a batch of 256 and of 32), gradient additivity over
batch halves, softmax normalisation, the fixed point of the forced weight normalisation, the fused optimiser against a flat
torch restatement over all 130 M parameters, and run-to-run bit-reproducibility of the full step."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
M_FULL, D, HM, HEADS, T = 65536, 768, 3072, 12, 256


@pytest.fixture(scope="module")
def L():
    import mapdit_amd
    return mapdit_amd._lib


def st():
    return torch.cuda.current_stream().cuda_stream


def small_ints(shape, lo, hi, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, device=DEV, generator=g).to(torch.bfloat16)


@pytest.mark.parametrize("layout,m,n,k", [(0, M_FULL, HM, D),        # fc1 forward   [M,D] x [4D,D]^T
                                          (0, M_FULL, D, HM),        # fc2 forward   [M,4D] x [D,4D]^T
                                          (1, M_FULL, D, HM),        # fc1 dX        [M,4D] x [4D,D]
                                          (1, M_FULL, HM, D),        # fc2 dX        [M,D] x [D,4D]
                                          (2, HM, D, M_FULL),        # fc1 dW        [M,4D]^T x [M,D], K = 65,536 tokens, split-K
                                          (2, D, D, M_FULL),         # proj dW
                                          # widths that are odd multiples of 128 (DiT-S: 384, DiT-XL: 1152): two launches, the 256^2
                                          # kernel on the first N - 128 columns and the 128^2 kernel on the last 128 (round 4)
                                          (0, M_FULL, 384, 1536), (1, M_FULL, 384, 1536), (0, 16384, 1152, 1152), (1, 16384, 1152, 4608),
                                          (0, M_FULL, 1152, 384)])
def test_gemm_exact_on_small_integers_at_full_size(L, layout, m, n, k):
    """Operands in {-2..2} x {-1,0,1}: every product and every partial sum is an integer below 2^24, so fp32 accumulation is
    exact in ANY order and the result must equal torch's fp32 matmul bit for bit - over all 65,536 rows, every tile, every
    split-K slab (the slabs are summed here in fp32, also exactly)."""
    a_shape = (k, m) if layout == 2 else (m, k)
    b_shape = (n, k) if layout == 0 else (k, n)
    a, b = small_ints(a_shape, -2, 2, 1), small_ints(b_shape, -1, 1, 2)
    am = a.float().t() if layout == 2 else a.float()
    bm = b.float().t() if layout == 0 else b.float()
    ref = am @ bm
    assert float(ref.abs().max()) < 2 ** 24
    e = L.Epilogue()
    if layout == 2:
        tiles = ((m + 255) // 256) * ((n + 255) // 256)
        split = max(256 // tiles, 1)
        slabs = torch.empty(split, m, n, device=DEV)
        e.kind, e.out, e.ldo, e.alpha, e.split_k, e.slab_stride = L.EPI_STORE_F32, slabs.data_ptr(), n, 1.0, split, m * n
        L.lib().gemm_bf16(2, m, n, k, a.data_ptr(), m, b.data_ptr(), n, C.byref(e), st())
        torch.cuda.synchronize()
        got = slabs.sum(0)
    else:
        got = torch.empty(m, n, device=DEV)
        e.kind, e.out, e.ldo, e.alpha = L.EPI_STORE_F32, got.data_ptr(), n, 1.0
        L.lib().gemm_bf16(layout, m, n, k, a.data_ptr(), k, b.data_ptr(), k if layout == 0 else n, C.byref(e), st())
        torch.cuda.synchronize()
    assert torch.equal(got, ref)


def test_attention_full_size_normalisation_and_backward_linearity(L):
    """3,072 heads x 256 tokens (DiT-B/2, 256 samples).  Forward: with v = per-head constant rows the output must be that
    constant (softmax rows sum to one; tolerance = bf16 rounding of the probabilities, 4e-3).  Backward: scaling dO by 2 is
    exact in bf16 and fp32, so every gradient must double bit for bit."""
    B = 256
    g = torch.Generator(device=DEV).manual_seed(3)
    rows = B * HEADS * T
    q = torch.randn(rows, 64, device=DEV, generator=g)
    k = torch.randn(rows, 64, device=DEV, generator=g)
    qn = (q * 8 / (q.norm(dim=1, keepdim=True) + 1e-4)).bfloat16()
    kn = (k * 8 / (k.norm(dim=1, keepdim=True) + 1e-4)).bfloat16()
    const = torch.randn(B * HEADS, 1, 64, device=DEV, generator=g).bfloat16()
    v = const.expand(B * HEADS, T, 64).contiguous().view(rows, 64)
    o = torch.empty(B * T, HEADS * 64, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(rows, device=DEV)
    L.lib().attn_cos_fwd(qn.data_ptr(), kn.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, T, HEADS, 64, st())
    torch.cuda.synchronize()
    want = const.view(B, HEADS, 1, 64).expand(B, HEADS, T, 64).permute(0, 2, 1, 3).reshape(B * T, HEADS * 64).float()
    err = (o.float() - want).abs().max() / want.abs().max()
    assert float(err) < 4e-3, float(err)
    # backward linearity (general v)
    v2 = torch.randn(rows, 64, device=DEV, generator=g).bfloat16()
    L.lib().attn_cos_fwd(qn.data_ptr(), kn.data_ptr(), v2.data_ptr(), o.data_ptr(), lse.data_ptr(), B, T, HEADS, 64, st())
    dO = torch.randn(B * T, HEADS * 64, device=DEV, generator=g).bfloat16()
    outs = []
    for scale in (1.0, 2.0):
        d = (dO.float() * scale).bfloat16()
        delta = torch.empty(rows, device=DEV)
        dq, dk, dv = (torch.empty(rows, 64, device=DEV, dtype=torch.bfloat16) for _ in range(3))
        L.lib().attn_cos_bwd(qn.data_ptr(), kn.data_ptr(), v2.data_ptr(), d.data_ptr(), o.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                             dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, T, HEADS, 64, st())
        torch.cuda.synchronize()
        outs.append((dq.float(), dk.float(), dv.float()))
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and float(a.abs().max()) > 0
        assert torch.equal(2 * a, b)
    # the one-launch streaming backward (what the engine runs: 12 heads per persistent workgroup at this size), same properties:
    # doubling dO doubles dqkv bit for bit, a repeated launch gives the same bits, and it agrees with the two-pass result
    scales = torch.stack([8.0 / (q.norm(dim=1) + 1e-4), 8.0 / (k.norm(dim=1) + 1e-4)]).contiguous()
    fused = []
    for scale in (1.0, 2.0, 1.0):
        d = (dO.float() * scale).bfloat16()
        delta = torch.empty(rows, device=DEV)
        dqkv = torch.empty(B * T, 3 * HEADS * 64, device=DEV, dtype=torch.bfloat16)
        L.lib().attn_cos_bwd_fused(qn.data_ptr(), kn.data_ptr(), v2.data_ptr(), d.data_ptr(), o.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                   scales.data_ptr(), dqkv.data_ptr(), B, T, HEADS, 64, st())
        torch.cuda.synchronize()
        fused.append(dqkv.float())
    assert torch.isfinite(fused[0]).all() and torch.equal(2 * fused[0], fused[1]) and torch.equal(fused[0], fused[2])
    dv_two_pass = outs[0][2].view(B, HEADS, T, 64).transpose(1, 2).reshape(B * T, HEADS * 64)
    dv_fused = fused[0].view(B * T, 3, HEADS * 64)[:, 2]
    assert float((dv_fused - dv_two_pass).norm() / dv_two_pass.norm()) < 6e-3


@pytest.fixture(scope="module")
def b2():
    from mapdit_amd.src.models import DIT_MODELS
    torch.manual_seed(0)
    m = DIT_MODELS["DiT-B/2"](in_channels=4, input_size=32, num_classes=1000).to(DEV)
    # bf16: fp32's exponent range.  (The tests below back-propagate SUMS of per-sample losses, 256 x the mean the automatic fp16 loss
    # scale is chosen for: in "f16" that overflows - which is what the optimiser's non-finite guard exists for, test_f16_gpu.py.)
    m.gemm_precision = "bf16"
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(256, 4, 32, 32, device=DEV, generator=g)
    y = torch.randint(0, 1000, (256,), device=DEV, generator=g)
    t = torch.randint(0, 1000, (256,), device=DEV, generator=g)
    noise = torch.randn(256, 4, 32, 32, device=DEV, generator=g)
    return m, x, y, t, noise


def test_forced_weightnorm_converges_to_its_fixed_point_at_full_size(b2):
    """A training-mode forward rewrites every weight row to w sqrt(in) / (|w| + 1e-4) (mp_linear.py:38-40).  That map contracts
    towards |w| = sqrt(in) - 1e-4 with factor ~1e-4 / |w|: the second forward may still move rows whose initial norm was off by
    O(1) (in = 17: by ~2e-5 relative), the third one only by fp32 rounding."""
    m, x, y, t, _ = b2
    m.train()
    with torch.no_grad():
        m(x[:8], t[:8], y[:8])
        w1 = m._pflat.clone()
        m(x[:8], t[:8], y[:8])
        w2 = m._pflat.clone()
        m(x[:8], t[:8], y[:8])
    scale = w1.abs().max()
    assert float((w2 - w1).abs().max() / scale) < 5e-5
    assert float((m._pflat - w2).abs().max() / scale) < 5e-7
    rows = m.blocks[3].mlp.net[0].weight
    assert float((rows.detach().norm(dim=1) - math.sqrt(rows.shape[1])).abs().max()) < 2e-3     # |row| = sqrt(in) - 1e-4 up to rounding


def test_logits_do_not_depend_on_the_batch_they_are_computed_in(b2):
    """Samples are independent through the network: the logits of samples 0..31 out of a batch of 256 and out of a batch of 32
    must be the same BITS (every row of every GEMM accumulates in the same order whatever M is; attention is per head)."""
    m, x, y, t, _ = b2
    m.eval()
    with torch.no_grad():
        big = m(x, t, y)
        small = m(x[:32].contiguous(), t[:32].contiguous(), y[:32].contiguous())
    assert torch.isfinite(big).all()
    assert torch.equal(big[:32], small)


def test_full_size_step_is_reproducible_and_gradients_add_over_batch_halves(b2):
    """DiT-B/2, 256 samples: (1) the same step twice gives identical bits (loss, every gradient); (2) the gradient of the sum of
    per-sample losses over the batch equals the sum of the gradients of its two halves up to fp32 summation order in the weight
    gradients' token / sample reductions (2e-4 relative per tensor; measured 1e-5 ... 5e-5 on the timestep-MLP weights, whose
    gradient is a strongly cancelling sum over the samples, 1e-6 elsewhere): per-row quantities do not depend on the batch."""
    from mapdit_amd.diffusion import create_diffusion
    m, x, y, t, noise = b2
    m.eval()                                   # no weight rewrite, no label drop: the three runs see identical weights
    diff = create_diffusion("")

    def grads(sl):
        for p in m.parameters():
            p.grad = None
        loss = diff.training_losses(m, x[sl].contiguous(), t[sl].contiguous(), dict(y=y[sl].contiguous()), noise=noise[sl].contiguous())["loss"]
        loss.sum().backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), m._gflat.clone()

    l1, g1 = grads(slice(0, 256))
    l2, g2 = grads(slice(0, 256))
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    la, ga = grads(slice(0, 128))
    lb, gb = grads(slice(128, 256))
    assert torch.equal(torch.cat([la, lb]), l1)                     # per-sample losses: bit-identical
    views = {k: (o, p.numel()) for (k, p), o in zip(m.named_parameters(), m._poffs)}
    worst = 0.0
    for k, (o, n) in views.items():
        whole, parts = g1[o:o + n].double(), (ga[o:o + n].double() + gb[o:o + n].double())
        if float(whole.norm()) < 1e-9 or n == 1:         # scalar gains: sums of ~1e7 cancelling terms, covered by the golden tests
            continue
        e = float((whole - parts).norm() / whole.norm())
        worst = max(worst, e)
        # Conditioning-path weights (timestep MLP, label table, modulation linears) see the per-sample sums dshift / dscale / dgate
        # after a rounding to bf16.  A batch of 256 and a batch of 128 may sum them in a different order (the pointwise pass cuts a
        # sample's rows into pieces at small batches; MAPDIT_FUSED_RMB=1 sums them in the dX GEMM's epilogue): a few bf16 roundings
        # flip, and these strongly cancelling gradients move by up to ~3e-4.  Everything else is per-row work: 2e-4.
        cond_path = k.startswith(("t_embedder.", "y_embedder.")) or ".modulation." in k
        assert e < (1e-3 if cond_path else 2e-4), (k, e)
    print(f"gradient additivity over batch halves: worst relative difference {worst:.2e}")


def test_xl_width_column_split_backward_adds_over_batch_halves():
    """DiT-XL's width (1152 = 4 x 256 + 128) at 64 samples takes the round-5 column split of the fused dX + residual / modulate backward
    (fused epilogue on the first 1024 columns, plain store + restricted pass on the last 128); its halves of 32 samples take the
    whole-width fused epilogue with a ragged fifth column tile.  Per-sample losses are the same bits either way, and the 64-sample
    gradient is the sum of the two halves' gradients up to fp32 summation order - i.e. the split changes no per-element value."""
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.src.dit import DiT
    torch.manual_seed(0)
    m = DiT(depth=2, hidden_size=1152, patch_size=2, input_size=32, in_channels=4, num_heads=16, num_classes=10).to(DEV).eval()
    m.gemm_precision = "bf16"
    with torch.no_grad():
        for k, p_ in m.named_parameters():
            if "gain_" in k:
                p_.fill_(0.2)
    g = torch.Generator(device=DEV).manual_seed(5)
    x, noise = torch.randn(64, 4, 32, 32, device=DEV, generator=g), torch.randn(64, 4, 32, 32, device=DEV, generator=g)
    y, t = torch.randint(0, 10, (64,), device=DEV, generator=g), torch.randint(0, 1000, (64,), device=DEV, generator=g)
    diff = create_diffusion("")

    def grads(sl):
        for p_ in m.parameters():
            p_.grad = None
        loss = diff.training_losses(m, x[sl].contiguous(), t[sl].contiguous(), dict(y=y[sl].contiguous()), noise=noise[sl].contiguous())["loss"]
        loss.sum().backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), m._gflat.clone()

    l1, g1 = grads(slice(0, 64))
    l2, g2 = grads(slice(0, 64))
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    la, ga = grads(slice(0, 32))
    lb, gb = grads(slice(32, 64))
    assert torch.equal(torch.cat([la, lb]), l1)
    worst = 0.0
    for (k, p_), o in zip(m.named_parameters(), m._poffs):
        n = p_.numel()
        whole, parts = g1[o:o + n].double(), (ga[o:o + n].double() + gb[o:o + n].double())
        if float(whole.norm()) < 1e-9 or n == 1:
            continue
        e = float((whole - parts).norm() / whole.norm())
        worst = max(worst, e)
        cond_path = k.startswith(("t_embedder.", "y_embedder.")) or ".modulation." in k       # (see the DiT-B/2 test above)
        assert e < (1e-3 if cond_path else 2e-4), (k, e)
    print(f"XL width, 64 = 32 + 32 samples: worst relative difference {worst:.2e}")


def test_fused_adam_ema_over_all_parameters(b2):
    """Two fused optimiser steps over the flat 130 M-parameter buffer against the same arithmetic in torch (Adam with bias
    correction, lr 1e-2, betas (0.9, 0.99), eps 1e-8; power-function EMA ema.lerp_(w, (1 - 1/t)^(gamma+1)) for sigma_rel 0.05
    and 0.1, src/ema.py:135-140): 1e-6."""
    from oracle import dit_oracle as O
    from mapdit_amd.optim import FusedAdamEMA
    m = b2[0]
    opt = FusedAdamEMA(m, lr=1e-2, betas=(0.9, 0.99), ema_stds=(0.05, 0.1))
    m._attach_grads()
    g = torch.Generator(device=DEV).manual_seed(5)
    w = m._pflat.clone()
    mom, var = torch.zeros_like(w), torch.zeros_like(w)
    ema = {std: w.clone() for std in (0.05, 0.1)}
    for step in (1, 2):
        m._gflat.copy_(torch.randn(m._gflat.shape, device=DEV, generator=g) * 1e-3)
        for p, gv in zip(m.parameters(), m._gviews):
            p.grad = gv
        grad = m._gflat.clone()
        opt.step()
        mom = 0.9 * mom + 0.1 * grad
        var = 0.99 * var + 0.01 * grad * grad
        w = w - 1e-2 * (mom / (1 - 0.9 ** step)) / ((var / (1 - 0.99 ** step)).sqrt() + 1e-8)
        for std in ema:
            ema[std] = ema[std] + O.ema_beta(std, step) * (w - ema[std])
    torch.cuda.synchronize()
    mask = torch.zeros_like(w, dtype=torch.bool)
    for p, o in zip(m.parameters(), m._poffs):
        mask[o:o + p.numel()] = True                                    # alignment padding between slots is not a parameter
    err = ((m._pflat - w)[mask]).norm() / w[mask].norm()
    assert float(err) < 1e-6, float(err)
    names = [k for k, _ in m.named_parameters()]
    for std in (0.05, 0.1):
        sd = opt.ema_state_dict(std)
        got = torch.cat([sd[k].reshape(-1) for k in names])
        ref = torch.cat([ema[std][o:o + p.numel()] for p, o in zip(m.parameters(), m._poffs)])
        assert float((got - ref).norm() / ref.norm()) < 1e-6, std
        assert float((got - m._pflat[mask]).norm()) > 0                 # the EMA lags the weights: the comparison is not vacuous


def test_bf16_and_bf16x3_engines_agree_at_scale(b2):
    """DiT-B/2, 32 samples, eval: the bf16 fast path against the fp32-accurate bf16x3 engine (which follows the reference to 2e-5 on
    the fixtures): 3e-2 on the logits, the bf16 operand-rounding level of a 12-layer model (measured ~1e-2)."""
    m, x, y, t, _ = b2
    m.eval()
    with torch.no_grad():
        fast = m(x[:32].contiguous(), t[:32].contiguous(), y[:32].contiguous())
        m.gemm_precision = "bf16x3"
        try:
            exact = m(x[:32].contiguous(), t[:32].contiguous(), y[:32].contiguous())
        finally:
            m.gemm_precision = "bf16"
    e = float((fast.double() - exact.double()).norm() / exact.double().norm())
    print(f"bf16 vs bf16x3 logits, DiT-B/2 x 32: {e:.3e}")
    assert torch.isfinite(exact).all() and e < 3e-2


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_xl2_graphed_denoise_step_equals_eager_at_full_size(precision):
    """BASELINE config 5 (DiT-XL/2, 250-step schedule, cfg 1.5, batch 2 x 128): one replay of the captured hipGraph at t = 0 (no noise
    added) must equal the eager p_sample step bit for bit, and the device-side step counter must move."""
    from mapdit_amd.diffusion import create_diffusion
    from mapdit_amd.sampling import GraphedSampler
    from mapdit_amd.src.models import DIT_MODELS
    torch.manual_seed(0)
    m = DIT_MODELS["DiT-XL/2"](in_channels=4, input_size=32, num_classes=1000).to(DEV).eval()
    m.gemm_precision = precision
    d = create_diffusion("250")
    n = 128
    g = torch.Generator(device=DEV).manual_seed(2)
    z = torch.randn(n, 4, 32, 32, device=DEV, generator=g)
    z = torch.cat([z, z], 0)
    y = torch.cat([torch.randint(0, 1000, (n,), device=DEV, generator=g), torch.full((n,), 1000, device=DEV)])
    s = GraphedSampler(m, d, z.shape, y, cfg_scale=1.5)
    s.img.copy_(z)
    s.t.fill_(0)
    s.graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        t0 = torch.zeros(2 * n, dtype=torch.int64, device=DEV)
        mo = d._wrap_model(m.forward_with_cfg)(z, t0, y=y, cfg_scale=1.5)
        ref, _ = d._step_math(mo, z, t0, torch.zeros_like(z), False)
    assert torch.isfinite(ref).all()
    assert torch.equal(s.img, ref)
    assert int(s.t[0]) == -1
