"""Per-kernel parity tests on the MI355X, called through the C ABI (ctypes).

Checker: the CPU oracle (oracle/) and plain fp32/fp64 torch restatements of single ops.
Where a kernel takes bf16 operands the inputs are generated bf16-exact, so the only
differences left are fp32 accumulation order and the final bf16 rounding of outputs;
tolerances are stated per test.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda"


# Every test of this file runs twice: on the bf16 entry points and on their IEEE fp16 twins (mapdit.h, "16-bit operand format").
# MODE["dt"] is the torch dtype of the 16-bit tensors of the current pass; the library view maps a call to its _f16 form.
MODE = {"dt": torch.bfloat16, "f16": False}
_F16_NAMES = {"gemm_bf16": "gemm_f16", "gemm_group_tn_bf16": "gemm_group_tn_f16", "f32_to_bf16": "f32_to_f16", "f32_to_bf16_2d": "f32_to_f16_2d", "mpsilu_to_bf16": "mpsilu_to_f16"}


class _F16Calls:
    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        twin = _F16_NAMES.get(name, name + "_f16")
        return getattr(self._real, twin if hasattr(self._real, twin) else name)


class _LibView:
    def __init__(self, mod, f16):
        self._mod, self._f16 = mod, f16

    def __getattr__(self, name):
        return getattr(self._mod, name)

    def lib(self):
        real = self._mod.lib()
        return _F16Calls(real) if self._f16 else real


@pytest.fixture(scope="module", params=["bf16", "f16"])
def L(request):
    import mapdit_amd
    MODE["f16"] = request.param == "f16"
    MODE["dt"] = torch.float16 if MODE["f16"] else torch.bfloat16
    yield _LibView(mapdit_amd._lib, MODE["f16"])
    MODE["f16"], MODE["dt"] = False, torch.bfloat16


def bf16_exact(*shape, seed=0, scale=1.0):
    """Values exact in bf16 - and in fp16 too (8 significant bits, exponents far inside fp16's range at these scales)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().float()


def to_bf(x):
    return x.to(DEV).to(MODE["dt"]).contiguous()


def p(t):
    return None if t is None else t.data_ptr()


def st():
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------------
def run_gemm(L, layout, a, b, kind, M, N, K, **kw):
    ep = L.Epilogue()
    ep.kind = kind
    for k, v in kw.items():
        setattr(ep, k, v)
    lda = a.shape[1]
    ldb = b.shape[1]
    L.lib().gemm_bf16(layout, M, N, K, p(a), lda, p(b), ldb, C.byref(ep), st())
    torch.cuda.synchronize()


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 384, 128), (200, 136, 192), (1024, 768, 768), (77, 32, 64),
                                   (64, 24, 256), (8, 128, 32), (130, 128, 8),
                                   # 256x256 staggered kernel (M >= 512, N >= 256): 1..7 K-tiles, ragged edges
                                   (512, 256, 64), (512, 256, 128), (768, 512, 192), (600, 264, 448), (2048, 3072, 768),
                                   # K not a multiple of the 64-deep K-tile (zero-sourced tail): both MFMA kernels
                                   (512, 256, 32), (640, 384, 96), (1024, 768, 200), (96, 64, 40), (4608, 768, 32),
                                   # K % 8 != 0: the scalar fallback
                                   (64, 32, 12)])
def test_gemm_layouts(L, layout, shape):
    """C = A B^T in the three storage layouts, incl. ragged M/N, K tails (K % 64 != 0) and the non-MFMA fallback (K % 8 != 0)."""
    M, N, K = shape
    if layout == 2 and M % 8:
        pytest.skip("TN needs M % 8 == 0 (rows of the K-major operand are 16-byte chunks)")
    A = bf16_exact(M, K, seed=1)
    B = bf16_exact(N, K, seed=2)          # asymmetric, random: catches transposed C
    ref = A.double() @ B.double().t()
    a = to_bf(A if layout != 2 else A.t())
    b = to_bf(B if layout == 0 else B.t())
    out = torch.full((M, N), float("nan"), device=DEV)
    run_gemm(L, layout, a, b, L.EPI_STORE_F32, M, N, K, out=p(out), ldo=N, alpha=1.0)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 2e-6
    assert float((out.cpu().double() - ref).abs().max()) < 1e-3


def test_rmb_column_split_equals_the_whole_width_pass(L):
    """Round 5 (DiT-XL: hidden 1152 = 4 x 256 + 128): the dX GEMM + residual / modulate backward of a width that is an odd multiple
    of 128 runs as THREE launches - the fused EPI_RMB epilogue on the first D - 128 columns (N < D with the tensors' row stride in ldo),
    a plain 16-bit store of the last 128 columns, and mapdit_resid_mod_bwd restricted to that column range (ldx = D, every column-indexed
    pointer advanced) - and must give what the whole-width unfused pass gives: per-element outputs bit for bit, column sums to
    summation order, the scalar gain gradient from both parts' partials."""
    N, T, D, K = 4, 256, 384, 128
    D1 = D - 128
    M = N * T
    g = torch.Generator().manual_seed(91)
    xp = torch.randn(M, D, generator=g).to(DEV)
    y_up = to_bf(bf16_exact(M, D, seed=12))
    mod, mod_up = torch.randn(N, 6 * D, generator=g).to(DEV), torch.randn(N, 6 * D, generator=g).to(DEV)
    gain = torch.tensor(0.37, device=DEV)
    dxo16 = torch.randn(M, D, generator=g).to(DEV).to(MODE["dt"])
    ad, bd = to_bf(bf16_exact(M, K, seed=13)), to_bf(bf16_exact(K, D, seed=14) * 0.25)          # NN: dX = dy W
    ca, cb = 0.7 / math.sqrt(0.58), 0.3 / math.sqrt(0.58)

    def args(dxm, dx_bf, dy, dmod, dmod_up, part, c0=0, width=D, ldx=0):
        a = L.ResidModBwd()
        el = lambda t, n: t.data_ptr() + n * t.element_size()
        a.dxo_bf, a.dxm, a.x = el(dxo16, c0), (el(dxm, c0) if dxm is not None else None), el(xp, c0)
        a.shift, a.scale, a.gain, a.ldmod = el(mod, 3 * D + c0), el(mod, 4 * D + c0), p(gain), 6 * D
        a.y_up, a.g_up, a.ldg_up = el(y_up, c0), el(mod_up, 5 * D + c0), 6 * D
        a.dy_up, a.dg_up, a.ldd_up = el(dy, c0), el(dmod_up, 5 * D + c0), 6 * D
        a.dx_bf = el(dx_bf, c0)
        a.dshift, a.dscale, a.ldd = el(dmod, 3 * D + c0), el(dmod, 4 * D + c0), 6 * D
        a.dgain_part = p(part)
        a.n_samples, a.T, a.D, a.ca, a.cb, a.ldx = N, T, width, ca, cb, ldx
        return a

    mk = lambda: (torch.zeros(M, D, device=DEV, dtype=MODE["dt"]), torch.zeros(M, D, device=DEV, dtype=MODE["dt"]),
                  torch.zeros(N, 6 * D, device=DEV), torch.zeros(N, 6 * D, device=DEV), torch.zeros(4096, device=DEV))
    # whole width: plain GEMM store + the pass
    gm = torch.zeros(M, D, device=DEV, dtype=MODE["dt"])
    run_gemm(L, 1, ad, bd, L.EPI_STORE_BF16, M, D, K, out=p(gm), ldo=D, alpha=1.0)
    dx_w, dy_w, dmod_w, dmu_w, part_w = mk()
    a = args(gm, dx_w, dy_w, dmod_w, dmu_w, part_w)
    npart = C.c_int(0)
    a.gain_partials_out = C.pointer(npart)
    L.lib().resid_mod_bwd(C.byref(a), st())
    dgain_w = torch.zeros((), device=DEV)
    L.lib().reduce_partials(p(part_w), npart.value, p(dgain_w), 0, st())
    # split: fused epilogue on [0, D1), plain store + restricted pass on [D1, D)
    dx_s, dy_s, dmod_s, dmu_s, part_s = mk()
    a1 = args(None, dx_s, dy_s, dmod_s, dmu_s, part_s)
    e = L.Epilogue()
    e.kind, e.ldo, e.rmb = L.EPI_RMB, D, C.addressof(a1)
    L.lib().gemm_tuning(256, 2, 0)
    try:
        L.lib().gemm_bf16(1, M, D1, K, p(ad), K, p(bd), D, C.byref(e), st())
    finally:
        L.lib().gemm_tuning(0, 2, 0)
    n1 = (M // 256) * (D1 // 256)
    gm2 = torch.zeros(M, D, device=DEV, dtype=MODE["dt"])
    ep = L.Epilogue()
    ep.kind, ep.out, ep.ldo, ep.alpha = L.EPI_STORE_BF16, gm2.data_ptr() + D1 * 2, D, 1.0
    L.lib().gemm_bf16(1, M, 128, K, p(ad), K, bd.data_ptr() + D1 * 2, D, C.byref(ep), st())
    part2 = torch.zeros(4096, device=DEV)
    a2 = args(gm2, dx_s, dy_s, dmod_s, dmu_s, part2, c0=D1, width=128, ldx=D)
    npart2 = C.c_int(0)
    a2.gain_partials_out = C.pointer(npart2)
    L.lib().resid_mod_bwd(C.byref(a2), st())
    torch.cuda.synchronize()
    part_s[n1:n1 + npart2.value] = part2[:npart2.value]
    dgain_s = torch.zeros((), device=DEV)
    L.lib().reduce_partials(p(part_s), n1 + npart2.value, p(dgain_s), 0, st())
    torch.cuda.synchronize()
    assert torch.equal(gm2[:, D1:], gm[:, D1:])
    assert torch.equal(dx_s, dx_w) and torch.equal(dy_s, dy_w), "per-element outputs must not depend on the split"
    assert float(dx_w.float().abs().sum()) > 0 and float(dy_w[:, D1:].float().abs().sum()) > 0
    assert rel_err(dmod_s.cpu().numpy(), dmod_w.cpu().numpy()) < 1e-6 and rel_err(dmu_s.cpu().numpy(), dmu_w.cpu().numpy()) < 1e-6
    assert abs(dgain_s.item() - dgain_w.item()) < 1e-5 * abs(dgain_w.item()) + 1e-4


def test_gemm_column_split_shape_on_the_scalar_fallback(L):
    """a result whose width is an odd multiple of 128 on ~one round of 256^2 tiles is run as two launches (256^2 kernel on
    N - 128 columns + 128^2 kernel on the last 128).  A TN operand with M % 8 != 0 (padded leading dimension, so every alignment test
    passes) is NOT MFMA-eligible: both halves would fall to the scalar kernel, which knows no column offset - the second half would
    land on columns [0, 128).  The split must not be taken; the whole result comes from one scalar launch and is exact."""
    M, N, K = 49156, 384, 64                       # 193 row tiles x 1 column tile of 256 (+128): inside the split's window; M % 8 == 4
    lda = M + 4
    A = bf16_exact(K, M, seed=31)
    B = bf16_exact(K, N, seed=32)
    a = torch.zeros(K, lda, dtype=MODE["dt"], device=DEV)
    a[:, :M] = to_bf(A)
    b = to_bf(B)
    out = torch.full((M, N), float("nan"), device=DEV)
    ep = L.Epilogue()
    ep.kind, ep.out, ep.ldo, ep.alpha = L.EPI_STORE_F32, p(out), N, 1.0
    L.lib().gemm_bf16(2, M, N, K, p(a), lda, p(b), N, C.byref(ep), st())
    torch.cuda.synchronize()
    ref = (A.double().t() @ B.double()).float()
    assert torch.isfinite(out).all(), "columns left unwritten"
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 2e-6


def test_gemm_split_k_slabs_feed_weightnorm_bwd(L):
    """dW = dy^T x with K cut into slabs; mapdit_weightnorm_bwd adds the slabs in order (deterministic split-K)."""
    from oracle.dit_oracle import normalize
    _split_k_case(L, 256, 128, 1024, 8)       # 128^2 kernel
    _split_k_case(L, 768, 256, 2048, 4)       # 256^2 staggered kernel
    _split_k_case(L, 512, 512, 1344, 7)       # 3 K-tiles per slab (odd tile count)
    _split_k_case(L, 512, 256, 1216, 5)       # 19 K-tiles over 5 slabs: uneven ranges 4,4,4,4,3


def _split_k_case(L, rows, cols, K, S):
    from oracle.dit_oracle import normalize
    dy, x = bf16_exact(K, rows, seed=21), bf16_exact(K, cols, seed=22)
    G_ref = (dy.double().t() @ x.double()).float()
    slabs = torch.full((S, rows, cols), float("nan"), device=DEV)
    run_gemm(L, 2, to_bf(dy), to_bf(x), L.EPI_STORE_F32, rows, cols, K, out=p(slabs), ldo=cols, alpha=1.0, split_k=S,
             slab_stride=rows * cols)
    assert rel_err(slabs.sum(0).cpu().numpy(), G_ref.numpy()) < 2e-6
    k0 = 64 * ((K // 64) // S + (1 if (K // 64) % S else 0))          # slab 0 owns the first ceil(tiles/S) K-tiles
    assert float((slabs[0].cpu() - (dy[:k0].double().t() @ x[:k0].double()).float()).abs().max()) < 1e-3
    W = torch.randn(rows, cols, generator=torch.Generator().manual_seed(23))
    Wr = W.clone().requires_grad_(True)
    (normalize(Wr) / math.sqrt(cols)).backward(G_ref)
    dW = torch.zeros(rows, cols, device=DEV)
    Wd = W.to(DEV)
    L.lib().weightnorm_bwd(p(Wd), p(slabs), cols, S, rows * cols, p(dW), rows, cols, 1.0, 0, st())
    torch.cuda.synchronize()
    assert rel_err(dW.cpu().numpy(), Wr.grad.numpy()) < 2e-5
    with pytest.raises(L.MapditError):                      # more slabs than 64-wide K-tiles
        run_gemm(L, 2, to_bf(dy), to_bf(x), L.EPI_STORE_F32, rows, cols, K, out=p(slabs), ldo=cols, alpha=1.0,
                 split_k=K // 64 + 1, slab_stride=rows * cols)


@pytest.mark.parametrize("S", [1, 2])
def test_gemm_group_equals_single_launches(L, S):
    """mapdit_gemm_group_tn_*: the three large weight gradients of a DiT-XL block ([4608, 1152], [1152, 4608], [3456, 1152]: 90 + 90 + 70 tiles of
    256^2, ragged fifth tile column) plus a small fourth item as ONE launch == one launch each with the same split_k, bit for bit; and the sum
    of the slabs == the product."""
    import numpy as np
    K = 512
    shapes = [(4608, 1152), (1152, 4608), (3456, 1152), (256, 264)]
    dys = [to_bf(bf16_exact(K, r, seed=40 + i)) for i, (r, c) in enumerate(shapes)]
    xs = [to_bf(bf16_exact(K, c, seed=50 + i)) for i, (r, c) in enumerate(shapes)]
    single = [torch.full((S, r, c), float("nan"), device=DEV) for r, c in shapes]
    for (r, c), dy, x, o in zip(shapes, dys, xs, single):
        run_gemm(L, 2, dy, x, L.EPI_STORE_F32, r, c, K, out=p(o), ldo=c, alpha=0.5, split_k=S, slab_stride=r * c)
    grouped = [torch.full((S, r, c), float("nan"), device=DEV) for r, c in shapes]
    items = (L.GemmGroupItem * len(shapes))()
    for i, ((r, c), dy, x, o) in enumerate(zip(shapes, dys, xs, grouped)):
        items[i] = L.GemmGroupItem(A=p(dy), lda=r, B=p(x), ldb=c, M=r, N=c, out=p(o), ldo=c, alpha=0.5, slab_stride=r * c)
    L.lib().gemm_group_tn_bf16(len(shapes), C.cast(items, C.c_void_p), K, S, st())
    torch.cuda.synchronize()
    for (r, c), dy, x, a, b in zip(shapes, dys, xs, single, grouped):
        assert torch.equal(a, b), (r, c)
        ref = 0.5 * (dy.float().double().t() @ x.float().double()).float()
        assert rel_err(b.sum(0).cpu().numpy(), ref.cpu().numpy()) < 2e-6
    with pytest.raises(L.MapditError):                      # an item off the MFMA path (M % 8 != 0) is refused, nothing is launched
        items[3] = L.GemmGroupItem(A=p(dys[3]), lda=256, B=p(xs[3]), ldb=264, M=250, N=264, out=p(grouped[3]), ldo=264, alpha=1.0, slab_stride=256 * 264)
        L.lib().gemm_group_tn_bf16(len(shapes), C.cast(items, C.c_void_p), K, S, st())


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("M,N,K", [(2048, 128, 2304), (16384, 128, 1152), (1000, 256, 512), (320, 384, 64)])
def test_gemm_64_row_tiles_give_the_128_kernel_s_bits(L, layout, M, N, K):
    """Results of few 128^2 tiles with a 16-bit store or RESID epilogue run on 64 x 128 tiles (gemm_mfma64_kernel: DiT-XL's 128-column strips): every
    element is accumulated by the same MFMA steps in the same K order, so a STORE_BF16 result == the fp32 result of the 128^2 kernel (STORE_F32
    never takes the 64-row form) rounded to the operand format, bit for bit; ragged M included."""
    a = bf16_exact(M, K, seed=71)
    b = bf16_exact(N, K, seed=72) if layout == 0 else bf16_exact(K, N, seed=72)
    f32 = torch.zeros(M, N, device=DEV)
    run_gemm(L, layout, to_bf(a), to_bf(b), L.EPI_STORE_F32, M, N, K, out=p(f32), ldo=N, alpha=1.0)
    o16 = torch.zeros(M, N, device=DEV, dtype=MODE["dt"])
    run_gemm(L, layout, to_bf(a), to_bf(b), L.EPI_STORE_BF16, M, N, K, out=p(o16), ldo=N, alpha=1.0)
    assert torch.equal(o16, f32.to(MODE["dt"]))
    ref = (a.double() @ (b.double().t() if layout == 0 else b.double())).float()
    assert rel_err(f32.cpu().numpy(), ref.numpy()) < 2e-6


def test_gemm_identity_asymmetric(L):
    """A = I with an asymmetric B: output must be B^T exactly (guide: catches row/col swaps)."""
    N, K = 128, 128
    A = torch.eye(128)
    B = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251
    out = torch.zeros(128, N, device=DEV)
    run_gemm(L, 0, to_bf(A), to_bf(B), L.EPI_STORE_F32, 128, N, K, out=p(out), ldo=N, alpha=1.0)
    assert torch.equal(out.cpu(), B.t().contiguous())


def test_gemm_epilogues(L):
    M, N, K, T = 256, 256, 128, 64
    A, B = bf16_exact(M, K, seed=3), bf16_exact(N, K, seed=4, scale=0.2)
    acc = (A.double() @ B.double().t()).float()
    a, b = to_bf(A), to_bf(B)
    # bf16 store with alpha
    out = torch.zeros(M, N, device=DEV, dtype=MODE["dt"])
    run_gemm(L, 0, a, b, L.EPI_STORE_BF16, M, N, K, out=p(out), ldo=N, alpha=0.5)
    assert rel_err(out.float().cpu().numpy(), (0.5 * acc).numpy()) < 3e-3
    # accumulate fp32
    base = torch.randn(M, N)
    o2 = base.to(DEV).clone()
    run_gemm(L, 0, a, b, L.EPI_STORE_F32, M, N, K, out=p(o2), ldo=N, alpha=2.0, accumulate=1)
    assert rel_err(o2.cpu().numpy(), (base + 2 * acc).numpy()) < 2e-6
    # silu2
    pre = torch.zeros(M, N, device=DEV, dtype=MODE["dt"])
    act = torch.zeros_like(pre)
    run_gemm(L, 0, a, b, L.EPI_SILU2, M, N, K, out=p(pre), out2=p(act), ldo=N)
    assert rel_err(pre.float().cpu().numpy(), acc.numpy()) < 3e-3
    assert rel_err(act.float().cpu().numpy(), (torch.nn.functional.silu(acc) / 0.596).numpy()) < 3e-3
    # silu2 with the derivative factor instead of the pre-activation, and the product epilogue that consumes it
    dfac = torch.zeros_like(pre)
    act2 = torch.zeros_like(pre)
    run_gemm(L, 0, a, b, L.EPI_SILU2_GRAD, M, N, K, out=p(dfac), out2=p(act2), ldo=N)
    leaf = acc.clone().requires_grad_(True)
    (torch.nn.functional.silu(leaf) / 0.596).sum().backward()
    assert rel_err(dfac.float().cpu().numpy(), leaf.grad.numpy()) < 3e-3
    assert rel_err(act2.float().cpu().numpy(), (torch.nn.functional.silu(acc) / 0.596).numpy()) < 3e-3
    act3 = torch.zeros_like(pre)
    run_gemm(L, 0, a, b, L.EPI_SILU2_GRAD, M, N, K, out=None, out2=p(act3), ldo=N)          # inference form: no factor
    # (two instruction sequences for the same sigmoid: an output now and then rounds the other way - 2^-9 in bf16, 2^-12 in fp16)
    assert rel_err(act3.float().cpu().numpy(), act2.float().cpu().numpy()) < 1e-5
    prod = torch.zeros_like(pre)
    run_gemm(L, 0, a, b, L.EPI_MUL_AUX, M, N, K, out=p(prod), aux=p(dfac), ldo=N)
    assert rel_err(prod.float().cpu().numpy(), (acc * dfac.float().cpu()).numpy()) < 3e-3
    # residual: xout = ca*x + cb*gate[m/T]*acc
    x = torch.randn(M, N)
    gate = torch.randn(M // T, 3 * N)
    xo = torch.zeros(M, N, device=DEV)
    y = torch.zeros(M, N, device=DEV, dtype=MODE["dt"])
    xd, gd = x.to(DEV), gate.to(DEV)
    run_gemm(L, 0, a, b, L.EPI_RESID, M, N, K, out=p(y), out2=p(xo), aux=p(xd), gate=gd.data_ptr() + 4 * N, ldg=3 * N,
             rows_per_sample=T, ldo=N, alpha=0.7, beta=0.3)
    ref = 0.7 * x + 0.3 * gate[:, N:2 * N].repeat_interleave(T, 0) * acc
    assert rel_err(xo.cpu().numpy(), ref.numpy()) < 2e-6
    assert rel_err(y.float().cpu().numpy(), acc.numpy()) < 3e-3
    # dsilu
    h = bf16_exact(M, N, seed=9)
    o3 = torch.zeros(M, N, device=DEV, dtype=MODE["dt"])
    hd = to_bf(h)
    run_gemm(L, 0, a, b, L.EPI_DSILU, M, N, K, out=p(o3), aux=p(hd), ldo=N)
    hh = h.clone().requires_grad_(True)
    (torch.nn.functional.silu(hh) / 0.596).backward(acc)
    assert rel_err(o3.float().cpu().numpy(), hh.grad.numpy()) < 3e-3


@pytest.mark.parametrize("T", [256, 64])
def test_resid_epilogue_same_bits_on_every_kernel(L, T):
    """RESID with the fused next-branch modulate (dit_block.py:35-36 + utils.py:11-16) on [2048, 512] x K = 192: the 128^2 kernel, the
    256^2 kernel's guarded epilogue (T = 64: four samples per tile) and its straight-line instantiation (T = 256: interior tiles, each
    inside one sample, every optional output present) must agree bit for bit - a sample's activations do not depend on which kernel
    its batch size selects."""
    M, N, K = 2048, 512, 192
    A, B = bf16_exact(M, K, seed=21), bf16_exact(N, K, seed=22, scale=0.2)
    a, b = to_bf(A), to_bf(B)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(M, N, generator=g).to(DEV)
    gate = torch.randn(M // T, 3 * N, generator=g).to(DEV)
    mod = torch.randn(M // T, 2 * N, generator=g).to(DEV)
    gain = torch.tensor([0.3], device=DEV)
    res = []
    for tile in (128, 256):
        L.lib().gemm_tuning(tile, 2, 0)
        xo = torch.zeros(M, N, device=DEV)
        y = torch.zeros(M, N, device=DEV, dtype=MODE["dt"])
        xm = torch.zeros(M, N, device=DEV, dtype=MODE["dt"])
        run_gemm(L, 0, a, b, L.EPI_RESID, M, N, K, out=p(y), out2=p(xo), aux=p(x), gate=gate.data_ptr() + 4 * N, ldg=3 * N,
                 rows_per_sample=T, ldo=N, alpha=0.7, beta=0.3, out3=p(xm), shift2=p(mod), scale2=mod.data_ptr() + 4 * N, gain2=p(gain),
                 ld2=2 * N)
        res.append((xo, y, xm))
    L.lib().gemm_tuning(0, 2, 0)
    for name, u, v in zip(("xout", "y", "xm"), res[0], res[1]):
        assert torch.equal(u, v), (name, int((u != v).sum()), float((u.float() - v.float()).abs().max()))
    acc = (A.double() @ B.double().t()).float()
    ref = 0.7 * x.cpu() + 0.3 * gate.cpu()[:, N:2 * N].repeat_interleave(T, 0) * acc
    assert rel_err(res[1][0].cpu().numpy(), ref.numpy()) < 2e-6
    den = math.sqrt(0.7 ** 2 + 0.3 ** 2)
    xm_ref = (0.7 * ref * mod.cpu()[:, N:].repeat_interleave(T, 0) + 0.3 * mod.cpu()[:, :N].repeat_interleave(T, 0)) / den
    assert rel_err(res[1][2].float().cpu().numpy(), xm_ref.numpy()) < 3e-3


def test_gemm_rejects_bad_args(L):
    a = torch.zeros(64, 64, device=DEV, dtype=MODE["dt"])
    ep = L.Epilogue()
    ep.kind = L.EPI_STORE_F32
    ep.out = None
    with pytest.raises(L.MapditError):
        L.lib().gemm_bf16(0, 64, 64, 64, p(a), 64, p(a), 64, C.byref(ep), st())
    ep.out = p(a)
    ep.ldo = 64
    with pytest.raises(L.MapditError):
        L.lib().gemm_bf16(0, 64, 60, 64, p(a), 64, p(a), 64, C.byref(ep), st())   # N % 8


# ---------------------------------------------------------------------------------------------------
# weight norm / optimiser
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(384, 128), (16, 17), (11, 128), (768, 3072)])
@pytest.mark.parametrize("forced", [0, 1])
def test_weightnorm(L, rows, cols, forced):
    from oracle.dit_oracle import normalize
    g = torch.Generator().manual_seed(5)
    W = torch.randn(rows, cols, generator=g) * 1.7
    Wd = W.to(DEV).clone()
    wb = torch.zeros(rows, cols, device=DEV, dtype=MODE["dt"])
    wf = torch.zeros(rows, cols, device=DEV)
    inv = torch.zeros(rows, device=DEV)
    L.lib().weightnorm_fwd(p(Wd), rows, cols, forced, 1.0, p(wb), p(wf), p(inv), st())
    torch.cuda.synchronize()
    Wn = normalize(W) if forced else W
    w_eff = normalize(Wn) / math.sqrt(cols)
    assert rel_err(Wd.cpu().numpy(), Wn.numpy()) < 1e-6
    assert rel_err(wf.cpu().numpy(), w_eff.numpy()) < 1e-6
    assert rel_err(wb.float().cpu().numpy(), w_eff.numpy()) < 3e-3
    # backward
    G = torch.randn(rows, cols + 3, generator=g)
    Wr = Wn.clone().requires_grad_(True)
    (normalize(Wr) / math.sqrt(cols)).backward(G[:, :cols])
    dW = torch.zeros(rows, cols, device=DEV)
    Gd = G.to(DEV).contiguous()
    Wnd = Wn.to(DEV)
    L.lib().weightnorm_bwd(p(Wnd), p(Gd), cols + 3, 1, 0, p(dW), rows, cols, 1.0, 0, st())
    torch.cuda.synchronize()
    assert rel_err(dW.cpu().numpy(), Wr.grad.numpy()) < 2e-5


@pytest.mark.parametrize("rows,cols", [(384, 128), (16, 17), (768, 3072)])
@pytest.mark.parametrize("forced", [0, 1])
def test_weightnorm_plain_flag(L, rows, cols, forced):
    """MAPDIT_WN_PLAIN (bit 1 of `forced` / `accumulate` / job.flags; README.md:60 off form, parity unpinned): image = W / sqrt(cols) of the (optionally
    rewritten) master, backward dW = G / sqrt(cols) (+ the accumulate bit), slim and batch forms the same bits as the plain launch."""
    import numpy as np
    from oracle.dit_oracle import normalize
    PLAIN = L.WN_PLAIN
    g = torch.Generator().manual_seed(6)
    W = torch.randn(rows, cols, generator=g) * 1.7
    Wd = W.to(DEV).clone()
    wf = torch.zeros(rows, cols, device=DEV)
    L.lib().weightnorm_fwd(p(Wd), rows, cols, forced | PLAIN, 1.0, None, p(wf), None, st())
    torch.cuda.synchronize()
    Wn = normalize(W) if forced else W
    assert rel_err(Wd.cpu().numpy(), Wn.numpy()) < 1e-6
    assert rel_err(wf.cpu().numpy(), (Wn / math.sqrt(cols)).numpy()) < 1e-6
    G = torch.randn(3, rows, cols, generator=g)               # three split-K slabs
    want = G.sum(0) / math.sqrt(cols)
    outs = []
    for fn in ("weightnorm_bwd", "weightnorm_bwd_slim"):
        if fn.endswith("slim") and cols % 4:
            continue
        Gd = G.to(DEV).contiguous()
        dW = torch.ones(rows, cols, device=DEV)
        getattr(L.lib(), fn)(p(Wd), p(Gd), cols, 3, rows * cols, p(dW), rows, cols, 1.0, PLAIN | 1, st())
        torch.cuda.synchronize()
        assert rel_err(dW.cpu().numpy(), (want + 1.0).numpy()) < 1e-6, fn
        outs.append(dW.cpu())
    assert len(outs) < 2 or torch.equal(outs[0], outs[1])
    if cols % 4 == 0:                                         # the in-place batch form on one summed slab
        Gs = G.sum(0).to(DEV).contiguous()
        ref = torch.zeros(rows, cols, device=DEV)
        L.lib().weightnorm_bwd(p(Wd), p(Gs.clone()), cols, 1, 0, p(ref), rows, cols, 1.0, PLAIN, st())
        jobs = (L.WnJob * 1)()
        jobs[0] = L.WnJob(W=p(Wd), rows=rows, cols=cols, out_scale=1.0, first_block=0, w_bf16=None, w_f32=p(Gs), flags=PLAIN)
        raw = torch.from_numpy(np.frombuffer(bytes(jobs), dtype=np.uint8).copy()).to(DEV)
        L.lib().weightnorm_bwd_batch(p(raw), 1, (rows + 3) // 4, st())
        torch.cuda.synchronize()
        assert torch.equal(Gs, ref)


@pytest.mark.parametrize("N,T,D", [(2, 64, 128), (3, 16, 384), (1, 256, 1152), (2, 8, 2048)])
def test_layernorm_modulate_and_its_backward(L, N, T, D):
    """mapdit_ln_modulate_fwd / mapdit_ln_bwd_merge (README.md:64 off form, parity unpinned: upstream DiT's LayerNorm without affine, eps 1e-6,
    in front of modulate) against torch.nn.functional.layer_norm + the oracle's modulate and their autograd, with the pass-through term as an
    fp32 tensor, a 16-bit tensor, and absent."""
    from oracle.dit_oracle import modulate
    g = torch.Generator().manual_seed(31)
    M = N * T
    x = torch.randn(M, D, generator=g) * 2.0 + 0.7
    sh, sc = torch.randn(N, D + 4, generator=g), 1.0 + 0.3 * torch.randn(N, D + 4, generator=g)
    gain = torch.tensor(0.3)
    xr = x.clone().requires_grad_(True)
    xh_ref = torch.nn.functional.layer_norm(xr.view(N, T, D), (D,), eps=1e-6)
    u_ref = modulate(xh_ref, sh[:, :D], sc[:, :D], gain)
    xd, shd, scd, gd = x.to(DEV), sh.to(DEV), sc.to(DEV), gain.to(DEV)
    xh, rstd = torch.zeros(M, D, device=DEV), torch.zeros(M, device=DEV)
    out = torch.zeros(M, D, device=DEV, dtype=MODE["dt"])
    lib = L.lib()
    lib.ln_modulate_fwd(p(xd), p(shd), p(scd), D + 4, p(gd), p(xh), p(rstd), p(out), N, T, D, st())
    out2 = torch.zeros_like(out)
    lib.ln_modulate_fwd(p(xd), p(shd), p(scd), D + 4, p(gd), None, None, p(out2), N, T, D, st())          # inference: nothing kept
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    assert rel_err(xh.cpu().numpy(), xh_ref.detach().reshape(M, D).numpy()) < 2e-6
    assert rel_err(rstd.cpu().numpy(), (1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-6)).numpy()) < 2e-6
    assert rel_err(out.float().cpu().numpy(), u_ref.detach().reshape(M, D).numpy()) < (4e-3 if MODE["dt"] == torch.bfloat16 else 5e-4)
    # backward of the normalisation alone, merged with ca * dxo
    gh = torch.randn(M, D, generator=g)
    xh_ref.backward(gh.view(N, T, D))
    dxo = torch.randn(M, D, generator=g).to(MODE["dt"]).float()
    ghd, d32, d16 = gh.to(DEV), dxo.to(DEV), dxo.to(DEV).to(MODE["dt"])
    for kind in ("f32", "16", "none"):
        o = torch.zeros(M, D, device=DEV)
        lib.ln_bwd_merge(p(ghd), p(xh), p(rstd), p(d32) if kind == "f32" else None, p(d16) if kind == "16" else None, 0.9, p(o), M, D, st())
        torch.cuda.synchronize()
        want = xr.grad + (0.9 * dxo if kind != "none" else 0.0)
        assert rel_err(o.cpu().numpy(), want.numpy()) < 1e-5, kind


def test_weightnorm_bwd_group_equals_single_launches(L):
    """mapdit_weightnorm_bwd_group (up to four weights, each over its own slabs, one launch) == a launch each, bit for bit - with the accumulate and
    MAPDIT_WN_PLAIN bits per item and ragged row counts."""
    shapes = [(768, 3072, 2, 0), (3072, 768, 3, 1), (2304, 768, 1, L.WN_PLAIN), (10, 768, 2, 0)]
    g = torch.Generator().manual_seed(9)
    Ws = [torch.randn(r, c, generator=g).to(DEV) for r, c, _, _ in shapes]
    Gs = [torch.randn(S, r, c, generator=g) for r, c, S, _ in shapes]
    outs = {}
    for which in ("single", "group"):
        Gd = [x.to(DEV).contiguous() for x in Gs]
        dWs = [torch.full((r, c), 0.5, device=DEV) for r, c, _, _ in shapes]
        if which == "single":
            for W, G_, d, (r, c, S, fl) in zip(Ws, Gd, dWs, shapes):
                L.lib().weightnorm_bwd(p(W), p(G_), c, S, r * c, p(d), r, c, 1.25, fl, st())
        else:
            items = (L.WnBwdItem * len(shapes))()
            for i, (W, G_, d, (r, c, S, fl)) in enumerate(zip(Ws, Gd, dWs, shapes)):
                items[i] = L.WnBwdItem(W=p(W), G=p(G_), ldg=c, nslabs=S, slab_stride=r * c, dW=p(d), rows=r, cols=c, out_scale=1.25, flags=fl)
            L.lib().weightnorm_bwd_group(len(shapes), C.cast(items, C.c_void_p), st())
        torch.cuda.synchronize()
        outs[which] = [d.cpu() for d in dWs] + [x[0].cpu() for x in Gd]      # (slab 0 holds the summed slabs afterwards)
    for a, b in zip(outs["single"], outs["group"]):
        assert torch.equal(a, b)


def test_reduce_slabs_group_equals_single_launches(L):
    """mapdit_reduce_slabs_group == mapdit_reduce_slabs per buffer, bit for bit."""
    sizes, S = [768 * 3072, 3072 * 768, 2304 * 768, 768 * 768], 2
    g = torch.Generator().manual_seed(10)
    slabs = [torch.randn(S, n, generator=g).to(DEV) for n in sizes]
    single = [torch.zeros(n, device=DEV) for n in sizes]
    group = [torch.zeros(n, device=DEV) for n in sizes]
    for o, sl, n in zip(single, slabs, sizes):
        L.lib().reduce_slabs(p(o), p(sl), S, n, n, st())
    arr = lambda ty, vals: (ty * len(vals))(*vals)
    L.lib().reduce_slabs_group(len(sizes), C.cast(arr(C.c_void_p, [p(o) for o in group]), C.c_void_p), C.cast(arr(C.c_void_p, [p(s_) for s_ in slabs]), C.c_void_p),
                               C.cast(arr(C.c_long, sizes), C.c_void_p), C.cast(arr(C.c_long, sizes), C.c_void_p), S, st())
    torch.cuda.synchronize()
    for a, b, sl in zip(single, group, slabs):
        assert torch.equal(a, b) and torch.equal(a, sl[0] + sl[1])


def test_weightnorm_batch_equals_single_launches(L):
    """mapdit_weightnorm_fwd_batch (one launch for every weight, device job table) against one launch per weight: bit-equal
    rewritten masters and images, for ragged row counts (rows % 4 != 0) and both output kinds."""
    import numpy as np
    shapes = [(384, 128, 1.0, "bf"), (10, 257, 1.0, "f32"), (1001, 64, 8.0, "f32"), (7, 128, 1.0, "bf"), (2304, 768, 1.0, "bf")]
    g = torch.Generator().manual_seed(5)
    masters = [torch.randn(r, c, generator=g) for r, c, _, _ in shapes]
    for forced in (0, 1):
        singles, batch = [], []
        for which in ("single", "batch"):
            Ws = [m.clone().to(DEV) for m in masters]
            outs = [torch.zeros(r, c, device=DEV, dtype=MODE["dt"] if kind == "bf" else torch.float32) for r, c, _, kind in shapes]
            if which == "single":
                for W, o, (r, c, sc, kind) in zip(Ws, outs, shapes):
                    L.lib().weightnorm_fwd(p(W), r, c, forced, sc, p(o) if kind == "bf" else None, p(o) if kind == "f32" else None, None, st())
            else:
                jobs = (L.WnJob * len(shapes))()
                blocks = 0
                for j, (W, o, (r, c, sc, kind)) in enumerate(zip(Ws, outs, shapes)):
                    jobs[j] = L.WnJob(W=p(W), rows=r, cols=c, out_scale=sc, first_block=blocks, w_bf16=p(o) if kind == "bf" else None,
                                      w_f32=p(o) if kind == "f32" else None)
                    blocks += (r + 3) // 4
                raw = torch.from_numpy(np.frombuffer(bytes(jobs), dtype=np.uint8).copy()).to(DEV)
                L.lib().weightnorm_fwd_batch(p(raw), len(shapes), blocks, forced, st())
            torch.cuda.synchronize()
            (singles if which == "single" else batch).extend([w.cpu() for w in Ws] + [o.cpu() for o in outs])
        for a, b in zip(singles, batch):
            assert torch.equal(a, b)
    # the optional two-term split image [hi | lo | hi] of the effective weight (conditioning weights of the bf16 engine):
    # hi = the bf16 image, hi + lo = the fp32 effective weight to 2^-16 relative, for a vector-width and a ragged column count
    for r, c in ((96, 256), (10, 257)):
        W = torch.randn(r, c, generator=g).to(DEV)
        wb = torch.zeros(r, c, device=DEV, dtype=MODE["dt"])
        wf = torch.zeros(r, c, device=DEV)
        w3 = torch.zeros(r, 3 * c, device=DEV, dtype=torch.bfloat16)
        jobs = (L.WnJob * 1)()
        jobs[0] = L.WnJob(W=p(W), rows=r, cols=c, out_scale=1.0, first_block=0, w_bf16=p(wb), w_f32=p(wf), w_split3=p(w3))
        raw = torch.from_numpy(np.frombuffer(bytes(jobs), dtype=np.uint8).copy()).to(DEV)
        L.lib().weightnorm_fwd_batch(p(raw), 1, (r + 3) // 4, 1, st())
        torch.cuda.synchronize()
        # (the split image is bf16 in both builds; the plain image next to it is bf16 or fp16)
        hi = wb if not MODE["f16"] else wf.bfloat16()
        assert torch.equal(w3[:, :c], hi) and torch.equal(w3[:, 2 * c:], hi)
        assert float((wb.float() - wf).abs().max()) <= float(wf.abs().max()) * 2.0 ** (-11 if MODE["f16"] else -8)
        assert float(((w3[:, :c].float() + w3[:, c:2 * c].float()) - wf).abs().max()) <= float(wf.abs().max()) * 2.0 ** -16


def test_adam_ema(L):
    from oracle.dit_oracle import adam_step, ema_beta
    n = 4096 + 64
    g = torch.Generator().manual_seed(6)
    P, Gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.1
    m, v = torch.zeros(n), torch.zeros(n)
    e1, e2 = P.clone(), P.clone()
    d = [t.to(DEV).clone() for t in (P, Gr, m, v, e1, e2)]
    for step in (1, 2, 3):
        b1, b2 = 0.9, 0.99
        hp = torch.tensor([1e-2 / (1 - b1 ** step), 1 / math.sqrt(1 - b2 ** step), ema_beta(0.05, step), ema_beta(0.1, step), 1.0],
                          device=DEV)
        L.lib().adam_ema_step(p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(d[4]), p(d[5]), n, p(hp), b1, b2, 1e-8, st())
        adam_step(P, Gr, m, v, step, lr=1e-2)
        e1.lerp_(P, ema_beta(0.05, step))
        e2.lerp_(P, ema_beta(0.1, step))
    torch.cuda.synchronize()
    assert rel_err(d[0].cpu().numpy(), P.numpy()) < 1e-6
    assert rel_err(d[4].cpu().numpy(), e1.numpy()) < 1e-6
    assert rel_err(d[5].cpu().numpy(), e2.numpy()) < 1e-6


def test_round5_weight_pass_kernels(L):
    """The kernels behind the sharded weight passes and the side-stream Jacobian (abi 5), each against the entry point it replaces:
    * mapdit_weightnorm_bwd_slim (48 registers, no LDS): the bits of mapdit_weightnorm_bwd, for 1 / 3 / 7 / 9 split-K slabs;
    * mapdit_reduce_slabs + an in-place one-slab mapdit_weightnorm_bwd on a ROW RANGE == the whole-weight Jacobian on those rows
      (the raw sum is what a reduce-scatter moves; the Jacobian of a row needs that row only);
    * mapdit_weightnorm_bwd_batch (row ranges of several weights in one launch, in place) == per-weight launches, bit for bit;
    * mapdit_adam_ema_step_ranges / mapdit_grad_nonfinite_check_ranges == the contiguous entry points on each range, bit for bit,
      elements outside the ranges untouched, a NaN inside a range flagged and outside it ignored;
    * mapdit_sum_bf16_chunks == the fp32 sum of the bf16 values in chunk order; mapdit_scale_copy."""
    import numpy as np
    if MODE["f16"]:
        pytest.skip("fp32 kernels: one copy in the library")
    g = torch.Generator().manual_seed(77)
    rows, cols = 96, 768
    W = torch.randn(rows, cols, generator=g).to(DEV)
    for S in (1, 3, 7, 9):
        slabs = torch.randn(S, rows, cols, generator=g).to(DEV)
        a, b = slabs.clone(), slabs.clone()
        da, db = torch.zeros(rows, cols, device=DEV), torch.zeros(rows, cols, device=DEV)
        L.lib().weightnorm_bwd(p(W), p(a), cols, S, rows * cols, p(da), rows, cols, 1.0, 0, st())
        L.lib().weightnorm_bwd_slim(p(W), p(b), cols, S, rows * cols, p(db), rows, cols, 1.0, 0, st())
        torch.cuda.synchronize()
        assert torch.equal(da, db) and torch.equal(a[0], b[0]), S
        # raw sum, then the Jacobian of rows [32, 64) in place
        raw = torch.full((rows, cols), float("nan"), device=DEV)
        L.lib().reduce_slabs(p(raw), p(slabs), S, rows * cols, rows * cols, st())
        torch.cuda.synchronize()
        assert torch.equal(raw, a[0] if S > 1 else slabs[0])                     # (the Jacobian parks the same ordered sum in slab 0)
        lo, hi = 32, 64
        L.lib().weightnorm_bwd(W.data_ptr() + lo * cols * 4, raw.data_ptr() + lo * cols * 4, cols, 1, 0, raw.data_ptr() + lo * cols * 4,
                               hi - lo, cols, 1.0, 0, st())
        torch.cuda.synchronize()
        assert torch.equal(raw[lo:hi], da[lo:hi]) and torch.equal(raw[:lo], (a[0] if S > 1 else slabs[0])[:lo])
    # batch of row ranges
    shapes = [(64, 256), (24, 768), (7, 128)]
    Ws = [torch.randn(r, c, generator=g).to(DEV) for r, c in shapes]
    Gs = [torch.randn(r, c, generator=g).to(DEV) for r, c in shapes]
    single = []
    for Wt, Gt, (r, c) in zip(Ws, Gs, shapes):
        d = torch.zeros(r, c, device=DEV)
        L.lib().weightnorm_bwd(p(Wt), p(Gt.clone()), c, 1, 0, p(d), r, c, 1.0, 0, st())
        single.append(d)
    jobs = (L.WnJob * len(shapes))()
    blocks = 0
    Gb = [t.clone() for t in Gs]
    for j, (Wt, Gt, (r, c)) in enumerate(zip(Ws, Gb, shapes)):
        jobs[j] = L.WnJob(W=p(Wt), rows=r, cols=c, out_scale=1.0, first_block=blocks, w_bf16=None, w_f32=p(Gt))
        blocks += (r + 3) // 4
    raw = torch.from_numpy(np.frombuffer(bytes(jobs), dtype=np.uint8).copy()).to(DEV)
    L.lib().weightnorm_bwd_batch(p(raw), len(shapes), blocks, st())
    torch.cuda.synchronize()
    for a, b in zip(single, Gb):
        assert torch.equal(a, b)
    # multi-range optimiser step and check
    n = 40_000
    ranges = [(0, 4096), (8192, 8192 + 5000), (20_000, 20_004), (30_000, 39_996)]
    init = [torch.randn(n, generator=g) for _ in range(6)]
    init[3] = init[3].abs()                                                      # second moment
    hyper = L.AdamScalars(1e-2 / (1 - 0.9), 1 / math.sqrt(1 - 0.99), 0.3, 0.6, 0.5)
    A = [t.to(DEV).clone() for t in init]
    B = [t.to(DEV).clone() for t in init]
    for lo, hi in ranges:
        L.lib().adam_ema_step_scalars(*(t.data_ptr() + 4 * lo for t in (A[0], A[1], A[2], A[3], A[4], A[5])), hi - lo, C.byref(hyper), 0.9, 0.99,
                                      1e-8, st())
    tab, blk = [], 0
    for lo, hi in ranges:
        tab.append((lo, hi, blk))
        blk += (hi - lo + 4095) // 4096
    tabd = torch.tensor(tab, dtype=torch.int64, device=DEV)
    status = torch.zeros(2, dtype=torch.int32, device=DEV)
    L.lib().grad_nonfinite_check_ranges(p(B[1]), p(tabd), len(ranges), blk, p(status), 5, st())
    L.lib().adam_ema_step_ranges(p(B[0]), p(B[1]), p(B[2]), p(B[3]), p(B[4]), p(B[5]), p(tabd), len(ranges), blk, C.byref(hyper), 0.9, 0.99, 1e-8,
                                 p(status), 5, st())
    torch.cuda.synchronize()
    assert status.tolist() == [0, 0]
    for a, b, i0 in zip(A, B, init):
        assert torch.equal(a, b)
        assert torch.equal(b[4096:8192].cpu(), i0[4096:8192])                    # outside every range: untouched
    assert not torch.equal(B[0][:4096].cpu(), init[0][:4096])
    B[1][5000] = float("nan")                                                    # outside the ranges: ignored
    L.lib().grad_nonfinite_check_ranges(p(B[1]), p(tabd), len(ranges), blk, p(status), 6, st())
    torch.cuda.synchronize()
    assert status.tolist() == [0, 0]
    B[1][8192 + 4999] = float("inf")                                             # last element of a range: flagged, and the step is refused
    keep = B[0].clone()
    L.lib().grad_nonfinite_check_ranges(p(B[1]), p(tabd), len(ranges), blk, p(status), 7, st())
    L.lib().adam_ema_step_ranges(p(B[0]), p(B[1]), p(B[2]), p(B[3]), p(B[4]), p(B[5]), p(tabd), len(ranges), blk, C.byref(hyper), 0.9, 0.99, 1e-8,
                                 p(status), 7, st())
    torch.cuda.synchronize()
    assert status.tolist() == [7, 1] and torch.equal(B[0], keep)
    # 16-bit exchange, receiving side; scale copy
    chunks = torch.randn(4, 1024, generator=g).bfloat16().to(DEV)
    out = torch.zeros(1024, device=DEV)
    L.lib().sum_bf16_chunks(p(out), p(chunks), 4, 1024, 1024, st())
    want = chunks[0].float()
    for c_ in chunks[1:]:
        want = want + c_.float()
    sc = torch.zeros(1024, device=DEV)
    L.lib().scale_copy(p(sc), p(out), 1024, 0.25, st())
    torch.cuda.synchronize()
    assert torch.equal(out, want) and torch.equal(sc, out * 0.25)


# ---------------------------------------------------------------------------------------------------
# token-stream kernels
# ---------------------------------------------------------------------------------------------------
def test_modulate_and_fused_backward(L):
    from oracle.dit_oracle import modulate, mp_sum
    N, T, D = 3, 64, 256
    g = torch.Generator().manual_seed(7)
    x_up = torch.randn(N, T, D, generator=g)
    y_up = bf16_exact(N, T, D, seed=8)
    mod_up = torch.randn(N, 6 * D, generator=g)
    mod = torch.randn(N, 6 * D, generator=g)
    gain = torch.tensor(0.37)
    dxm = bf16_exact(N, T, D, seed=9)
    dxo = torch.randn(N, T, D, generator=g)
    # reference chain: x' = mp_sum(x_up, g_up*y_up, .3); u = modulate(x', shift, scale, gain); loss = <u, dxm> + <x', dxo'>
    ca, cb = 0.7 / math.sqrt(0.58), 0.3 / math.sqrt(0.58)
    leaves = [t.clone().requires_grad_(True) for t in (x_up, y_up, mod_up, mod, gain)]
    xu, yu, mu, mm, gg = leaves
    xp = mp_sum(xu, mu[:, 5 * D:].unsqueeze(1) * yu, 0.3)
    u = modulate(xp, mm[:, 3 * D:4 * D], mm[:, 4 * D:5 * D], gg)
    # downstream residual contributes ca*dxo to d x'
    ((u * dxm).sum() + (xp * (ca * dxo)).sum()).backward()
    # forward kernel
    xpd = xp.detach().to(DEV).contiguous()
    modd = mod.to(DEV)
    gd = gain.to(DEV)
    out = torch.zeros(N * T, D, device=DEV, dtype=MODE["dt"])
    L.lib().modulate_fwd(p(xpd), modd.data_ptr() + 4 * 3 * D, modd.data_ptr() + 4 * 4 * D, 6 * D, p(gd), p(out), N, T, D, st())
    torch.cuda.synchronize()
    assert rel_err(out.float().cpu().numpy().reshape(N, T, D), u.detach().numpy()) < 3e-3
    # fused backward kernel
    a = L.ResidModBwd()
    dxod, dxmd, yud, mud = dxo.to(DEV).contiguous(), to_bf(dxm), to_bf(y_up), mod_up.to(DEV)
    dx = torch.zeros(N, T, D, device=DEV)
    dxbf = torch.zeros(N, T, D, device=DEV, dtype=MODE["dt"])
    dmod = torch.zeros(N, 6 * D, device=DEV)
    dmod_up = torch.zeros(N, 6 * D, device=DEV)
    part = torch.zeros(N * (D // 128), device=DEV)
    dy = torch.zeros(N, T, D, device=DEV, dtype=MODE["dt"])
    a.dxo, a.dxm, a.x = p(dxod), p(dxmd), p(xpd)
    a.shift, a.scale, a.gain, a.ldmod = modd.data_ptr() + 4 * 3 * D, modd.data_ptr() + 4 * 4 * D, p(gd), 6 * D
    a.y_up, a.g_up, a.ldg_up = p(yud), mud.data_ptr() + 4 * 5 * D, 6 * D
    a.dx, a.dx_bf = p(dx), p(dxbf)
    a.dshift, a.dscale, a.ldd = dmod.data_ptr() + 4 * 3 * D, dmod.data_ptr() + 4 * 4 * D, 6 * D
    a.dgain_part, a.dy_up, a.dg_up, a.ldd_up = p(part), p(dy), dmod_up.data_ptr() + 4 * 5 * D, 6 * D
    a.n_samples, a.T, a.D, a.ca, a.cb = N, T, D, ca, cb
    L.lib().resid_mod_bwd(C.byref(a), st())
    dgain = torch.zeros((), device=DEV)
    L.lib().reduce_partials(p(part), N * (D // 128), p(dgain), 0, st())
    torch.cuda.synchronize()
    # d x' = ca*dxo + k*scale*dxm ; autograd: xu.grad = ca * d x'
    assert rel_err((dx.cpu() * ca).numpy(), xu.grad.numpy()) < 1e-5
    assert rel_err(dxbf.float().cpu().numpy(), dx.cpu().numpy()) < 3e-3
    assert rel_err(dy.float().cpu().numpy(), yu.grad.numpy()) < 3e-3
    assert rel_err(dmod.cpu().numpy(), mm.grad.numpy()) < 1e-5
    assert rel_err(dmod_up.cpu().numpy(), mu.grad.numpy()) < 1e-5
    assert abs(dgain.item() - gg.grad.item()) < 1e-4 * abs(gg.grad.item()) + 1e-5
    # the row-split form small batches take (scratch given: a sample's rows are cut into Z pieces, sums added by a second kernel)
    scratch = torch.full((8 * N * 3 * D,), float("nan"), device=DEV)
    part8 = torch.full((8 * N * (D // 128),), float("nan"), device=DEV)
    nparts = C.c_int(0)
    dmod2, dmod_up2, dx2, dy2 = torch.zeros_like(dmod), torch.zeros_like(dmod_up), torch.zeros_like(dx), torch.zeros_like(dy)
    a.part_scratch, a.part_scratch_bytes, a.gain_partials_out = p(scratch), scratch.numel() * 4, C.pointer(nparts)
    a.dgain_part, a.dx, a.dy_up = p(part8), p(dx2), p(dy2)
    a.dshift, a.dscale = dmod2.data_ptr() + 4 * 3 * D, dmod2.data_ptr() + 4 * 4 * D
    a.dg_up = dmod_up2.data_ptr() + 4 * 5 * D
    L.lib().resid_mod_bwd(C.byref(a), st())
    dgain2 = torch.zeros((), device=DEV)
    L.lib().reduce_partials(p(part8), nparts.value, p(dgain2), 0, st())
    torch.cuda.synchronize()
    assert nparts.value == 8 * N * (D // 128)                 # T = 64 rows -> 8 pieces of 8 rows
    assert torch.equal(dx2, dx) and torch.equal(dy2, dy)      # per-element work does not depend on the split
    assert rel_err(dmod2.cpu().numpy(), mm.grad.numpy()) < 1e-5 and rel_err(dmod_up2.cpu().numpy(), mu.grad.numpy()) < 1e-5
    assert abs(dgain2.item() - gg.grad.item()) < 1e-4 * abs(gg.grad.item()) + 1e-5
    # round 4: the downstream gradient as a 16-bit tensor (dxo_bf; the fp16 engine's block-to-block gradient stream).  Given the
    # 16-bit rounding of dxo it must produce the bits the fp32 form produces from those rounded values; and the two are alternatives.
    dxo16 = dxod.to(MODE["dt"])
    dxr = dxo16.float().contiguous()
    outs = []
    for form in ("fp32 of the rounded values", "16-bit"):
        dxf, dyf, dmf = torch.zeros_like(dx), torch.zeros_like(dy), torch.zeros_like(dmod)
        a.part_scratch, a.part_scratch_bytes, a.gain_partials_out = None, 0, None
        a.dxo, a.dxo_bf = (p(dxr), None) if form.startswith("fp32") else (None, p(dxo16))
        a.dgain_part, a.dx, a.dy_up = p(part), p(dxf), p(dyf)
        a.dshift, a.dscale = dmf.data_ptr() + 4 * 3 * D, dmf.data_ptr() + 4 * 4 * D
        L.lib().resid_mod_bwd(C.byref(a), st())
        torch.cuda.synchronize()
        outs.append((dxf, dyf, dmf))
    for u_, v_ in zip(*outs):
        assert torch.equal(u_, v_)
    a.dxo = p(dxr)
    with pytest.raises(L.MapditError):
        L.lib().resid_mod_bwd(C.byref(a), st())


@pytest.mark.parametrize("N,T,D,K,layout,with_up,with_dxo", [(2, 256, 256, 128, 1, True, True), (8, 64, 512, 192, 1, True, True),
                                                            (4, 128, 256, 64, 0, True, False), (3, 256, 256, 128, 1, False, True),
                                                            (5, 128, 512, 320, 1, True, True)])
def test_gemm_fused_resid_mod_backward(L, N, T, D, K, layout, with_up, with_dxo):
    """EPI_RMB: the dX GEMM whose epilogue is the backward of modulate() + the residual mp_sum above it (what
    mapdit_resid_mod_bwd does as a separate pass), against autograd of the oracle ops.  The gradient wrt the modulated input is
    the accumulator rounded to bf16 (exactly what the unfused path stores and re-reads).  Covers one / two / four samples per
    256-row tile, a ragged last tile (N*T not a multiple of 256), NT and NN layouts, with and without the upstream residual."""
    from oracle.dit_oracle import modulate, mp_sum
    M = N * T
    g = torch.Generator().manual_seed(70 + N)
    x_up = torch.randn(N, T, D, generator=g)
    y_up = bf16_exact(N, T, D, seed=8)
    mod_up = torch.randn(N, 6 * D, generator=g)
    mod = torch.randn(N, 6 * D, generator=g)
    gain = torch.tensor(0.37)
    dyo = bf16_exact(M, K, seed=9)                                    # gradient entering the dX GEMM
    w = bf16_exact(K, D, seed=10) * 0.25 if layout == 1 else bf16_exact(D, K, seed=10) * 0.25
    dxm = (dyo @ (w if layout == 1 else w.t())).to(MODE["dt"]).float()    # the GEMM result as the backward sees it: bf16
    dxo = torch.randn(N, T, D, generator=g)
    ca, cb = 0.7 / math.sqrt(0.58), 0.3 / math.sqrt(0.58)
    leaves = [t.clone().requires_grad_(True) for t in (x_up, y_up, mod_up, mod, gain)]
    xu, yu, mu, mm, gg = leaves
    xp = mp_sum(xu, mu[:, 5 * D:].unsqueeze(1) * yu, 0.3) if with_up else xu * 1.0
    u = modulate(xp, mm[:, 3 * D:4 * D], mm[:, 4 * D:5 * D], gg)
    loss = (u * dxm.view(N, T, D)).sum()
    if with_dxo:
        loss = loss + (xp * (ca * dxo)).sum()
    loss.backward()
    xpd = xp.detach().to(DEV).contiguous()
    modd, gd, mud = mod.to(DEV), gain.to(DEV), mod_up.to(DEV)
    dxod, yud = dxo.to(DEV).contiguous(), to_bf(y_up)
    dx = torch.zeros(N, T, D, device=DEV)
    dxbf = torch.zeros(N, T, D, device=DEV, dtype=MODE["dt"])
    dmod = torch.zeros(N, 6 * D, device=DEV)
    dmod_up = torch.zeros(N, 6 * D, device=DEV)
    tiles = ((M + 255) // 256) * (D // 256)
    part = torch.full((tiles + 4,), float("nan"), device=DEV)
    dy = torch.zeros(N, T, D, device=DEV, dtype=MODE["dt"])
    a = L.ResidModBwd()
    a.dxo, a.dxm, a.x = (p(dxod) if with_dxo else None), None, p(xpd)
    a.shift, a.scale, a.gain, a.ldmod = modd.data_ptr() + 4 * 3 * D, modd.data_ptr() + 4 * 4 * D, p(gd), 6 * D
    if with_up:
        a.y_up, a.g_up, a.ldg_up = p(yud), mud.data_ptr() + 4 * 5 * D, 6 * D
        a.dy_up, a.dg_up, a.ldd_up = p(dy), dmod_up.data_ptr() + 4 * 5 * D, 6 * D
    a.dx, a.dx_bf = p(dx), p(dxbf)
    a.dshift, a.dscale, a.ldd = dmod.data_ptr() + 4 * 3 * D, dmod.data_ptr() + 4 * 4 * D, 6 * D
    a.dgain_part = p(part)
    a.n_samples, a.T, a.D, a.ca, a.cb = N, T, D, ca, cb
    e = L.Epilogue()
    e.kind, e.ldo, e.rmb = L.EPI_RMB, D, C.addressof(a)
    ad, bd = to_bf(dyo), to_bf(w)
    L.lib().gemm_tuning(256, 2, 0)            # these small results would take the 128^2 kernel by themselves (too few 256^2 tiles)
    try:                                        # (the tile override is process-global state: always undone)
        L.lib().gemm_bf16(layout, M, D, K, p(ad), K, p(bd), D if layout == 1 else K, C.byref(e), st())
        dgain = torch.zeros((), device=DEV)
        L.lib().reduce_partials(p(part), tiles, p(dgain), 0, st())
        torch.cuda.synchronize()
        assert torch.isnan(part[tiles:]).all() and torch.isfinite(part[:tiles]).all()          # exactly one partial per tile
        want_dxp = xu.grad / ca if with_up else xu.grad            # d x' (autograd: xu.grad = ca * d x' through the mp_sum)
        # (the reference rounds a CPU fp32 product to bf16, the kernel its own fp32 accumulation: a few roundings flip -> ~2e-5)
        assert rel_err(dx.cpu().numpy(), want_dxp.numpy()) < 1e-4
        assert rel_err(dxbf.float().cpu().numpy(), dx.cpu().numpy()) < 3e-3
        assert rel_err(dmod[:, 3 * D:5 * D].cpu().numpy(), mm.grad[:, 3 * D:5 * D].numpy()) < 1e-4
        assert abs(dgain.item() - gg.grad.item()) < 1e-4 * (dxm.abs().sum().item() ** 0.5 + 1) + 1e-3 * abs(gg.grad.item())
        if with_up:
            assert rel_err(dy.float().cpu().numpy(), yu.grad.numpy()) < 3e-3
            assert rel_err(dmod_up[:, 5 * D:].cpu().numpy(), mu.grad[:, 5 * D:].numpy()) < 1e-4
        # bit-reproducible (no atomics): a second launch gives the same bits
        dx2, dmod2 = dx.clone(), dmod.clone()
        L.lib().gemm_bf16(layout, M, D, K, p(ad), K, p(bd), D if layout == 1 else K, C.byref(e), st())
        torch.cuda.synchronize()
        assert torch.equal(dx, dx2) and torch.equal(dmod, dmod2)
        # round 4: the downstream gradient as a 16-bit tensor (dxo_bf: what the 16-bit engines' fused backward runs on) gives the bits
        # of the fp32 form on the same rounded values, and the bits of the UNFUSED pass (mapdit_resid_mod_bwd on the stored GEMM result)
        if with_dxo:
            dxo16 = dxod.to(MODE["dt"])
            dxr = dxo16.float().contiguous()
            res = []
            for form in ("fp32 of the rounded values", "16-bit"):
                dxf, dxbf_f, dmf, dyf = torch.zeros_like(dx), torch.zeros_like(dxbf), torch.zeros_like(dmod), torch.zeros_like(dy)
                a.dxo, a.dxo_bf = (p(dxr), None) if form.startswith("fp32") else (None, p(dxo16))
                a.dx, a.dx_bf = p(dxf), p(dxbf_f)
                a.dshift, a.dscale = dmf.data_ptr() + 4 * 3 * D, dmf.data_ptr() + 4 * 4 * D
                if with_up:
                    a.dy_up = p(dyf)
                L.lib().gemm_bf16(layout, M, D, K, p(ad), K, p(bd), D if layout == 1 else K, C.byref(e), st())
                torch.cuda.synchronize()
                res.append((dxf, dxbf_f, dmf[:, 3 * D:5 * D].clone(), dyf))
            for u_, v_ in zip(*res):
                assert torch.equal(u_, v_)
            # the unfused pass on the bf16 GEMM result the fused epilogue sees
            gm = torch.zeros(M, D, device=DEV, dtype=MODE["dt"])
            run_gemm(L, layout, ad, bd, L.EPI_STORE_BF16, M, D, K, out=p(gm), ldo=D, alpha=1.0)
            dxu, dmu, dyu = torch.zeros_like(dxbf), torch.zeros_like(dmod), torch.zeros_like(dy)
            partu = torch.zeros(N * (D // 128) * 8, device=DEV)
            a.dxm, a.dx, a.dx_bf, a.dgain_part = p(gm), None, p(dxu), p(partu)
            a.dshift, a.dscale = dmu.data_ptr() + 4 * 3 * D, dmu.data_ptr() + 4 * 4 * D
            if with_up:
                a.dy_up = p(dyu)
            L.lib().resid_mod_bwd(C.byref(a), st())
            torch.cuda.synchronize()
            assert torch.equal(dxu, res[1][1]) and torch.equal(dyu, res[1][3])                   # per-element work: the same bits
            assert rel_err(dmu[:, 3 * D:5 * D].cpu().numpy(), res[1][2].cpu().numpy()) < 1e-6    # (column sums: another summation order)
            a.dxm, a.dxo, a.dxo_bf, a.dx, a.dx_bf, a.dgain_part = None, p(dxod), None, p(dx), p(dxbf), p(part)
            a.dshift, a.dscale = dmod.data_ptr() + 4 * 3 * D, dmod.data_ptr() + 4 * 4 * D
            if with_up:
                a.dy_up = p(dy)
    finally:
        L.lib().gemm_tuning(0, 2, 0)
    # shapes the 256x256 path does not take are refused, not mis-computed
    with pytest.raises(L.MapditError):
        L.lib().gemm_bf16(layout, 256, D, K, p(ad), K, p(bd), D if layout == 1 else K, C.byref(e), st())
    # a sample whose rows would cross a 256-row tile (T = 192, 512: the per-sample sums are stored once per tile) is refused too
    for bad_t in (192, 512):
        if M % bad_t == 0:
            a.T = bad_t
            L.lib().gemm_tuning(256, 2, 0)
            try:
                with pytest.raises(L.MapditError, match="RMB needs T"):
                    L.lib().gemm_bf16(layout, M, D, K, p(ad), K, p(bd), D if layout == 1 else K, C.byref(e), st())
            finally:
                L.lib().gemm_tuning(0, 2, 0)
    a.T = T


@pytest.mark.parametrize("B,T,H", [(2, 64, 2), (1, 256, 3), (3, 128, 1), (1, 512, 2), (2, 1024, 1)])     # > 256 tokens: 256-token tiles
def test_attention_fwd_bwd(L, B, T, H):
    """qkv split + cosine normalise + attention forward and the whole backward chain vs autograd of the oracle ops."""
    from oracle.dit_oracle import normalize
    D = H * 64
    qkv = bf16_exact(B * T, 3 * D, seed=10)
    dO = bf16_exact(B * T, D, seed=11)
    leaf = qkv.clone().requires_grad_(True)
    q, k, v = leaf.view(B, T, 3 * D).chunk(3, dim=-1)
    sp = lambda z: z.reshape(B, T, H, 64).transpose(1, 2)
    qn, kn = normalize(sp(q)), normalize(sp(k))
    att = torch.softmax(qn @ kn.transpose(-1, -2) / 8.0, dim=-1) @ sp(v)
    o_ref = att.transpose(1, 2).reshape(B * T, D)
    o_ref.backward(dO)

    qkvd = to_bf(qkv)
    mk = lambda *s: torch.zeros(*s, device=DEV, dtype=MODE["dt"])
    qn_d, kn_d, v_d = mk(B * H, T, 64), mk(B * H, T, 64), mk(B * H, T, 64)
    lib = L.lib()
    lib.qkv_split(p(qkvd), B, T, H, 64, p(qn_d), p(kn_d), p(v_d), st())
    o_d = mk(B * T, D)
    lse = torch.zeros(B * H, T, device=DEV)
    lib.attn_cos_fwd(p(qn_d), p(kn_d), p(v_d), p(o_d), p(lse), B, T, H, 64, st())
    torch.cuda.synchronize()
    assert rel_err(qn_d.float().cpu().numpy(), qn.detach().reshape(B * H, T, 64).numpy()) < 3e-3
    assert torch.equal(v_d.cpu().view(B, H, T, 64), sp(to_bf(qkv).cpu().view(B, T, 3 * D)[..., 2 * D:]))
    # forward: bf16 q^,k^ and bf16 P in the PV product -> ~1e-2 relative
    assert rel_err(o_d.float().cpu().numpy(), o_ref.detach().numpy()) < 1e-2
    dOd = to_bf(dO)
    delta = torch.zeros(B * H, T, device=DEV)
    dqn, dkn, dv = mk(B * H, T, 64), mk(B * H, T, 64), mk(B * H, T, 64)
    lib.attn_cos_bwd(p(qn_d), p(kn_d), p(v_d), p(dOd), p(o_d), p(lse), p(delta), p(dqn), p(dkn), p(dv), B, T, H, 64, st())
    torch.cuda.synchronize()
    d_ref = (dO * o_d.float().cpu()).view(B, T, H, 64).sum(-1).transpose(1, 2).reshape(B * H, T)
    assert rel_err(delta.cpu().numpy(), d_ref.numpy()) < 1e-5        # delta = rowsum(dO * O), written by the dQ pass
    dqkv = mk(B * T, 3 * D)
    lib.qkv_merge_bwd(p(qkvd), B, T, H, 64, p(dqn), p(dkn), p(dv), p(dqkv), st())
    torch.cuda.synchronize()
    got = dqkv.float().cpu().view(B * T, 3, D)
    ref = leaf.grad.view(B * T, 3, D)
    assert rel_err(got[:, 2].numpy(), ref[:, 2].numpy()) < 1.5e-2     # dV
    assert rel_err(got[:, 0].numpy(), ref[:, 0].numpy()) < 3e-2       # dQ (through the cosine-norm Jacobian)
    assert rel_err(got[:, 1].numpy(), ref[:, 1].numpy()) < 3e-2       # dK


@pytest.mark.parametrize("B,T,H,hd,amp", [(2, 64, 2, 64, 1.0), (1, 256, 3, 64, 5.0), (3, 128, 1, 64, 5.0), (2, 256, 2, 72, 5.0), (1, 128, 2, 72, 1.0),
                                            (2, 16, 4, 32, 3.0), (1, 200, 2, 96, 5.0)])
def test_plain_sdpa_attention_fwd_bwd(L, B, T, H, hd, amp):
    """mapdit_attn_sdpa_fwd (README.md:58 off form, parity unpinned): F.scaled_dot_product_attention on q, k that were NOT normalised - logits up to
    a hundred and more (amp = 5: q.k / sqrt(hd) has standard deviation 25), which the cosine kernels' max-free exponentials would overflow on.
    MFMA kernels for head_dim 64 / 72, the generic fp32 kernel otherwise; backward = mapdit_attn_cos_bwd (exp(s - lse)) + mapdit_heads_merge_bwd,
    against autograd."""
    D = H * hd
    g = torch.Generator().manual_seed(20 + T + hd)
    rb = lambda *s, a=1.0: (torch.randn(*s, generator=g) * a).to(MODE["dt"]).float()          # (exact in the operand format)
    q, k, v = rb(B, H, T, hd, a=amp), rb(B, H, T, hd, a=amp), rb(B, H, T, hd)
    dO = rb(B * T, D)
    leaves = [z.clone().requires_grad_(True) for z in (q, k, v)]
    logits = leaves[0] @ leaves[1].transpose(-1, -2) / math.sqrt(hd)
    o_ref = (torch.softmax(logits, dim=-1) @ leaves[2]).transpose(1, 2).reshape(B * T, D)
    o_ref.backward(dO)
    assert amp < 5 or float(logits.detach().max()) > 89.0                # exp() of these overflows fp32 without the maximum taken out
    dev = lambda z: z.to(DEV).to(MODE["dt"]).reshape(B * H, T, hd).contiguous()
    qd, kd, vd = dev(q), dev(k), dev(v)
    mk = lambda *s: torch.zeros(*s, device=DEV, dtype=MODE["dt"])
    o_d, lse = mk(B * T, D), torch.zeros(B * H, T, device=DEV)
    lib = L.lib()
    lib.attn_sdpa_fwd(p(qd), p(kd), p(vd), p(o_d), p(lse), B, T, H, hd, st())
    torch.cuda.synchronize()
    assert torch.isfinite(o_d.float()).all() and torch.isfinite(lse).all()
    assert rel_err(lse.cpu().numpy(), torch.logsumexp(logits.detach(), -1).reshape(B * H, T).numpy()) < 1e-5
    assert rel_err(o_d.float().cpu().numpy(), o_ref.detach().numpy()) < 1e-2
    dOd = dO.to(DEV).to(MODE["dt"])
    delta = torch.zeros(B * H, T, device=DEV)
    dqn, dkn, dv = mk(B * H, T, hd), mk(B * H, T, hd), mk(B * H, T, hd)
    lib.attn_cos_bwd(p(qd), p(kd), p(vd), p(dOd), p(o_d), p(lse), p(delta), p(dqn), p(dkn), p(dv), B, T, H, hd, st())
    dqkv = mk(B * T, 3 * D)
    lib.heads_merge_bwd(p(dqn), p(dkn), p(dv), B, T, H, hd, p(dqkv), st())
    torch.cuda.synchronize()
    got = dqkv.float().cpu().view(B, T, 3, H, hd)
    for i, name in enumerate("qkv"):
        ref = leaves[i].grad.transpose(1, 2)                     # [B, T, H, hd]
        assert rel_err(got[:, :, i].numpy(), ref.numpy()) < (1.5e-2 if name == "v" else 3e-2), name
        assert torch.equal(dqkv.view(B, T, 3, H, hd)[:, :, i].cpu(), (dqn, dkn, dv)[i].view(B, H, T, hd).transpose(1, 2).cpu()), "the merge moves bits"


@pytest.mark.parametrize("B,T,H,K", [(2, 64, 2, 128), (1, 256, 3, 192), (8, 128, 4, 256), (3, 64, 1, 64), (1, 512, 2, 128), (1, 1024, 1, 64)])
def test_fused_qkv_epilogue_and_fused_attention_backward(L, B, T, H, K):
    """QKV GEMM with the head split + cosine normalisation in its epilogue (MAPDIT_EPI_QKV_HEADS), attention forward, and
    the backward that writes dqkv directly (normalisation Jacobian + head merge inside the two passes) against autograd
    over the oracle ops.  The shapes cover the 256x256, the 128x128 and the generic GEMM kernel."""
    from oracle.dit_oracle import normalize
    D = H * 64
    M = B * T
    x = bf16_exact(M, K, seed=20, scale=0.25)
    w = bf16_exact(3 * D, K, seed=21, scale=0.25)
    dO = bf16_exact(M, D, seed=22)
    qkv_ref = (x @ w.t()).requires_grad_(True)                    # fp32 product of bf16-exact operands
    q, k, v = qkv_ref.view(B, T, 3 * D).chunk(3, dim=-1)
    sp = lambda z: z.reshape(B, T, H, 64).transpose(1, 2)
    qn, kn = normalize(sp(q)), normalize(sp(k))
    att = torch.softmax(qn @ kn.transpose(-1, -2) / 8.0, dim=-1) @ sp(v)
    o_ref = att.transpose(1, 2).reshape(M, D)
    o_ref.backward(dO)

    mk = lambda *s: torch.zeros(*s, device=DEV, dtype=MODE["dt"])
    xd, wd = to_bf(x), to_bf(w)
    qn_d, kn_d, v_d = mk(B * H, T, 64), mk(B * H, T, 64), mk(B * H, T, 64)
    scales = torch.zeros(2, B * H, T, device=DEV)
    run_gemm(L, 0, xd, wd, L.EPI_QKV_HEADS, M, 3 * D, K, out=p(qn_d), out2=p(kn_d), out3=p(v_d), out4=p(scales),
             rows_per_sample=T, alpha=1.0)
    qn_r, kn_r = qn.detach().reshape(B * H, T, 64), kn.detach().reshape(B * H, T, 64)
    # one bf16 rounding of the fp32-normalised rows: 2^-9 relative per element
    assert rel_err(qn_d.float().cpu().numpy(), qn_r.numpy()) < 2.5e-3
    assert rel_err(kn_d.float().cpu().numpy(), kn_r.numpy()) < 2.5e-3
    assert rel_err(v_d.float().cpu().numpy(), sp(v).detach().reshape(B * H, T, 64).numpy()) < 2.5e-3
    s_ref = torch.stack([8.0 / (torch.linalg.vector_norm(sp(z).detach(), dim=-1) + 1e-4) for z in (q, k)]).reshape(2, B * H, T)
    assert rel_err(scales.cpu().numpy(), s_ref.numpy()) < 1e-5

    lib = L.lib()
    o_d = mk(M, D)
    lse = torch.zeros(B * H, T, device=DEV)
    lib.attn_cos_fwd(p(qn_d), p(kn_d), p(v_d), p(o_d), p(lse), B, T, H, 64, st())
    dOd = to_bf(dO)
    delta = torch.zeros(B * H, T, device=DEV)
    dqkv = mk(M, 3 * D)
    lib.attn_cos_bwd_fused(p(qn_d), p(kn_d), p(v_d), p(dOd), p(o_d), p(lse), p(delta), p(scales), p(dqkv), B, T, H, 64, st())
    torch.cuda.synchronize()
    assert rel_err(o_d.float().cpu().numpy(), o_ref.detach().numpy()) < 1e-2
    got = dqkv.float().cpu().view(M, 3, D)
    ref = qkv_ref.grad.view(M, 3, D)
    assert rel_err(got[:, 2].numpy(), ref[:, 2].numpy()) < 1.5e-2     # dV
    assert rel_err(got[:, 0].numpy(), ref[:, 0].numpy()) < 3e-2       # dQ through the cosine-norm Jacobian
    assert rel_err(got[:, 1].numpy(), ref[:, 1].numpy()) < 3e-2       # dK
    # the fused backward equals the two-kernel chain (attn_cos_bwd + qkv_merge_bwd on the bf16 qkv) up to its extra roundings
    qkvd = to_bf(qkv_ref.detach())
    dqn, dkn, dv = mk(B * H, T, 64), mk(B * H, T, 64), mk(B * H, T, 64)
    lib.attn_cos_bwd(p(qn_d), p(kn_d), p(v_d), p(dOd), p(o_d), p(lse), p(delta), p(dqn), p(dkn), p(dv), B, T, H, 64, st())
    dqkv2 = mk(M, 3 * D)
    lib.qkv_merge_bwd(p(qkvd), B, T, H, 64, p(dqn), p(dkn), p(dv), p(dqkv2), st())
    torch.cuda.synchronize()
    assert rel_err(dqkv.float().cpu().numpy(), dqkv2.float().cpu().numpy()) < 1e-2
    with pytest.raises(L.MapditError):                                   # head_dim 80 has no fused path (64 and 72 do)
        lib.attn_cos_bwd_fused(p(qn_d), p(kn_d), p(v_d), p(dOd), p(o_d), p(lse), p(delta), p(scales), p(dqkv), B, T, H, 80, st())


@pytest.mark.parametrize("B,T,H,K", [(2, 256, 3, 128), (3, 64, 2, 192), (1, 128, 16, 256)])
def test_head_dim_72_inference_path_without_the_split_pass(L, B, T, H, K):
    """Inference at head_dim 72 (DiT-XL sampling): the QKV GEMM writes q, k, v head-major and unnormalised (MAPDIT_EPI_QKV_HEADS_RAW)
    and mapdit_attn_cos_fwd_rawqk normalises q, k while staging them - against the training-path chain (plain QKV store,
    mapdit_qkv_split, mapdit_attn_cos_fwd) on the same operands and against the oracle ops."""
    from oracle.dit_oracle import normalize
    hd, D, M = 72, H * 72, B * T
    x = bf16_exact(M, K, seed=40, scale=0.25)
    w = bf16_exact(3 * D, K, seed=41, scale=0.25)
    lib = L.lib()
    mk = lambda *s: torch.zeros(*s, device=DEV, dtype=MODE["dt"])
    xd, wd = to_bf(x), to_bf(w)
    # reference chain of the library
    qkv = mk(M, 3 * D)
    run_gemm(L, 0, xd, wd, L.EPI_STORE_BF16, M, 3 * D, K, out=p(qkv), ldo=3 * D, alpha=1.0)
    qn, kn, v = mk(B * H, T, hd), mk(B * H, T, hd), mk(B * H, T, hd)
    lib.qkv_split(p(qkv), B, T, H, hd, p(qn), p(kn), p(v), st())
    o1, lse1 = mk(M, D), torch.zeros(B * H, T, device=DEV)
    lib.attn_cos_fwd(p(qn), p(kn), p(v), p(o1), p(lse1), B, T, H, hd, st())
    # fused chain
    qr, kr, vr = mk(B * H, T, hd), mk(B * H, T, hd), mk(B * H, T, hd)
    run_gemm(L, 0, xd, wd, L.EPI_QKV_HEADS_RAW, M, 3 * D, K, out=p(qr), out2=p(kr), out3=p(vr), rows_per_sample=T, ld2=hd, alpha=1.0)
    o2, lse2 = mk(M, D), torch.zeros(B * H, T, device=DEV)
    lib.attn_cos_fwd_rawqk(p(qr), p(kr), p(vr), p(o2), p(lse2), B, T, H, hd, st())
    torch.cuda.synchronize()
    heads = lambda z: z.view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).reshape(3, B * H, T, hd)
    raw = heads(qkv)
    assert torch.equal(qr, raw[0]) and torch.equal(kr, raw[1]) and torch.equal(vr, raw[2]) and torch.equal(vr, v)   # the split is exact
    assert rel_err(o2.float().cpu().numpy(), o1.float().cpu().numpy()) < 4e-3      # same roundings, other summation order of the norms
    assert rel_err(lse2.cpu().numpy(), lse1.cpu().numpy()) < 1e-3
    qkv_ref = x @ w.t()
    q, k, vv = qkv_ref.view(B, T, 3 * D).chunk(3, dim=-1)
    sp = lambda z: z.reshape(B, T, H, hd).transpose(1, 2)
    att = torch.softmax(normalize(sp(q)) @ normalize(sp(k)).transpose(-1, -2) / hd ** 0.5, dim=-1) @ sp(vv)
    assert rel_err(o2.float().cpu().numpy(), att.transpose(1, 2).reshape(M, D).numpy()) < 1.2e-2
    with pytest.raises(L.MapditError):
        lib.attn_cos_fwd_rawqk(p(qr), p(kr), p(vr), p(o2), p(lse2), B, T, H, 64, st())


@pytest.mark.parametrize("B,T,H,K", [(2, 256, 3, 128), (3, 64, 2, 192), (1, 128, 16, 256)])
def test_head_dim_72_training_path_without_split_and_merge_passes(L, B, T, H, K):
    """Training at head_dim 72 (round 4): raw head-major q, k, v from the QKV GEMM, mapdit_attn_cos_fwd_rawqk_save normalises q, k IN
    PLACE and keeps their scales, mapdit_attn_cos_bwd_fused (head_dim 72) applies the normalisation Jacobian and writes dqkv [M, 3D]
    directly - against the chain it replaces (plain QKV store, mapdit_qkv_split, mapdit_attn_cos_fwd, mapdit_attn_cos_bwd,
    mapdit_qkv_merge_bwd) on the same operands, and against autograd over the oracle ops."""
    from oracle.dit_oracle import normalize
    hd, D, M = 72, H * 72, B * T
    x = bf16_exact(M, K, seed=50, scale=0.25)
    w = bf16_exact(3 * D, K, seed=51, scale=0.25)
    dO = bf16_exact(M, D, seed=52)
    lib = L.lib()
    mk = lambda *s: torch.zeros(*s, device=DEV, dtype=MODE["dt"])
    xd, wd, dOd = to_bf(x), to_bf(w), to_bf(dO)
    # the chain it replaces
    qkv = mk(M, 3 * D)
    run_gemm(L, 0, xd, wd, L.EPI_STORE_BF16, M, 3 * D, K, out=p(qkv), ldo=3 * D, alpha=1.0)
    qn, kn, v = mk(B * H, T, hd), mk(B * H, T, hd), mk(B * H, T, hd)
    lib.qkv_split(p(qkv), B, T, H, hd, p(qn), p(kn), p(v), st())
    o1, lse1 = mk(M, D), torch.zeros(B * H, T, device=DEV)
    lib.attn_cos_fwd(p(qn), p(kn), p(v), p(o1), p(lse1), B, T, H, hd, st())
    dqn, dkn, dv = mk(B * H, T, hd), mk(B * H, T, hd), mk(B * H, T, hd)
    delta1 = torch.zeros(B * H, T, device=DEV)
    lib.attn_cos_bwd(p(qn), p(kn), p(v), p(dOd), p(o1), p(lse1), p(delta1), p(dqn), p(dkn), p(dv), B, T, H, hd, st())
    dqkv1 = mk(M, 3 * D)
    lib.qkv_merge_bwd(p(qkv), B, T, H, hd, p(dqn), p(dkn), p(dv), p(dqkv1), st())
    # the fused chain
    qr, kr, vr = mk(B * H, T, hd), mk(B * H, T, hd), mk(B * H, T, hd)
    run_gemm(L, 0, xd, wd, L.EPI_QKV_HEADS_RAW, M, 3 * D, K, out=p(qr), out2=p(kr), out3=p(vr), rows_per_sample=T, ld2=hd, alpha=1.0)
    o2, lse2 = mk(M, D), torch.zeros(B * H, T, device=DEV)
    scales = torch.zeros(2, B * H, T, device=DEV)
    lib.attn_cos_fwd_rawqk_save(p(qr), p(kr), p(vr), p(o2), p(lse2), p(scales), B, T, H, hd, st())
    delta2 = torch.zeros(B * H, T, device=DEV)
    dqkv2 = torch.full((M, 3 * D), float("nan"), device=DEV).to(MODE["dt"])                     # every element must be written
    lib.attn_cos_bwd_fused(p(qr), p(kr), p(vr), p(dOd), p(o2), p(lse2), p(delta2), p(scales), p(dqkv2), B, T, H, hd, st())
    torch.cuda.synchronize()
    # forward: the rows written back in place are the split kernel's normalised rows (same roundings, other summation order of the norm)
    assert rel_err(qr.float().cpu().numpy(), qn.float().cpu().numpy()) < 2e-3
    assert rel_err(kr.float().cpu().numpy(), kn.float().cpu().numpy()) < 2e-3
    assert torch.equal(vr, v)
    assert rel_err(o2.float().cpu().numpy(), o1.float().cpu().numpy()) < 4e-3
    qkv_ref = (x @ w.t()).requires_grad_(True)
    q, k, vv = qkv_ref.view(B, T, 3 * D).chunk(3, dim=-1)
    sp = lambda z: z.reshape(B, T, H, hd).transpose(1, 2)
    s_ref = torch.stack([hd ** 0.5 / (torch.linalg.vector_norm(sp(z).detach(), dim=-1) + 1e-4) for z in (q, k)]).reshape(2, B * H, T)
    assert rel_err(scales.cpu().numpy(), s_ref.numpy()) < 3e-3                  # (of the 16-bit q, k the GEMM stored)
    att = torch.softmax(normalize(sp(q)) @ normalize(sp(k)).transpose(-1, -2) / hd ** 0.5, dim=-1) @ sp(vv)
    att.transpose(1, 2).reshape(M, D).backward(dO)
    # backward: against the chain it replaces and against autograd
    assert torch.isfinite(dqkv2.float()).all()
    assert rel_err(delta2.cpu().numpy(), delta1.cpu().numpy()) < 5e-3
    assert rel_err(dqkv2.float().cpu().numpy(), dqkv1.float().cpu().numpy()) < 1e-2
    got, ref = dqkv2.float().cpu().view(M, 3, D), qkv_ref.grad.view(M, 3, D)
    assert rel_err(got[:, 2].numpy(), ref[:, 2].numpy()) < 1.5e-2     # dV
    assert rel_err(got[:, 0].numpy(), ref[:, 0].numpy()) < 3e-2       # dQ through the cosine-norm Jacobian
    assert rel_err(got[:, 1].numpy(), ref[:, 1].numpy()) < 3e-2       # dK
    with pytest.raises(L.MapditError):
        lib.attn_cos_fwd_rawqk_save(p(qr), p(kr), p(vr), p(o2), p(lse2), p(scales), B, T, H, 64, st())


@pytest.mark.parametrize("B,H", [(30, 12), (64, 12), (23, 13)])
def test_streaming_attention_backward_many_heads(L, B, H):
    """The one-launch attention backward of 256-token heads keeps one workgroup per CU and walks several heads per workgroup (K
    images, scalars and V fragments of the next head fetched during the current one, the dS^T -> dQ hand-over running across the
    head boundary): 360 heads (one or two per workgroup on 256 CUs), 768 (three each) and 299 (uneven, 13 heads per sample) against
    the two-pass kernels (each verified against autograd above) on the same inputs, plus delta = rowsum(dO * O)."""
    T, D = 256, H * 64
    lib = L.lib()
    g = torch.Generator(device=DEV).manual_seed(100 + B)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    raw_q, raw_k = rn(B * H, T, 64), rn(B * H, T, 64)
    nq, nk = raw_q.norm(dim=-1, keepdim=True), raw_k.norm(dim=-1, keepdim=True)
    qn, kn, v = (8 * raw_q / (nq + 1e-4)).to(MODE["dt"]), (8 * raw_k / (nk + 1e-4)).to(MODE["dt"]), rn(B * H, T, 64).to(MODE["dt"])
    scales = torch.stack([8.0 / (nq.squeeze(-1) + 1e-4), 8.0 / (nk.squeeze(-1) + 1e-4)]).contiguous()
    dO = (rn(B * T, D) * 0.05).to(MODE["dt"])
    mk = lambda *s: torch.zeros(*s, device=DEV, dtype=MODE["dt"])
    o, lse = mk(B * T, D), torch.zeros(B * H, T, device=DEV)
    lib.attn_cos_fwd(p(qn), p(kn), p(v), p(o), p(lse), B, T, H, 64, st())
    delta1, delta2 = torch.zeros(B * H, T, device=DEV), torch.zeros(B * H, T, device=DEV)
    dqkv = torch.full((B * T, 3 * D), float("nan"), device=DEV).to(MODE["dt"])                  # every element must be written
    lib.attn_cos_bwd_fused(p(qn), p(kn), p(v), p(dO), p(o), p(lse), p(delta1), p(scales), p(dqkv), B, T, H, 64, st())
    dqn, dkn, dv = mk(B * H, T, 64), mk(B * H, T, 64), mk(B * H, T, 64)
    lib.attn_cos_bwd(p(qn), p(kn), p(v), p(dO), p(o), p(lse), p(delta2), p(dqn), p(dkn), p(dv), B, T, H, 64, st())
    torch.cuda.synchronize()
    assert torch.isfinite(dqkv.float()).all()
    d_ref = (dO.float() * o.float()).view(B, T, H, 64).sum(-1).transpose(1, 2).reshape(B * H, T)
    assert rel_err(delta1.cpu().numpy(), d_ref.cpu().numpy()) < 1e-5 and rel_err(delta2.cpu().numpy(), d_ref.cpu().numpy()) < 1e-5
    got = dqkv.float().view(B, T, 3, H, 64)
    heads = lambda z: z.float().view(B, H, T, 64).transpose(1, 2)                                 # -> [B, T, H, 64]
    assert rel_err(got[:, :, 2].cpu().numpy(), heads(dv).cpu().numpy()) < 6e-3                    # dV: the same products, other order
    # dQ, dK: the two-pass kernels return the gradient w.r.t. the normalised rows; apply the Jacobian of x^ = x s here
    for idx, dn, xh, sc, raw_n in ((0, dqn, qn, scales[0], nq), (1, dkn, kn, scales[1], nk)):
        gg, xx = dn.float(), xh.float()
        dx = sc.unsqueeze(-1) * gg - xx * (gg * xx).sum(-1, keepdim=True) / (8.0 * raw_n)
        assert rel_err(got[:, :, idx].cpu().numpy(), heads(dx).cpu().numpy()) < 1.2e-2, idx


@pytest.mark.parametrize("B,T,H,hd", [(2, 64, 2, 72), (1, 256, 2, 72), (3, 16, 2, 64), (2, 48, 1, 40)])
def test_generic_attention_fwd_bwd(L, B, T, H, hd):
    """The fp32 VALU path for head sizes / token counts the MFMA kernels do not take (DiT-XL head_dim 72, patch-8 T = 16)."""
    from oracle.dit_oracle import normalize
    D = H * hd
    qkv = bf16_exact(B * T, 3 * D, seed=30)
    dO = bf16_exact(B * T, D, seed=31)
    leaf = qkv.clone().requires_grad_(True)
    q, k, v = leaf.view(B, T, 3 * D).chunk(3, dim=-1)
    sp = lambda z: z.reshape(B, T, H, hd).transpose(1, 2)
    qn, kn = normalize(sp(q)), normalize(sp(k))
    att = torch.softmax(qn @ kn.transpose(-1, -2) / math.sqrt(hd), dim=-1) @ sp(v)
    o_ref = att.transpose(1, 2).reshape(B * T, D)
    o_ref.backward(dO)
    mk = lambda *s: torch.zeros(*s, device=DEV, dtype=MODE["dt"])
    qkvd, dOd = to_bf(qkv), to_bf(dO)
    qn_d, kn_d, v_d = mk(B * H, T, hd), mk(B * H, T, hd), mk(B * H, T, hd)
    lib = L.lib()
    lib.qkv_split_generic(p(qkvd), B, T, H, hd, p(qn_d), p(kn_d), p(v_d), st())
    o_d, lse, delta = mk(B * T, D), torch.zeros(B * H, T, device=DEV), torch.zeros(B * H, T, device=DEV)
    lib.attn_generic_fwd(p(qn_d), p(kn_d), p(v_d), p(o_d), p(lse), B, T, H, hd, st())
    dqn, dkn, dv, dqkv = mk(B * H, T, hd), mk(B * H, T, hd), mk(B * H, T, hd), mk(B * T, 3 * D)
    lib.attn_generic_bwd(p(qn_d), p(kn_d), p(v_d), p(dOd), p(o_d), p(lse), p(delta), p(dqn), p(dkn), p(dv), B, T, H, hd, st())
    lib.qkv_merge_bwd_generic(p(qkvd), B, T, H, hd, p(dqn), p(dkn), p(dv), p(dqkv), st())
    torch.cuda.synchronize()
    assert rel_err(qn_d.float().cpu().numpy(), qn.detach().reshape(B * H, T, hd).numpy()) < 3e-3
    assert rel_err(o_d.float().cpu().numpy(), o_ref.detach().numpy()) < 1e-2
    got, ref = dqkv.float().cpu().view(B * T, 3, D), leaf.grad.view(B * T, 3, D)
    assert rel_err(got[:, 2].numpy(), ref[:, 2].numpy()) < 1.5e-2
    assert rel_err(got[:, 0].numpy(), ref[:, 0].numpy()) < 3e-2
    assert rel_err(got[:, 1].numpy(), ref[:, 1].numpy()) < 3e-2


@pytest.mark.parametrize("B,T,H", [(2, 64, 2), (1, 128, 3), (2, 256, 2), (1, 256, 16)])
def test_attention_head_dim_72_mfma(L, B, T, H):
    """DiT-XL's head_dim 72 on the MFMA kernels (attention72.hip; zero-padded to 80 on the reduction side, 96 on the output
    side) through the public entry points: against autograd over the oracle ops AND against the generic fp32 kernels on
    the same bf16 inputs (same rounding points, so the two agree much tighter than either agrees with fp32 autograd)."""
    from oracle.dit_oracle import normalize
    hd = 72
    D = H * hd
    qkv = bf16_exact(B * T, 3 * D, seed=40)
    dO = bf16_exact(B * T, D, seed=41)
    leaf = qkv.clone().requires_grad_(True)
    q, k, v = leaf.view(B, T, 3 * D).chunk(3, dim=-1)
    sp = lambda z: z.reshape(B, T, H, hd).transpose(1, 2)
    qn, kn = normalize(sp(q)), normalize(sp(k))
    att = torch.softmax(qn @ kn.transpose(-1, -2) / math.sqrt(hd), dim=-1) @ sp(v)
    o_ref = att.transpose(1, 2).reshape(B * T, D)
    o_ref.backward(dO)
    mk = lambda *s: torch.full(s, float("nan"), device=DEV, dtype=MODE["dt"])      # NaN-filled: every element must be written
    qkvd, dOd = to_bf(qkv), to_bf(dO)
    qn_d, kn_d, v_d = mk(B * H, T, hd), mk(B * H, T, hd), mk(B * H, T, hd)
    lib = L.lib()
    lib.qkv_split(p(qkvd), B, T, H, hd, p(qn_d), p(kn_d), p(v_d), st())
    outs = {}
    for path in ("mfma", "generic"):
        o_d = mk(B * T, D)
        lse = torch.full((B * H, T), float("nan"), device=DEV)
        delta = torch.full((B * H, T), float("nan"), device=DEV)
        dqn, dkn, dv, dqkv = mk(B * H, T, hd), mk(B * H, T, hd), mk(B * H, T, hd), mk(B * T, 3 * D)
        if path == "mfma":
            lib.attn_cos_fwd(p(qn_d), p(kn_d), p(v_d), p(o_d), p(lse), B, T, H, hd, st())
            lib.attn_cos_bwd(p(qn_d), p(kn_d), p(v_d), p(dOd), p(o_d), p(lse), p(delta), p(dqn), p(dkn), p(dv), B, T, H, hd, st())
        else:
            lib.attn_generic_fwd(p(qn_d), p(kn_d), p(v_d), p(o_d), p(lse), B, T, H, hd, st())
            lib.attn_generic_bwd(p(qn_d), p(kn_d), p(v_d), p(dOd), p(o_d), p(lse), p(delta), p(dqn), p(dkn), p(dv), B, T, H, hd, st())
        lib.qkv_merge_bwd(p(qkvd), B, T, H, hd, p(dqn), p(dkn), p(dv), p(dqkv), st())
        torch.cuda.synchronize()
        outs[path] = tuple(t.float().cpu() for t in (o_d, lse, delta, dqn, dkn, dv, dqkv))
        for t in outs[path]:
            assert torch.isfinite(t).all()
    o_d, lse, delta, dqn, dkn, dv, dqkv = outs["mfma"]
    assert rel_err(o_d.numpy(), o_ref.detach().numpy()) < 1e-2
    got, ref = dqkv.view(B * T, 3, D), leaf.grad.view(B * T, 3, D)
    assert rel_err(got[:, 2].numpy(), ref[:, 2].numpy()) < 1.5e-2
    assert rel_err(got[:, 0].numpy(), ref[:, 0].numpy()) < 3e-2
    assert rel_err(got[:, 1].numpy(), ref[:, 1].numpy()) < 3e-2
    g = outs["generic"]
    assert rel_err(lse.numpy(), g[1].numpy()) < 1e-5
    assert rel_err(delta.numpy(), g[2].numpy()) < 1e-2                     # delta = rowsum(dO * O) of each path's own O
    assert rel_err(o_d.numpy(), g[0].numpy()) < 6e-3                       # MFMA rounds P to bf16, the generic path does not
    for a, b_ in zip((dqn, dkn, dv), g[3:6]):
        assert rel_err(a.numpy(), b_.numpy()) < 1.2e-2


def test_attention_exact_small_integers(L):
    """Uniform attention (all logits equal) with integer V: O must be the exact key-mean of V."""
    B, T, H = 1, 64, 1
    mk = lambda *s: torch.zeros(*s, device=DEV, dtype=MODE["dt"])
    qn = mk(1, T, 64)          # zero queries -> all logits 0 -> uniform softmax
    kn = to_bf(bf16_exact(1, T, 64, seed=12))
    V = (torch.arange(T * 64).reshape(1, T, 64) % 7).float()
    vd = to_bf(V)
    o = mk(T, 64)
    lse = torch.zeros(1, T, device=DEV)
    L.lib().attn_cos_fwd(p(qn), p(kn), p(vd), p(o), p(lse), B, T, H, 64, st())
    torch.cuda.synchronize()
    ref = V.mean(1).expand(T, 64)
    assert rel_err(o.float().cpu().numpy(), ref.numpy()) < 4e-3
    assert np.allclose(lse.cpu().numpy(), math.log(T), atol=1e-5)


# ---------------------------------------------------------------------------------------------------
# embedding / output side
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("p_,S", [(2, 16), (4, 32)])
def test_patch_embed(L, p_, S):
    from oracle.dit_oracle import patchify, mp_sum
    N, C_, D = 3, 4, 128
    g = torch.Generator().manual_seed(13)
    x = torch.randn(N, C_, S, S, generator=g)
    P = p_ * p_ * C_
    w = torch.randn(D, P + 1, generator=g)
    T = (S // p_) ** 2
    pos = torch.randn(T, D, generator=g)
    h = patchify(x, p_)
    h = torch.cat([h, torch.ones_like(h[:, :, :1])], -1)
    ref = mp_sum(h @ w.t(), pos.unsqueeze(0), 0.5)
    ldp = (P + 1 + 7) // 8 * 8
    out = torch.zeros(N * T, D, device=DEV)
    patches = torch.full((N * T, ldp), 7.0, device=DEV, dtype=MODE["dt"])
    xd, wd, pd = x.to(DEV), w.to(DEV), pos.to(DEV)
    L.lib().patch_embed_fwd(p(xd), p(wd), p(pd), p(out), p(patches), ldp, N, C_, S, p_, D, 0.0, st())
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy().reshape(N, T, D), ref.numpy()) < 1e-6
    out1 = torch.zeros_like(out)                   # out_scale 1: the plain sum (off form of --use-mp-pos-enc)
    L.lib().patch_embed_fwd(p(xd), p(wd), p(pd), p(out1), None, ldp, N, C_, S, p_, D, 1.0, st())
    torch.cuda.synchronize()
    assert rel_err(out1.cpu().numpy().reshape(N, T, D), (h @ w.t() + pos.unsqueeze(0)).numpy()) < 1e-6
    assert rel_err(patches[:, :P + 1].float().cpu().numpy().reshape(N, T, P + 1), h.numpy()) < 3e-3
    assert float(patches[:, P + 1:].float().abs().max()) == 0.0


def test_conditioning_kernels(L):
    N, D = 5, 128
    g = torch.Generator().manual_seed(14)
    t = torch.randint(0, 1000, (N,), generator=g)
    scale, shift = 2 * math.pi * torch.randn(256, generator=g), 2 * math.pi * torch.rand(256, generator=g)
    ref = math.sqrt(2) * torch.cos(torch.outer(t.float(), scale) + shift)
    four = torch.zeros(N, 256, device=DEV, dtype=torch.bfloat16)
    td, sd, hd = t.to(DEV), scale.to(DEV), shift.to(DEV)
    L.lib().fourier_fwd(p(td), p(sd), p(hd), p(four), N, 256, st())
    torch.cuda.synchronize()
    # arguments reach ~2e4 rad: the kernel must round t*scale before adding shift exactly like torch (no fused multiply-add),
    # then device cosf agrees with host cos to 1 ulp(fp32) and the bf16 results are identical up to rare 1-ulp(bf16) ties
    got, want = four.float().cpu(), ref.bfloat16().float()
    assert float((got != want).float().mean()) < 2e-3
    assert float((got - want).abs().max()) < 2e-2
    temb, table = torch.randn(N, D, generator=g), torch.randn(11, D, generator=g)
    y = torch.randint(0, 11, (N,), generator=g)
    cref = (temb + table[y]) * 0.5 / math.sqrt(0.5)
    c = torch.zeros(N, D, device=DEV)
    cs = torch.zeros(N, D, device=DEV, dtype=MODE["dt"])
    cb = torch.zeros_like(cs)
    tb, tab, yd = temb.to(DEV), table.to(DEV), y.to(DEV)
    L.lib().cond_combine_fwd(p(tb), p(tab), p(yd), p(c), p(cs), p(cb), N, D, tab.shape[0], st())
    torch.cuda.synchronize()
    assert rel_err(c.cpu().numpy(), cref.numpy()) < 1e-6
    assert rel_err(cs.float().cpu().numpy(), (torch.nn.functional.silu(cref) / 0.596).numpy()) < 3e-3
    # backward
    dcs, dcd = torch.randn(N, D, generator=g), torch.randn(N, D, generator=g)
    leaf_t, leaf_tab = temb.clone().requires_grad_(True), table.clone().requires_grad_(True)
    cc = (leaf_t + leaf_tab[y]) * 0.5 / math.sqrt(0.5)
    ((torch.nn.functional.silu(cc) / 0.596 * dcs).sum() + (cc * dcd).sum()).backward()
    dtemb = torch.zeros(N, D, device=DEV, dtype=MODE["dt"])
    dtable = torch.zeros(11, D, device=DEV)
    a, b = dcs.to(DEV), dcd.to(DEV)
    L.lib().cond_combine_bwd(p(c), p(a), p(b), p(yd), p(dtemb), p(dtable), N, D, dtable.shape[0], st())
    torch.cuda.synchronize()
    assert rel_err(dtemb.float().cpu().numpy(), leaf_t.grad.numpy()) < 3e-3
    assert rel_err(dtable.cpu().numpy(), leaf_tab.grad.numpy()) < 1e-5


@pytest.mark.parametrize("p_,S", [(2, 16), (4, 32)])
def test_final_out(L, p_, S):
    from oracle.dit_oracle import unpatchify
    N, C_ = 3, 4
    P = p_ * p_ * C_
    T = (S // p_) ** 2
    g = torch.Generator().manual_seed(15)
    lin = torch.randn(N * T, 2 * P, generator=g)
    am, asg = torch.randn(N, 8, generator=g), torch.randn(N, 8, generator=g)
    rm, rs = torch.randn(8, generator=g), torch.randn(8, generator=g)
    leaves = [z.clone().requires_grad_(True) for z in (lin, am, asg, rm, rs)]
    l_, a1, a2, r1, r2 = leaves
    mean, sigma = l_.view(N, T, 2 * P).chunk(2, -1)
    gm = torch.sigmoid(a1 @ r1 / math.sqrt(8)).view(-1, 1, 1)
    gs = torch.sigmoid(a2 @ r2 / math.sqrt(8)).view(-1, 1, 1)
    ref = torch.cat([unpatchify(mean * gm, S, p_), unpatchify(sigma * gs, S, p_)], 1)
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout)
    d = [z.to(DEV).contiguous() for z in (lin, am, asg, rm, rs)]
    out = torch.zeros(N, 2 * C_, S, S, device=DEV)
    L.lib().final_out_fwd(p(d[0]), 2 * P, p(d[1]), p(d[2]), p(d[3]), p(d[4]), p(out), N, C_, S, p_, st())
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), ref.detach().numpy()) < 1e-6
    ldd = 2 * P if 2 * P > 64 else 64
    dlin = torch.zeros(N * T, ldd, device=DEV, dtype=MODE["dt"])
    da = torch.zeros(2, N, 8, device=DEV, dtype=MODE["dt"])
    drm, drs = torch.zeros(8, device=DEV), torch.zeros(8, device=DEV)
    part = torch.zeros(N, 2, 8, device=DEV)
    dd = dout.to(DEV).contiguous()
    L.lib().final_out_bwd(p(dd), p(d[0]), 2 * P, p(d[1]), p(d[2]), p(d[3]), p(d[4]), p(dlin), ldd, p(da), p(part), p(drm),
                          p(drs), 1.0, N, C_, S, p_, st())
    torch.cuda.synchronize()
    assert rel_err(dlin[:, :2 * P].float().cpu().numpy(), l_.grad.numpy()) < 3e-3
    assert rel_err(da[0].float().cpu().numpy(), a1.grad.numpy()) < 4e-3
    assert rel_err(da[1].float().cpu().numpy(), a2.grad.numpy()) < 4e-3
    assert rel_err(drm.cpu().numpy(), r1.grad.numpy()) < 1e-5
    assert rel_err(drs.cpu().numpy(), r2.grad.numpy()) < 1e-5


# ---------------------------------------------------------------------------------------------------
# diffusion math
# ---------------------------------------------------------------------------------------------------
def _tables(d):
    import numpy as np
    rows = [d.sqrt_alphas_cumprod, d.sqrt_one_minus_alphas_cumprod, d.sqrt_recip_alphas_cumprod, d.sqrt_recipm1_alphas_cumprod,
            d.posterior_log_variance_clipped, d.log_betas, d.posterior_mean_coef1, d.posterior_mean_coef2]
    return torch.from_numpy(np.stack(rows)).float().to(DEV).contiguous()


def test_loss_and_sampling_math(L):
    from oracle.diffusion_oracle import DiffusionOracle
    d = DiffusionOracle("")
    tab = _tables(d)
    N, C_, S = 6, 4, 16
    per = C_ * S * S
    g = torch.Generator().manual_seed(16)
    x0 = torch.randn(N, C_, S, S, generator=g)
    x0[0, 0, 0, :4] = torch.tensor([-1.5, 1.5, 0.9995, -0.9995])
    noise = torch.randn(N, C_, S, S, generator=g)
    t = torch.tensor([0, 5, 999, 400, 0, 37])
    mo = torch.randn(N, 2 * C_, S, S, generator=g) * 0.7
    # q_sample
    xt_ref = d.q_sample(x0, t, noise)
    xt = torch.zeros(N, C_, S, S, device=DEV)
    d_ = [z.to(DEV).contiguous() for z in (x0, noise, t, mo)]
    L.lib().q_sample(p(d_[0]), p(d_[1]), p(d_[2]), p(tab), 1000, p(xt), N, per, st())
    torch.cuda.synchronize()
    assert rel_err(xt.cpu().numpy(), xt_ref.numpy()) < 1e-6
    # loss forward + gradient
    leaf = mo.clone().requires_grad_(True)
    ref = d.training_losses(lambda xx, tt, **kw: leaf, x0, t, noise=noise)
    wl, wm, wv = torch.randn(N, generator=g), torch.randn(N, generator=g), torch.randn(N, generator=g)
    ((ref["loss"] * wl).sum() + (ref["mse"] * wm).sum() + (ref["vb"] * wv).sum()).backward()
    mse, vb, loss = (torch.zeros(N, device=DEV) for _ in range(3))
    G = torch.zeros(N, 2 * C_, S, S, device=DEV)
    xtd = xt_ref.to(DEV).contiguous()
    L.lib().loss_fwd(p(d_[3]), p(d_[0]), p(xtd), p(d_[1]), p(d_[2]), p(tab), 1000, p(mse), p(vb), p(loss), p(G), N, per, st())
    dout = torch.zeros_like(G)
    w = [z.to(DEV) for z in (wl, wm, wv)]
    L.lib().loss_bwd(p(G), p(w[0]), p(w[1]), p(w[2]), p(dout), N, per, st())
    torch.cuda.synchronize()
    assert rel_err(mse.cpu().numpy(), ref["mse"].detach().numpy()) < 1e-5
    assert rel_err(vb.cpu().numpy(), ref["vb"].detach().numpy()) < 2e-5
    assert rel_err(loss.cpu().numpy(), ref["loss"].detach().numpy()) < 1e-5
    assert rel_err(dout.cpu().numpy(), leaf.grad.numpy()) < 1e-4
    # p_sample at t > 0 and t == 0
    d250 = DiffusionOracle("250")
    tab250 = _tables(d250)
    for tv in (137, 0):
        ts = torch.full((N,), tv)
        r = d250.p_sample(lambda xx, tt, **kw: mo, x0, ts, noise, clip_denoised=bool(tv == 0))
        smp, xs = torch.zeros(N, C_, S, S, device=DEV), torch.zeros(N, C_, S, S, device=DEV)
        tsd = ts.to(DEV)
        L.lib().psample_step(p(d_[3]), p(d_[0]), p(d_[1]), p(tsd), p(tab250), 250, int(tv == 0), p(smp), p(xs), N, per, st())
        torch.cuda.synchronize()
        assert rel_err(smp.cpu().numpy(), r["sample"].numpy()) < 1e-5
        assert rel_err(xs.cpu().numpy(), r["pred_xstart"].numpy()) < 1e-5


def test_cfg_combine(L):
    n, C_, HW = 6, 4, 64
    g = torch.Generator().manual_seed(17)
    mo = torch.randn(n, 2 * C_, 8, 8, generator=g)
    eps, rest = mo[:, :C_], mo[:, C_:]
    cond, unc = eps[:3], eps[3:]
    half = unc + 1.5 * (cond - unc)
    ref = torch.cat([torch.cat([half, half], 0), rest], 1)
    out = torch.zeros(n, 2 * C_, 8, 8, device=DEV)
    md = mo.to(DEV).contiguous()
    L.lib().cfg_combine(p(md), p(out), n, C_, HW, 1.5, st())
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-6


def test_c_abi_collectives_single_rank(L):
    """mapdit_comm_* / mapdit_*_bucket: RCCL bound at run time behind the C ABI (for a host program without torch.distributed).
    One rank here (a one-GPU box): the id / communicator life cycle and the in-place layouts of the three collectives - an
    all-reduce, reduce-scatter and all-gather over one rank must leave the buffer unchanged - through RCCL's own kernels on the
    caller's stream, followed by an engine kernel on the same stream."""
    lib = L.lib()
    uid = C.create_string_buffer(128)
    lib.comm_unique_id(uid)
    assert any(b != 0 for b in uid.raw)
    comm = C.c_void_p()
    lib.comm_create(uid, 0, 1, C.byref(comm))
    try:
        g = torch.Generator(device=DEV).manual_seed(3)
        buf = torch.randn(1 << 20, device=DEV, generator=g)
        want = buf.clone()
        lib.allreduce_bucket(comm, p(buf), buf.numel(), st())
        lib.reduce_scatter_bucket(comm, p(buf), buf.numel(), st())
        lib.allgather_bucket(comm, p(buf), buf.numel(), st())
        out = torch.empty(buf.numel(), device=DEV, dtype=MODE["dt"])
        lib.f32_to_bf16(p(buf), p(out), buf.numel(), 1.0, st())          # an engine kernel queued behind the collectives
        torch.cuda.synchronize()
        assert torch.equal(buf, want) and torch.equal(out, want.to(MODE["dt"]))
        with pytest.raises(L.MapditError):
            lib.reduce_scatter_bucket(comm, p(buf), 0, st())
    finally:
        lib.comm_destroy(comm)
