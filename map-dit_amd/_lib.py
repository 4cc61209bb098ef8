"""ctypes binding of ``libmapdit_hip.so`` (C ABI declared in ``include/mapdit.h``).

The product path has no CPU fallback: if the shared library is missing or a call returns a
non-zero status, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAPDIT_LIB") or os.path.join(_HERE, "libmapdit_hip.so")      # MAPDIT_LIB: A/B runs against another build
CSRC = os.path.join(_HERE, "csrc")

vp, ci, cf, cl = C.c_void_p, C.c_int, C.c_float, C.c_long


class MapditError(RuntimeError):
    pass


class Epilogue(C.Structure):
    _fields_ = [("kind", ci), ("out", vp), ("ldo", ci), ("out2", vp), ("aux", vp), ("gate", vp), ("ldg", ci),
                ("rows_per_sample", ci), ("alpha", cf), ("beta", cf), ("accumulate", ci), ("out3", vp), ("shift2", vp), ("scale2", vp),
                ("gain2", vp), ("ld2", ci), ("split_k", ci), ("slab_stride", cl), ("out4", vp), ("rmb", vp), ("rot2", ci)]


class ResidModBwd(C.Structure):
    _fields_ = [("dxo", vp), ("dxm", vp), ("x", vp), ("shift", vp), ("scale", vp), ("gain", vp), ("y_up", vp),
                ("g_up", vp), ("dx", vp), ("dx_bf", vp), ("dshift", vp), ("dscale", vp), ("dgain_part", vp),
                ("dy_up", vp), ("dg_up", vp), ("ldmod", ci), ("ldg_up", ci), ("ldd", ci), ("ldd_up", ci),
                ("n_samples", ci), ("T", ci), ("D", ci), ("ca", cf), ("cb", cf), ("part_scratch", vp), ("part_scratch_bytes", C.c_size_t),
                ("gain_partials_out", C.POINTER(ci)), ("dgain_out", vp), ("rot", ci), ("dgain_scale", cf), ("dxo_bf", vp), ("ldx", ci)]


class GemmGroupItem(C.Structure):  # mapdit_gemm_group_item_t
    _fields_ = [("A", vp), ("lda", ci), ("B", vp), ("ldb", ci), ("M", ci), ("N", ci), ("out", vp), ("ldo", ci), ("alpha", cf), ("slab_stride", cl)]


class WnBwdItem(C.Structure):      # mapdit_wn_bwd_item_t
    _fields_ = [("W", vp), ("G", vp), ("ldg", ci), ("nslabs", ci), ("slab_stride", cl), ("dW", vp), ("rows", ci), ("cols", ci), ("out_scale", cf),
                ("flags", ci)]


class WnJob(C.Structure):          # mapdit_wn_job_t
    _fields_ = [("W", vp), ("rows", ci), ("cols", ci), ("out_scale", cf), ("first_block", ci), ("w_bf16", vp), ("w_f32", vp),
                ("w_split3", vp), ("flags", ci)]


class AdamScalars(C.Structure):    # mapdit_adam_scalars_t
    _fields_ = [("step_size", cf), ("inv_sqrt_bc2", cf), ("ema_beta_a", cf), ("ema_beta_b", cf), ("grad_scale", cf)]


class Config(C.Structure):
    _fields_ = [("depth", ci), ("hidden", ci), ("patch", ci), ("input_size", ci), ("in_channels", ci),
                ("num_heads", ci), ("mlp_hidden", ci), ("table_rows", ci), ("max_batch", ci), ("precision", ci), ("rotation", ci),
                ("mp_off", ci), ("loss_scale", cf)]


# mapdit_config_t.mp_off bits (mapdit.h MAPDIT_OFF_*): off forms of the README's --use-* flags (parity unpinned); key = the facade's
# constructor argument, True (the default) = the snapshot's arithmetic
MP_OFF = {"mp_silu": 1, "mp_residual": 2, "mp_pos_enc": 4, "mp_embedding": 8, "weight_normalization": 16, "cosine_attention": 32, "no_layernorm": 64}
WN_PLAIN = 2          # mapdit.h MAPDIT_WN_PLAIN


# engine precisions (mapdit.h MAPDIT_PREC_*).  "f16": the bf16 engine with IEEE fp16 operands - same speed, 10 mantissa bits
# (forward logits within 1e-3 of the fp32 reference); "bf16x3": the fp32-accurate parity instrument.
PRECISIONS = {"bf16": 0, "bf16x3": 1, "f16": 2}


NT, NN, TN = 0, 1, 2
EPI_STORE_BF16, EPI_STORE_F32, EPI_SILU2, EPI_RESID, EPI_DSILU, EPI_SILU2_COND, EPI_QKV_HEADS, EPI_SILU2_GRAD, EPI_MUL_AUX, EPI_RMB, EPI_QKV_HEADS_RAW = range(11)
PROF_FC1_FWD = 0
PEEK_IDS = {name: i for i, name in enumerate(
    ["four", "temb", "c", "mod_all", "x0", "xmodf", "lin", "xm", "qkv", "qn", "kn", "v", "o", "xm2", "hact", "xmid", "xout"])}

# parameter-table indices (mapdit.h)
(P_X_EMB, P_T0, P_T2, P_Y_EMB, P_F_LIN, P_F_MOD, P_MS_LIN, P_MS_REF, P_SS_LIN, P_SS_REF, P_F_GAIN, P_FOURIER_SCALE,
 P_FOURIER_SHIFT, P_POS_EMBED, NUM_GLOBAL) = range(15)
(B_QKV, B_PROJ, B_FC1, B_FC2, B_MOD, B_GAIN_MSA, B_GAIN_MLP, NUM_BLOCK) = range(8)

# name -> argtypes for every status-returning entry point of include/mapdit.h
_SIGS = {
    "mapdit_gemm_bf16": [ci, ci, ci, ci, vp, ci, vp, ci, C.POINTER(Epilogue), vp],
    "mapdit_gemm_group_tn_bf16": [ci, vp, ci, ci, vp],
    "mapdit_weightnorm_fwd": [vp, ci, ci, ci, cf, vp, vp, vp, vp],
    "mapdit_weightnorm_bwd": [vp, vp, ci, ci, cl, vp, ci, ci, cf, ci, vp],
    "mapdit_weightnorm_bwd_slim": [vp, vp, ci, ci, cl, vp, ci, ci, cf, ci, vp],
    "mapdit_weightnorm_bwd_batch": [vp, ci, ci, vp],
    "mapdit_weightnorm_bwd_group": [ci, vp, vp],
    "mapdit_reduce_slabs_group": [ci, vp, vp, vp, vp, ci, vp],
    "mapdit_weightnorm_fwd_batch": [vp, ci, ci, ci, vp],
    "mapdit_adam_ema_step": [vp, vp, vp, vp, vp, vp, cl, vp, cf, cf, cf, vp],
    "mapdit_adam_ema_step_scalars": [vp, vp, vp, vp, vp, vp, cl, C.POINTER(AdamScalars), cf, cf, cf, vp],
    "mapdit_adam_ema_step_guarded": [vp, vp, vp, vp, vp, vp, cl, C.POINTER(AdamScalars), cf, cf, cf, vp, ci, vp],
    "mapdit_grad_nonfinite_check": [vp, cl, vp, ci, vp],
    "mapdit_adam_ema_step_ranges": [vp, vp, vp, vp, vp, vp, vp, ci, cl, C.POINTER(AdamScalars), cf, cf, cf, vp, ci, vp],
    "mapdit_grad_nonfinite_check_ranges": [vp, vp, ci, cl, vp, ci, vp],
    "mapdit_modulate_fwd": [vp, vp, vp, ci, vp, vp, ci, ci, ci, vp],
    "mapdit_resid_mod_bwd": [C.POINTER(ResidModBwd), vp],
    "mapdit_reduce_partials": [vp, ci, vp, ci, vp],
    "mapdit_rot_coef_fwd": [vp, vp, ci, vp, vp, vp, ci, ci, ci, vp],
    "mapdit_rot_coef_fwd_all": [vp, ci, vp, vp, vp, ci, vp, vp, ci, ci, ci, vp],
    "mapdit_rot_coef_bwd": [vp, vp, ci, vp, vp, ci, vp, vp, vp, ci, vp, cf, ci, ci, vp],
    "mapdit_rot_modulate_fwd": [vp, vp, vp, ci, vp, ci, ci, ci, vp],
    "mapdit_mpsilu_to_bf16": [vp, vp, cl, vp],
    "mapdit_f32_to_bf16": [vp, vp, cl, cf, vp],
    "mapdit_f32_to_bf16_2d": [vp, ci, vp, ci, ci, ci, cf, vp],
    "mapdit_sum_slabs": [vp, vp, ci, cl, cl, vp],
    "mapdit_scale_copy": [vp, vp, cl, cf, vp],
    "mapdit_reduce_slabs": [vp, vp, ci, cl, cl, vp],
    "mapdit_sum_bf16_chunks": [vp, vp, ci, cl, cl, vp],
    "mapdit_qkv_split": [vp, ci, ci, ci, ci, vp, vp, vp, vp],
    "mapdit_qkv_merge_bwd": [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp],
    "mapdit_attn_cos_fwd": [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "mapdit_attn_sdpa_fwd": [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "mapdit_ln_modulate_fwd": [vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, ci, vp],
    "mapdit_ln_bwd_merge": [vp, vp, vp, vp, vp, cf, vp, cl, ci, vp],
    "mapdit_heads_merge_bwd": [vp, vp, vp, ci, ci, ci, ci, vp, vp],
    "mapdit_attn_cos_fwd_rawqk": [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "mapdit_attn_cos_fwd_rawqk_save": [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "mapdit_attn_cos_bwd": [vp] * 10 + [ci, ci, ci, ci, vp],
    "mapdit_attn_cos_bwd_fused": [vp] * 9 + [ci, ci, ci, ci, vp],
    "mapdit_qkv_split_generic": [vp, ci, ci, ci, ci, vp, vp, vp, vp],
    "mapdit_qkv_merge_bwd_generic": [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp],
    "mapdit_attn_generic_fwd": [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "mapdit_attn_generic_bwd": [vp] * 10 + [ci, ci, ci, ci, vp],
    "mapdit_patch_embed_fwd": [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, vp],
    "mapdit_fourier_fwd": [vp, vp, vp, vp, ci, ci, vp],
    "mapdit_cond_combine_fwd": [vp, vp, vp, vp, vp, vp, ci, ci, ci, vp],
    "mapdit_cond_combine_bwd": [vp, vp, vp, vp, vp, vp, ci, ci, ci, vp],
    "mapdit_device_error_poll": [vp],
    "mapdit_comm_unique_id": [vp],
    "mapdit_comm_create": [vp, ci, ci, C.POINTER(vp)],
    "mapdit_allreduce_bucket": [vp, vp, cl, vp],
    "mapdit_reduce_scatter_bucket": [vp, vp, cl, vp],
    "mapdit_allgather_bucket": [vp, vp, cl, vp],
    "mapdit_final_out_fwd": [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "mapdit_final_out_bwd": [vp, vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, cf, ci, ci, ci, ci, vp],
    "mapdit_cfg_combine": [vp, vp, ci, ci, ci, cf, vp],
    "mapdit_q_sample": [vp, vp, vp, vp, ci, vp, ci, ci, vp],
    "mapdit_loss_fwd": [vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, vp],
    "mapdit_loss_bwd": [vp, vp, vp, vp, vp, ci, ci, vp],
    "mapdit_psample_step": [vp, vp, vp, vp, vp, ci, ci, vp, vp, ci, ci, vp],
    "mapdit_ddim_step": [vp, vp, vp, vp, vp, vp, ci, ci, cf, ci, vp, vp, ci, ci, vp],
    "mapdit_engine_create": [C.POINTER(Config), ci, vp, C.c_size_t, vp, C.POINTER(vp)],
    "mapdit_engine_bind": [vp, C.POINTER(vp), C.POINTER(vp)],
    "mapdit_engine_prepare_weights": [vp, ci, vp],
    "mapdit_engine_forward": [vp, vp, vp, vp, ci, ci, vp, vp],
    "mapdit_engine_backward": [vp, vp, vp],
    "mapdit_engine_backward_stages": [vp, vp, ci, ci, vp],
    "mapdit_engine_set_shard": [vp, ci, ci],
    "mapdit_engine_set_block_fences": [vp, C.POINTER(vp), ci],
    "mapdit_engine_weight_image": [vp, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)],
    "mapdit_engine_jacobian_shard": [vp, vp],
    "mapdit_engine_set_loss_scale": [vp, cf],
    "mapdit_engine_loss_scale": [vp, C.POINTER(cf)],
    "mapdit_engine_profile_begin": [vp, ci, ci],
    "mapdit_engine_profile_begin_strided": [vp, ci, ci, ci],
    "mapdit_engine_profile_end": [vp, C.POINTER(ci), C.POINTER(C.c_double)],
    "mapdit_engine_peek": [vp, ci, ci, C.POINTER(vp), C.POINTER(C.c_long), C.POINTER(ci), C.POINTER(ci)],
}
# IEEE fp16 operand forms: same signatures (mapdit.h, "16-bit operand format")
for _n in ("weightnorm_fwd", "weightnorm_fwd_batch", "modulate_fwd", "resid_mod_bwd", "rot_modulate_fwd", "qkv_split",
           "qkv_merge_bwd", "attn_cos_fwd", "attn_cos_fwd_rawqk", "attn_cos_fwd_rawqk_save", "attn_cos_bwd", "attn_cos_bwd_fused", "attn_sdpa_fwd", "heads_merge_bwd", "ln_modulate_fwd", "ln_bwd_merge", "qkv_split_generic", "qkv_merge_bwd_generic",
           "attn_generic_fwd", "attn_generic_bwd", "patch_embed_fwd", "cond_combine_fwd", "cond_combine_bwd", "final_out_bwd"):
    _SIGS[f"mapdit_{_n}_f16"] = _SIGS[f"mapdit_{_n}"]
for _b, _h in (("gemm_bf16", "gemm_f16"), ("gemm_group_tn_bf16", "gemm_group_tn_f16"), ("f32_to_bf16", "f32_to_f16"), ("f32_to_bf16_2d", "f32_to_f16_2d"),
               ("mpsilu_to_bf16", "mpsilu_to_f16")):
    _SIGS[f"mapdit_{_h}"] = _SIGS[f"mapdit_{_b}"]
# entry points that do not return a status
_OTHER = {
    "mapdit_last_error": (C.c_char_p, []),
    "mapdit_abi_version": (ci, []),
    "mapdit_gemm_tile_size": (ci, [ci, ci]),
    "mapdit_gemm_tile_size_ex": (ci, [ci, ci, ci]),
    "mapdit_gemm_tile_size_k": (ci, [ci, ci, ci, ci]),
    "mapdit_gemm_tuning": (None, [ci, ci, cl]),
    "mapdit_engine_workspace_bytes": (C.c_size_t, [C.POINTER(Config), ci]),
    "mapdit_engine_destroy": (None, [vp]),
    "mapdit_comm_destroy": (None, [vp]),
}
EXPORTS = sorted(list(_SIGS) + list(_OTHER))

_lock = threading.Lock()
_handle = None


def _source_digest() -> str:
    """SHA-256 over everything the library is built from (sources, headers, Makefile)."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(os.path.dirname(_HERE), "include", "mapdit.h")
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) or f == "Makefile") + [inc]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU).  Staleness is decided by a digest of the sources
    kept next to the library, not by file times: a repo snapshot copied to another machine does not keep them."""
    stamp = LIB_PATH + ".src-sha256"
    digest = _source_digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return LIB_PATH
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-j8"])
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    return LIB_PATH


class _Lib:
    def __init__(self, cdll):
        self._cdll = cdll
        for name, argtypes in _SIGS.items():
            fn = getattr(cdll, name)
            fn.argtypes = argtypes
            fn.restype = ci
            setattr(self, name[len("mapdit_"):], self._wrap(name, fn))
        for name, (res, argtypes) in _OTHER.items():
            fn = getattr(cdll, name)
            fn.argtypes = argtypes
            fn.restype = res
            setattr(self, name[len("mapdit_"):], fn)

    def _wrap(self, name, fn):
        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise MapditError(f"{name} failed ({rc}): {self._cdll.mapdit_last_error().decode()}")
        call.__name__ = name
        return call


def lib() -> _Lib:
    """The loaded library; raises if it has not been built (no fallback path exists)."""
    global _handle
    with _lock:
        if _handle is None:
            if not os.path.exists(LIB_PATH):
                raise MapditError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                  f"or `make -C {CSRC}`; there is no CPU fallback")
            # torch first: its wheel carries its own HIP runtime (torch/lib/libamdhip64.so, loaded into the global scope), and
            # the library's HIP calls must bind to THAT one.  Loaded before torch, they bind to the ROCm installation's
            # libamdhip64.so.7 instead - a second runtime in the process, which does not see the device torch holds (every
            # launch then fails with hipErrorNoDevice).
            import torch  # noqa: F401
            _handle = _Lib(C.CDLL(LIB_PATH))
        return _handle


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def cur_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
