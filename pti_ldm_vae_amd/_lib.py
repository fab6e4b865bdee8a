"""ctypes binding of ``libpti_vae_hip.so`` (C-ABI declared in ``include/pti_vae.h``).

The product path has no CPU or PyTorch fallback: if the shared object is missing or a
symbol cannot be resolved this module raises, loudly, at first use.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTI_VAE_LIB") or os.path.join(_HERE, "libpti_vae_hip.so")   # env: kernel-variant A/B runs

def env_overrides() -> dict:
    """Every ``PTI_*`` environment variable that is set (tuning / A-B knobs): recorded by bench.py in its JSON line."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("PTI_")}


def refuse_wrong_result_env(who: str) -> None:
    """Measurement and training entry points call this first: a variable that makes kernels skip work or produce
    garbage (``PTI_*DIAG*``, ``PTI_ALLOW_WRONG_RESULTS``) must never reach a timed region or a training run."""
    bad = [k for k in os.environ if k.startswith("PTI_") and ("DIAG" in k or k == "PTI_ALLOW_WRONG_RESULTS")]
    if bad:
        raise SystemExit(f"{who}: refusing to run with wrong-result diagnostic variable(s) set: {sorted(bad)} "
                         "(timing diagnostics live in tools/diag_skip.py)")


ABI_VERSION = 5   # PTI_ABI_VERSION of include/pti_vae.h
PTI_CONV_S1, PTI_CONV_S2PAD, PTI_CONV_UP2, PTI_CONV_ZINS = 0, 1, 2, 3
PTI_PRO_NONE, PTI_PRO_GN, PTI_PRO_GN_SILU = 0, 1, 2


class ConvDesc(C.Structure):
    """``pti_conv_desc`` of include/pti_vae.h (field order must match)."""

    _fields_ = [
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32),
        ("ho", C.c_int32), ("wo", C.c_int32), ("cout", C.c_int32),
        ("ksize", C.c_int32), ("mode", C.c_int32), ("prologue", C.c_int32), ("groups", C.c_int32),
        ("add_residual", C.c_int32), ("accum_stats", C.c_int32), ("out_groups", C.c_int32),
        ("eps", C.c_float), ("in_f32", C.c_int32), ("out_f32", C.c_int32),
        ("in_stride", C.c_int64 * 4), ("out_stride", C.c_int64 * 4),
        ("in_f16", C.c_int32), ("res_f16", C.c_int32), ("out_f16", C.c_int32), ("pool2x2_out", C.c_int32),
        ("w_f16", C.c_int32), ("relu_out", C.c_int32),
    ]


class WgradJob(C.Structure):
    """``pti_wgrad_job`` of include/pti_vae.h."""

    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p),
                ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("accumulate", C.c_int32)]


WGRAD_BATCH_MAX = 16   # PTI_WGRAD_BATCH_MAX
DIRECT_REPACK_MAX = 8  # PTI_DIRECT_REPACK_MAX


class DirectRepackEntry(C.Structure):
    """``pti_direct_repack_entry`` of include/pti_vae.h."""

    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("w_tck", C.c_void_p), ("w_tck_t", C.c_void_p), ("wpad", C.c_void_p),
                ("bpad", C.c_void_p), ("cout", C.c_int32), ("cin", C.c_int32), ("pad_cin", C.c_int32), ("reserved", C.c_int32)]


class DirectRepackTable(C.Structure):
    """``pti_direct_repack_table`` of include/pti_vae.h."""

    _fields_ = [("e", DirectRepackEntry * DIRECT_REPACK_MAX), ("n", C.c_int32)]


_P = C.c_void_p
_I = C.c_int
_I64 = C.c_int64
_F = C.c_float

# symbol -> (restype, argtypes); every symbol include/pti_vae.h declares is listed here and
# tests/test_abi.py checks the two lists against each other.
SIGNATURES = {
    "pti_abi_version": (_I, []),
    "pti_last_error_string": (C.c_char_p, []),
    "pti_last_kernel_name": (C.c_char_p, []),
    "pti_conv_packed_bytes": (_I64, [_I, _I, _I, _I]),
    "pti_conv_pack_weights": (_I, [C.POINTER(_P), _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "pti_conv_pack_entry_bytes": (_I, []),
    "pti_conv_pack_table_fill": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(_I64)]),
    "pti_conv_pack_weights_batched": (_I, [_P, _P, _I, _I, _P]),
    "pti_gn_stats": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "pti_conv2d_mfma": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(ConvDesc), _P]),
    "pti_conv2d_mfma_saveact": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(ConvDesc), _P]),
    "pti_conv2d_direct": (_I, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(ConvDesc), _P]),
    "pti_wgrad_direct": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I,
                              C.POINTER(_I64), _I64, _I64, _I64, _P, _I64, _P]),
    "pti_conv_wgrad_workspace_bytes": (_I64, [_I, _I, _I, _I]),
    "pti_conv_wgrad_mfma": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I, C.POINTER(ConvDesc), _P]),
    "pti_conv_wgrad_mfma_partials": (_I, [_P, _P, _P, _P, _P, _P, _I64, C.POINTER(ConvDesc), C.POINTER(_I), _P]),
    "pti_conv_wgrad_mfma_batched": (_I, [C.POINTER(WgradJob), _I, _P, _I64, _P]),
    "pti_direct_repack": (_I, [C.POINTER(DirectRepackTable), _P]),
    "pti_conv_wgrad_reduce": (_I, [_P, _I, _P, _P, _I, C.POINTER(ConvDesc), _P]),
    "pti_gn_bwd_blocks": (_I, [_I, _I, _I]),
    "pti_gn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P]),
    "pti_conv_gnbwd_tiles": (_I, [C.POINTER(ConvDesc)]),
    "pti_gn_sums_finalize": (_I, [_P, _P, _I, _I, _I, _P]),
    "pti_conv2d_mfma_gnbwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(ConvDesc), _I, _P]),
    "pti_conv_gnbwd_chain_supported": (_I, [_I, _I, _I]),
    "pti_conv2d_mfma_gnbwd_chain": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(ConvDesc), _I, _P]),
    "pti_gn_affine_grads": (_I, [_P, _P, _P, _I, _I, _P]),
    "pti_gn_sums_finalize_affine": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _I, _P]),
    "pti_gn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "pti_pool2x2_sum": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "pti_attention_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "pti_attention_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "pti_latent_head_fwd": (_I, [_P] * 12 + [_I, _I, _I, _P]),
    "pti_post_quant": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "pti_post_quant_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "pti_latent_head_bwd": (_I, [_P] * 19 + [_I, _I, _I, _P]),
    "pti_vae_loss": (_I, [_P, _P, _I64, _P, _P, _I64, _I, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "pti_ar_vae_loss": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _P, _F, _P, _P, _P, _P]),
    "pti_pd_im2col_image": (_I, [_P, _P, _I, _I, _I, _P]),
    "pti_pd_im2col": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "pti_pd_in_stats": (_I, [_P, _P, _I, _I, _I, _F, _P]),
    "pti_pd_col2im_blocks": (_I, [_I, _I, _I]),
    "pti_pd_col2im": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "pti_pd_col2im_image": (_I, [_P, _P, _I, _I, _I, _F, _I, _P]),
    "pti_pd_in_bwd_apply": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "pti_pd_lsgan_blocks": (_I, [_I]),
    "pti_pd_lsgan": (_I, [_P, _I, _I, _I, _F, _F, _F, _P, _P, _P]),
    "pti_pd_final_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "pti_pd_final_dgrad": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "pti_pd_final_wgrad_blocks": (_I, [_I, _I, _I]),
    "pti_pd_final_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "pti_lpips_tap_blocks": (_I, [_I, _I]),
    "pti_lpips_tap_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "pti_lpips_tap_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "pti_relu_f16": (_I, [_P, _I64, _P]),
    "pti_relu_bwd": (_I, [_P, _P, _I64, _P]),
    "pti_relu_bwd_add": (_I, [_P, _P, _P, _I64, _P]),
    "pti_maxpool3s2_out": (_I, [_I]),
    "pti_maxpool3s2_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "pti_maxpool3s2_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "pti_squeeze_conv1_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "pti_squeeze_conv1_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "pti_nchw_f32_to_nhwc_f16": (_I, [_P, _P, _I, _I, _I, _P]),
    "pti_pad_nchw_to_nhwc32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "pti_slice_nhwc32_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "pti_nhwc_bf16_add_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "pti_lpips_tap_nhwc_blocks": (_I, [_I, _I]),
    "pti_lpips_tap_nhwc_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "pti_lpips_tap_nhwc_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "pti_adam_step": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _I, _F, _P]),
    "pti_preprocess_batch": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "pti_cast_nchw_f32_to_nhwc_bf16": (_I, [_P, _P, _I, _I, _I, _P]),
    "pti_cast_nhwc_bf16_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _P]),
}

_lib = None


class PtiError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtiError(
                f"{LIB_PATH} is missing: the HIP extension was not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or pti_ldm_vae_amd/csrc/build.sh). "
                "There is no CPU/PyTorch fallback for the VAE hot path.")
        # torch must load ITS HIP runtime first: the extension then binds to the same libamdhip64 instead of
        # pulling a second copy from /opt/rocm (two runtimes in one process => "no ROCm-capable device").
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        ver = handle.pti_abi_version()
        if ver != ABI_VERSION:
            raise PtiError(f"libpti_vae_hip.so ABI version {ver}, expected {ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().pti_last_error_string()
        raise PtiError(f"{what or 'pti call'} failed (code {rc}): {msg.decode() if msg else ''}")
