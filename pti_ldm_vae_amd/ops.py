"""Host-side launchers: validate torch tensors, hand raw pointers to the C-ABI.

Every function launches on ``torch.cuda.current_stream()`` of the calling thread (so the
autograd engine's backward thread and side streams are respected) and never synchronises.
Shapes are checked HERE, before any pointer reaches a kernel.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from ._lib import (PTI_CONV_S1, PTI_CONV_S2PAD, PTI_CONV_UP2, PTI_CONV_ZINS, PTI_PRO_GN, PTI_PRO_GN_SILU,
                   PTI_PRO_NONE, ConvDesc)

BF16 = torch.bfloat16
F32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype, name, dims=None):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA(HIP) tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if dims is not None and t.dim() != dims:
        raise ValueError(f"{name}: expected {dims} dims, got {tuple(t.shape)}")


def conv_out_hw(h, w, mode):
    if mode == PTI_CONV_S1:
        return h, w
    if mode == PTI_CONV_S2PAD:
        return (h + 1 - 3) // 2 + 1, (w + 1 - 3) // 2 + 1
    return 2 * h, 2 * w


def pack_conv_weight(ws, ksize, mode=PTI_CONV_S1, flip=False, out=None):
    """fp32 [cout,cin,k,k] (or nn.Linear [cout,cin]) master weight(s) -> MFMA-packed bf16."""
    ws = list(ws) if isinstance(ws, (list, tuple)) else [ws]
    w0 = ws[0]
    cout, cin = w0.shape[0], w0.shape[1]
    for w in ws:
        _chk(w, F32, "weight")
        if w.shape[0] != cout or w.shape[1] != cin or w.numel() != cout * cin * ksize * ksize:
            raise ValueError("pack_conv_weight: inconsistent weight shapes")
    nbytes = L.lib().pti_conv_packed_bytes(cout * len(ws), cin, ksize, mode)
    if nbytes == 0:
        raise ValueError(f"pack_conv_weight: unsupported cout={cout} cin={cin} k={ksize}")
    if out is None:
        out = torch.empty(nbytes // 2, dtype=BF16, device=w0.device)
    elif out.numel() * 2 != nbytes:
        raise ValueError("pack_conv_weight: bad out size")
    arr = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
    L.check(L.lib().pti_conv_pack_weights(arr, len(ws), _ptr(out), cout, cin, ksize, mode, int(flip), _stream()),
            "pti_conv_pack_weights")
    return out


def gn_stats(x, groups, stats=None):
    """x: [N,H,W,C] bf16 -> stats [N,G,2] fp32 {sum, sumsq} (accumulated into ``stats`` if given)."""
    _chk(x, BF16, "x", 4)
    n, h, w, c = x.shape
    if stats is None:
        stats = torch.zeros(n, groups, 2, dtype=F32, device=x.device)
    else:
        _chk(stats, F32, "stats")
        if stats.numel() != n * groups * 2:
            raise ValueError("gn_stats: stats size")
    L.check(L.lib().pti_gn_stats(_ptr(x), _ptr(stats), n, h * w, c, groups, _stream()), "pti_gn_stats")
    return stats


def conv_mfma(x, w_packed, bias, y, *, cout, ksize=3, mode=PTI_CONV_S1, prologue=PTI_PRO_NONE, in_stats=None,
              gamma=None, beta=None, groups=0, eps=1e-6, residual=None, out_stats=None, out_groups=0):
    _chk(x, BF16, "x", 4)
    _chk(y, BF16, "y", 4)
    n, h, w, cin = x.shape
    ho, wo = conv_out_hw(h, w, mode)
    if tuple(y.shape) != (n, ho, wo, cout):
        raise ValueError(f"conv_mfma: y shape {tuple(y.shape)} != {(n, ho, wo, cout)}")
    if w_packed.numel() != cout * cin * ksize * ksize or w_packed.dtype != BF16:
        raise ValueError("conv_mfma: packed weight size/dtype mismatch")
    if bias is not None:
        _chk(bias, F32, "bias")
        if bias.numel() != cout:
            raise ValueError("conv_mfma: bias size")
    if prologue != PTI_PRO_NONE:
        for t, nm, cnt in ((in_stats, "in_stats", n * groups * 2), (gamma, "gamma", cin), (beta, "beta", cin)):
            _chk(t, F32, nm)
            if t.numel() != cnt:
                raise ValueError(f"conv_mfma: {nm} size")
    if residual is not None:
        _chk(residual, BF16, "residual", 4)
        if residual.shape != y.shape:
            raise ValueError("conv_mfma: residual shape")
    if out_stats is not None:
        _chk(out_stats, F32, "out_stats")
        if out_stats.numel() != n * out_groups * 2:
            raise ValueError("conv_mfma: out_stats size")
    d = ConvDesc(n=n, h=h, w=w, cin=cin, ho=ho, wo=wo, cout=cout, ksize=ksize, mode=mode, prologue=prologue,
                 groups=groups, add_residual=int(residual is not None), accum_stats=int(out_stats is not None),
                 out_groups=out_groups, eps=eps)
    L.check(L.lib().pti_conv2d_mfma(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(in_stats), _ptr(gamma), _ptr(beta),
                                    _ptr(residual), _ptr(y), _ptr(out_stats), C.byref(d), _stream()),
            "pti_conv2d_mfma")
    return y


def _strides4(t, layout):
    """element strides (n,h,w,c) of a 4-D tensor given as 'nchw' or 'nhwc'."""
    s = t.stride()
    return (s[0], s[2], s[3], s[1]) if layout == "nchw" else (s[0], s[1], s[2], s[3])


def conv_direct(x, w_tck, bias, y, *, n, h, w, cin, cout, ksize=3, x_layout="nhwc", y_layout="nhwc",
                prologue=PTI_PRO_NONE, in_stats=None, gamma=None, beta=None, groups=0, eps=1e-6):
    """Degenerate-channel stride-1 conv.  ``w_tck`` fp32 [k*k, cin, cout].  The narrow side may be
    fp32 (any strides, e.g. the user's NCHW tensor); the wide side is dense NHWC bf16."""
    _chk(w_tck, F32, "w_tck")
    if w_tck.numel() != ksize * ksize * cin * cout:
        raise ValueError("conv_direct: weight size")
    if x.numel() != n * h * w * cin or y.numel() != n * h * w * cout:
        raise ValueError("conv_direct: tensor sizes do not match n,h,w,cin,cout")
    d = ConvDesc(n=n, h=h, w=w, cin=cin, ho=h, wo=w, cout=cout, ksize=ksize, mode=PTI_CONV_S1, prologue=prologue,
                 groups=groups, eps=eps, in_f32=int(x.dtype == F32), out_f32=int(y.dtype == F32))
    d.in_stride = (C.c_int64 * 4)(*_strides4(x, x_layout))
    d.out_stride = (C.c_int64 * 4)(*_strides4(y, y_layout))
    if cout % 32 == 0 and cin <= 16:
        if y.dtype != BF16 or not y.is_contiguous() or y_layout != "nhwc":
            raise ValueError("conv_direct: wide output must be dense NHWC bf16")
    else:
        if x.dtype != BF16 or not x.is_contiguous() or x_layout != "nhwc":
            raise ValueError("conv_direct: wide input must be dense NHWC bf16")
    L.check(L.lib().pti_conv2d_direct(_ptr(x), _ptr(w_tck), _ptr(bias), _ptr(in_stats), _ptr(gamma), _ptr(beta),
                                      _ptr(y), C.byref(d), _stream()), "pti_conv2d_direct")
    return y


def wgrad_direct(wide, narrow, dw, *, n, h, w, cw, cn, ksize, sgn, narrow_layout, dw_strides, dbias_wide=None,
                 dbias_narrow=None, prologue=PTI_PRO_NONE, in_stats=None, gamma=None, beta=None, groups=0, eps=1e-6):
    """dw[tap,cw,k] += sum_p narrow[p,k] * T(wide)[p + sgn*tap, cw]; dw_strides = (tap, cw, k) element
    strides into the fp32 OIHW gradient ``dw`` (must be zero-initialised or hold a running sum)."""
    _chk(wide, BF16, "wide", 4)
    _chk(dw, F32, "dw")
    ns = (C.c_int64 * 4)(*_strides4(narrow, narrow_layout))
    L.check(L.lib().pti_wgrad_direct(_ptr(wide), _ptr(narrow), _ptr(dw), _ptr(dbias_wide), _ptr(dbias_narrow),
                                     _ptr(in_stats), _ptr(gamma), _ptr(beta), n, h, w, cw, cn, ksize, sgn, prologue,
                                     groups, eps, int(narrow.dtype == F32), ns, dw_strides[0], dw_strides[1],
                                     dw_strides[2], _stream()), "pti_wgrad_direct")
    return dw
