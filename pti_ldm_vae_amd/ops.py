"""Host-side launchers: validate torch tensors, hand raw pointers to the C-ABI.

Every function launches on ``torch.cuda.current_stream()`` of the calling thread (so the
autograd engine's backward thread and side streams are respected) and never synchronises.
Shapes are checked HERE, before any pointer reaches a kernel.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import (PTI_CONV_S1, PTI_CONV_S2PAD, PTI_CONV_UP2, PTI_CONV_ZINS, PTI_PRO_GN, PTI_PRO_GN_SILU,
                   PTI_PRO_NONE, ConvDesc)

BF16 = torch.bfloat16
F16 = torch.float16
F32 = torch.float32
I64 = torch.int64
STAT_SCALE = 65536.0   # GroupNorm statistics are Q47.16 fixed-point int64 {sum, sum of squares} (pti_common.h)
ACT16 = (BF16, F16)   # storage formats of a forward activation (flag derived from the tensor's dtype)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """hipStream_t of torch's current stream on the current device.  The public route (torch.cuda.current_stream()
    builds a Stream object, resolves the device index through several Python layers) cost ~8 us per call x ~280 calls
    per training step = a quarter of the step's host time; the two C entry points behind it cost ~0.3 us."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype, name, dims=None):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA(HIP) tensor")
    if (t.dtype not in dtype) if isinstance(dtype, tuple) else (t.dtype != dtype):
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


def pack_conv_weight(ws, ksize, mode=PTI_CONV_S1, flip=False, out=None, f16=False):
    """fp32 [cout,cin,k,k] (or nn.Linear [cout,cin]) master weight(s) -> MFMA-packed bf16 (``f16``: IEEE fp16, the
    operand of forward launches with ``w_f16``)."""
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
        out = torch.empty(nbytes // 2, dtype=F16 if f16 else BF16, device=w0.device)
    elif out.numel() * 2 != nbytes or out.dtype != (F16 if f16 else BF16):
        raise ValueError("pack_conv_weight: bad out size")
    arr = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
    L.check(L.lib().pti_conv_pack_weights(arr, len(ws), _ptr(out), cout, cin, ksize, mode, int(flip), int(f16), _stream()),
            "pti_conv_pack_weights")
    return out


class DirectRepack:
    """The derived operands of up to DIRECT_REPACK_MAX degenerate-channel convs as ONE launch (csrc/conv_direct.hip,
    ``pti_direct_repack``).  ``entries``: dicts with w (fp32 [cout,cin,3,3]) and optionally b, w_tck, w_tck_t, wpad (zeroed
    fp32 [cout', pad_cin, 3, 3], cout' >= cout), bpad.  Every tensor lives at a fixed address (parameter arena views /
    buffers allocated once), so the table is built once."""

    def __init__(self, entries):
        if not 1 <= len(entries) <= L.DIRECT_REPACK_MAX:
            raise ValueError(f"DirectRepack: 1..{L.DIRECT_REPACK_MAX} entries")
        self.table = L.DirectRepackTable()
        self.table.n = len(entries)
        self._keep = entries
        for i, e in enumerate(entries):
            w = e["w"]
            _chk(w, F32, "w", 4)
            cout, cin = w.shape[0], w.shape[1]
            if w.shape[2] != 3 or w.shape[3] != 3:
                raise ValueError("DirectRepack: 3x3 weights")
            for k in ("b", "w_tck", "w_tck_t", "wpad", "bpad"):
                if e.get(k) is not None:
                    _chk(e[k], F32, k)
            wpad = e.get("wpad")
            pad_cin = 0
            if wpad is not None:
                pad_cin = wpad.shape[1]
                if wpad.dim() != 4 or wpad.shape[0] < cout or pad_cin < cin or wpad.shape[2:] != w.shape[2:]:
                    raise ValueError("DirectRepack: wpad shape")
            for k, size in (("w_tck", cout * cin * 9), ("w_tck_t", cout * cin * 9), ("bpad", cout)):
                if e.get(k) is not None and e[k].numel() < size:
                    raise ValueError(f"DirectRepack: {k} too small")
            if e.get("bpad") is not None and (e.get("b") is None or e["b"].numel() != cout):
                raise ValueError("DirectRepack: bpad needs b")
            self.table.e[i] = L.DirectRepackEntry(_ptr(w), _ptr(e.get("b")), _ptr(e.get("w_tck")), _ptr(e.get("w_tck_t")),
                                                  _ptr(wpad), _ptr(e.get("bpad")), cout, cin, pad_cin, 0)

    def run(self):
        L.check(L.lib().pti_direct_repack(C.byref(self.table), _stream()), "pti_direct_repack")


def _chk_stats(t, count, name):
    _chk(t, I64, name)
    if t.numel() != count:
        raise ValueError(f"{name}: expected {count} fixed-point sums, got {t.numel()}")


def stats_to_float(stats):
    """Fixed-point {sum, sumsq} -> float64 tensor of the same shape (tests, diagnostics)."""
    return stats.double() / STAT_SCALE


def gn_stats(x, groups, stats=None):
    """x: [N,H,W,C] bf16|fp16 -> stats [N,G,2] int64 Q47.16 {sum, sumsq} (accumulated into ``stats`` if given)."""
    _chk(x, ACT16, "x", 4)
    n, h, w, c = x.shape
    if stats is None:
        stats = torch.zeros(n, groups, 2, dtype=I64, device=x.device)
    else:
        _chk_stats(stats, n * groups * 2, "stats")
    L.check(L.lib().pti_gn_stats(_ptr(x), _ptr(stats), n, h * w, c, groups, int(x.dtype == F16), _stream()),
            "pti_gn_stats")
    return stats


def conv_mfma(x, w_packed, bias, y, *, cout, ksize=3, mode=PTI_CONV_S1, prologue=PTI_PRO_NONE, in_stats=None,
              gamma=None, beta=None, groups=0, eps=1e-6, residual=None, out_stats=None, out_groups=0, act_out=None,
              pool2=False, relu=False):
    """``act_out`` (optional, bf16, x's shape): also write prologue(x) for the weight-gradient pass to reuse.
    ``pool2``: y is [n, ho/2, wo/2, cout], the 2x2 sum pool of the conv output (fused nearest-2x up-sampling backward).
    ``relu``: y = max(conv + bias, 0) (plain fp16 forward launches: the perceptual network's Fire modules)."""
    _chk(x, ACT16, "x", 4)
    _chk(y, ACT16, "y", 4)
    n, h, w, cin = x.shape
    if act_out is not None:
        _chk(act_out, BF16, "act_out", 4)
        if act_out.shape != x.shape:
            raise ValueError("conv_mfma: act_out shape")
    ho, wo = conv_out_hw(h, w, mode)
    yshape = (n, ho // 2, wo // 2, cout) if pool2 else (n, ho, wo, cout)
    if tuple(y.shape) != yshape:
        raise ValueError(f"conv_mfma: y shape {tuple(y.shape)} != {yshape}")
    if w_packed.numel() != cout * cin * ksize * ksize or w_packed.dtype not in (BF16, F16):
        raise ValueError("conv_mfma: packed weight size/dtype mismatch")
    w_f16 = w_packed.dtype == F16   # fp16-packed weights: the MFMA multiplies fp16 operands (forward, fp16 storage)
    if bias is not None:
        _chk(bias, F32, "bias")
        if bias.numel() != cout:
            raise ValueError("conv_mfma: bias size")
    if prologue != PTI_PRO_NONE:
        _chk_stats(in_stats, n * groups * 2, "in_stats")
        for t, nm, cnt in ((gamma, "gamma", cin), (beta, "beta", cin)):
            _chk(t, F32, nm)
            if t.numel() != cnt:
                raise ValueError(f"conv_mfma: {nm} size")
    if residual is not None:
        _chk(residual, ACT16, "residual", 4)
        if residual.shape != y.shape or pool2:
            raise ValueError("conv_mfma: residual shape")
    if out_stats is not None:
        _chk_stats(out_stats, n * out_groups * 2, "out_stats")
    d = ConvDesc(n=n, h=h, w=w, cin=cin, ho=ho, wo=wo, cout=cout, ksize=ksize, mode=mode, prologue=prologue,
                 groups=groups, add_residual=int(residual is not None), accum_stats=int(out_stats is not None),
                 out_groups=out_groups, eps=eps, in_f16=int(x.dtype == F16),
                 res_f16=int(residual is not None and residual.dtype == F16), out_f16=int(y.dtype == F16),
                 pool2x2_out=int(pool2), w_f16=int(w_f16), relu_out=int(relu))
    prof = KERNEL_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if act_out is not None:
        L.check(L.lib().pti_conv2d_mfma_saveact(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(in_stats), _ptr(gamma),
                                                _ptr(beta), _ptr(residual), _ptr(y), _ptr(out_stats), _ptr(act_out),
                                                C.byref(d), _stream()), "pti_conv2d_mfma_saveact")
    else:
        L.check(L.lib().pti_conv2d_mfma(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(in_stats), _ptr(gamma), _ptr(beta),
                                        _ptr(residual), _ptr(y), _ptr(out_stats), C.byref(d), _stream()),
                "pti_conv2d_mfma")
    if prof is not None:
        e1.record()
        # algorithmic work; the zero-insert data gradient only has 1/4 useful taps per output pixel
        flops = 2.0 * n * ho * wo * cout * cin * ksize * ksize * (0.25 if mode == PTI_CONV_ZINS else 1.0)
        # algorithmic bytes: read the input once (16-bit), write the output once (+ residual read, + side output)
        nbytes = 2.0 * (x.numel() * (2 if act_out is not None else 1) + y.numel() * (2 if residual is not None else 1))   # (pooled y counted as stored)
        kind = "conv fwd" if x.dtype == F16 or prologue != PTI_PRO_NONE else "conv dgrad"
        prof.append((last_kernel_name(), flops, nbytes, e0, e1,
                     (kind, cin, cout, ho, wo, ksize, {PTI_CONV_S1: "s1", PTI_CONV_S2PAD: "s2", PTI_CONV_UP2: "up2",
                                                       PTI_CONV_ZINS: "zins"}[mode], n)))
    return y


# Set to a list to make the MFMA conv / weight-gradient launchers record (kernel name, algorithmic flops, bytes, start
# event, end event, shape) per launch on the current stream -- used by bench.py for the roofline line and its per-shape
# table; None costs nothing.  The kernel name is the symbol the HIP runtime reports for the launch
# (pti_last_kernel_name), shortened the way tools/pmc_traffic.py shortens rocprofv3's Kernel_Name column.
KERNEL_PROFILE = None


def last_kernel_name() -> str:
    name = (L.lib().pti_last_kernel_name() or b"").decode()
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):       # drop the trailing argument list, keep template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i].strip()
    return name.strip()


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
    if prologue != PTI_PRO_NONE:
        _chk_stats(in_stats, n * groups * 2, "in_stats")
    d = ConvDesc(n=n, h=h, w=w, cin=cin, ho=h, wo=w, cout=cout, ksize=ksize, mode=PTI_CONV_S1, prologue=prologue,
                 groups=groups, eps=eps, in_f32=int(x.dtype == F32), out_f32=int(y.dtype == F32),
                 in_f16=int(x.dtype == F16), out_f16=int(y.dtype == F16))
    d.in_stride = (C.c_int64 * 4)(*_strides4(x, x_layout))
    d.out_stride = (C.c_int64 * 4)(*_strides4(y, y_layout))
    if cout % 32 == 0 and cin <= 16:
        if y.dtype not in ACT16 or not y.is_contiguous() or y_layout != "nhwc":
            raise ValueError("conv_direct: wide output must be dense NHWC bf16/fp16")
    else:
        if x.dtype not in ACT16 or not x.is_contiguous() or x_layout != "nhwc":
            raise ValueError("conv_direct: wide input must be dense NHWC bf16/fp16")
    L.check(L.lib().pti_conv2d_direct(_ptr(x), _ptr(w_tck), _ptr(bias), _ptr(in_stats), _ptr(gamma), _ptr(beta),
                                      _ptr(y), C.byref(d), _stream()), "pti_conv2d_direct")
    return y


def wgrad_direct(wide, narrow, dw, *, n, h, w, cw, cn, ksize, sgn, narrow_layout, dw_strides, dbias_wide=None,
                 dbias_narrow=None, prologue=PTI_PRO_NONE, in_stats=None, gamma=None, beta=None, groups=0, eps=1e-6,
                 workspace=None):
    """dw[tap,cw,k] += sum_p narrow[p,k] * T(wide)[p + sgn*tap, cw]; dw_strides = (tap, cw, k) element
    strides into the fp32 OIHW gradient ``dw`` (must be zero-initialised or hold a running sum)."""
    _chk(wide, ACT16, "wide", 4)
    _chk(dw, F32, "dw")
    if prologue != PTI_PRO_NONE:
        _chk_stats(in_stats, n * groups * 2, "in_stats")
    ns = (C.c_int64 * 4)(*_strides4(narrow, narrow_layout))
    ws = workspace if workspace is not None else wgrad_workspace(wide.device)
    L.check(L.lib().pti_wgrad_direct(_ptr(wide), _ptr(narrow), _ptr(dw), _ptr(dbias_wide), _ptr(dbias_narrow),
                                     _ptr(in_stats), _ptr(gamma), _ptr(beta), n, h, w, cw, cn, ksize, sgn, prologue,
                                     groups, eps, int(narrow.dtype == F32), int(wide.dtype == F16), ns, dw_strides[0],
                                     dw_strides[1],
                                     dw_strides[2], _ptr(ws), ws.numel() * 4, _stream()), "pti_wgrad_direct")
    return dw


_WS = {}


def wgrad_workspace(device, nbytes=int(os.environ.get("PTI_WGRAD_WORKSPACE_MB", "256")) << 20):
    """One reusable split-K workspace per device (slabs of fp32 partial weight gradients).  256 MB by default: a 256 -> 256
    layer's slab is 2.4 MB per pixel split, and a batched launch of 16 such layers that can afford only one or two splits
    per layer has fewer workgroups than the chip has CUs (round 2's 48 MB did that to the AR model's batches)."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    ws = _WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(nbytes // 4, dtype=F32, device=device)
        _WS[key] = ws
    return ws


def conv_wgrad_mfma(x, dy, dw, dbias, *, ksize=3, mode=PTI_CONV_S1, prologue=PTI_PRO_NONE, in_stats=None, gamma=None,
                    beta=None, groups=0, eps=1e-6, accumulate=False, workspace=None):
    _chk(x, ACT16, "x", 4)
    _chk(dy, BF16, "dy", 4)
    _chk(dw, F32, "dw")
    n, h, w, cin = x.shape
    ho, wo = conv_out_hw(h, w, mode)
    cout = dy.shape[3]
    if tuple(dy.shape) != (n, ho, wo, cout):
        raise ValueError(f"conv_wgrad_mfma: dy shape {tuple(dy.shape)} != {(n, ho, wo, cout)}")
    if dw.numel() != cout * cin * ksize * ksize:
        raise ValueError("conv_wgrad_mfma: dw size")
    if dbias is not None:
        _chk(dbias, F32, "dbias")
        if dbias.numel() != cout:
            raise ValueError("conv_wgrad_mfma: dbias size")
    if prologue != PTI_PRO_NONE:
        _chk_stats(in_stats, n * groups * 2, "in_stats")
        for t, nm, cnt in ((gamma, "gamma", cin), (beta, "beta", cin)):
            _chk(t, F32, nm)
            if t.numel() != cnt:
                raise ValueError(f"conv_wgrad_mfma: {nm} size")
    ws = workspace if workspace is not None else wgrad_workspace(x.device)
    d = ConvDesc(n=n, h=h, w=w, cin=cin, ho=ho, wo=wo, cout=cout, ksize=ksize, mode=mode, prologue=prologue,
                 groups=groups, eps=eps, in_f16=int(x.dtype == F16))
    prof = KERNEL_PROFILE
    if prof is None:
        L.check(L.lib().pti_conv_wgrad_mfma(_ptr(x), _ptr(dy), _ptr(in_stats), _ptr(gamma), _ptr(beta), _ptr(dw),
                                            _ptr(dbias), _ptr(ws), ws.numel() * 4, int(accumulate), C.byref(d),
                                            _stream()), "pti_conv_wgrad_mfma")
        return dw
    # profiling: the same two launches through the two-call form, with events around the partial (MFMA) kernel only
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    splits = C.c_int(0)
    e0.record()
    L.check(L.lib().pti_conv_wgrad_mfma_partials(_ptr(x), _ptr(dy), _ptr(in_stats), _ptr(gamma), _ptr(beta), _ptr(ws),
                                                 ws.numel() * 4, C.byref(d), C.byref(splits), _stream()),
            "pti_conv_wgrad_mfma_partials")
    e1.record()
    name = last_kernel_name()
    L.check(L.lib().pti_conv_wgrad_reduce(_ptr(ws), splits.value, _ptr(dw), _ptr(dbias), int(accumulate), C.byref(d),
                                          _stream()), "pti_conv_wgrad_reduce")
    flops = 2.0 * n * ho * wo * cout * cin * ksize * ksize
    # algorithmic bytes: x and dy read once (16-bit); dw itself is negligible (the split-K slabs are not algorithmic)
    prof.append((name, flops, 2.0 * (x.numel() + dy.numel()), e0, e1,
                 ("conv wgrad", cin, cout, ho, wo, ksize, {PTI_CONV_S1: "s1", PTI_CONV_S2PAD: "s2", PTI_CONV_UP2: "up2"}[mode], n)))
    return dw


def wgrad_batch_eligible(x, dy, ksize, mode, prologue):
    """Whether pti_conv_wgrad_mfma_batched can take this weight gradient: plain stride-1 3x3, bf16 x without prologue."""
    return (ksize == 3 and mode == PTI_CONV_S1 and prologue == PTI_PRO_NONE and x.dtype == BF16 and dy.dtype == BF16
            and x.shape[3] % 32 == 0 and dy.shape[3] % 32 == 0 and x.numel() * 2 < (1 << 31) and dy.numel() * 2 < (1 << 31))


def conv_wgrad_mfma_batched(jobs, workspace=None, accumulate=True):
    """``jobs``: up to WGRAD_BATCH_MAX tuples (x [n,h,w,cin] bf16, dy [n,h,w,cout] bf16, dw fp32 [cout*cin*9], dbias fp32
    [cout] | None) of plain stride-1 3x3 convs -> ONE partial launch + ONE reduction launch on the current stream."""
    if not 1 <= len(jobs) <= L.WGRAD_BATCH_MAX:
        raise ValueError(f"conv_wgrad_mfma_batched: 1..{L.WGRAD_BATCH_MAX} jobs, got {len(jobs)}")
    arr = (L.WgradJob * len(jobs))()
    flops = nbytes = 0.0
    for i, (x, dy, dw, db) in enumerate(jobs):
        _chk(x, BF16, "x", 4)
        _chk(dy, BF16, "dy", 4)
        _chk(dw, F32, "dw")
        n, h, w, cin = x.shape
        cout = dy.shape[3]
        if tuple(dy.shape) != (n, h, w, cout) or dw.numel() != cout * cin * 9 or (db is not None and db.numel() != cout):
            raise ValueError(f"conv_wgrad_mfma_batched: job {i}: shapes")
        if db is not None:
            _chk(db, F32, "dbias")
        arr[i] = L.WgradJob(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None if db is None else db.data_ptr(), n, h, w, cin,
                            cout, int(accumulate))
        flops += 2.0 * n * h * w * cout * cin * 9
        nbytes += 2.0 * (x.numel() + dy.numel())
    ws = workspace if workspace is not None else wgrad_workspace(jobs[0][0].device)
    prof = KERNEL_PROFILE
    if prof is None:
        L.check(L.lib().pti_conv_wgrad_mfma_batched(arr, len(jobs), _ptr(ws), ws.numel() * 4, _stream()),
                "pti_conv_wgrad_mfma_batched")
        return
    # profiling: the library launches one kernel per mode (wgrad_mfma.hip, w4_fill_job: the v6 kernel's two shapes, two
    # output-channel blocks per workgroup for Cout % 64 == 0, tile pairs otherwise); issue the groups as separate calls
    # so that each kernel gets its own record (its own algorithmic work, its own duration = partial launch + the <1 %
    # reduction launch) under its own name
    cob2 = os.environ.get("PTI_WGRAD_V4_COB2", "1") != "0"
    v6 = int(os.environ.get("PTI_WGRAD_V6", "3") or 0)

    def mode_of(x, cout):
        cin = x.shape[3]
        if v6 >= 1 and cout % 128 == 0 and cin % 64 == 0:
            return 2
        if v6 >= 2 and cout % 64 == 0 and cin % 64 == 0 and (v6 == 2 or x.shape[1] * x.shape[2] <= 128 * 128):
            return 3
        return 1 if cob2 and cout % 64 == 0 else 0
    groups = {}
    for i, (x, dy, dw, db) in enumerate(jobs):
        groups.setdefault(mode_of(x, dy.shape[3]), []).append(i)
    for mode in (3, 2, 1, 0):
        idx = groups.get(mode)
        if not idx:
            continue
        sub = (L.WgradJob * len(idx))(*[arr[i] for i in idx])
        fl = sum(2.0 * jobs[i][0].shape[0] * jobs[i][0].shape[1] * jobs[i][0].shape[2] * jobs[i][1].shape[3] * jobs[i][0].shape[3] * 9
                 for i in idx)
        nb = sum(2.0 * (jobs[i][0].numel() + jobs[i][1].numel()) for i in idx)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().pti_conv_wgrad_mfma_batched(sub, len(idx), _ptr(ws), ws.numel() * 4, _stream()),
                "pti_conv_wgrad_mfma_batched")
        e1.record()
        prof.append((last_kernel_name(), fl, nb, e0, e1, ("conv wgrad (batched)", 0, 0, 0, 0, 3, "s1", len(idx))))


def gn_bwd(x, da, dx, stats, gamma, beta, sums, dgamma, dbeta, *, groups, eps=1e-6, silu=True, dres=None):
    """dx <- backward of act(GroupNorm(x)); ``sums`` is an fp32 [n,c,2] scratch (written; no zeroing needed)."""
    _chk(x, ACT16, "x", 4)
    _chk(da, BF16, "da", 4)
    _chk(dx, BF16, "dx", 4)
    n, h, w, c = x.shape
    if da.shape != x.shape or dx.shape != x.shape or (dres is not None and dres.shape != x.shape):
        raise ValueError("gn_bwd: shape mismatch")
    _chk_stats(stats, n * groups * 2, "stats")
    if sums.numel() != n * c * 2:
        raise ValueError("gn_bwd: scratch sizes")
    blocks = L.lib().pti_gn_bwd_blocks(n, h * w, c)
    if blocks <= 0:
        raise ValueError(f"gn_bwd: unsupported channel count {c}")
    part = torch.empty(n * blocks * c * 2, dtype=torch.float32, device=x.device)
    L.check(L.lib().pti_gn_bwd(_ptr(x), _ptr(da), _ptr(dres), _ptr(dx), _ptr(stats), _ptr(gamma), _ptr(beta),
                               _ptr(sums), _ptr(part), _ptr(dgamma), _ptr(dbeta), n, h * w, c, groups, eps, int(silu),
                               int(x.dtype == F16), _stream()), "pti_gn_bwd")
    return dx


def conv_mfma_gnbwd(dy_in, w_packed_t, gx, gstats, ggamma, gbeta, dy_out, gsums, *, cout, ksize=3, mode=PTI_CONV_S1,
                    groups=0, eps=1e-6, silu=True):
    """Data-gradient conv with the GroupNorm(+SiLU) backward reduction fused into its epilogue:
    dy_out = conv^T(dy_in) * act'(GN(gx)); gsums[n,c] = {sum dy_out, sum dy_out*xhat} (one partial row per pixel tile
    from the conv, added up in a fixed order by pti_gn_sums_finalize: no float atomics, bitwise reproducible)."""
    _chk(dy_in, BF16, "dy_in", 4)
    _chk(gx, ACT16, "gx", 4)
    _chk(dy_out, BF16, "dy_out", 4)
    n, h, w, cin = dy_in.shape
    ho, wo = conv_out_hw(h, w, mode)
    if tuple(dy_out.shape) != (n, ho, wo, cout) or gx.shape != dy_out.shape:
        raise ValueError("conv_mfma_gnbwd: shapes")
    _chk_stats(gstats, n * groups * 2, "gstats")
    if gsums.numel() != n * cout * 2 or ggamma.numel() != cout:
        raise ValueError("conv_mfma_gnbwd: GroupNorm buffers")
    d = ConvDesc(n=n, h=h, w=w, cin=cin, ho=ho, wo=wo, cout=cout, ksize=ksize, mode=mode, groups=groups, eps=eps,
                 res_f16=int(gx.dtype == F16))
    tiles = L.lib().pti_conv_gnbwd_tiles(C.byref(d))
    if tiles <= 0:
        raise ValueError("conv_mfma_gnbwd: unsupported shape")
    part = torch.empty(n * tiles * cout * 2, dtype=torch.float32, device=dy_in.device)
    prof = KERNEL_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.lib().pti_conv2d_mfma_gnbwd(_ptr(dy_in), _ptr(w_packed_t), _ptr(gx), _ptr(gstats), _ptr(ggamma),
                                          _ptr(gbeta), _ptr(dy_out), _ptr(part), C.byref(d), int(silu), _stream()),
            "pti_conv2d_mfma_gnbwd")
    if prof is not None:
        e1.record()
        name = last_kernel_name()
    L.check(L.lib().pti_gn_sums_finalize(_ptr(part), _ptr(gsums), n, tiles, 2 * cout, _stream()), "pti_gn_sums_finalize")
    if prof is not None:
        prof.append((name, 2.0 * n * ho * wo * cout * cin * ksize * ksize * (0.25 if mode == PTI_CONV_ZINS else 1.0),
                     2.0 * (dy_in.numel() + 2 * dy_out.numel()), e0, e1,
                     ("conv dgrad+GN bwd", cin, cout, ho, wo, ksize, "zins" if mode == PTI_CONV_ZINS else "s1", n)))
    return dy_out


def gnbwd_chain_supported(cin, cout, ksize, x_dtype, groups=16):
    """Whether conv_mfma_gnbwd_chain covers a data-gradient launch with ``cin`` input / ``cout`` output channels whose
    epilogue GroupNorm input is stored as ``x_dtype`` (the chained GroupNorm needs at least 8 channels per group: one
    16-byte piece of the loader lies inside one group)."""
    return (x_dtype == F16 and groups > 0 and cin % groups == 0 and cin // groups >= 8
            and bool(L.lib().pti_conv_gnbwd_chain_supported(int(cin), int(cout), int(ksize))))


def conv_mfma_gnbwd_chain(g_in, x_in, in_stats, in_gamma, in_sums, dx_in, w_packed_t, gx, gstats, ggamma, gbeta, dy_out, gsums,
                          *, cout, groups, eps=1e-6, silu=True, in_dgamma=None, in_dbeta=None):
    """conv_mfma_gnbwd whose INPUT is the un-applied GroupNorm backward of the layer above: ``g_in`` = dA * act'(GN(x_in)),
    ``x_in`` that GroupNorm's input, ``in_sums`` its finalized sums; the loader applies rstd*(gamma*g - c1 - xhat*c2) on the
    way in and writes that tensor to ``dx_in`` (bf16) for the weight gradient of the conv in between -- the
    pti_gn_bwd_apply launch of that GroupNorm disappears.  Its affine gradients are added to ``in_dgamma`` / ``in_dbeta`` by
    the finalize launch of this call when given (otherwise: gn_affine_grads)."""
    _chk(g_in, BF16, "g_in", 4)
    _chk(x_in, ACT16, "x_in", 4)
    _chk(dx_in, BF16, "dx_in", 4)
    _chk(gx, ACT16, "gx", 4)
    _chk(dy_out, BF16, "dy_out", 4)
    n, h, w, cin = g_in.shape
    if x_in.shape != g_in.shape or dx_in.shape != g_in.shape or tuple(dy_out.shape) != (n, h, w, cout) or gx.shape != dy_out.shape:
        raise ValueError("conv_mfma_gnbwd_chain: shapes")
    _chk_stats(in_stats, n * groups * 2, "in_stats")
    _chk_stats(gstats, n * groups * 2, "gstats")
    if in_sums.numel() != n * cin * 2 or gsums.numel() != n * cout * 2 or in_gamma.numel() != cin or ggamma.numel() != cout:
        raise ValueError("conv_mfma_gnbwd_chain: GroupNorm buffers")
    d = ConvDesc(n=n, h=h, w=w, cin=cin, ho=h, wo=w, cout=cout, ksize=3, mode=PTI_CONV_S1, groups=groups, eps=eps,
                 res_f16=int(gx.dtype == F16))
    tiles = L.lib().pti_conv_gnbwd_tiles(C.byref(d))
    if tiles <= 0:
        raise ValueError("conv_mfma_gnbwd_chain: unsupported shape")
    part = torch.empty(n * tiles * cout * 2, dtype=torch.float32, device=g_in.device)
    prof = KERNEL_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.lib().pti_conv2d_mfma_gnbwd_chain(_ptr(g_in), _ptr(x_in), int(x_in.dtype == F16), _ptr(in_stats), _ptr(in_gamma),
                                                _ptr(in_sums), _ptr(dx_in), _ptr(w_packed_t), _ptr(gx), _ptr(gstats),
                                                _ptr(ggamma), _ptr(gbeta), _ptr(dy_out), _ptr(part), C.byref(d), int(silu),
                                                _stream()), "pti_conv2d_mfma_gnbwd_chain")
    if prof is not None:
        e1.record()
        name = last_kernel_name()
    if in_dgamma is not None or in_dbeta is not None:
        L.check(L.lib().pti_gn_sums_finalize_affine(_ptr(part), _ptr(gsums), n, tiles, 2 * cout, _ptr(in_sums), _ptr(in_dgamma),
                                                    _ptr(in_dbeta), cin, _stream()), "pti_gn_sums_finalize_affine")
    else:
        L.check(L.lib().pti_gn_sums_finalize(_ptr(part), _ptr(gsums), n, tiles, 2 * cout, _stream()), "pti_gn_sums_finalize")
    if prof is not None:     # reads g, x_in, gx; writes dx_in and dy_out
        prof.append((name, 2.0 * n * h * w * cout * cin * 9, 2.0 * (3 * g_in.numel() + 2 * dy_out.numel()), e0, e1,
                     ("conv dgrad+GN bwd (chained)", cin, cout, h, w, 3, "s1", n)))
    return dy_out


def gn_affine_grads(sums, dgamma, dbeta, n, c):
    """dgamma[c] += sum_n sums[n][c][1]; dbeta[c] += sum_n sums[n][c][0] (the affine gradients pti_gn_bwd_apply would add)."""
    L.check(L.lib().pti_gn_affine_grads(_ptr(sums), _ptr(dgamma), _ptr(dbeta), int(n), int(c), _stream()), "pti_gn_affine_grads")


def gn_bwd_apply(x, dy, dx, stats, gamma, beta, sums, dgamma, dbeta, *, groups, eps=1e-6, dres=None):
    _chk(x, ACT16, "x", 4)
    _chk(dy, BF16, "dy", 4)
    _chk(dx, BF16, "dx", 4)
    n, h, w, c = x.shape
    if dy.shape != x.shape or dx.shape != x.shape or (dres is not None and dres.shape != x.shape):
        raise ValueError("gn_bwd_apply: shape mismatch")
    _chk_stats(stats, n * groups * 2, "stats")
    prof = KERNEL_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.lib().pti_gn_bwd_apply(_ptr(x), _ptr(dy), _ptr(dres), _ptr(dx), _ptr(stats), _ptr(gamma), _ptr(beta),
                                     _ptr(sums), _ptr(dgamma), _ptr(dbeta), n, h * w, c, groups, eps,
                                     int(x.dtype == F16), _stream()), "pti_gn_bwd_apply")
    if prof is not None:   # pure HBM pass: reads x, dy (+ dres), writes dx; ~8 flops per element
        e1.record()
        prof.append((last_kernel_name(), 8.0 * x.numel(), 2.0 * x.numel() * (4 if dres is not None else 3), e0, e1,
                     ("GroupNorm bwd apply", c, c, h, w, 0, "-", n)))
    return dx


def pool2x2_sum(x, y):
    _chk(x, BF16, "x", 4)
    _chk(y, BF16, "y", 4)
    n, h2, w2, c = x.shape
    if tuple(y.shape) != (n, h2 // 2, w2 // 2, c) or h2 % 2 or w2 % 2:
        raise ValueError("pool2x2_sum: shapes")
    L.check(L.lib().pti_pool2x2_sum(_ptr(x), _ptr(y), n, h2 // 2, w2 // 2, c, _stream()), "pti_pool2x2_sum")
    return y


def latent_head_fwd(h, eps, wm, bm, wl, bl, wp, bp, mu, sigma, logvar, zq):
    b, hw, l = h.shape
    for t, nm in ((h, "h"), (wm, "wm"), (bm, "bm"), (wl, "wl"), (bl, "bl"), (wp, "wp"), (bp, "bp"), (mu, "mu"),
                  (sigma, "sigma"), (zq, "zq")):
        _chk(t, F32, nm)
    if mu.numel() != b * hw * l or sigma.numel() != b * hw * l or zq.numel() != b * hw * l:
        raise ValueError("latent_head_fwd: output sizes")
    if eps is not None and (eps.numel() != b * hw * l or eps.dtype != F32 or not eps.is_contiguous()):
        raise ValueError("latent_head_fwd: eps must be contiguous fp32 of the latent shape")
    L.check(L.lib().pti_latent_head_fwd(_ptr(h), _ptr(eps), _ptr(wm), _ptr(bm), _ptr(wl), _ptr(bl), _ptr(wp), _ptr(bp),
                                        _ptr(mu), _ptr(sigma), _ptr(logvar), _ptr(zq), b, hw, l, _stream()),
            "pti_latent_head_fwd")


def post_quant(z_nchw, wp, bp, zq):
    b, l = z_nchw.shape[0], z_nchw.shape[1]
    hw = z_nchw.numel() // (b * l)
    _chk(z_nchw, F32, "z")
    L.check(L.lib().pti_post_quant(_ptr(z_nchw), _ptr(wp), _ptr(bp), _ptr(zq), b, hw, l, _stream()), "pti_post_quant")


def post_quant_bwd(dzq, z_nchw, wp, dz, gwp, gbp):
    b, l = z_nchw.shape[0], z_nchw.shape[1]
    hw = z_nchw.numel() // (b * l)
    _chk(dzq, F32, "dzq")
    _chk(z_nchw, F32, "z")
    if dzq.numel() != z_nchw.numel() or (dz is not None and dz.numel() != z_nchw.numel()):
        raise ValueError("post_quant_bwd: sizes")
    ws = torch.empty(POST_QUANT_BWD_MAX_BLOCKS * (l * l + l), dtype=torch.float32, device=dzq.device)
    L.check(L.lib().pti_post_quant_bwd(_ptr(dzq), _ptr(z_nchw), _ptr(wp), _ptr(dz), _ptr(gwp), _ptr(gbp), _ptr(ws), b, hw, l,
                                       _stream()), "pti_post_quant_bwd")


LATENT_BWD_MAX_BLOCKS = 512   # PTI_LATENT_BWD_MAX_BLOCKS / PTI_VAE_LOSS_MAX_BLOCKS of include/pti_vae.h
VAE_LOSS_MAX_BLOCKS = 1024
POST_QUANT_BWD_MAX_BLOCKS = 256


def latent_head_bwd(h, eps, wm, bm, wl, bl, wp, bp, dzq, dmu, dsigma, dh, gwm, gbm, gwl, gbl, gwp, gbp):
    b, hw, l = h.shape
    for t in (dzq, dmu, dsigma):
        if t is not None and (t.dtype != F32 or not t.is_contiguous() or t.numel() != b * hw * l):
            raise ValueError("latent_head_bwd: gradient inputs must be contiguous fp32 of the latent size")
    ws = torch.empty(LATENT_BWD_MAX_BLOCKS * 3 * (l * l + l), dtype=torch.float32, device=h.device)
    L.check(L.lib().pti_latent_head_bwd(_ptr(h), _ptr(eps), _ptr(wm), _ptr(bm), _ptr(wl), _ptr(bl), _ptr(wp), _ptr(bp),
                                        _ptr(dzq), _ptr(dmu), _ptr(dsigma), _ptr(dh), _ptr(gwm), _ptr(gbm), _ptr(gwl),
                                        _ptr(gbl), _ptr(gwp), _ptr(gbp), _ptr(ws), b, hw, l, _stream()),
            "pti_latent_head_bwd")


def vae_loss(recon, images, mu, third, out2, d_recon, d_mu, d_third, *, l2=False, third_mode=0, kl_weight=1e-3):
    for t, nm in ((recon, "recon"), (images, "images"), (mu, "mu"), (third, "third"), (out2, "out2")):
        _chk(t, F32, nm)
    if recon.shape != images.shape or mu.shape != third.shape or out2.numel() < 2:
        raise ValueError("vae_loss: shapes")
    ws = torch.empty(2 * VAE_LOSS_MAX_BLOCKS, dtype=torch.float32, device=recon.device)
    L.check(L.lib().pti_vae_loss(_ptr(recon), _ptr(images), recon.numel(), _ptr(mu), _ptr(third), mu.numel(),
                                 recon.shape[0], _ptr(out2), _ptr(d_recon), _ptr(d_mu), _ptr(d_third), _ptr(ws), int(l2),
                                 third_mode, kl_weight, _stream()), "pti_vae_loss")


def ar_vae_loss(mu, attrs, channels, deltas, per_attr, counts, *, gamma=0.0, d_mu=None, pair_mask=None):
    """AR-VAE term on the device (``pti_ar_vae_loss``): mu [b,l,h,w] fp32, attrs [na,b] fp32, channels int32 [na],
    deltas fp32 [na] -> per_attr fp32 [na], counts int32 [na]; ``d_mu`` (same shape as mu) += gamma * gradient."""
    _chk(mu, F32, "mu", 4)
    _chk(attrs, F32, "attrs", 2)
    _chk(deltas, F32, "deltas")
    _chk(per_attr, F32, "per_attr")
    b, l, h, w = mu.shape
    na = attrs.shape[0]
    if attrs.shape[1] != b or channels.numel() != na or deltas.numel() != na or per_attr.numel() != na or counts.numel() != na:
        raise ValueError("ar_vae_loss: attrs must be [na, b]; channels / deltas / per_attr / counts [na]")
    if channels.dtype != torch.int32 or counts.dtype != torch.int32 or not (channels.is_cuda and counts.is_cuda):
        raise TypeError("ar_vae_loss: channels / counts must be int32 device tensors")
    if d_mu is not None:
        _chk(d_mu, F32, "d_mu", 4)
        if d_mu.shape != mu.shape:
            raise ValueError("ar_vae_loss: d_mu shape")
    if pair_mask is not None:
        if pair_mask.dtype != torch.uint8 or tuple(pair_mask.shape) != (na, b, b) or not pair_mask.is_cuda or not pair_mask.is_contiguous():
            raise ValueError("ar_vae_loss: pair_mask must be a contiguous uint8 device tensor [na, b, b]")
    L.check(L.lib().pti_ar_vae_loss(_ptr(mu), b, l, h * w, _ptr(attrs), _ptr(channels), _ptr(deltas), na, _ptr(pair_mask),
                                    float(gamma), _ptr(per_attr), _ptr(counts), _ptr(d_mu), _stream()), "pti_ar_vae_loss")


def adam_step(p, g, m, v, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, step=1, grad_scale=1.0):
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, F32, nm)
        if t.numel() != p.numel():
            raise ValueError("adam_step: size mismatch")
    L.check(L.lib().pti_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps, step,
                                  grad_scale, _stream()), "pti_adam_step")


def cast_nchw_f32_to_nhwc_bf16(x, y):
    n, c = x.shape[0], x.shape[1]
    _chk(x, F32, "x")
    _chk(y, BF16, "y")
    L.check(L.lib().pti_cast_nchw_f32_to_nhwc_bf16(_ptr(x), _ptr(y), n, c, x.numel() // (n * c), _stream()), "cast")
    return y


def cast_nhwc_bf16_to_nchw_f32(x, y):
    n, c = y.shape[0], y.shape[1]
    _chk(x, BF16, "x")
    _chk(y, F32, "y")
    L.check(L.lib().pti_cast_nhwc_bf16_to_nchw_f32(_ptr(x), _ptr(y), n, c, y.numel() // (n * c), _stream()), "cast")
    return y


def attention_fwd(qkv, o, lse2):
    """qkv [B,L,3C] bf16 -> o [B,L,C] bf16, lse2 [B,L] fp32."""
    _chk(qkv, BF16, "qkv", 3)
    _chk(o, BF16, "o", 3)
    _chk(lse2, F32, "lse2")
    b, l, c3 = qkv.shape
    c = c3 // 3
    if tuple(o.shape) != (b, l, c) or lse2.numel() != b * l or c3 != 3 * c:
        raise ValueError("attention_fwd: shapes")
    L.check(L.lib().pti_attention_fwd(_ptr(qkv), _ptr(o), _ptr(lse2), b, l, c, _stream()), "pti_attention_fwd")
    return o


def attention_bwd(qkv, o, dout, lse2, delta, dqkv):
    _chk(qkv, BF16, "qkv", 3)
    _chk(o, BF16, "o", 3)
    _chk(dout, BF16, "dout", 3)
    _chk(dqkv, BF16, "dqkv", 3)
    b, l, c3 = qkv.shape
    c = c3 // 3
    if o.shape != dout.shape or tuple(o.shape) != (b, l, c) or dqkv.shape != qkv.shape:
        raise ValueError("attention_bwd: shapes")
    if lse2.numel() != b * l or delta.numel() != b * l:
        raise ValueError("attention_bwd: lse/delta sizes")
    L.check(L.lib().pti_attention_bwd(_ptr(qkv), _ptr(o), _ptr(dout), _ptr(lse2), _ptr(delta), _ptr(dqkv), b, l, c,
                                      _stream()), "pti_attention_bwd")
    return dqkv


class BatchedPacker:
    """All MFMA weight packs of a model as ONE kernel launch (the per-layer form costs ~110 launches per
    optimiser step).  Entries are (fp32 weight view [cout,cin,k,k], ksize, mode, flip[, f16]); the packed outputs are
    allocated here and live at fixed addresses, as do the weights (views of the parameter arena)."""

    def __init__(self, entries, device):
        lib = L.lib()
        esz = lib.pti_conv_pack_entry_bytes()
        host = bytearray(esz * len(entries))
        hbuf = (C.c_char * len(host)).from_buffer(host)
        first, total, self.outputs = [], 0, []
        for i, (w, ksize, mode, flip, *rest) in enumerate(entries):
            f16 = bool(rest[0]) if rest else False
            _chk(w, F32, "weight")
            cout, cin = w.shape[0], w.shape[1]
            out = torch.empty(w.numel(), dtype=F16 if f16 else BF16, device=device)
            nb = C.c_int64(0)
            L.check(lib.pti_conv_pack_table_fill(C.byref(hbuf, i * esz), _ptr(w), _ptr(out), cout, cin, ksize, mode,
                                                 int(flip), int(f16), C.byref(nb)), "pti_conv_pack_table_fill")
            first.append(total)
            total += nb.value
            self.outputs.append(out)
        self.n, self.total_blocks = len(entries), total
        self.table = torch.frombuffer(host, dtype=torch.uint8).clone().to(device)
        self.first = torch.tensor(first, dtype=torch.int32, device=device)
        self._keep = [e[0] for e in entries]

    def run(self):
        L.check(L.lib().pti_conv_pack_weights_batched(_ptr(self.table), _ptr(self.first), self.n, self.total_blocks,
                                                      _stream()), "pti_conv_pack_weights_batched")


def preprocess_batch(src, offsets, hw, out, stats=None):
    """Resize(area) + LocalNormalizeByMask of a batch of raw fp32 images (see include/pti_vae.h).
    src: flat fp32 device tensor; offsets int64 [B]; hw int32 [B,2]; out fp32 [B,1,Hp,Wp]."""
    _chk(src, F32, "src")
    _chk(out, F32, "out", 4)
    b, _, hp, wp = out.shape
    if offsets.dtype != torch.int64 or hw.dtype != torch.int32 or offsets.numel() != b or hw.numel() != 2 * b:
        raise ValueError("preprocess_batch: offsets must be int64 [B], hw int32 [B,2]")
    if not (offsets.is_cuda and hw.is_cuda and offsets.is_contiguous() and hw.is_contiguous()):
        raise ValueError("preprocess_batch: descriptor tables must be contiguous device tensors")
    if stats is None:
        stats = torch.empty(3 * b, dtype=torch.float64, device=out.device)
    L.check(L.lib().pti_preprocess_batch(_ptr(src), _ptr(offsets), _ptr(hw), b, hp, wp, _ptr(out), _ptr(stats), _stream()),
            "pti_preprocess_batch")
    return out


# ---- PatchDiscriminator passes (csrc/discriminator.hip; include/pti_vae.h "PatchDiscriminator") -----------------------
def pd_out_hw(h, w, stride):
    return (h + 2 - 4) // stride + 1, (w + 2 - 4) // stride + 1


def pd_im2col_image(img, patches):
    """img fp32 [B,1,H,W] (or [B,H,W]) -> patches bf16 [B,H/2,W/2,32] of the 4x4 stride-2 pad-1 window (+16 zero columns)."""
    _chk(img, F32, "img")
    _chk(patches, BF16, "patches", 4)
    b, h, w = img.shape[0], img.shape[-2], img.shape[-1]
    if img.numel() != b * h * w or tuple(patches.shape) != (b, h // 2, w // 2, 32):
        raise ValueError(f"pd_im2col_image: img {tuple(img.shape)} / patches {tuple(patches.shape)}")
    L.check(L.lib().pti_pd_im2col_image(_ptr(img), _ptr(patches), b, h, w, _stream()), "pti_pd_im2col_image")
    return patches


def pd_im2col(src, norm, patches, *, stride, act=True, slope=0.2):
    _chk(src, BF16, "src", 4)
    _chk(patches, BF16, "patches", 4)
    n, h, w, c = src.shape
    ho, wo = pd_out_hw(h, w, stride)
    if tuple(patches.shape) != (n, ho, wo, 16 * c):
        raise ValueError(f"pd_im2col: patches {tuple(patches.shape)} != {(n, ho, wo, 16 * c)}")
    if norm is not None:
        _chk(norm, F32, "norm")
        if norm.numel() != n * c * 2:
            raise ValueError("pd_im2col: norm table size")
    L.check(L.lib().pti_pd_im2col(_ptr(src), _ptr(norm), _ptr(patches), n, h, w, c, stride, int(act), float(slope), _stream()),
            "pti_pd_im2col")
    return patches


def pd_in_stats(y, eps=1e-5, table=None):
    _chk(y, BF16, "y", 4)
    n, h, w, c = y.shape
    if table is None:
        table = torch.empty(n, c, 2, dtype=F32, device=y.device)
    L.check(L.lib().pti_pd_in_stats(_ptr(y), _ptr(table), n, h * w, c, float(eps), _stream()), "pti_pd_in_stats")
    return table


def pd_col2im(d_patches, y_prev, norm, g, *, stride, slope=0.2):
    """-> (g, sums [n,c,2] | None): g = LeakyReLU'(norm(y_prev)) * col2im(d_patches); with ``norm`` also the
    InstanceNorm-backward sums {sum g, sum g*xhat} (block partials folded in fixed order by pti_gn_sums_finalize)."""
    _chk(d_patches, BF16, "d_patches", 4)
    _chk(y_prev, BF16, "y_prev", 4)
    _chk(g, BF16, "g", 4)
    n, h, w, c = y_prev.shape
    ho, wo = pd_out_hw(h, w, stride)
    if tuple(d_patches.shape) != (n, ho, wo, 16 * c) or g.shape != y_prev.shape:
        raise ValueError(f"pd_col2im: d_patches {tuple(d_patches.shape)} for y_prev {tuple(y_prev.shape)} stride {stride}")
    part = sums = None
    if norm is not None:
        _chk(norm, F32, "norm")
        bps = L.lib().pti_pd_col2im_blocks(n, h * w, c)
        part = torch.empty(n, bps, c, 2, dtype=F32, device=g.device)
    L.check(L.lib().pti_pd_col2im(_ptr(d_patches), _ptr(y_prev), _ptr(norm), _ptr(g), _ptr(part), n, h, w, c, stride,
                                  float(slope), _stream()), "pti_pd_col2im")
    if part is not None:
        sums = torch.empty(n, c, 2, dtype=F32, device=g.device)
        L.check(L.lib().pti_gn_sums_finalize(_ptr(part), _ptr(sums), n, part.shape[1], 2 * c, _stream()), "pti_gn_sums_finalize")
    return g, sums


def pd_col2im_image(d_patches, d_img, *, scale=1.0, accumulate=False):
    _chk(d_patches, BF16, "d_patches", 4)
    _chk(d_img, F32, "d_img")
    n, ho, wo, k = d_patches.shape
    if k != 32 or d_img.numel() != n * 4 * ho * wo:
        raise ValueError("pd_col2im_image: shapes")
    L.check(L.lib().pti_pd_col2im_image(_ptr(d_patches), _ptr(d_img), n, 2 * ho, 2 * wo, float(scale), int(accumulate),
                                        _stream()), "pti_pd_col2im_image")
    return d_img


def pd_in_bwd_apply(g, y, norm, sums, dy=None):
    _chk(g, BF16, "g", 4)
    _chk(y, BF16, "y", 4)
    n, h, w, c = y.shape
    dy = g if dy is None else dy
    L.check(L.lib().pti_pd_in_bwd_apply(_ptr(g), _ptr(y), _ptr(norm), _ptr(sums), _ptr(dy), n, h * w, c, _stream()),
            "pti_pd_in_bwd_apply")
    return dy


def pd_lsgan(logits_rows, *, target, slope=0.05, grad_scale=0.0, d_logits=None):
    """logits_rows: [M, stride], the logit in column 0 -- 16-bit padded rows, or fp32 (the direct final block: stride 1).
    Returns the one-element fp32 device tensor mean((LeakyReLU_slope(l) - target)^2); ``d_logits`` (same shape; bf16
    for 16-bit logits, fp32 for fp32 logits) gets grad_scale * (a - target) * LeakyReLU'(l) in column 0 (16-bit rows:
    zeros elsewhere)."""
    _chk(logits_rows, (BF16, F16, F32), "logits", 2)
    m, stride = logits_rows.shape
    fmt = 2 if logits_rows.dtype == F32 else int(logits_rows.dtype == F16)
    if d_logits is not None:
        _chk(d_logits, F32 if fmt == 2 else BF16, "d_logits", 2)
        if d_logits.shape != logits_rows.shape:
            raise ValueError("pd_lsgan: d_logits shape")
    buf = torch.empty(1 + L.lib().pti_pd_lsgan_blocks(m), dtype=F32, device=logits_rows.device)
    L.check(L.lib().pti_pd_lsgan(_ptr(logits_rows), fmt, stride, m, float(target), float(slope), float(grad_scale), _ptr(buf),
                                 _ptr(d_logits), _stream()), "pti_pd_lsgan")
    return buf[:1]


def pd_final_fwd(y_prev, norm, w16c, bias, logits, *, slope=0.2):
    """Final discriminator block without a patch matrix: logits fp32 [n, h-1, w-1] = conv4x4(LeakyReLU(norm(y_prev)), w) + b;
    w16c fp32 [16*c] (tap-major), bias fp32 [>=1]."""
    _chk(y_prev, BF16, "y_prev", 4)
    _chk(logits, F32, "logits", 3)
    n, h, w, c = y_prev.shape
    if tuple(logits.shape) != (n, h - 1, w - 1) or w16c.numel() != 16 * c:
        raise ValueError("pd_final_fwd: shapes")
    L.check(L.lib().pti_pd_final_fwd(_ptr(y_prev), _ptr(norm), _ptr(w16c), _ptr(bias), _ptr(logits), n, h, w, c, float(slope),
                                     _stream()), "pti_pd_final_fwd")
    return logits


def pd_final_dgrad(d_logits, y_prev, norm, w16c, g, *, slope=0.2):
    """-> (g, sums | None) like pd_col2im, for the direct final block (d_logits fp32 [n, h-1, w-1])."""
    _chk(d_logits, F32, "d_logits")
    _chk(y_prev, BF16, "y_prev", 4)
    _chk(g, BF16, "g", 4)
    n, h, w, c = y_prev.shape
    if d_logits.numel() != n * (h - 1) * (w - 1) or g.shape != y_prev.shape or w16c.numel() != 16 * c:
        raise ValueError("pd_final_dgrad: shapes")
    part = sums = None
    if norm is not None:
        part = torch.empty(n, L.lib().pti_pd_col2im_blocks(n, h * w, c), c, 2, dtype=F32, device=g.device)
    L.check(L.lib().pti_pd_final_dgrad(_ptr(d_logits), _ptr(y_prev), _ptr(norm), _ptr(w16c), _ptr(g), _ptr(part), n, h, w, c,
                                       float(slope), _stream()), "pti_pd_final_dgrad")
    if part is not None:
        sums = torch.empty(n, c, 2, dtype=F32, device=g.device)
        L.check(L.lib().pti_gn_sums_finalize(_ptr(part), _ptr(sums), n, part.shape[1], 2 * c, _stream()), "pti_gn_sums_finalize")
    return g, sums


def pd_final_wgrad(d_logits, y_prev, norm, *, slope=0.2):
    """-> fp32 [16*c + 8]: {dw[16][c], dbias, 0...} of the direct final block (block partials summed in block order)."""
    _chk(d_logits, F32, "d_logits")
    _chk(y_prev, BF16, "y_prev", 4)
    n, h, w, c = y_prev.shape
    if d_logits.numel() != n * (h - 1) * (w - 1):
        raise ValueError("pd_final_wgrad: shapes")
    blocks = L.lib().pti_pd_final_wgrad_blocks(n, h, w)
    row = 16 * c + 8
    part = torch.empty(blocks, row, dtype=F32, device=y_prev.device)
    L.check(L.lib().pti_pd_final_wgrad(_ptr(d_logits), _ptr(y_prev), _ptr(norm), _ptr(part), n, h, w, c, float(slope), _stream()),
            "pti_pd_final_wgrad")
    out = torch.empty(row, dtype=F32, device=y_prev.device)
    L.check(L.lib().pti_gn_sums_finalize(_ptr(part), _ptr(out), 1, blocks, row, _stream()), "pti_gn_sums_finalize")
    return out


# ---- LPIPS comparison tail (perceptual term) -------------------------------------------------------------
def lpips_tap_fwd(a, b, w):
    """a, b: fp32 NCHW feature maps [n, c, h, w_] (contiguous), w: fp32 [c] -> (value [n], saved [n, 3, h*w_]):
    value_i = mean_p sum_c w_c (a_c/(|a_p|+1e-10) - b_c/(|b_p|+1e-10))^2 (lpips normalize_tensor + lin layer + spatial mean)."""
    _chk(a, F32, "a", 4)
    _chk(b, F32, "b", 4)
    _chk(w, F32, "w", 1)
    if a.shape != b.shape or w.numel() != a.shape[1]:
        raise ValueError("lpips_tap_fwd: shapes")
    n, c, h, ww = a.shape
    hw = h * ww
    blocks = L.lib().pti_lpips_tap_blocks(c, hw)
    if blocks <= 0:
        raise ValueError("lpips_tap_fwd: empty feature map")
    saved = torch.empty(n, 3, hw, dtype=F32, device=a.device)
    part = torch.empty(n, blocks, dtype=F32, device=a.device)
    L.check(L.lib().pti_lpips_tap_fwd(_ptr(a), _ptr(b), _ptr(w), _ptr(saved), _ptr(part), n, c, hw, _stream()), "pti_lpips_tap_fwd")
    return part.sum(1) / hw, saved


def lpips_tap_bwd(a, b, w, saved, gout):
    """-> d(sum_i gout_i * value_i) / d a, fp32 like a."""
    _chk(a, F32, "a", 4)
    _chk(b, F32, "b", 4)
    _chk(gout, F32, "gout", 1)
    n, c, h, ww = a.shape
    if a.shape != b.shape or tuple(saved.shape) != (n, 3, h * ww) or gout.numel() != n or w.numel() != c:
        raise ValueError("lpips_tap_bwd: shapes")
    ga = torch.empty_like(a)
    L.check(L.lib().pti_lpips_tap_bwd(_ptr(a), _ptr(b), _ptr(w), _ptr(saved), _ptr(gout), _ptr(ga), n, c, h * ww, _stream()),
            "pti_lpips_tap_bwd")
    return ga


# ---- trunk of the perceptual network (csrc/squeeze.hip) -----------------------------------------------------------------
def relu_f16_(x):
    _chk(x, F16, "x")
    L.check(L.lib().pti_relu_f16(_ptr(x), x.numel(), _stream()), "pti_relu_f16")
    return x


def relu_bwd_(g, y):
    """g (bf16) = y > 0 ? g : 0 in place; y: the fp16 ReLU output of the same shape."""
    _chk(g, BF16, "g")
    _chk(y, F16, "y")
    if g.shape != y.shape:
        raise ValueError("relu_bwd_: shapes")
    L.check(L.lib().pti_relu_bwd(_ptr(g), _ptr(y), g.numel(), _stream()), "pti_relu_bwd")
    return g


def relu_bwd_add_(g, g2, y):
    """g (bf16) = y > 0 ? g + g2 : 0 in place."""
    _chk(g, BF16, "g")
    _chk(g2, BF16, "g2")
    _chk(y, F16, "y")
    if g.shape != y.shape or g2.shape != y.shape:
        raise ValueError("relu_bwd_add_: shapes")
    L.check(L.lib().pti_relu_bwd_add(_ptr(g), _ptr(g2), _ptr(y), g.numel(), _stream()), "pti_relu_bwd_add")
    return g


def maxpool3s2_fwd(x):
    """MaxPool2d(3, 2, ceil_mode=True) on NHWC fp16."""
    _chk(x, F16, "x", 4)
    n, h, w, c = x.shape
    y = torch.empty(n, L.lib().pti_maxpool3s2_out(h), L.lib().pti_maxpool3s2_out(w), c, dtype=F16, device=x.device)
    L.check(L.lib().pti_maxpool3s2_fwd(_ptr(x), _ptr(y), n, h, w, c, _stream()), "pti_maxpool3s2_fwd")
    return y


def maxpool3s2_bwd(gy, x, y, gx=None):
    """-> gx (bf16, x's shape): gradient of maxpool3s2_fwd; accumulated into ``gx`` when given."""
    _chk(gy, BF16, "gy", 4)
    _chk(x, F16, "x", 4)
    _chk(y, F16, "y", 4)
    if gy.shape != y.shape:
        raise ValueError("maxpool3s2_bwd: shapes")
    n, h, w, c = x.shape
    acc = gx is not None
    if acc:
        _chk(gx, BF16, "gx", 4)
        if gx.shape != x.shape:
            raise ValueError("maxpool3s2_bwd: gx shape")
    else:
        gx = torch.empty(x.shape, dtype=BF16, device=x.device)
    L.check(L.lib().pti_maxpool3s2_bwd(_ptr(gy), _ptr(x), _ptr(y), _ptr(gx), n, h, w, c, int(acc), _stream()), "pti_maxpool3s2_bwd")
    return gx


def nchw_f32_to_nhwc_f16(x):
    """fp32 [n, c, h, w] -> fp16 [n, h, w, c] (c a multiple of 64)."""
    _chk(x, F32, "x", 4)
    n, c, h, w = x.shape
    y = torch.empty(n, h, w, c, dtype=F16, device=x.device)
    L.check(L.lib().pti_nchw_f32_to_nhwc_f16(_ptr(x), _ptr(y), n, c, h * w, _stream()), "pti_nchw_f32_to_nhwc_f16")
    return y


def nhwc_bf16_add_to_nchw_f32_(g, y):
    """y (fp32 [n, c, h, w]) += g (bf16 [n, h, w, c])."""
    _chk(g, BF16, "g", 4)
    _chk(y, F32, "y", 4)
    n, c, h, w = y.shape
    if tuple(g.shape) != (n, h, w, c):
        raise ValueError("nhwc_bf16_add_to_nchw_f32_: shapes")
    L.check(L.lib().pti_nhwc_bf16_add_to_nchw_f32(_ptr(g), _ptr(y), n, c, h * w, _stream()), "pti_nhwc_bf16_add_to_nchw_f32")
    return y


def pad_nchw_to_nhwc32(x, dtype_a, dtype_b=None):
    """fp32 [n, c, h, w] (1 <= c <= 8) -> 16-bit [n, h, w, 32] with channels >= c zero; a second copy in ``dtype_b`` (the
    weight gradient's bf16 operand next to the forward conv's fp16 one) comes out of the same pass."""
    _chk(x, F32, "x", 4)
    n, c, h, w = x.shape
    ya = torch.empty(n, h, w, 32, dtype=dtype_a, device=x.device)
    yb = torch.empty(n, h, w, 32, dtype=dtype_b, device=x.device) if dtype_b is not None else None
    L.check(L.lib().pti_pad_nchw_to_nhwc32(_ptr(x), _ptr(ya), _ptr(yb), n, c, h * w, int(dtype_a == F16),
                                           int(dtype_b == F16), _stream()), "pti_pad_nchw_to_nhwc32")
    return ya, yb


def slice_nhwc32_to_nchw(x, c, out=None):
    """The first ``c`` channels of a 16-bit [n, h, w, 32] tensor -> fp32 [n, c, h, w]."""
    _chk(x, ACT16, "x", 4)
    n, h, w, c32 = x.shape
    if c32 != 32:
        raise ValueError("slice_nhwc32_to_nchw: expected 32 channels")
    y = out if out is not None else torch.empty(n, c, h, w, dtype=F32, device=x.device)
    _chk(y, F32, "y", 4)
    L.check(L.lib().pti_slice_nhwc32_to_nchw(_ptr(x), _ptr(y), n, c, h * w, int(x.dtype == F16), _stream()),
            "pti_slice_nhwc32_to_nchw")
    return y


def lpips_tap_nhwc_supported(c):
    return L.lib().pti_lpips_tap_nhwc_blocks(int(c), 1) > 0


def lpips_tap_nhwc_fwd(a, b, w):
    """lpips_tap_fwd on NHWC fp16 maps [n, h, w_, c] -> (value [n], saved [n, 3, h*w_])."""
    _chk(a, F16, "a", 4)
    _chk(b, F16, "b", 4)
    _chk(w, F32, "w", 1)
    if a.shape != b.shape or w.numel() != a.shape[3]:
        raise ValueError("lpips_tap_nhwc_fwd: shapes")
    n, h, ww, c = a.shape
    hw = h * ww
    blocks = L.lib().pti_lpips_tap_nhwc_blocks(c, hw)
    if blocks <= 0:
        raise ValueError(f"lpips_tap_nhwc_fwd: unsupported channel count {c}")
    saved = torch.empty(n, 3, hw, dtype=F32, device=a.device)
    part = torch.empty(n, blocks, dtype=F32, device=a.device)
    L.check(L.lib().pti_lpips_tap_nhwc_fwd(_ptr(a), _ptr(b), _ptr(w), _ptr(saved), _ptr(part), n, c, hw, _stream()),
            "pti_lpips_tap_nhwc_fwd")
    return part.sum(1) / hw, saved


def lpips_tap_nhwc_bwd(a, b, w, saved, gout):
    """-> d(sum_i gout_i * value_i) / d a as NHWC bf16."""
    _chk(a, F16, "a", 4)
    _chk(b, F16, "b", 4)
    _chk(gout, F32, "gout", 1)
    n, h, ww, c = a.shape
    if a.shape != b.shape or tuple(saved.shape) != (n, 3, h * ww) or gout.numel() != n or w.numel() != c:
        raise ValueError("lpips_tap_nhwc_bwd: shapes")
    ga = torch.empty(a.shape, dtype=BF16, device=a.device)
    L.check(L.lib().pti_lpips_tap_nhwc_bwd(_ptr(a), _ptr(b), _ptr(w), _ptr(saved), _ptr(gout), _ptr(ga), n, c, h * ww, _stream()),
            "pti_lpips_tap_nhwc_bwd")
    return ga


def squeeze_conv1_fwd(x, w10):
    """x fp32 [n, 1, h, w] (or [n, h, w]) -> tap 0 of the perceptual network, fp16 [n, (h-3)//2+1, (w-3)//2+1, 64];
    w10: the folded first layer, fp32 [10, 64] (perceptual_engine.fold_first_layer)."""
    _chk(x, F32, "x")
    _chk(w10, F32, "w10")
    if x.dim() == 4 and x.shape[1] == 1:
        x = x[:, 0]
    if x.dim() != 3 or w10.numel() != 640:
        raise ValueError("squeeze_conv1_fwd: expected a one-channel image batch and a [10, 64] table")
    n, h, w = x.shape
    y = torch.empty(n, (h - 3) // 2 + 1, (w - 3) // 2 + 1, 64, dtype=F16, device=x.device)
    L.check(L.lib().pti_squeeze_conv1_fwd(_ptr(x), _ptr(w10), _ptr(y), n, h, w, _stream()), "pti_squeeze_conv1_fwd")
    return y


def squeeze_conv1_bwd(g, t0, w10, h, w):
    """g bf16 [n, ho, wo, 64] (gradient w.r.t. tap 0), t0 = the forward's output (None: g already carries the ReLU
    mask) -> dx fp32 [n, 1, h, w]."""
    _chk(g, BF16, "g", 4)
    _chk(w10, F32, "w10")
    n = g.shape[0]
    if t0 is not None:
        _chk(t0, F16, "t0", 4)
    if (t0 is not None and g.shape != t0.shape) or tuple(g.shape[1:]) != ((h - 3) // 2 + 1, (w - 3) // 2 + 1, 64):
        raise ValueError("squeeze_conv1_bwd: shapes")
    dx = torch.empty(n, 1, h, w, dtype=F32, device=g.device)
    L.check(L.lib().pti_squeeze_conv1_bwd(_ptr(g), _ptr(t0), _ptr(w10), _ptr(dx), n, h, w, _stream()), "pti_squeeze_conv1_bwd")
    return dx
