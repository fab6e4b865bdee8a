"""HIP execution engine of the AutoencoderKL encoder / decoder (forward AND backward).

It walks the same block list MONAI's ``Encoder`` / ``Decoder`` iterate over (SURVEY.md Appendix
A.1/A.2) but never runs a PyTorch op on an activation: every step is one C-ABI launch
(``pti_ldm_vae_amd/ops.py``) on NHWC bf16 tensors, with GroupNorm(+SiLU) folded into the loader of
the consuming convolution, residual adds and the next GroupNorm's statistics folded into the
producing convolution's epilogue, nearest-2x up-sampling / asymmetric-pad stride-2 folded into the
conv's addressing, and the mid-block attention run flash-style.  PyTorch provides device memory
(caching allocator), streams and autograd plumbing only.

Two ``torch.autograd.Function`` s expose it: ``_EncodeFn`` (x -> z_mu, z_sigma) and ``_DecodeFn``
(z -> reconstruction).  Their backward passes write weight gradients straight into the model's
flat gradient arena and hand autograd views of it.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .models import autoencoderkl as M
from .ops import (BF16, F32, PTI_CONV_S1, PTI_CONV_S2PAD, PTI_CONV_UP2, PTI_CONV_ZINS, PTI_PRO_GN, PTI_PRO_GN_SILU,
                  PTI_PRO_NONE)


class _Act:
    """An NHWC bf16 activation plus (optionally) its GroupNorm {sum,sumsq} statistics."""

    __slots__ = ("t", "stats")

    def __init__(self, t, stats=None):
        self.t, self.stats = t, stats


def _empty(shape, like, dtype=BF16):
    return torch.empty(shape, dtype=dtype, device=like.device)


# =================================================================================================
# layers
# =================================================================================================
class _MfmaConv:
    """One MFMA convolution's parameters, packed operands and gradient slots."""

    def __init__(self, net, wname, bname, ksize, mode, fused_names=None, fwd_f16=True):
        self.net, self.ksize, self.mode = net, ksize, mode
        # forward operands: fp16 weights + fp16 activations on the MFMA when the engine stores fp16 (Engine sets
        # .f16 after construction); convs whose input or output is a bf16 tensor (q|k|v, out_proj) stay bf16
        self.fwd_f16, self.f16 = fwd_f16, False
        self.wname, self.bname, self.fused = wname, bname, fused_names
        if fused_names:  # q|k|v: three adjacent [C,C] weights seen as one [3C,C,1,1]
            w0 = net._param_by_name[fused_names[0][0]]
            self.cout, self.cin = 3 * w0.shape[0], w0.shape[1]
        else:
            w = net._param_by_name[wname]
            self.cout, self.cin = w.shape[0], w.shape[1]
        self.wp = self.wpt = None

    def _w(self):
        if self.fused:
            o, n, _ = self.net._slots[self.fused[0][0]]
            return self.net._arena[o:o + 3 * n].view(self.cout, self.cin, 1, 1)
        return self.net._param_by_name[self.wname].data.view(self.cout, self.cin, self.ksize, self.ksize)

    def bias(self):
        if self.fused:
            o, n, _ = self.net._slots[self.fused[0][1]]
            return self.net._arena[o:o + 3 * n]
        return self.net._param_by_name[self.bname].data

    def grads(self):
        if self.fused:
            o, n, _ = self.net._slots[self.fused[0][0]]
            ob, nb, _ = self.net._slots[self.fused[0][1]]
            g = self.net.grad_arena
            return g[o:o + 3 * n], g[ob:ob + 3 * nb]
        return self.net.grad_view(self.wname).view(-1), self.net.grad_view(self.bname)

    def repack(self):
        w = self._w()
        self.wp = ops.pack_conv_weight(w, self.ksize, self.mode, out=self.wp, f16=self.f16)
        dmode = PTI_CONV_ZINS if self.mode == PTI_CONV_S2PAD else PTI_CONV_S1
        self.wpt = ops.pack_conv_weight(w, self.ksize, dmode, flip=True, out=self.wpt)

    # y = conv(prologue(x)) + b [+ residual]; optional fused stats of y
    def fwd(self, x, *, pro=PTI_PRO_NONE, norm=None, residual=None, want_stats=False, eng=None, act_out=None,
            out_dtype=None):
        n, h, w, _ = x.t.shape
        ho, wo = ops.conv_out_hw(h, w, self.mode)
        y = _empty((n, ho, wo, self.cout), x.t, out_dtype or eng.act_dtype)
        st = eng.new_stats(n) if want_stats else None
        g, b = (norm.weight.data, norm.bias.data) if norm is not None else (None, None)
        ops.conv_mfma(x.t, self.wp, self.bias(), y, cout=self.cout, ksize=self.ksize, mode=self.mode, prologue=pro,
                      in_stats=x.stats if pro else None, gamma=g, beta=b, groups=eng.G, eps=eng.eps,
                      residual=residual, out_stats=st, out_groups=eng.G, act_out=act_out)
        return _Act(y, st)

    # data gradient w.r.t. the (post-prologue) input
    def dgrad(self, dy, *, residual=None):
        n, ho, wo, _ = dy.shape
        if self.mode == PTI_CONV_S2PAD:
            out = _empty((n, 2 * ho, 2 * wo, self.cin), dy)
            ops.conv_mfma(dy, self.wpt, None, out, cout=self.cin, ksize=3, mode=PTI_CONV_ZINS, residual=residual)
            return out
        if self.mode == PTI_CONV_UP2:   # gradient w.r.t. the pre-up-sampling map: 2x2 sum pool fused into the epilogue
            pooled = _empty((n, ho // 2, wo // 2, self.cin), dy)
            ops.conv_mfma(dy, self.wpt, None, pooled, cout=self.cin, ksize=self.ksize, mode=PTI_CONV_S1, pool2=True)
            return pooled
        out = _empty((n, ho, wo, self.cin), dy)
        ops.conv_mfma(dy, self.wpt, None, out, cout=self.cin, ksize=self.ksize, mode=PTI_CONV_S1, residual=residual)
        return out

    def dgrad_gn(self, dy, x, norm, *, silu, dres, eng):
        """Data gradient through  conv(act(GN(x)))  down to dx: the conv^T launch also does the GroupNorm
        backward reduction (fused epilogue), pti_gn_bwd_apply finishes.  Stride-1 convs only."""
        n, ho, wo, _ = dy.shape
        dyt = _empty((n, ho, wo, self.cin), dy)
        sums = eng.zeros(n * self.cin * 2)
        ops.conv_mfma_gnbwd(dy, self.wpt, x.t, x.stats, norm.weight.data, norm.bias.data, dyt, sums, cout=self.cin,
                            ksize=self.ksize, mode=PTI_CONV_S1, groups=eng.G, eps=eng.eps, silu=silu)
        dx = _empty(x.t.shape, x.t)
        ops.gn_bwd_apply(x.t, dyt, dx, x.stats, norm.weight.data, norm.bias.data, sums,
                         norm.net.grad_view(norm.prefix + ".weight"), norm.net.grad_view(norm.prefix + ".bias"),
                         groups=eng.G, eps=eng.eps, dres=dres)
        return dx

    def dgrad_gn_raw(self, dy, x, norm, *, silu, eng):
        """First half of ``dgrad_gn``: conv^T with the GroupNorm(+SiLU) backward reduction in its epilogue.  Returns
        (g = dA * act'(GN(x)), finalized sums) WITHOUT the pti_gn_bwd_apply pass -- for a consumer that applies it in its
        loader (``dgrad_gn_chain``)."""
        n, ho, wo, _ = dy.shape
        g = _empty((n, ho, wo, self.cin), dy)
        sums = eng.zeros(n * self.cin * 2)
        ops.conv_mfma_gnbwd(dy, self.wpt, x.t, x.stats, norm.weight.data, norm.bias.data, g, sums, cout=self.cin,
                            ksize=self.ksize, mode=PTI_CONV_S1, groups=eng.G, eps=eng.eps, silu=silu)
        return g, sums

    def dgrad_gn_chain(self, g_in, x_in, norm_in, sums_in, x, norm, *, silu, dres, eng):
        """``dgrad_gn`` whose input gradient arrives un-applied: (g_in, sums_in) from ``dgrad_gn_raw`` of the conv above,
        x_in / norm_in that conv's GroupNorm input and parameters.  The loader of this launch computes the GroupNorm
        backward on the way in and writes it out once for the weight gradient.  Returns (dx, d x_in)."""
        n, ho, wo, _ = g_in.shape
        dxin = _empty(g_in.shape, g_in)
        dyt = _empty((n, ho, wo, self.cin), g_in)
        sums = eng.zeros(n * self.cin * 2)
        ops.conv_mfma_gnbwd_chain(g_in, x_in.t, x_in.stats, norm_in.weight.data, sums_in, dxin, self.wpt, x.t, x.stats,
                                  norm.weight.data, norm.bias.data, dyt, sums, cout=self.cin, groups=eng.G, eps=eng.eps,
                                  silu=silu, in_dgamma=norm_in.net.grad_view(norm_in.prefix + ".weight"),
                                  in_dbeta=norm_in.net.grad_view(norm_in.prefix + ".bias"))
        dx = _empty(x.t.shape, x.t)
        ops.gn_bwd_apply(x.t, dyt, dx, x.stats, norm.weight.data, norm.bias.data, sums,
                         norm.net.grad_view(norm.prefix + ".weight"), norm.net.grad_view(norm.prefix + ".bias"),
                         groups=eng.G, eps=eng.eps, dres=dres)
        return dx, dxin

    def wgrad(self, x, dy, *, pro=PTI_PRO_NONE, norm=None, eng=None):
        dw, db = self.grads()
        if eng.batch_wgrad and ops.wgrad_batch_eligible(x.t, dy, self.ksize, self.mode, pro):
            eng.defer_wgrad(x.t, dy, dw, db)     # launched with other layers' weight gradients (Engine.flush_wgrad)
            return
        g, b = (norm.weight.data, norm.bias.data) if norm is not None else (None, None)
        ws = eng.wgrad_stream
        if ws is None:
            ops.conv_wgrad_mfma(x.t, dy, dw, db, ksize=self.ksize, mode=self.mode, prologue=pro,
                                in_stats=x.stats if pro else None, gamma=g, beta=b, groups=eng.G, eps=eng.eps,
                                accumulate=True, workspace=eng.workspace)
            return
        # weight gradients are off the data-gradient chain: they run on a side stream, behind an event that marks
        # "x and dy exist", and are joined before their gradients are used (Engine.join_wgrad)
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        ws.wait_event(ev)
        x.t.record_stream(ws)
        dy.record_stream(ws)
        with torch.cuda.stream(ws):
            ops.conv_wgrad_mfma(x.t, dy, dw, db, ksize=self.ksize, mode=self.mode, prologue=pro,
                                in_stats=x.stats if pro else None, gamma=g, beta=b, groups=eng.G, eps=eng.eps,
                                accumulate=True, workspace=eng.workspace_side)
        eng._wgrad_pending = True


class _Norm:
    def __init__(self, net, prefix):
        self.net, self.prefix = net, prefix
        self.weight = net._param_by_name[prefix + ".weight"]
        self.bias = net._param_by_name[prefix + ".bias"]

    def bwd(self, x, da, *, silu, dres, eng):
        n, h, w, c = x.t.shape
        dx = _empty(x.t.shape, x.t)
        sums = eng.zeros(n * c * 2)
        ops.gn_bwd(x.t, da, dx, x.stats, self.weight.data, self.bias.data, sums,
                   self.net.grad_view(self.prefix + ".weight"), self.net.grad_view(self.prefix + ".bias"),
                   groups=eng.G, eps=eng.eps, silu=silu, dres=dres)
        return dx


class _ResBlock:
    def __init__(self, net, p, blk):
        self.norm1, self.norm2 = _Norm(net, p + ".norm1"), _Norm(net, p + ".norm2")
        self.conv1 = _MfmaConv(net, p + ".conv1.conv.weight", p + ".conv1.conv.bias", 3, PTI_CONV_S1)
        self.conv2 = _MfmaConv(net, p + ".conv2.conv.weight", p + ".conv2.conv.bias", 3, PTI_CONV_S1)
        self.nin = None
        if blk.in_channels != blk.out_channels:
            self.nin = _MfmaConv(net, p + ".nin_shortcut.conv.weight", p + ".nin_shortcut.conv.bias", 1, PTI_CONV_S1)
        self.convs = [c for c in (self.conv1, self.conv2, self.nin) if c is not None]
        self.needs_in_stats = True
        self.prefix = p + "."

    def fwd(self, x, eng, want_stats, save):
        # training: the convs also write their activated inputs SiLU(GN(.)) (what autograd saves for the weight
        # gradient), so that the weight-gradient kernels do not redo the normalisation + SiLU on every halo tile
        a1 = a2 = None
        if save is not None and eng.saves_activated_input(x.t):
            a1 = _empty(x.t.shape, x.t)
        h1 = self.conv1.fwd(x, pro=PTI_PRO_GN_SILU, norm=self.norm1, want_stats=True, eng=eng, act_out=a1)
        sc = x.t if self.nin is None else self.nin.fwd(x, eng=eng).t
        if save is not None and eng.saves_activated_input(h1.t):
            a2 = _empty(h1.t.shape, h1.t)
        out = self.conv2.fwd(h1, pro=PTI_PRO_GN_SILU, norm=self.norm2, residual=sc, want_stats=want_stats, eng=eng,
                             act_out=a2)
        if save is not None:
            save.append((x, h1, a1, a2))
        return out

    def bwd(self, dout, saved, eng):
        x, h1, a1, a2 = saved
        if a2 is not None:
            self.conv2.wgrad(_Act(a2), dout, eng=eng)
        else:
            self.conv2.wgrad(h1, dout, pro=PTI_PRO_GN_SILU, norm=self.norm2, eng=eng)
        if eng.gnbwd_chain and ops.gnbwd_chain_supported(self.conv1.cout, self.conv1.cin, 3, x.t.dtype, eng.G):
            # the GroupNorm backward between the two convs is applied inside conv1's data-gradient launch: no
            # pti_gn_bwd_apply for norm2 (its affine gradients ride the finalize launch of conv1's data gradient)
            g2, sums2 = self.conv2.dgrad_gn_raw(dout, h1, self.norm2, silu=True, eng=eng)
            if self.nin is None:
                dres = dout
            else:
                dres = self.nin.dgrad(dout)
                self.nin.wgrad(x, dout, eng=eng)
            dx, dh1 = self.conv1.dgrad_gn_chain(g2, h1, self.norm2, sums2, x, self.norm1, silu=True, dres=dres, eng=eng)
            if a1 is not None:
                self.conv1.wgrad(_Act(a1), dh1, eng=eng)
            else:
                self.conv1.wgrad(x, dh1, pro=PTI_PRO_GN_SILU, norm=self.norm1, eng=eng)
            return dx
        dh1 = self.conv2.dgrad_gn(dout, h1, self.norm2, silu=True, dres=None, eng=eng)
        if a1 is not None:
            self.conv1.wgrad(_Act(a1), dh1, eng=eng)
        else:
            self.conv1.wgrad(x, dh1, pro=PTI_PRO_GN_SILU, norm=self.norm1, eng=eng)
        if self.nin is None:
            dres = dout
        else:
            dres = self.nin.dgrad(dout)
            self.nin.wgrad(x, dout, eng=eng)
        return self.conv1.dgrad_gn(dh1, x, self.norm1, silu=True, dres=dres, eng=eng)


class _Resample:
    """AEKLDownsample (pad (0,1,0,1) + 3x3 stride 2) or Upsample (nearest 2x + 3x3)."""

    def __init__(self, net, wprefix, mode, prefix):
        self.conv = _MfmaConv(net, wprefix + ".weight", wprefix + ".bias", 3, mode)
        self.convs = [self.conv]
        self.needs_in_stats = False
        self.prefix = prefix

    def fwd(self, x, eng, want_stats, save):
        out = self.conv.fwd(x, want_stats=want_stats, eng=eng)
        if save is not None:
            save.append(x)
        return out

    def bwd(self, dout, saved, eng):
        x = saved
        self.conv.wgrad(x, dout, eng=eng)
        return self.conv.dgrad(dout)


class _Attention:
    def __init__(self, net, p):
        self.norm = _Norm(net, p + ".norm")
        a = p + ".attn."
        self.qkv = _MfmaConv(net, None, None, 1, PTI_CONV_S1,
                             fused_names=[(a + "to_q.weight", a + "to_q.bias")], fwd_f16=False)
        self.proj = _MfmaConv(net, a + "out_proj.weight", a + "out_proj.bias", 1, PTI_CONV_S1, fwd_f16=False)
        self.convs = [self.qkv, self.proj]
        self.needs_in_stats = True
        self.prefix = p + "."

    def fwd(self, x, eng, want_stats, save):
        n, h, w, c = x.t.shape
        qkv = self.qkv.fwd(x, pro=PTI_PRO_GN, norm=self.norm, eng=eng, out_dtype=BF16).t   # MFMA operand of attention
        o = _empty((n, h * w, c), x.t)
        lse = _empty((n, h * w), x.t, F32)
        ops.attention_fwd(qkv.view(n, h * w, 3 * c), o, lse)
        out = self.proj.fwd(_Act(o.view(n, h, w, c)), residual=x.t, want_stats=want_stats, eng=eng)
        if save is not None:
            save.append((x, qkv, o, lse))
        return out

    def bwd(self, dout, saved, eng):
        x, qkv, o, lse = saved
        n, h, w, c = x.t.shape
        do = self.proj.dgrad(dout)
        self.proj.wgrad(_Act(o.view(n, h, w, c)), dout, eng=eng)
        dqkv = _empty(qkv.shape, qkv)
        delta = _empty((n, h * w), qkv, F32)
        ops.attention_bwd(qkv.view(n, h * w, 3 * c), o, do.view(n, h * w, c), lse, delta, dqkv.view(n, h * w, 3 * c))
        self.qkv.wgrad(x, dqkv, pro=PTI_PRO_GN, norm=self.norm, eng=eng)
        return self.qkv.dgrad_gn(dqkv, x, self.norm, silu=False, dres=dout, eng=eng)


class _DirectConv:
    """Degenerate-channel 3x3 conv (conv_in / conv_out of Encoder and Decoder)."""

    def __init__(self, net, prefix, norm_prefix=None):
        self.net, self.prefix = net, prefix
        self.w = net._param_by_name[prefix + ".weight"]
        self.b = net._param_by_name[prefix + ".bias"]
        self.cout, self.cin = self.w.shape[0], self.w.shape[1]
        self.norm = _Norm(net, norm_prefix) if norm_prefix else None
        self.few_cin = self.cin <= 16
        self.w_tck = self.w_tck_t = None
        self.convs = []
        self.needs_in_stats = self.norm is not None
        # The launch that PRODUCES the narrow side (conv_out forward: wide -> narrow; conv_in data gradient: wide ->
        # narrow) is a reduction over the wide channels per output.  The direct kernel does it with VALU dot products
        # reduced across lanes, which is fine for 1-4 narrow channels and collapses for 16 (AR config: 256 -> 16 at 64^2
        # took 2.7 ms per launch, 25 % of the step; config A: 128 -> 4 at 32^2 took 70 us for an 8 MB map).  For 4..32 narrow
        # channels it goes to the MFMA conv instead, with
        # the narrow channels zero-padded to one 32-wide MFMA tile (`wpad` = fp32 master copy with the padding, `wp_mfma`
        # its packed operand; Engine.refresh_weights keeps both current) and the result sliced back.
        narrow, wide = min(self.cin, self.cout), max(self.cin, self.cout)
        self.mfma_narrow = 4 <= narrow <= 32 and wide % 32 == 0
        self.wpad = self.wp_mfma = None
        self.pack_f16 = False
        # IMAGE side (encoder conv_in, decoder conv_out) with 2..8 image channels: both the forward launch and the
        # gradients run on the MFMA kernels with the image channels zero-padded to one 32-wide tile (csrc/narrow_pad.hip
        # is the boundary with the [N,C,H,W] fp32 images).  One image channel -- every shipped config -- stays on the
        # direct kernels, which do it near their byte floors; at three channels those cost 1.65 ms more per step.
        # Engine sets ``img_mfma`` (it knows which two convs face the image).
        self.img_mfma = False
        self.wp_img = self.wpt_img = self.wpad_img = self.bpad_img = None

    def alloc(self):
        """Buffers of the derived operands, allocated ONCE: captured HIP graphs (inference encode / decode, the training
        step) hold these addresses, so a re-pack must never move them (fresh tensors per re-pack left replayed
        graphs reading freed memory after a native optimiser step)."""
        w = self.w.data
        if self.w_tck is not None:
            return
        self.w_tck = torch.empty(9, self.cin, self.cout, dtype=F32, device=w.device)
        self.w_tck_t = torch.empty(9, self.cout, self.cin, dtype=F32, device=w.device)
        if self.mfma_narrow:   # narrow channels padded to 32: [32, wide, 3, 3] (conv_out) or [wide, 32, 3, 3] (conv_in)
            shape = (32, self.cin, 3, 3) if self.cout < self.cin else (self.cout, 32, 3, 3)
            self.wpad = torch.zeros(shape, dtype=F32, device=w.device)
            self.bpad = torch.zeros(32, dtype=F32, device=w.device)
        if self.img_mfma:
            shape = (self.cout, 32, 3, 3) if self.cin < self.cout else (32, self.cin, 3, 3)
            self.wpad_img = torch.zeros(shape, dtype=F32, device=w.device)
            self.bpad_img = torch.zeros(32, dtype=F32, device=w.device)

    def repack_entries(self):
        """Entries of the one-launch re-pack (ops.DirectRepack): w -> w_tck, its tap-reversed transpose (the
        data-gradient operand w'[tap'][co][ci] = w[co][ci][8 - tap']) and the zero-padded master copies."""
        w, b = self.w.data, self.b.data
        out = [dict(w=w, b=b, w_tck=self.w_tck, w_tck_t=self.w_tck_t,
                    wpad=self.wpad, bpad=self.bpad if (self.mfma_narrow and self.cout < self.cin) else None)]
        if self.img_mfma:
            out.append(dict(w=w, b=b, wpad=self.wpad_img, bpad=self.bpad_img if self.cout < self.cin else None))
        return out

    def pack_entries(self):
        """(attribute, BatchedPacker entry) of the MFMA operands packed from the zero-padded master copies."""
        out = []
        if self.mfma_narrow:
            if self.cout < self.cin:    # forward operand of conv_out (fp16 when the forward pass runs fp16 operands)
                out.append(("wp_mfma", (self.wpad, 3, PTI_CONV_S1, False, self.pack_f16)))
            else:                       # data-gradient operand of conv_in: W'[ci(pad 32)][co]
                out.append(("wp_mfma", (self.wpad, 3, PTI_CONV_S1, True, False)))
        if self.img_mfma:
            out.append(("wp_img", (self.wpad_img, 3, PTI_CONV_S1, False, self.pack_f16)))
            if self.cout < self.cin:    # conv_out: its data-gradient operand as well
                out.append(("wpt_img", (self.wpad_img, 3, PTI_CONV_S1, True, False)))
        return out


class _Plan:
    """Shared forward/backward driver over a block list."""

    def __init__(self, net, eng):
        self.net, self.eng = net, eng
        self.layers = []

    def all_convs(self):
        out = []
        for l in self.layers:
            out += l.convs
        return out


# =================================================================================================
# engine
# =================================================================================================
class Engine:
    def __init__(self, net: "M.AutoencoderKL"):
        self.net = net
        self.G, self.eps = net.norm_num_groups, float(net.norm_eps)
        self.dev = net.param_arena.device
        self.workspace = ops.wgrad_workspace(self.dev)
        self.packed_version = -1
        self.pack_gen = 0            # number of re-packs so far (monotonic; tests and diagnostics)
        self.grad_ready_cb = None
        self._range_cache = {}
        self._zpool, self._zoff, self._zpool_size = None, 0, 1 << 16
        self._packer = None
        self.save_act_min_hw = int(os.environ.get("PTI_SAVE_ACT_MIN_HW", "0"))
        # storage format of the forward activations (residual stream, conv outputs): fp16 = same bytes as bf16 with
        # an 8x finer rounding step (GroupNorm keeps the range far inside fp16's); MFMA operands, saved activated
        # inputs, attention tensors and every gradient are bf16 either way.  PTI_FWD_ACT_DTYPE=bf16 restores bf16.
        self.act_dtype = {"fp16": torch.float16, "bf16": BF16}[os.environ.get("PTI_FWD_ACT_DTYPE", "fp16")]
        # MFMA weight-gradient launches go to a side stream with its own split-K workspace (PTI_WGRAD_STREAM=0: same
        # stream): they are off the data-gradient chain, and their ramp-up / drain overlaps it (-3.4 % per step)
        self.wgrad_stream = torch.cuda.Stream(device=self.dev) if os.environ.get("PTI_WGRAD_STREAM", "1") == "1" else None
        self.workspace_side = torch.empty_like(self.workspace) if self.wgrad_stream is not None else None
        self._wgrad_pending = False
        # weight gradients of the plain 3x3 convs are collected while backward walks the layers and launched a batch
        # at a time (one launch has ~11 us of fixed cost against 10-45 us of streaming per layer, and the layers'
        # weight gradients are independent): PTI_WGRAD_BATCH = jobs per launch (default 16 = the C-ABI's maximum; 1 = off).
        # With a gradient exchange attached (data parallel) "gradients ready" notifications wait for the batch that
        # holds their layers (see _ready / flush_wgrad).
        self.wgrad_batch_max = max(1, min(ops.L.WGRAD_BATCH_MAX, int(os.environ.get("PTI_WGRAD_BATCH", "16"))))
        self.batch_wgrad = self.wgrad_batch_max > 1
        self._wgrad_jobs, self._ready_queue = [], []
        # inference encodes replay a HIP graph per input shape (see _encode_graphed)
        self.encode_graphs = os.environ.get("PTI_ENCODE_GRAPH", "1") == "1"
        self._enc_graphs, self._dec_graphs, self._enc_graph_version = {}, {}, None
        ops.L.lib()  # fail loudly now if the HIP extension is missing
        for c in net.channels:
            if c % 32:
                raise ValueError(f"HIP path needs channel counts that are multiples of 32, got {net.channels}")
            cpg = c // self.G
            if cpg not in (2, 4, 8, 16, 32) or (c & (c - 1)):
                raise ValueError(f"HIP GroupNorm path supports power-of-two channels with 2..32 channels per group; "
                                 f"got channels={c}, groups={self.G}")
        # ---- encoder plan ----
        eb = net.encoder.blocks
        self.enc_in = _DirectConv(net, "encoder.blocks.0.conv")
        self.enc_layers = []
        last = len(eb) - 1
        for i in range(1, last - 1):
            self.enc_layers.append(self._make_layer(eb[i], f"encoder.blocks.{i}"))
        self.enc_out = _DirectConv(net, f"encoder.blocks.{last}.conv", norm_prefix=f"encoder.blocks.{last - 1}")
        # ---- decoder plan ----
        db = net.decoder.blocks
        self.dec_in = _DirectConv(net, "decoder.blocks.0.conv")
        self.dec_layers = []
        last = len(db) - 1
        for i in range(1, last - 1):
            self.dec_layers.append(self._make_layer(db[i], f"decoder.blocks.{i}"))
        self.dec_out = _DirectConv(net, f"decoder.blocks.{last}.conv", norm_prefix=f"decoder.blocks.{last - 1}")
        self.mfma_convs = [c for l in self.enc_layers + self.dec_layers for c in l.convs]
        for c in self.mfma_convs:
            c.f16 = c.fwd_f16 and self.act_dtype == torch.float16
        self.direct_convs = [self.enc_in, self.enc_out, self.dec_in, self.dec_out]
        self.enc_out.pack_f16 = self.act_dtype == torch.float16     # forward launch: fp16 operands like every forward conv
        # 2..8 image channels: conv_in / conv_out on the MFMA kernels (see _DirectConv.img_mfma); PTI_IMG_MFMA=0 keeps the
        # direct kernels
        img = os.environ.get("PTI_IMG_MFMA", "1") == "1"
        self.enc_in.img_mfma = img and 2 <= net.in_channels <= 8 and net.channels[0] % 32 == 0
        self.dec_out.img_mfma = img and 2 <= net.out_channels <= 8 and net.channels[0] % 32 == 0
        self.enc_in.pack_f16 = self.dec_out.pack_f16 = self.act_dtype == torch.float16
        self._wgrad_posts = []
        self._flush_up = int(os.environ.get("PTI_WGRAD_FLUSH_UP", "1"))
        # GroupNorm backward of a ResBlock's second norm applied inside the first conv's data-gradient launch (128-wide
        # tiles; csrc/conv_mfma.hip PRO_GNB) instead of a pti_gn_bwd_apply pass.  OFF by default (PTI_GNBWD_CHAIN=1): built
        # and measured in round 3 -- it removes 11 launches and 0.2 ms of apply time per step on config A, the chained
        # convs (VALU-bound) get 0.15 ms slower, and the step ends up +0.5..0.9 % SLOWER (same box, interleaved); AR model
        # -0.2 %.  See DESIGN.md section 4, round 3 (xi).
        self.gnbwd_chain = os.environ.get("PTI_GNBWD_CHAIN", "0") == "1"
        self.Lc = net.latent_channels
        self._plist = list(net._param_by_name.values())

    def saves_activated_input(self, t) -> bool:
        """Whether a ResBlock conv on input ``t`` (NHWC) also writes SiLU(GN(t)) for its weight-gradient pass.
        Pays where the weight gradient is bound by the prologue's VALU work (large maps); costs one extra bf16
        copy of ``t`` kept until backward.  PTI_SAVE_ACT_MIN_HW overrides the pixel threshold (0 = always)."""
        return t.shape[1] * t.shape[2] >= self.save_act_min_hw

    def _make_layer(self, blk, prefix):
        if isinstance(blk, M.AEKLResBlock):
            return _ResBlock(self.net, prefix, blk)
        if isinstance(blk, M.AEKLDownsample):
            return _Resample(self.net, prefix + ".conv.conv", PTI_CONV_S2PAD, prefix + ".")
        if isinstance(blk, M.Upsample):
            return _Resample(self.net, prefix + ".postconv.conv", PTI_CONV_UP2, prefix + ".")
        if isinstance(blk, M.SpatialAttentionBlock):
            return _Attention(self.net, prefix)
        raise TypeError(f"unsupported block {type(blk)} at {prefix}")

    # ---- helpers -------------------------------------------------------------------------------
    def new_stats(self, n):
        """Zeroed GroupNorm statistics [n, G, 2]: fixed-point int64 sums (see pti_common.h), carved from the fp32 pool."""
        return self.zeros(n * self.G * 4).view(torch.int64).view(n, self.G, 2)

    def zeros(self, count):
        """Zero-initialised fp32 scratch carved from one pool per pass (one memset instead of ~100 fills)."""
        count = (count + 3) // 4 * 4
        pool = self._zpool
        if pool is None or self._zoff + count > pool.numel():
            pool = self._zpool = torch.zeros(max(count, self._zpool_size), dtype=F32, device=self.dev)
            self._zoff = 0
        out = pool[self._zoff:self._zoff + count]
        self._zoff += count
        return out

    def begin_pass(self, batch, backward=False):
        cmax = max(self.net.channels)
        self._zpool_size = (64 * batch * cmax * 2) if backward else (96 * batch * self.G * 2)
        self._zpool = None

    def _pack_buffer_ids(self):
        """Addresses of every derived weight operand a captured inference graph reads."""
        ids = [t.data_ptr() for c in self.mfma_convs for t in (c.wp, c.wpt) if t is not None]
        for c in self.direct_convs:
            ids += [t.data_ptr() for t in (c.w_tck, c.w_tck_t, c.wp_mfma, c.wpad, c.wp_img, c.wpt_img, c.wpad_img) if t is not None]
        return tuple(ids)

    def _drop_stale_graphs(self):
        """Inference graphs stay valid across re-packs because every pack writes in place (``pack_gen`` counts the
        re-packs; the replays see the new contents at the old addresses).  They are dropped only if an operand buffer
        was (re)allocated since the capture -- keyed on the buffers' addresses, not on the parameters' version sum,
        which a native optimiser step leaves unchanged."""
        ids = self._pack_buffer_ids()
        if ids != self._enc_graph_version:
            self._enc_graphs.clear()
            self._dec_graphs.clear()
            self._enc_graph_version = ids

    def refresh_weights(self):
        """Re-derive the bf16 MFMA-packed / transposed operands when the fp32 masters changed.  In-place
        updates through the nn.Parameters (torch optimisers, load_state_dict) bump the parameters' own
        version counters; kernels that write the arena directly call ``net.mark_weights_dirty()``."""
        v = sum(p._version for p in self._plist)
        if v == self.packed_version:
            return
        if self._packer is None:
            entries = []
            for c in self.mfma_convs:
                dmode = PTI_CONV_ZINS if c.mode == PTI_CONV_S2PAD else PTI_CONV_S1
                entries += [(c._w(), c.ksize, c.mode, False, c.f16), (c._w(), c.ksize, dmode, True, False)]
            # the degenerate-channel convs: one launch for their fp32 operands / zero-padded master copies
            # (ops.DirectRepack), and their MFMA operands ride the batched pack launch below
            rep, targets = [], []
            for c in self.direct_convs:
                c.alloc()
                rep += c.repack_entries()
                for attr, e in c.pack_entries():
                    targets.append((c, attr, len(entries)))
                    entries.append(e)
            self._direct_repack = ops.DirectRepack(rep)
            self._packer = ops.BatchedPacker(entries, self.dev)
            for i, c in enumerate(self.mfma_convs):
                c.wp, c.wpt = self._packer.outputs[2 * i], self._packer.outputs[2 * i + 1]
            for c, attr, i in targets:
                setattr(c, attr, self._packer.outputs[i])
        self._direct_repack.run()     # (before the pack launch: it fills the padded copies that launch reads)
        self._packer.run()
        self.packed_version = v
        self.pack_gen += 1

    def _qp(self, name):
        L = self.Lc
        w = self.net._param_by_name[name + ".conv.weight"].data.view(L, L)
        b = self.net._param_by_name[name + ".conv.bias"].data
        return w, b

    def _check_input(self, x, channels, what):
        if not x.is_cuda:
            raise RuntimeError(f"{what}: expected a cuda (HIP) tensor; the VAE hot path has no CPU fallback")
        if x.dim() != 4 or x.shape[1] != channels:
            raise ValueError(f"{what}: expected [B,{channels},H,W], got {tuple(x.shape)}")
        if x.dtype != F32:
            x = x.float()
        return x.contiguous()

    # ---- block-list walkers ---------------------------------------------------------------------
    def _walk_fwd(self, layers, act, tail_needs_stats, save):
        for i, l in enumerate(layers):
            nxt_needs = layers[i + 1].needs_in_stats if i + 1 < len(layers) else tail_needs_stats
            act = l.fwd(act, self, nxt_needs, save)
        return act

    def _walk_bwd(self, layers, dout, saved):
        for l, s in zip(reversed(layers), reversed(saved)):
            dout = l.bwd(dout, s, self)
            self._ready(l.prefix)
        return dout

    def _wgrad_direct(self, wide, narrow, *args, **kw):
        """ops.wgrad_direct on the weight-gradient side stream (same protocol as _MfmaConv.wgrad)."""
        ws = self.wgrad_stream
        if ws is None:
            ops.wgrad_direct(wide, narrow, *args, **kw)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        ws.wait_event(ev)
        wide.record_stream(ws)
        narrow.record_stream(ws)
        with torch.cuda.stream(ws):
            ops.wgrad_direct(wide, narrow, *args, workspace=self.workspace_side, **kw)
        self._wgrad_pending = True

    def side_call(self, fn, *tensors):
        """Run ``fn`` (small launches that only the optimiser step waits for) on the weight-gradient stream, behind
        everything the current stream has issued so far; ``tensors`` are the buffers it reads."""
        ws = self.wgrad_stream
        if ws is None:
            fn()
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        ws.wait_event(ev)
        for t in tensors:
            t.record_stream(ws)
        with torch.cuda.stream(ws):
            fn()
        self._wgrad_pending = True

    def defer_wgrad(self, x, dy, dw, db, post=None):
        """``post``: called right behind the batched launch, on its stream (the image-side convs copy the real rows /
        columns of their zero-padded weight gradient into the gradient arena there)."""
        # The encoder's backward climbs from the smallest maps to the largest: the jobs still queued from the level below are
        # launched when the first job of a LARGER level arrives, instead of waiting for the batch to fill or for the
        # optimiser's join -- without this the final flush carried 128- and 64-channel jobs that had been ready for a
        # millisecond and ran them after the main stream had finished (0.2 ms of the 0.6-ms weight-gradient tail).
        # Same-box A/B: config A -0.3 %, AR model -1.2..1.8 %.  PTI_WGRAD_FLUSH_UP: 0 = off, 1 = when climbing (default),
        # 2 = at every change of map size (config A +0.7 %: the decoder's descent splits well-filled batches)
        if self._flush_up and self._wgrad_jobs:
            new, old = x.shape[1] * x.shape[2], self._wgrad_jobs[-1][0].shape[1] * self._wgrad_jobs[-1][0].shape[2]
            if new > old or (self._flush_up == 2 and new != old):
                self.flush_wgrad()
        self._wgrad_jobs.append((x, dy, dw, db))
        if post is not None:
            self._wgrad_posts.append(post)
        if len(self._wgrad_jobs) >= self.wgrad_batch_max:
            self.flush_wgrad()

    def wgrad_padded(self, x, dy, dw, db, post):
        """Weight gradient of an image-side conv on its zero-padded 32-channel operands (same protocol as
        _MfmaConv.wgrad: batched and on the side stream when those are on)."""
        if self.batch_wgrad and ops.wgrad_batch_eligible(x, dy, 3, PTI_CONV_S1, PTI_PRO_NONE):
            self.defer_wgrad(x, dy, dw, db, post)
            return
        ws = self.wgrad_stream
        if ws is None:
            ops.conv_wgrad_mfma(x, dy, dw, db, accumulate=True, workspace=self.workspace)
            post()
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        ws.wait_event(ev)
        x.record_stream(ws)
        dy.record_stream(ws)
        with torch.cuda.stream(ws):
            ops.conv_wgrad_mfma(x, dy, dw, db, accumulate=True, workspace=self.workspace_side)
            post()
        self._wgrad_pending = True

    def flush_wgrad(self):
        """Launch the collected weight gradients (side stream if there is one), then hand the queued "gradients ready"
        ranges to the exchange -- in their original order, behind that launch."""
        jobs, self._wgrad_jobs = self._wgrad_jobs, []
        posts, self._wgrad_posts = self._wgrad_posts, []
        ws = self.wgrad_stream
        if jobs:
            if ws is None:
                ops.conv_wgrad_mfma_batched(jobs, workspace=self.workspace)
                for p in posts:
                    p()
            else:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                ws.wait_event(ev)
                for x, dy, _, _ in jobs:
                    x.record_stream(ws)
                    dy.record_stream(ws)
                with torch.cuda.stream(ws):
                    ops.conv_wgrad_mfma_batched(jobs, workspace=self.workspace_side)
                    for p in posts:
                        p()
                self._wgrad_pending = True
        queue, self._ready_queue = self._ready_queue, []
        for rng in queue:
            self._emit_ready(rng)

    def join_wgrad(self):
        """Make the current stream wait for the side-stream weight gradients issued so far."""
        self.flush_wgrad()
        if self.wgrad_stream is not None and self._wgrad_pending:
            ev = torch.cuda.Event()
            ev.record(self.wgrad_stream)
            torch.cuda.current_stream().wait_event(ev)
            self._wgrad_pending = False

    def _ready(self, *prefixes):
        """Tell the data-parallel exchange that the gradients of these blocks are final."""
        cb = self.grad_ready_cb
        if cb is None:
            return
        rng = self._range_cache.get(prefixes)
        if rng is None:
            lo, hi = None, None
            for name, (o, n, _) in self.net._slots.items():
                if name.startswith(prefixes):
                    lo = o if lo is None else min(lo, o)
                    hi = o + (n + 3) // 4 * 4 if hi is None else max(hi, o + (n + 3) // 4 * 4)
            rng = self._range_cache[prefixes] = (lo, hi)
        if rng[0] is None:
            return
        if self._wgrad_jobs or self._ready_queue:   # some of this range's weight gradients may still be waiting in the batch
            self._ready_queue.append(rng)
            if len(self._ready_queue) >= 6:         # do not let the exchange fall far behind backward
                self.flush_wgrad()
            return
        self._emit_ready(rng)

    def _emit_ready(self, rng):
        cb = self.grad_ready_cb
        if cb is None:
            return
        ws = self.wgrad_stream
        if ws is None:
            cb(*rng)
            return
        # the block's gradients come from BOTH streams (weight gradients on the side stream, norm / bias / direct-conv
        # gradients on the main one): the side stream waits for the main stream up to here and the exchange is
        # enqueued behind the side stream, so the data-gradient chain on the main stream is never held up
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        ws.wait_event(ev)
        with torch.cuda.stream(ws):
            cb(*rng)
        self._wgrad_pending = True

    # ---- encoder --------------------------------------------------------------------------------
    def encode_forward(self, x, save):
        self.refresh_weights()
        x = self._check_input(x, self.net.in_channels, "encode")
        n, cin, h, w = x.shape
        down = 2 ** (len(self.net.channels) - 1)
        if h % down or w % down:
            raise ValueError(f"encode: H,W must be multiples of {down}, got {h}x{w}")
        self.begin_pass(n)
        c0 = self.net.channels[0]
        t0 = _empty((n, h, w, c0), x, self.act_dtype)
        xpad_b = None
        if self.enc_in.img_mfma:   # image channels zero-padded to 32: MFMA conv with the next GroupNorm's statistics fused
            xpad, xpad_b = ops.pad_nchw_to_nhwc32(x, self.act_dtype, BF16 if (save and self.act_dtype != BF16) else None)
            if save and xpad_b is None:
                xpad_b = xpad
            st0 = self.new_stats(n)
            ops.conv_mfma(xpad, self.enc_in.wp_img, self.enc_in.b.data, t0, cout=c0, ksize=3, out_stats=st0, out_groups=self.G)
            a0 = _Act(t0, st0)
        else:
            ops.conv_direct(x, self.enc_in.w_tck, self.enc_in.b.data, t0, n=n, h=h, w=w, cin=cin, cout=c0, x_layout="nchw")
            a0 = _Act(t0, ops.gn_stats(t0, self.G, self.new_stats(n)))
        saved = [] if save else None
        act = self._walk_fwd(self.enc_layers, a0, True, saved)
        hl, wl, L = act.t.shape[1], act.t.shape[2], self.Lc
        nm = self.enc_out.norm
        if self.enc_out.mfma_narrow:   # latent_channels in 5..32: MFMA conv on the zero-padded 32-channel tile
            ypad = _empty((n, hl, wl, 32), x, self.act_dtype)
            ops.conv_mfma(act.t, self.enc_out.wp_mfma, self.enc_out.bpad, ypad, cout=32, ksize=3, prologue=PTI_PRO_GN,
                          in_stats=act.stats, gamma=nm.weight.data, beta=nm.bias.data, groups=self.G, eps=self.eps)
            hlat = ypad.view(n, hl * wl, 32)[..., :L].float()
        else:
            hlat = _empty((n, hl * wl, L), x, F32)
            ops.conv_direct(act.t, self.enc_out.w_tck, self.enc_out.b.data, hlat.view(n, hl, wl, L), n=n, h=hl, w=wl,
                            cin=act.t.shape[3], cout=L, prologue=PTI_PRO_GN, in_stats=act.stats, gamma=nm.weight.data,
                            beta=nm.bias.data, groups=self.G, eps=self.eps)
        mu = _empty((n, L, hl, wl), x, F32)
        sigma = _empty((n, L, hl, wl), x, F32)
        zq_unused = _empty((n, hl * wl, L), x, F32)
        wm, bm = self._qp("quant_conv_mu")
        wl_, bl = self._qp("quant_conv_log_sigma")
        wp, bp = self._qp("post_quant_conv")
        ops.latent_head_fwd(hlat, None, wm, bm, wl_, bl, wp, bp, mu, sigma, None, zq_unused)
        ctx = (x, a0, saved, act, hlat, xpad_b) if save else None
        return mu, sigma, ctx

    def encode_backward(self, ctx, dmu, dsigma, want_dx=False):
        x, a0, saved, act, hlat, xpad_b = ctx
        net = self.net
        n, L = hlat.shape[0], self.Lc
        hl, wl = act.t.shape[1], act.t.shape[2]
        wm, bm = self._qp("quant_conv_mu")
        wl_, bl = self._qp("quant_conv_log_sigma")
        wp, bp = self._qp("post_quant_conv")
        self.begin_pass(n, backward=True)
        dh = torch.empty_like(hlat)
        gv = net.grad_view
        scratch = self.zeros(L * L + L)   # post_quant grads are not produced here
        ops.latent_head_bwd(hlat, None, wm, bm, wl_, bl, wp, bp, None,
                            None if dmu is None else dmu.contiguous().float(),
                            None if dsigma is None else dsigma.contiguous().float(), dh,
                            gv("quant_conv_mu.conv.weight"), gv("quant_conv_mu.conv.bias"),
                            gv("quant_conv_log_sigma.conv.weight"), gv("quant_conv_log_sigma.conv.bias"),
                            scratch[:L * L], scratch[L * L:])
        self._ready("quant_conv_mu.", "quant_conv_log_sigma.")
        # encoder conv_out (norm + 3x3, C -> L): weight grads, then data grad through the norm
        eo, nm = self.enc_out, self.enc_out.norm
        C = act.t.shape[3]
        self._wgrad_direct(act.t, dh.view(n, hl, wl, L), gv(eo.prefix + ".weight"), n=n, h=hl, w=wl, cw=C, cn=L, ksize=3,
                         sgn=1, narrow_layout="nhwc", dw_strides=(1, 9, C * 9), dbias_narrow=gv(eo.prefix + ".bias"),
                         prologue=PTI_PRO_GN, in_stats=act.stats, gamma=nm.weight.data, beta=nm.bias.data, groups=self.G,
                         eps=self.eps)
        da = _empty(act.t.shape, x)
        ops.conv_direct(dh.view(n, hl, wl, L), eo.w_tck_t, None, da, n=n, h=hl, w=wl, cin=L, cout=C)
        dout = nm.bwd(act, da, silu=False, dres=None, eng=self)
        self._ready(eo.prefix[:-4], nm.prefix + ".")
        dout = self._walk_bwd(self.enc_layers, dout, saved)
        # encoder conv_in (cin -> C0): weight grads; data grad only on request
        ei = self.enc_in
        _, cin, h, w = x.shape
        c0 = dout.shape[3]
        if ei.img_mfma:   # x = the saved zero-padded image (bf16): dW[c0][32 (pad)] -> its first cin columns
            dwp = torch.zeros(c0, 32, 3, 3, dtype=F32, device=x.device)
            gw = gv(ei.prefix + ".weight")
            self.wgrad_padded(xpad_b, dout, dwp, gv(ei.prefix + ".bias"),
                              lambda: gw.view(c0, cin, 3, 3).add_(dwp[:, :cin]))
        else:
            self._wgrad_direct(dout, x, gv(ei.prefix + ".weight"), n=n, h=h, w=w, cw=c0, cn=cin, ksize=3, sgn=-1,
                               narrow_layout="nchw", dw_strides=(1, cin * 9, 9), dbias_wide=gv(ei.prefix + ".bias"))
        self._ready("encoder.blocks.0.")
        self.join_wgrad()
        if not want_dx:
            return None
        dx = torch.empty_like(x)
        ops.conv_direct(dout, ei.w_tck_t, None, dx, n=n, h=h, w=w, cin=c0, cout=cin, y_layout="nchw")
        return dx

    # ---- decoder --------------------------------------------------------------------------------
    def decode_forward(self, z, save):
        self.refresh_weights()
        z = self._check_input(z, self.Lc, "decode")
        n, L, hl, wl = z.shape
        self.begin_pass(n)
        wp, bp = self._qp("post_quant_conv")
        zq = _empty((n, hl * wl, L), z, F32)
        ops.post_quant(z, wp, bp, zq)
        di = self.dec_in
        t0 = _empty((n, hl, wl, di.cout), z, self.act_dtype)
        ops.conv_direct(zq.view(n, hl, wl, L), di.w_tck, di.b.data, t0, n=n, h=hl, w=wl, cin=L, cout=di.cout)
        a0 = _Act(t0, ops.gn_stats(t0, self.G, self.new_stats(n)))
        saved = [] if save else None
        act = self._walk_fwd(self.dec_layers, a0, True, saved)
        h, w, C = act.t.shape[1], act.t.shape[2], act.t.shape[3]
        do, nm = self.dec_out, self.dec_out.norm
        recon = _empty((n, do.cout, h, w), z, F32)
        gact = None
        if do.img_mfma:   # output channels zero-padded to 32: MFMA conv (GroupNorm in its loader; the normalised input is
            ypad = _empty((n, h, w, 32), z, self.act_dtype)      # kept for the weight gradient), then the image slice
            gact = _empty(act.t.shape, z) if save else None
            ops.conv_mfma(act.t, do.wp_img, do.bpad_img, ypad, cout=32, ksize=3, prologue=PTI_PRO_GN, in_stats=act.stats,
                          gamma=nm.weight.data, beta=nm.bias.data, groups=self.G, eps=self.eps, act_out=gact)
            ops.slice_nhwc32_to_nchw(ypad, do.cout, out=recon)
        else:
            ops.conv_direct(act.t, do.w_tck, do.b.data, recon, n=n, h=h, w=w, cin=C, cout=do.cout, y_layout="nchw",
                            prologue=PTI_PRO_GN, in_stats=act.stats, gamma=nm.weight.data, beta=nm.bias.data, groups=self.G,
                            eps=self.eps)
        ctx = (z, zq, a0, saved, act, gact) if save else None
        return recon, ctx

    def decode_backward(self, ctx, drecon, want_dz=True, join=True):
        """``join=False``: leave the side-stream weight gradients running (the caller goes on to
        ``encode_backward``, whose final join covers them: one in-order side stream)."""
        z, zq, a0, saved, act, gact = ctx
        net, gv = self.net, self.net.grad_view
        n, L, hl, wl = z.shape
        self.begin_pass(n, backward=True)
        drecon = drecon.contiguous().float()
        do, nm = self.dec_out, self.dec_out.norm
        h, w, C = act.t.shape[1], act.t.shape[2], act.t.shape[3]
        co = do.cout
        da = _empty(act.t.shape, z)
        if do.img_mfma:   # d recon zero-padded to 32 channels (bf16): weight gradient dW[32 (pad)][C] -> its first co rows
            dpad, _ = ops.pad_nchw_to_nhwc32(drecon, BF16)
            dwp = torch.zeros(32, C, 3, 3, dtype=F32, device=z.device)
            dbp = torch.zeros(32, dtype=F32, device=z.device)
            gw, gb = gv(do.prefix + ".weight"), gv(do.prefix + ".bias")

            def post():
                gw.view(co, C, 3, 3).add_(dwp[:co])
                gb.add_(dbp[:co])
            self.wgrad_padded(gact, dpad, dwp, dbp, post)
            ops.conv_mfma(dpad, do.wpt_img, None, da, cout=C, ksize=3)
        else:
            self._wgrad_direct(act.t, drecon, gv(do.prefix + ".weight"), n=n, h=h, w=w, cw=C, cn=co, ksize=3, sgn=1,
                               narrow_layout="nchw", dw_strides=(1, 9, C * 9), dbias_narrow=gv(do.prefix + ".bias"),
                               prologue=PTI_PRO_GN, in_stats=act.stats, gamma=nm.weight.data, beta=nm.bias.data,
                               groups=self.G, eps=self.eps)
            ops.conv_direct(drecon, do.w_tck_t, None, da, n=n, h=h, w=w, cin=co, cout=C, x_layout="nchw")
        dout = nm.bwd(act, da, silu=False, dres=None, eng=self)
        self._ready(do.prefix[:-4], nm.prefix + ".")
        dout = self._walk_bwd(self.dec_layers, dout, saved)
        di = self.dec_in
        self._wgrad_direct(dout, zq.view(n, hl, wl, L), gv(di.prefix + ".weight"), n=n, h=hl, w=wl, cw=di.cout, cn=L,
                         ksize=3, sgn=-1, narrow_layout="nhwc", dw_strides=(1, L * 9, 9),
                         dbias_wide=gv(di.prefix + ".bias"))
        self._ready("decoder.blocks.0.")
        if di.mfma_narrow:
            dpad = _empty((n, hl, wl, 32), z)
            ops.conv_mfma(dout, di.wp_mfma, None, dpad, cout=32, ksize=3)
            dzq = dpad.view(n, hl * wl, 32)[..., :L].float()
        else:
            dzq = _empty((n, hl * wl, L), z, F32)
            ops.conv_direct(dout, di.w_tck_t, None, dzq.view(n, hl, wl, L), n=n, h=hl, w=wl, cin=di.cout, cout=L)
        wp, _ = self._qp("post_quant_conv")
        dz = torch.empty_like(z) if want_dz else None
        ops.post_quant_bwd(dzq, z, wp, dz, gv("post_quant_conv.conv.weight"), gv("post_quant_conv.conv.bias"))
        self._ready("post_quant_conv.")
        if join:
            self.join_wgrad()
        return dz

    # ---- autograd entry points --------------------------------------------------------------------
    def _region_params(self, which):
        pre = ("encoder.", "quant_conv_") if which == 0 else ("post_quant_conv.", "decoder.")
        return [(n, p) for n, p in self.net._param_by_name.items() if n.startswith(pre)]

    def encode(self, x):
        params = self._region_params(0)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for _, p in params)):
            return _EncodeFn.apply(self, x, *[p for _, p in params])
        if self.encode_graphs:
            out = self._encode_graphed(x)
            if out is not None:
                return out
        mu, sigma, _ = self.encode_forward(x, save=False)
        return mu, sigma

    def _encode_graphed(self, x):
        """Inference encode (no autograd) replayed from a HIP graph: the encoder forward is ~70 launches, and at the
        batch sizes the regression / inference scripts use (reg_edente_from_dente.json: 8) the host needs longer to
        enqueue them (~1.1 ms) than the GPU to run them.  One graph per input shape (at most four; further shapes run
        eagerly), captured after an eager warm-up; dropped and re-captured when the weights were re-packed (the direct
        convs' operands move).  Outputs are copies of the graph's static buffers.  PTI_ENCODE_GRAPH=0 turns it off."""
        if ops.KERNEL_PROFILE is not None:      # per-kernel timing wants the eager launches
            return None
        x = self._check_input(x, self.net.in_channels, "encode")
        self.refresh_weights()
        self._drop_stale_graphs()
        key = tuple(x.shape)
        ent = self._enc_graphs.get(key)
        if ent is None:
            if len(self._enc_graphs) >= 4:
                return None
            sx = x.clone()
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):           # warm-up outside the capture (allocator, lazy state)
                self.encode_forward(sx, save=False)
            cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):   # loader threads may issue copies meanwhile
                    mu, sigma, _ = self.encode_forward(sx, save=False)
            except Exception as ex:   # eager from here on
                import warnings
                warnings.warn(f"HIP-graph capture of the inference encode failed ({ex!r}); continuing with eager launches")
                self.encode_graphs = False
                torch.cuda.synchronize()
                return None
            ent = self._enc_graphs[key] = (g, sx, mu, sigma)
        g, sx, mu, sigma = ent
        sx.copy_(x)
        g.replay()
        return mu.clone(), sigma.clone()

    def decode(self, z):
        params = self._region_params(1)
        if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for _, p in params)):
            return _DecodeFn.apply(self, z, *[p for _, p in params])
        if self.encode_graphs:
            out = self._decode_graphed(z)
            if out is not None:
                return out
        recon, _ = self.decode_forward(z, save=False)
        return recon

    def _decode_graphed(self, z):
        """Inference decode replayed from a HIP graph per latent shape -- the mirror of ``_encode_graphed``."""
        if ops.KERNEL_PROFILE is not None:
            return None
        z = self._check_input(z, self.Lc, "decode")
        self.refresh_weights()
        self._drop_stale_graphs()
        key = tuple(z.shape)
        ent = self._dec_graphs.get(key)
        if ent is None:
            if len(self._dec_graphs) >= 4:
                return None
            sz = z.clone()
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self.decode_forward(sz, save=False)
            cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):   # loader threads may issue copies meanwhile
                    recon, _ = self.decode_forward(sz, save=False)
            except Exception as ex:   # eager from here on
                import warnings
                warnings.warn(f"HIP-graph capture of the inference decode failed ({ex!r}); continuing with eager launches")
                self.encode_graphs = False
                torch.cuda.synchronize()
                return None
            ent = self._dec_graphs[key] = (g, sz, recon)
        g, sz, recon = ent
        sz.copy_(z)
        g.replay()
        return recon.clone()

    def _prepare_grads(self, which):
        """Zero this region of the gradient arena unless the parameters' .grad already alias it (then
        the kernels keep accumulating in place).  Returns True when autograd must be handed views."""
        net = self.net
        params = self._region_params(which)
        aliased = all(p.grad is not None and p.grad.data_ptr() == net.grad_view(n).data_ptr() for n, p in params
                      if p.requires_grad)
        if aliased and any(p.requires_grad for _, p in params):
            return False
        s, e = net.arena_regions()[which]
        net.grad_arena[s:e].zero_()
        return True

    def _grad_outputs(self, which, hand_views, needs):
        out = []
        for (n, p), need in zip(self._region_params(which), needs):
            out.append(self.net.grad_view(n) if (hand_views and need) else None)
        return out


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, *params):
        mu, sigma, c = eng.encode_forward(x, save=True)
        ctx.eng, ctx.c = eng, c
        return mu, sigma

    @staticmethod
    def backward(ctx, dmu, dsigma):
        eng = ctx.eng
        hand = eng._prepare_grads(0)
        want_dx = ctx.needs_input_grad[1]
        dx = eng.encode_backward(ctx.c, dmu, dsigma, want_dx=want_dx)
        ctx.c = None
        return (None, dx, *eng._grad_outputs(0, hand, ctx.needs_input_grad[2:]))


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, z, *params):
        recon, c = eng.decode_forward(z, save=True)
        ctx.eng, ctx.c = eng, c
        return recon

    @staticmethod
    def backward(ctx, drecon):
        eng = ctx.eng
        hand = eng._prepare_grads(1)
        dz = eng.decode_backward(ctx.c, drecon, want_dz=ctx.needs_input_grad[1])
        ctx.c = None
        return (None, dz, *eng._grad_outputs(1, hand, ctx.needs_input_grad[2:]))
