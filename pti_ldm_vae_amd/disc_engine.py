"""HIP engine of the PatchDiscriminator (``models/patch_discriminator.py``): forward, data gradient and weight gradient
of the five 4x4 convolution blocks of MONAI's PatchDiscriminator(norm="INSTANCE") as the reference runs them in the
adversarial branch of its training step (``vae_scripts/train_vae.py:399-401`` generator term, ``:447-458``
discriminator step, ``:564-571`` validation).

Every block but the last is  patches = im2col(LeakyReLU(InstanceNorm(previous conv output)))  ->  1x1 convolution on the
MFMA kernel  (``ops.conv_mfma`` ksize 1): the pre-normalisation conv outputs ``y`` and the patch matrices ``P`` are what
is kept for backward.  The last block has ONE output channel -- a patch matrix would move ~15x the bytes of its input --
and runs on direct kernels (``ops.pd_final_*``) with fp32 logits.  Backward per block: weight gradient = 1x1 weight gradient of (P, dy) (``ops.conv_wgrad_mfma``),
data gradient = 1x1 conv with the transposed weight -> ``pd_col2im`` (gather + LeakyReLU' + InstanceNorm-backward sums)
-> ``pd_in_bwd_apply``.  All on the caller's current stream; no host synchronisation; no float atomics.
"""
from __future__ import annotations

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
LRELU, IN_EPS = 0.2, 1e-5


class DiscCtx:
    """What one forward pass keeps: patches P[l] (None for the direct final block), conv outputs y[l] (pre-norm; the
    last one = fp32 logits [B,Ho,Wo]), norm tables t[l] (None where the block has no InstanceNorm)."""
    __slots__ = ("P", "y", "t", "shape")

    def __init__(self):
        self.P, self.y, self.t, self.shape = [], [], [], None


class DiscEngine:
    def __init__(self, net):
        self.net = net
        self.dev = net.param_arena.device
        self.layers = net.layers
        self._plist = list(net.parameters())
        self.packed_version = -1
        self._packer = None
        self.wp, self.wpt = [], []
        # its own split-K workspace: the discriminator step may run on its own stream beside the VAE's weight gradients
        self.workspace = torch.empty_like(ops.wgrad_workspace(self.dev)) if self.dev.type == "cuda" else None

    # ---- weights ---------------------------------------------------------------------------------------------------
    def _w(self, lay, arena):
        return arena[lay["w_off"]:lay["w_off"] + lay["rows"] * lay["k"]].view(lay["rows"], lay["k"], 1, 1)

    def _w16c(self, lay, arena):
        """[16][cin] fp32 weight of the one-output-channel final block (row 0 of its padded arena slot)."""
        return arena[lay["w_off"]:lay["w_off"] + lay["k"]]

    def _b(self, lay, arena):
        return None if lay["b_off"] is None else arena[lay["b_off"]:lay["b_off"] + lay["rows"]]

    def refresh_weights(self):
        v = sum(p._version for p in self._plist)
        if v == self.packed_version:
            return
        if self._packer is None:
            entries = []
            for lay in self.layers:
                if lay["cout"] == 1:
                    continue          # direct final block: reads the fp32 master weights
                w = self._w(lay, self.net.param_arena)
                entries += [(w, 1, ops.PTI_CONV_S1, False, False), (w, 1, ops.PTI_CONV_S1, True, False)]
            self._packer = ops.BatchedPacker(entries, self.dev)
            self.wp = self._packer.outputs[0::2] + [None]
            self.wpt = self._packer.outputs[1::2] + [None]
        self._packer.run()
        self.packed_version = v

    # ---- forward ---------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, save: bool = True) -> DiscCtx:
        """x: fp32 [B,1,H,W] on the device (H, W even, >= 32).  Returns the context holding the padded logit rows
        ``ctx.y[-1]`` = bf16 [B,Ho,Wo,32] (column 0 = the logit)."""
        if not x.is_cuda:
            raise RuntimeError("PatchDiscriminator: expected a cuda (HIP) tensor; there is no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"PatchDiscriminator: expected [B,1,H,W], got {tuple(x.shape)}")
        x = x.float().contiguous()
        self.refresh_weights()
        b, _, h, w = x.shape
        ctx = DiscCtx()
        ctx.shape = (b, h, w)
        arena = self.net.param_arena
        prev, prev_t = None, None
        for i, lay in enumerate(self.layers):
            if lay["cout"] == 1:      # final block: direct, fp32 logits
                logits = torch.empty(b, prev.shape[1] - 1, prev.shape[2] - 1, dtype=F32, device=self.dev)
                if logits.shape[1] < 1 or logits.shape[2] < 1:
                    raise ValueError(f"PatchDiscriminator: input {h}x{w} is too small for {len(self.layers)} blocks")
                ops.pd_final_fwd(prev, prev_t, self._w16c(lay, arena), self._b(lay, arena), logits, slope=LRELU)
                ctx.P.append(None)
                ctx.y.append(logits)
                ctx.t.append(None)
                break
            if i == 0:
                P = torch.empty(b, h // 2, w // 2, 32, dtype=BF16, device=self.dev)
                ops.pd_im2col_image(x, P)
            else:
                ph, pw = prev.shape[1], prev.shape[2]
                ho, wo = ops.pd_out_hw(ph, pw, lay["stride"])
                if ho < 1 or wo < 1:
                    raise ValueError(f"PatchDiscriminator: input {h}x{w} is too small for {len(self.layers)} blocks")
                P = torch.empty(b, ho, wo, lay["k"], dtype=BF16, device=self.dev)
                ops.pd_im2col(prev, prev_t, P, stride=lay["stride"], act=True, slope=LRELU)
            y = torch.empty(P.shape[0], P.shape[1], P.shape[2], lay["rows"], dtype=BF16, device=self.dev)
            ops.conv_mfma(P, self.wp[i], self._b(lay, arena), y, cout=lay["rows"], ksize=1)
            t = ops.pd_in_stats(y, IN_EPS) if lay["norm"] else None
            ctx.P.append(P if save else None)
            ctx.y.append(y if (save or i == len(self.layers) - 1) else None)
            ctx.t.append(t)
            prev, prev_t = y, t
        return ctx

    @staticmethod
    def logit_rows(ctx: DiscCtx) -> torch.Tensor:
        return ctx.y[-1].view(-1, 1)

    def logits(self, ctx: DiscCtx) -> torch.Tensor:
        """fp32 [B,1,Ho,Wo] like the reference's ``discriminator(x)[-1]``."""
        return ctx.y[-1].unsqueeze(1)

    def lsgan(self, ctx: DiscCtx, *, target_is_real: bool, weight: float = 1.0, want_grad: bool = True, slope: float = 0.05):
        """PatchAdversarialLoss("least_squares") of this pass's logits -> (loss [1] fp32 device tensor, unweighted;
        d(weight * loss)/d logits as fp32 [M,1] or None)."""
        rows = self.logit_rows(ctx)
        d = torch.empty_like(rows) if want_grad else None
        loss = ops.pd_lsgan(rows, target=1.0 if target_is_real else 0.0, slope=slope, grad_scale=2.0 * weight / rows.shape[0],
                            d_logits=d)
        return loss, d

    # ---- backward --------------------------------------------------------------------------------------------------
    def backward(self, ctx: DiscCtx, d_rows: torch.Tensor, *, want_wgrad: bool, d_img: torch.Tensor | None = None,
                 dx_scale: float = 1.0, accumulate_dx: bool = False):
        """d_rows: fp32 [B*Ho*Wo, 1] gradient w.r.t. the logits.  ``want_wgrad``: accumulate (+=) the
        parameter gradients into ``net.grad_arena``.  ``d_img`` (fp32 [B,1,H,W]): receives dx_scale * gradient w.r.t. the
        input image (added to its contents when ``accumulate_dx``)."""
        garena = self.net.grad_arena
        dy = None
        for i in range(len(self.layers) - 1, -1, -1):
            lay, P = self.layers[i], ctx.P[i]
            if lay["cout"] == 1:      # direct final block
                yp, tp = ctx.y[i - 1], ctx.t[i - 1]
                if yp is None:
                    raise RuntimeError("PatchDiscriminator.backward: the forward pass was run with save=False")
                dl = d_rows.reshape(-1).float().contiguous()
                if want_wgrad:
                    gw = ops.pd_final_wgrad(dl, yp, tp, slope=LRELU)
                    self._w16c(lay, garena).add_(gw[:lay["k"]])
                    self._b(lay, garena)[:1].add_(gw[lay["k"]:lay["k"] + 1])
                g = torch.empty_like(yp)
                g, sums = ops.pd_final_dgrad(dl, yp, tp, self._w16c(lay, self.net.param_arena), g, slope=LRELU)
                dy = ops.pd_in_bwd_apply(g, yp, tp, sums) if tp is not None else g
                continue
            if P is None:
                raise RuntimeError("PatchDiscriminator.backward: the forward pass was run with save=False")
            if want_wgrad:
                ops.conv_wgrad_mfma(P, dy, self._w(lay, garena), self._b(lay, garena), ksize=1, accumulate=True,
                                    workspace=self.workspace)
            if i == 0:
                if d_img is not None:
                    dP = torch.empty_like(P)
                    ops.conv_mfma(dy, self.wpt[0], None, dP, cout=lay["k"], ksize=1)
                    ops.pd_col2im_image(dP, d_img, scale=dx_scale, accumulate=accumulate_dx)
                break
            dP = torch.empty_like(P)
            ops.conv_mfma(dy, self.wpt[i], None, dP, cout=lay["k"], ksize=1)
            yp, tp = ctx.y[i - 1], ctx.t[i - 1]
            g = torch.empty_like(yp)
            g, sums = ops.pd_col2im(dP, yp, tp, g, stride=lay["stride"], slope=LRELU)
            dy = ops.pd_in_bwd_apply(g, yp, tp, sums) if tp is not None else g
        return d_img

    # ---- drop-in autograd path -----------------------------------------------------------------------------------------
    def apply(self, x, return_intermediates=False):
        params = list(self.net._param_by_name.values())
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            logits, ctx = _DiscFn.apply(self, x, *params), None
        else:
            ctx = self.forward(x, save=return_intermediates)
            logits = self.logits(ctx)
        if not return_intermediates:
            return [logits]
        if ctx is None:
            with torch.no_grad():
                ctx = self.forward(x, save=True)
        outs = []
        for y, t, lay in zip(ctx.y[:-1], ctx.t[:-1], self.layers[:-1]):
            a = y[..., :lay["cout"]].float()
            if t is not None:
                a = (a - t[:, None, None, :, 0]) * t[:, None, None, :, 1]
            outs.append(torch.nn.functional.leaky_relu(a, LRELU).permute(0, 3, 1, 2).contiguous())
        return outs + [logits]


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, *params):
        c = eng.forward(x, save=True)
        ctx.eng, ctx.c, ctx.xshape = eng, c, x.shape
        return eng.logits(c)

    @staticmethod
    def backward(ctx, dlogits):
        eng, c = ctx.eng, ctx.c
        net = eng.net
        rows = eng.logit_rows(c)
        d = dlogits.reshape(-1, 1).float().contiguous()
        names = list(net._param_by_name)
        need_w = any(ctx.needs_input_grad[2:])
        aliased = all(p.grad is not None and p.grad.data_ptr() == net.grad_view(n).data_ptr()
                      for n, p in net._param_by_name.items())
        if need_w and not aliased:
            net.grad_arena.zero_()
        d_img = torch.empty(ctx.xshape, dtype=F32, device=rows.device) if ctx.needs_input_grad[1] else None
        eng.backward(c, d, want_wgrad=need_w, d_img=d_img)
        ctx.c = None
        # copies, not views: two of these nodes may sit in one graph (fake + real pass) and the second one clears the arena
        grads = [net.grad_view(n).clone(memory_format=torch.contiguous_format) if (need and not aliased) else None
                 for n, need in zip(names, ctx.needs_input_grad[2:])]
        return (None, d_img, *grads)
