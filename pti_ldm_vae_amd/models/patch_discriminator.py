"""``PatchDiscriminator`` — drop-in for ``monai.networks.nets.PatchDiscriminator`` as the reference builds it
(``vae_scripts/train_vae.py:266-279``: 2-D, ``num_layers_d=3``, ``channels=32``, ``in_channels=1``, ``out_channels=1``,
``norm="INSTANCE"``), running on the HIP engine of ``disc_engine.py``.

Parameter holder with MONAI's key names (``initial_conv.conv.{weight,bias}``, ``{0..}.conv.weight``,
``final_conv.conv.{weight,bias}``), so ``discriminator_*.pt`` / ``discriminator_state_dict`` files are interchangeable
with the reference's (train_vae.py:696-698,741-758).  All parameters are views of ONE flat fp32 arena laid out the way
the kernels read it: a 4x4 convolution is a 1x1 convolution over 16*Cin patch columns in (ky, kx, c) order, so a weight
lives as ``[cout][ky][kx][cin]`` and the ``nn.Parameter`` is its ``permute(0, 3, 1, 2)`` view (logical ``[cout, cin, 4,
4]``).  The first layer's 16 patch columns are padded to 32 and the last layer's single output channel to 32 rows (MFMA
tiles are 32 wide); the padding stays zero (zero weight, zero gradient => Adam leaves it at zero).

``forward(x)`` returns ``[logits]`` (MONAI returns every block's output; the reference only ever takes ``[-1]``);
``return_intermediates=True`` gives the full five-element list (the first four derived on the fly, not differentiable).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class _Conv(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError("parameter holder; arithmetic lives in pti_ldm_vae_amd.disc_engine")


class _Block(_Conv):
    def __init__(self):
        super().__init__()
        self.conv = _Conv()


class PatchDiscriminator(nn.Module):
    def __init__(self, spatial_dims: int = 2, num_layers_d: int = 3, channels: int = 32, in_channels: int = 1,
                 out_channels: int = 1, kernel_size: int = 4, norm: str = "INSTANCE", bias: bool = False,
                 padding: int = 1, dropout: float = 0.0, return_intermediates: bool = False):
        super().__init__()
        if spatial_dims != 2:
            raise ValueError("pti_ldm_vae_amd PatchDiscriminator: only spatial_dims=2 has a HIP path")
        if in_channels != 1 or out_channels != 1 or kernel_size != 4 or padding != 1 or bias or dropout:
            raise ValueError("pti_ldm_vae_amd PatchDiscriminator: built for the reference's call (in/out channels 1, "
                             "4x4 kernels, padding 1, no bias on the normalised layers, no dropout)")
        if str(norm).upper() != "INSTANCE":
            raise ValueError("pti_ldm_vae_amd PatchDiscriminator: norm='INSTANCE' only (what train_vae.py:274 passes)")
        if channels not in (32, 64, 128) or num_layers_d < 1 or channels * 2 ** num_layers_d > 256:
            raise ValueError("pti_ldm_vae_amd PatchDiscriminator: channels * 2**num_layers_d must not exceed 256 "
                             "(HIP kernels cover 32..256 channels)")
        self.num_layers_d, self.num_channels, self.return_intermediates = num_layers_d, channels, return_intermediates
        # (name, cin, cout, stride, has_bias, norm, act)
        spec = [("initial_conv", 1, channels, 2, True, False, True)]
        cin = channels
        for l_ in range(num_layers_d):
            spec.append((str(l_), cin, cin * 2, 1 if l_ == num_layers_d - 1 else 2, False, True, True))
            cin *= 2
        spec.append(("final_conv", cin, 1, 1, True, False, False))
        self.spec = spec
        for name, *_ in spec:
            self.add_module(name, _Block())
        self._engine = None
        self._build_arena()

    # ---- flat arena ----------------------------------------------------------------------------------------------
    def _build_arena(self):
        """slots[name] = (offset, elements in the arena, logical shape, kind); kinds: "w_first" ([cout][32] rows, 16
        real columns), "w" ([cout][4][4][cin]), "w_last" (row 0 of [32][16*cin]), "b" (dense), "b_last" ([32], 1 real)."""
        slots, layers, off = {}, [], 0
        for name, cin, cout, stride, has_bias, norm, act in self.spec:
            k = 32 if cin == 1 else 16 * cin          # patch columns (first layer: 16 taps padded to 32)
            rows = 32 if cout == 1 else cout          # output channels (last layer: 1 padded to 32)
            kind = "w_first" if cin == 1 else ("w_last" if cout == 1 else "w")
            slots[f"{name}.conv.weight"] = (off, rows * k, (cout, cin, 4, 4), kind)
            lay = {"name": name, "cin": cin, "cout": cout, "stride": stride, "norm": norm, "act": act, "k": k, "rows": rows,
                   "w_off": off, "b_off": None}
            off += rows * k
            if has_bias:
                slots[f"{name}.conv.bias"] = (off, rows, (cout,), "b_last" if cout == 1 else "b")
                lay["b_off"] = off
                off += rows
            layers.append(lay)
        self._slots, self.layers = slots, layers
        arena = torch.zeros(off, dtype=torch.float32)
        self._arena, self._grad_arena = arena, None
        self._param_by_name = {}
        for name, (o, n, shp, kind) in slots.items():
            p = nn.Parameter(self.slot_view(arena, name))
            mod = self.get_submodule(name.rsplit(".", 1)[0])
            mod.register_parameter(name.rsplit(".", 1)[1], p)
            self._param_by_name[name] = p
        with torch.no_grad():   # MONAI: Conv weights normal(0, 0.02) (initialise_weights); biases keep nn.Conv2d's default
            for name, p in self._param_by_name.items():
                if name.endswith("weight"):
                    p.normal_(0.0, 0.02)
                else:
                    lay = next(l for l in layers if name.startswith(l["name"] + "."))
                    bound = 1.0 / math.sqrt(lay["cin"] * 16)
                    p.uniform_(-bound, bound)

    def slot_view(self, arena: torch.Tensor, name: str) -> torch.Tensor:
        """The view of ``arena`` (parameter arena, gradient arena, an Adam moment buffer ...) with the parameter's
        logical shape."""
        o, n, shp, kind = self._slots[name]
        if kind == "w_first":
            return arena.as_strided(shp, (32, 16, 4, 1), o)
        if kind == "w":
            cout, cin = shp[0], shp[1]
            return arena[o:o + n].view(cout, 4, 4, cin).permute(0, 3, 1, 2)
        if kind == "w_last":
            cin = shp[1]
            return arena[o:o + 16 * cin].view(1, 4, 4, cin).permute(0, 3, 1, 2)
        if kind == "b_last":
            return arena[o:o + 1]
        return arena[o:o + n]

    def _repoint(self):
        for name, p in self._param_by_name.items():
            p.data = self.slot_view(self._arena, name)
            p.grad = None
        self._engine = None

    def _apply(self, fn, recurse=True):
        new = fn(self._arena)
        if new.dtype != torch.float32:
            raise TypeError("pti_ldm_vae_amd PatchDiscriminator keeps fp32 master weights; bf16 copies are derived")
        self._arena = new
        self._grad_arena = None
        self._repoint()
        return self

    @property
    def param_arena(self) -> torch.Tensor:
        return self._arena

    @property
    def grad_arena(self) -> torch.Tensor:
        if self._grad_arena is None or self._grad_arena.device != self._arena.device:
            self._grad_arena = torch.zeros_like(self._arena)
        return self._grad_arena

    def grad_view(self, name):
        return self.slot_view(self.grad_arena, name)

    def attach_grads(self):
        for name, p in self._param_by_name.items():
            p.grad = self.grad_view(name)

    def mark_weights_dirty(self):
        if self._engine is not None:
            self._engine.packed_version = -1

    def engine(self):
        if self._engine is None:
            if not self._arena.is_cuda:
                raise RuntimeError("pti_ldm_vae_amd PatchDiscriminator runs on MI355X only: move it to a cuda (HIP) "
                                   "device first. There is no CPU fallback.")
            from ..disc_engine import DiscEngine
            self._engine = DiscEngine(self)
        return self._engine

    # ---- (de)serialisation: dense tensors in MONAI's shapes, not views of the padded arena ------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else {}
        for name, p in self._param_by_name.items():
            out[prefix + name] = p if keep_vars else p.detach().clone(memory_format=torch.contiguous_format)
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        missing = [k for k in self._param_by_name if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._param_by_name]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for PatchDiscriminator: missing {missing}, "
                               f"unexpected {unexpected}")
        with torch.no_grad():
            for k, p in self._param_by_name.items():
                if k in state_dict:
                    if tuple(state_dict[k].shape) != tuple(p.shape):
                        raise RuntimeError(f"size mismatch for {k}: {tuple(state_dict[k].shape)} vs {tuple(p.shape)}")
                    p.copy_(state_dict[k])
        self.mark_weights_dirty()
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ---- MONAI API ------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        return self.engine().apply(x, self.return_intermediates)
