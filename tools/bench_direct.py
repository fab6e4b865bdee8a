#!/usr/bin/env python3
"""Micro-benchmark of the degenerate-channel (direct) conv kernels on the config-A shapes at batch 32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd import ops  # noqa: E402

dev, B, G = torch.device("cuda:0"), 32, 16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    for (cn, cw, s) in ((1, 32, 256), (4, 128, 32)):
        wide = torch.randn(B, s, s, cw, device=dev).to(torch.float16)
        st = ops.gn_stats(wide, G)
        gamma, beta = torch.ones(cw, device=dev), torch.zeros(cw, device=dev)
        nar_nchw = torch.randn(B, cn, s, s, device=dev)
        nar_nhwc = torch.randn(B, s, s, cn, device=dev)
        w_fc = torch.randn(9, cn, cw, device=dev)      # few-cin
        w_fo = torch.randn(9, cw, cn, device=dev)      # few-cout
        bias_w, bias_n = torch.zeros(cw, device=dev), torch.zeros(cn, device=dev)
        mb = wide.numel() * 2 / 1e6
        yw = torch.empty_like(wide)
        t = timeit(lambda: ops.conv_direct(nar_nchw, w_fc, bias_w, yw, n=B, h=s, w=s, cin=cn, cout=cw, x_layout="nchw"))
        print(f"few-cin  {cn:3d}->{cw:3d} @{s}: {t:7.1f} us  {mb / t:6.2f} TB/s(wide)")
        yn = torch.empty_like(nar_nchw)
        t = timeit(lambda: ops.conv_direct(wide, w_fo, bias_n, yn, n=B, h=s, w=s, cin=cw, cout=cn, y_layout="nchw",
                                           prologue=1, in_stats=st, gamma=gamma, beta=beta, groups=G))
        print(f"few-cout {cw:3d}->{cn:3d} @{s}: {t:7.1f} us  {mb / t:6.2f} TB/s(wide)  (GN prologue)")
        wb = wide.to(torch.bfloat16)
        yn2 = torch.empty_like(nar_nhwc)
        t = timeit(lambda: ops.conv_direct(wb, w_fo, None, yn2.view(B, s, s, cn), n=B, h=s, w=s, cin=cw, cout=cn))
        print(f"few-cout {cw:3d}->{cn:3d} @{s}: {t:7.1f} us  {mb / t:6.2f} TB/s(wide)  (plain, data gradient)")
        dw = torch.zeros(cn, cw, 3, 3, device=dev)
        t = timeit(lambda: ops.wgrad_direct(wide, nar_nchw, dw, n=B, h=s, w=s, cw=cw, cn=cn, ksize=3, sgn=1, narrow_layout="nchw",
                                            dw_strides=(1, 9, cw * 9), dbias_narrow=bias_n, prologue=1, in_stats=st,
                                            gamma=gamma, beta=beta, groups=G))
        print(f"wgrad_direct few-cout layer cw={cw} cn={cn} @{s}: {t:7.1f} us  {mb / t:6.2f} TB/s(wide)")
        dw2 = torch.zeros(cw, cn, 3, 3, device=dev)
        t = timeit(lambda: ops.wgrad_direct(wb, nar_nchw, dw2, n=B, h=s, w=s, cw=cw, cn=cn, ksize=3, sgn=-1,
                                            narrow_layout="nchw", dw_strides=(1, cn * 9, 9), dbias_wide=bias_w))
        print(f"wgrad_direct few-cin  layer cw={cw} cn={cn} @{s}: {t:7.1f} us  {mb / t:6.2f} TB/s(wide)")


if __name__ == "__main__":
    main()
