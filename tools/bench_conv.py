#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv / wgrad kernels on the layer shapes of config A (batch 32).
usage: python tools/bench_conv.py [fwd|wgrad|all]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = 32
ADT = torch.float16 if os.environ.get("BENCH_ACT", "bf16") == "fp16" else torch.bfloat16   # forward storage format
SHAPES = [  # cin, cout, h, w, mode
    (32, 32, 256, 256, ops.PTI_CONV_S1), (64, 64, 128, 128, ops.PTI_CONV_S1), (128, 128, 64, 64, ops.PTI_CONV_S1),
    (128, 128, 32, 32, ops.PTI_CONV_S1), (64, 32, 256, 256, ops.PTI_CONV_S1), (64, 64, 128, 128, ops.PTI_CONV_UP2),
    (128, 128, 32, 32, ops.PTI_CONV_UP2),
]


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
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    G = 16
    for cin, cout, h, w, mode in SHAPES:
        x = torch.randn(B, h, w, cin, device=dev).to(ADT if what == "fwd" else torch.bfloat16)
        wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        bias = torch.zeros(cout, device=dev)
        gamma, beta = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
        wp = ops.pack_conv_weight(wt, 3, mode)
        ho, wo = ops.conv_out_hw(h, w, mode)
        y = torch.empty(B, ho, wo, cout, dtype=ADT if what == "fwd" else torch.bfloat16, device=dev)
        res = torch.randn_like(y)
        st = ops.gn_stats(x, G)
        ost = torch.zeros(B, G, 2, dtype=torch.int64, device=dev)
        flops = 2.0 * B * ho * wo * cout * cin * 9
        nbytes = 2.0 * (x.numel() + y.numel())
        tag = f"{cin:3d}->{cout:3d} @{h}x{w} mode{mode}"
        if what in ("fwd", "all"):
            t0 = timeit(lambda: ops.conv_mfma(x, wp, bias, y, cout=cout, mode=mode))
            t1 = timeit(lambda: ops.conv_mfma(x, wp, bias, y, cout=cout, mode=mode, prologue=2, in_stats=st, gamma=gamma,
                                              beta=beta, groups=G))
            t2 = timeit(lambda: ops.conv_mfma(x, wp, bias, y, cout=cout, mode=mode, prologue=2, in_stats=st, gamma=gamma,
                                              beta=beta, groups=G, residual=res, out_stats=ost, out_groups=G))
            t3 = timeit(lambda: ops.conv_mfma(x, wp, bias, y, cout=cout, mode=mode, residual=res))
            t4 = timeit(lambda: ops.conv_mfma(x, wp, bias, y, cout=cout, mode=mode, out_stats=ost, out_groups=G))
            print(f"      {tag}: res-only {t3 * 1e6:7.1f} us | stats-only {t4 * 1e6:7.1f} us")
            print(f"conv  {tag}: plain {t0 * 1e6:7.1f} us {flops / t0 / 1e12:6.0f} TF/s {nbytes / t0 / 1e9:6.0f} GB/s | "
                  f"+GN/SiLU {t1 * 1e6:7.1f} us {flops / t1 / 1e12:6.0f} TF/s | +res+stats {t2 * 1e6:7.1f} us "
                  f"{flops / t2 / 1e12:6.0f} TF/s", flush=True)
        if what in ("wgrad", "all"):
            dy = torch.randn(B, ho, wo, cout, device=dev).to(torch.bfloat16)
            dw = torch.zeros(cout, cin, 3, 3, device=dev)
            db = torch.zeros(cout, device=dev)
            t0 = timeit(lambda: ops.conv_wgrad_mfma(x, dy, dw, db, mode=mode))
            t1 = timeit(lambda: ops.conv_wgrad_mfma(x, dy, dw, db, mode=mode, prologue=2, in_stats=st, gamma=gamma,
                                                    beta=beta, groups=G))
            print(f"wgrad {tag}: plain {t0 * 1e6:7.1f} us {flops / t0 / 1e12:6.0f} TF/s | +GN/SiLU {t1 * 1e6:7.1f} us "
                  f"{flops / t1 / 1e12:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
