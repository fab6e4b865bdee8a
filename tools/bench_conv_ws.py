#!/usr/bin/env python3
"""Interleaved A/B (one process, same box) of the weight-stationary conv kernel (csrc/conv_ws.hip) against the v2 kernel on
config A's 128 -> 128 launches at batch 32: forward (GN+SiLU prologue, residual, statistics, side output) and the data
gradient fused with the GroupNorm backward, at 32^2 / 64^2 / 128^2.  PTI_CONV_WS is read per launch.
usage: python tools/bench_conv_ws.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd import ops  # noqa: E402

dev, B, G, C = torch.device("cuda:0"), int(os.environ.get("BATCH", "32")), 16, 128
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def cases(hw):
    x = (torch.randn(B, hw, hw, C, device=dev) * 1.3).half()
    wt = torch.randn(C, C, 3, 3, device=dev) * 0.03
    wp, wpt = ops.pack_conv_weight(wt, 3, f16=True), ops.pack_conv_weight(wt, 3, flip=True)
    y = torch.empty(B, hw, hw, C, dtype=torch.float16, device=dev)
    res = torch.randn_like(y)
    st = ops.gn_stats(x, G)
    ost = torch.zeros(B, G, 2, dtype=torch.int64, device=dev)
    act = torch.empty(B, hw, hw, C, dtype=torch.bfloat16, device=dev)
    g, b, bias = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy = torch.randn(B, hw, hw, C, device=dev).bfloat16()
    out = torch.empty_like(dy)
    sums = torch.zeros(B, C, 2, device=dev)
    return {
        "fwd full": lambda: ops.conv_mfma(x, wp, bias, y, cout=C, prologue=2, in_stats=st, gamma=g, beta=b, groups=G, residual=res,
                                          out_stats=ost, out_groups=G, act_out=act),
        "fwd gn": lambda: ops.conv_mfma(x, wp, bias, y, cout=C, prologue=2, in_stats=st, gamma=g, beta=b, groups=G),
        "fwd plain": lambda: ops.conv_mfma(x, wp, bias, y, cout=C),
        "dgrad+gnbwd": lambda: ops.conv_mfma_gnbwd(dy, wpt, x, st, g, b, out, sums, cout=C, groups=G, silu=True),
        "dgrad plain": lambda: ops.conv_mfma(dy, wpt, None, out, cout=C),
    }


for hw in (32, 64, 128):
    cs = cases(hw)
    flops = 2.0 * B * hw * hw * C * C * 9
    for name, fn in cs.items():
        t = {"0": [], "1": []}
        for _ in range(rounds):
            for k in ("0", "1"):
                os.environ["PTI_CONV_WS"] = k
                t[k].append(timeit(fn))
        v2, ws = min(t["0"]), min(t["1"])
        print(f"128->128 @{hw}^2 {name:12s}: v2 {v2:7.1f} us {flops / v2 / 1e6:6.0f} TF/s | ws {ws:7.1f} us {flops / ws / 1e6:6.0f} TF/s "
              f"({ws / v2:.2f}x)  frac of 2.5 PF: {flops / ws / 1e6 / 2500:.3f}", flush=True)
os.environ.pop("PTI_CONV_WS", None)   # (default: off)
