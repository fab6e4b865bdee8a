#!/usr/bin/env python3
"""Interleaved A/B (one process) of the one-wave-per-SIMD weight-gradient kernel (wgrad_mfma5_kernel) against the v4 kernel's
two-block mode on the >= 64-channel 3x3 layers of config A (batch 32) and the AR model, plus a bit-level comparison of the
two results (same slab layout, same fixed-order reduction: the kernels differ only in fp32 summation order over pixel rows).
PTI_WGRAD_V5 is read per call.  usage: python tools/bench_wgrad_v5.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd import ops  # noqa: E402

os.environ["PTI_WGRAD_V6"] = "0"      # compare v5 with v4 (v6 would otherwise take the >= 128-channel shapes)
dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "32"))
SHAPES = [(64, 64, 128), (128, 64, 128), (128, 128, 64), (64, 128, 64), (128, 128, 32), (128, 128, 128), (256, 256, 64), (128, 256, 64)]


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


for cin, cout, hw in SHAPES:
    n = B if hw * hw * max(cin, cout) * B * 2 < (1 << 31) else B // 2
    x = torch.randn(n, hw, hw, cin, device=dev).bfloat16()
    dy = torch.randn(n, hw, hw, cout, device=dev).bfloat16()
    res, t = {}, {}
    for k in ("0", "1"):
        os.environ["PTI_WGRAD_V5"] = k
        dw, db = torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev)
        ops.conv_wgrad_mfma(x, dy, dw, db)
        torch.cuda.synchronize()
        res[k] = (dw.clone(), db.clone())
    for _ in range(2):
        for k in ("0", "1"):
            os.environ["PTI_WGRAD_V5"] = k
            dw, db = torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev)
            t.setdefault(k, []).append(timeit(lambda: ops.conv_wgrad_mfma(x, dy, dw, db)))
    rel = ((res["1"][0] - res["0"][0]).norm() / res["0"][0].norm()).item()
    relb = ((res["1"][1] - res["0"][1]).norm() / res["0"][1].norm()).item()
    flops = 2.0 * n * hw * hw * cin * cout * 9
    v4, v5 = min(t["0"]), min(t["1"])
    print(f"{cin:3d}->{cout:3d} @{hw:3d}^2 b{n}: v4 {v4:7.1f} us {flops / v4 / 1e6:5.0f} TF/s | v5 {v5:7.1f} us {flops / v5 / 1e6:5.0f} TF/s "
          f"({v5 / v4:.2f}x) | dw relL2 v5 vs v4 {rel:.1e}, dbias {relb:.1e}", flush=True)
os.environ.pop("PTI_WGRAD_V5", None)
