#!/usr/bin/env python3
"""Interleaved A/B (one process) of the 128co x 64ci weight-gradient kernel (wgrad_mfma6_kernel, both shapes: PTI_WGRAD_V6=2) against
the v4 kernel's two-block mode on the >= 128-channel 3x3 layers of config A (batch 32) and the AR model, with the results
compared (same slab layout and reduction; fp32 summation order over pixels differs) and, on small ragged shapes, checked
against torch's fp32 weight gradient.  PTI_WGRAD_V6 is read per call.  usage: python tools/bench_wgrad_v6.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "32"))
SHAPES = [(64, 64, 128), (128, 64, 128), (64, 64, 256), (128, 128, 32), (128, 128, 64), (64, 128, 128), (128, 128, 128), (256, 256, 64), (128, 256, 64), (256, 256, 32)]


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


def run(x, dy, v6):
    os.environ["PTI_WGRAD_V6"] = "2" if v6 else "0"
    cout, cin = dy.shape[3], x.shape[3]
    dw, db = torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev)
    ops.conv_wgrad_mfma(x, dy, dw, db)
    torch.cuda.synchronize()
    return dw, db


# ---- correctness on ragged shapes against torch (fp32 conv of the same bf16 values) ----
CHECK = os.environ.get("SKIP_CHECK", "0") != "1"      # (skipped under the wrong-result tuning aids)
for n, h, w, cin, cout in [] if not CHECK else [(2, 8, 16, 64, 128), (3, 13, 21, 64, 128), (2, 30, 20, 128, 128), (1, 4, 16, 128, 256), (5, 7, 5, 64, 128),
                             (2, 8, 16, 64, 64), (3, 13, 21, 128, 64), (2, 30, 20, 64, 64), (1, 3, 40, 64, 192), (4, 17, 9, 192, 64)]:
    g = torch.Generator(device=dev).manual_seed(h * 100 + w)
    x = torch.randn(n, h, w, cin, device=dev, generator=g).bfloat16()
    dy = torch.randn(n, h, w, cout, device=dev, generator=g).bfloat16()
    xr = x.float().permute(0, 3, 1, 2)
    wt = torch.zeros(cout, cin, 3, 3, device=dev, requires_grad=True)
    y = torch.nn.functional.conv2d(xr, wt, padding=1)
    y.backward(dy.float().permute(0, 3, 1, 2))
    ref, refb = wt.grad, dy.float().sum((0, 1, 2))
    dw6, db6 = run(x, dy, True)
    dw4, db4 = run(x, dy, False)
    e6 = ((dw6 - ref).norm() / ref.norm()).item()
    e4 = ((dw4 - ref).norm() / ref.norm()).item()
    eb = ((db6 - refb).norm() / refb.norm()).item()
    dw6b, _ = run(x, dy, True)
    print(f"check n={n} {h}x{w} {cin}->{cout}: v6 vs torch {e6:.2e} (v4 {e4:.2e}), dbias {eb:.2e}, run-to-run identical {torch.equal(dw6, dw6b)}",
          flush=True)
    assert e6 < 1e-5 and eb < 1e-5, "v6 mismatch"

for cin, cout, hw in SHAPES:
    n = B if hw * hw * max(cin, cout) * B * 2 < (1 << 31) else B // 2
    x = torch.randn(n, hw, hw, cin, device=dev).bfloat16()
    dy = torch.randn(n, hw, hw, cout, device=dev).bfloat16()
    res, t = {}, {}
    for k in (False, True):
        res[k] = run(x, dy, k)
    for _ in range(2):
        for k in (False, True):
            os.environ["PTI_WGRAD_V6"] = "2" if k else "0"
            dw, db = torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev)
            t.setdefault(k, []).append(timeit(lambda: ops.conv_wgrad_mfma(x, dy, dw, db)))
    rel = ((res[True][0] - res[False][0]).norm() / res[False][0].norm()).item()
    relb = ((res[True][1] - res[False][1]).norm() / res[False][1].norm()).item()
    flops = 2.0 * n * hw * hw * cin * cout * 9
    v4, v6 = min(t[False]), min(t[True])
    print(f"{cin:3d}->{cout:3d} @{hw:3d}^2 b{n}: v4 {v4:7.1f} us {flops / v4 / 1e6:5.0f} TF/s | v6 {v6:7.1f} us {flops / v6 / 1e6:5.0f} TF/s "
          f"({v6 / v4:.2f}x) | dw relL2 v6 vs v4 {rel:.1e}, dbias {relb:.1e}", flush=True)
os.environ.pop("PTI_WGRAD_V6", None)
