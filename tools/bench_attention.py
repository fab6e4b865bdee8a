#!/usr/bin/env python3
"""Isolated timings of the self-attention kernels (csrc/attention.hip) at the mid-block shapes of config A (L = 1024, C = 128,
batch 32) and of the AR model (L = 4096, C = 256, batch 8): forward, backward (delta + dq + dk/dv) with the executed FLOP
rate (forward 4 L^2 C, backward 10 L^2 C per sample incl. the recomputed scores).  usage: python tools/bench_attention.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


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


for b, l, c in [(32, 1024, 128), (8, 4096, 256), (32, 4096, 256)]:
    qkv = (torch.randn(b, l, 3 * c, device=dev) * 0.5).bfloat16()
    o = torch.empty(b, l, c, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(b, l, dtype=torch.float32, device=dev)
    do = torch.randn(b, l, c, device=dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(b, l, dtype=torch.float32, device=dev)
    tf = timeit(lambda: ops.attention_fwd(qkv, o, lse))
    tb = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, delta, dqkv))
    ff, fb = 4.0 * b * l * l * c, 10.0 * b * l * l * c
    print(f"b={b} L={l} C={c}: fwd {tf:7.1f} us {ff / tf / 1e6:5.0f} TF/s | bwd {tb:7.1f} us {fb / tb / 1e6:5.0f} TF/s", flush=True)
