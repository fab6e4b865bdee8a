#!/usr/bin/env python3
"""Bitwise run-to-run reproducibility of individual kernels on fixed inputs (statistics are integer fixed-point sums, every other reduction stores per-workgroup partials: all outputs compared).
A kernel whose OUTPUT TENSOR differs between two launches on identical inputs has a race."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, G = 8, 16


def same(name, f, n=4):
    outs = [f() for _ in range(n)]
    torch.cuda.synchronize()
    ok = all(all(torch.equal(a, b) for a, b in zip(o, outs[0])) for o in outs[1:])
    worst = max((a.float() - b.float()).abs().max().item() for o in outs[1:] for a, b in zip(o, outs[0]))
    print(f"{'OK  ' if ok else 'DIFF'} {name}  max|diff| {worst:.3e}")


for (cin, cout, s) in ((32, 32, 128), (64, 64, 64), (128, 128, 32), (64, 32, 128)):
    x = torch.randn(B, s, s, cin, device=dev).half()
    st = ops.gn_stats(x, G)
    gamma, beta = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    wp = ops.pack_conv_weight(w, 3, ops.PTI_CONV_S1)
    res = torch.randn(B, s, s, cout, device=dev).half()
    bias = torch.randn(cout, device=dev)

    def fwd():
        y = torch.empty(B, s, s, cout, dtype=torch.float16, device=dev)
        a = torch.empty(B, s, s, cin, dtype=torch.bfloat16, device=dev)
        ost = torch.zeros(B, G, 2, dtype=torch.int64, device=dev)
        ops.conv_mfma(x, wp, bias, y, cout=cout, prologue=2, in_stats=st, gamma=gamma, beta=beta, groups=G, residual=res,
                      out_stats=ost, out_groups=G, act_out=a)
        return y, a
    same(f"conv fwd {cin}->{cout}@{s} (y, act_out)", fwd)
    dy = torch.randn(B, s, s, cout, device=dev).bfloat16()
    wpt = ops.pack_conv_weight(w, 3, ops.PTI_CONV_S1, flip=True)

    def gnb():
        dyt = torch.empty(B, s, s, cin, dtype=torch.bfloat16, device=dev)
        sums = torch.zeros(B, cin, 2, device=dev)
        ops.conv_mfma_gnbwd(dy, wpt, x, st, gamma, beta, dyt, sums, cout=cin, groups=G, silu=True)
        return (dyt,)
    same(f"conv gnbwd {cout}->{cin}@{s} (dy)", gnb)

    def wg():
        dw, db = torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev)
        ops.conv_wgrad_mfma(x.bfloat16(), dy, dw, db)
        return dw, db
    same(f"wgrad {cin}->{cout}@{s} (dw, db)", wg)

    def apply_():
        dx = torch.empty(B, s, s, cin, dtype=torch.bfloat16, device=dev)
        dyt = torch.randn(B, s, s, cin, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).bfloat16()
        sums = torch.ones(B, cin, 2, device=dev)
        dg, db_ = torch.zeros(cin, device=dev), torch.zeros(cin, device=dev)
        ops.gn_bwd_apply(x, dyt, dx, st, gamma, beta, sums, dg, db_, groups=G)
        return (dx,)
    same(f"gn_bwd_apply c={cin}@{s} (dx)", apply_)

qkv = torch.randn(4, 1024, 3 * 128, device=dev).bfloat16()


def att():
    o = torch.empty(4, 1024, 128, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(4, 1024, device=dev)
    ops.attention_fwd(qkv, o, lse)
    dq = torch.empty_like(qkv)
    delta = torch.empty(4, 1024, device=dev)
    ops.attention_bwd(qkv, o, torch.ones_like(o), lse, delta, dq)
    return o, lse, dq
same("attention fwd+bwd (o, lse, dqkv)", att)
