#!/usr/bin/env python3
"""Launch one MFMA conv shape a few times (for `rocprofv3 --pmc ... -- python3 tools/pmc_conv.py cin cout h w [mode] [variant]`).
variant: plain | gn | full (GN+SiLU prologue, residual, fused stats); PMC_ACT=fp16 stores x / y / residual as fp16
(the training step's forward format)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd import ops  # noqa: E402

cin, cout, h, w = (int(v) for v in sys.argv[1:5])
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
variant = sys.argv[6] if len(sys.argv) > 6 else "plain"
dev, B, G = torch.device("cuda:0"), 32, 16
ADT = torch.float16 if os.environ.get("PMC_ACT", "bf16") == "fp16" else torch.bfloat16
x = torch.randn(B, h, w, cin, device=dev).to(ADT)
wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
wp = ops.pack_conv_weight(wt, 3, mode)
ho, wo = ops.conv_out_hw(h, w, mode)
y = torch.empty(B, ho, wo, cout, dtype=ADT, device=dev)
kw = {}
if variant in ("gn", "full"):
    kw.update(prologue=2, in_stats=ops.gn_stats(x, G), gamma=torch.ones(cin, device=dev), beta=torch.zeros(cin, device=dev), groups=G)
if variant == "full":
    kw.update(residual=torch.randn_like(y), out_stats=torch.zeros(B, G, 2, dtype=torch.int64, device=dev), out_groups=G)
for _ in range(5):
    ops.conv_mfma(x, wp, torch.zeros(cout, device=dev), y, cout=cout, mode=mode, **kw)
torch.cuda.synchronize()
