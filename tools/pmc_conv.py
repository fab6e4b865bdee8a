#!/usr/bin/env python3
"""Launch one MFMA conv shape a few times (for `rocprofv3 --pmc ... -- python3 tools/pmc_conv.py cin cout h w [mode] [variant]`).
variant: plain | gn | full (GN+SiLU prologue, residual, fused stats, activated-input side output = the training step's
forward launch) | dgrad (data gradient fused with the GroupNorm+SiLU backward reduction) | wgrad (weight gradient of the
plain stride-1 conv on bf16 operands: the kernel the library picks for the shape -- v6 / v4 -- plus its slab reduction).
PMC_ACT=fp16 (default): x / y / residual stored fp16 and fp16-packed weights = the training step's forward format;
PMC_ACT=bf16: all-bf16 storage and operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd import ops  # noqa: E402

cin, cout, h, w = (int(v) for v in sys.argv[1:5])
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
variant = sys.argv[6] if len(sys.argv) > 6 else "plain"
dev, B, G = torch.device("cuda:0"), int(os.environ.get("BATCH", "32")), 16
f16 = os.environ.get("PMC_ACT", "fp16") == "fp16"
ADT = torch.float16 if f16 else torch.bfloat16
x = (torch.randn(B, h, w, cin, device=dev) * 1.3).to(ADT)
wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
if variant == "wgrad":
    xb = x.to(torch.bfloat16)
    dy = torch.randn(B, h, w, cout, device=dev).to(torch.bfloat16)
    dw, db = torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev)
    for _ in range(5):
        ops.conv_wgrad_mfma(xb, dy, dw, db)
elif variant == "dgrad":
    dy = torch.randn(B, h, w, cout, device=dev).to(torch.bfloat16)
    wpt = ops.pack_conv_weight(wt, 3, ops.PTI_CONV_S1, flip=True)
    st = ops.gn_stats(x, G)
    out = torch.empty(B, h, w, cin, dtype=torch.bfloat16, device=dev)
    sums = torch.zeros(B, cin, 2, device=dev)
    g, b = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    for _ in range(5):
        ops.conv_mfma_gnbwd(dy, wpt, x, st, g, b, out, sums, cout=cin, groups=G, silu=True)
else:
    wp = ops.pack_conv_weight(wt, 3, mode, f16=f16)
    ho, wo = ops.conv_out_hw(h, w, mode)
    y = torch.empty(B, ho, wo, cout, dtype=ADT, device=dev)
    kw = {}
    if variant in ("gn", "full"):
        kw.update(prologue=2, in_stats=ops.gn_stats(x, G), gamma=torch.ones(cin, device=dev), beta=torch.zeros(cin, device=dev), groups=G)
    if variant == "full":
        kw.update(residual=torch.randn_like(y), out_stats=torch.zeros(B, G, 2, dtype=torch.int64, device=dev), out_groups=G,
                  act_out=torch.empty(B, h, w, cin, dtype=torch.bfloat16, device=dev))
    for _ in range(5):
        ops.conv_mfma(x, wp, torch.zeros(cout, device=dev), y, cout=cout, mode=mode, **kw)
torch.cuda.synchronize()
