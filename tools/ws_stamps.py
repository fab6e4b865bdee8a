#!/usr/bin/env python3
"""Where a tile of the weight-stationary conv kernel spends its cycles (diagnostic build with -DWS_STAMPS, loaded through
PTI_VAE_LIB; see csrc/conv_ws.hip).  Prints per phase the median cycles over workgroups 0..7 x tiles 2..11 (steady state)
for a few launch variants at 128 -> 128 @128^2 batch 32 (16 tiles per workgroup)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTI_CONV_WS"] = "1"
from pti_ldm_vae_amd import _lib as L, ops  # noqa: E402

dev, B, G, Cc, hw = torch.device("cuda:0"), 32, 16, 128, int(os.environ.get("HW", "128"))
lib = C.CDLL(L.LIB_PATH)
x = (torch.randn(B, hw, hw, Cc, device=dev) * 1.3).half()
wt = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.03
wp, wpt = ops.pack_conv_weight(wt, 3, f16=True), ops.pack_conv_weight(wt, 3, flip=True)
y = torch.empty(B, hw, hw, Cc, dtype=torch.float16, device=dev)
res = torch.randn_like(y)
st = ops.gn_stats(x, G)
ost = torch.zeros(B, G, 2, dtype=torch.int64, device=dev)
act = torch.empty(B, hw, hw, Cc, dtype=torch.bfloat16, device=dev)
g, b, bias = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
dy = torch.randn(B, hw, hw, Cc, device=dev).bfloat16()
out = torch.empty_like(dy)
sums = torch.zeros(B, Cc, 2, device=dev)
cases = {
    "fwd plain": lambda: ops.conv_mfma(x, wp, bias, y, cout=Cc),
    "fwd gn": lambda: ops.conv_mfma(x, wp, bias, y, cout=Cc, prologue=2, in_stats=st, gamma=g, beta=b, groups=G),
    "fwd full": lambda: ops.conv_mfma(x, wp, bias, y, cout=Cc, prologue=2, in_stats=st, gamma=g, beta=b, groups=G, residual=res,
                                      out_stats=ost, out_groups=G, act_out=act),
    "dgrad plain": lambda: ops.conv_mfma(dy, wpt, None, out, cout=Cc),
    "dgrad+gnbwd": lambda: ops.conv_mfma_gnbwd(dy, wpt, x, st, g, b, out, sums, cout=Cc, groups=G, silu=True),
}
buf = (C.c_longlong * (8 * 12 * 5))()
names = ["tile start", "MFMA loop", "barrier a + epilogue", "stores", "(whole tile)"]
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    assert lib.pti_debug_ws_stamps(buf) == 0
    t = torch.tensor(list(buf), dtype=torch.float64).view(8, 12, 5)
    d = torch.stack([t[:, 2:, 1] - t[:, 2:, 0], t[:, 2:, 2] - t[:, 2:, 1], t[:, 2:, 3] - t[:, 2:, 2], t[:, 2:, 4] - t[:, 2:, 3],
                     t[:, 2:, 4] - t[:, 2:, 0]], -1).reshape(-1, 5)
    med = d.median(0).values
    gap = (t[:, 3:, 0] - t[:, 2:-1, 4]).reshape(-1).median().item()
    print(f"{name:12s}: " + " | ".join(f"{n} {int(v):6d}" for n, v in zip(names, med.tolist())) + f" | between tiles {int(gap)}  (cycles; MFMA-only = 9216)")
