"""Weight-gradient kernels at the training step's shapes (config A, batch 32; AR shapes with --ar): partial launch and
slab reduction timed separately with events (two-call C-ABI form), 20 rounds after 3 warm-ups, median.
PTI_WGRAD_V4=0 selects the register-staged v3 kernel; run both in separate processes to compare."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES_A = [(32, 32, 256), (64, 32, 256), (64, 64, 128), (32, 64, 128), (128, 64, 128), (128, 128, 64), (64, 128, 64), (128, 128, 32)]
SHAPES_AR = [(64, 64, 256), (128, 128, 128), (256, 256, 64), (128, 256, 64)]
batch = int(os.environ.get("BATCH", "32"))
shapes = SHAPES_AR if "--ar" in sys.argv else SHAPES_A
if os.environ.get("SHAPE"):
    shapes = [tuple(int(v) for v in os.environ["SHAPE"].split(","))]
ws = ops.wgrad_workspace(dev)
lib = L.lib()
for cin, cout, h in shapes:
    x = torch.randn(batch, h, h, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(batch, h, h, cout, device=dev).to(torch.bfloat16)
    dw = torch.zeros(cout, cin, 3, 3, device=dev)
    db = torch.zeros(cout, device=dev)
    d = L.ConvDesc(n=batch, h=h, w=h, cin=cin, ho=h, wo=h, cout=cout, ksize=3, mode=L.PTI_CONV_S1)
    tp, tr = [], []
    splits = C.c_int(0)
    for it in range(23):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        L.check(lib.pti_conv_wgrad_mfma_partials(x.data_ptr(), dy.data_ptr(), None, None, None, ws.data_ptr(), ws.numel() * 4,
                                                 C.byref(d), C.byref(splits), torch.cuda.current_stream().cuda_stream))
        e[1].record()
        L.check(lib.pti_conv_wgrad_reduce(ws.data_ptr(), splits.value, dw.data_ptr(), db.data_ptr(), 0, C.byref(d),
                                          torch.cuda.current_stream().cuda_stream))
        e[2].record()
        torch.cuda.synchronize()
        if it >= 3:
            tp.append(e[0].elapsed_time(e[1]) * 1e3)
            tr.append(e[1].elapsed_time(e[2]) * 1e3)
    tp.sort(); tr.sort()
    p, r = tp[len(tp) // 2], tr[len(tr) // 2]
    flop = 2.0 * batch * h * h * cin * cout * 9
    nbytes = 2.0 * batch * h * h * (cin + cout)
    print(f"{cin:4d}->{cout:4d} @{h:3d}^2 b{batch}: splits {splits.value & 0xffff:4d}{'*' if splits.value >> 30 else ' '} partials {p:7.1f} us (min {tp[0]:6.1f}) reduce {r:5.1f} us | "
          f"{flop / p / 1e6:6.0f} TFLOP/s {nbytes / p / 1e3:6.0f} GB/s alg (partials only); total {p + r:7.1f} us", flush=True)
