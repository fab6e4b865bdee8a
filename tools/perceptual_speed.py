"""Timing probe: the LPIPS module (torch ops) forward + input gradient at batch 32 x 256^2 under a few execution modes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd.models import PerceptualLoss
dev = torch.device("cuda:0")
pl = PerceptualLoss(allow_random_init=True).to(dev)
x = torch.randn(32, 1, 256, 256, device=dev, requires_grad=True)
y = torch.randn(32, 1, 256, 256, device=dev)


def run(tag, fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); print(f"{tag}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)


def plain():
    l = pl(x, y); torch.autograd.grad(l, x)


def amp(dtype):
    def f():
        with torch.autocast("cuda", dtype=dtype):
            l = pl(x, y)
        torch.autograd.grad(l, x)
    return f


run("fp32 NCHW", plain)
run("autocast bf16", amp(torch.bfloat16))
run("autocast fp16", amp(torch.float16))
pl2 = PerceptualLoss(allow_random_init=True).to(dev).to(memory_format=torch.channels_last)
def cl():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l = pl2(x, y)
    torch.autograd.grad(l, x)
run("autocast bf16 + channels_last weights", cl)
