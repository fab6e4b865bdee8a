"""Timing probe: where the LPIPS term's time goes at batch 32 x 256^2 (feature passes, normalise/diff/lin tail, backward)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd.models import PerceptualLoss
from pti_ldm_vae_amd.utils.losses import ensure_three_channels
dev = torch.device("cuda:0")
pl = PerceptualLoss(allow_random_init=True).to(dev)
net = pl.net
B = int(os.environ.get("B", "32"))
x = torch.randn(B, 1, 256, 256, device=dev, requires_grad=True)
y = torch.randn(B, 1, 256, 256, device=dev)


def timed(tag, fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); print(f"{tag}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)


def prep(t):
    return (ensure_three_channels(t.float()) - net.shift) / net.scale


def tail(f0, f1):
    total = 0.0
    for k, (a, b) in enumerate(zip(f0, f1)):
        a = a / (a.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        b = b / (b.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        total = total + getattr(net, f"lin{k}")((a - b) ** 2).mean((2, 3), keepdim=True)
    return total.mean()


def feats_nograd():
    with torch.no_grad():
        net._taps(prep(y))


def feats_grad():
    net._taps(prep(x))


timed("whole term fwd+bwd", lambda: torch.autograd.grad(pl(x, y), x))
timed("features, no grad (target)", feats_nograd)
timed("features, grad graph (recon)", feats_grad)
with torch.no_grad():
    f1 = net._taps(prep(y))
f0 = net._taps(prep(x))
f0d = [t.detach().requires_grad_(True) for t in f0]
timed("tail fwd", lambda: tail(f0d, f1))
timed("tail fwd+bwd (to the taps)", lambda: torch.autograd.grad(tail(f0d, f1), f0d))
g = torch.autograd.grad(tail(f0d, f1), f0d)
timed("features bwd (taps -> input)", lambda: torch.autograd.grad(f0, x, g, retain_graph=True))
xc = torch.cat([x.detach(), y])
timed("features, no grad, both inputs as one batch", lambda: net._taps(prep(xc)) if not torch.is_grad_enabled() else torch.no_grad()(lambda: net._taps(prep(xc)))())
