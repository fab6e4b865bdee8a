"""Per-layer forward / backward deviation of the PatchDiscriminator engine against the oracle (diagnostic)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.patch_discriminator import PatchDiscriminator as Oracle, patch_adversarial_loss as pal
from pti_ldm_vae_amd.models import PatchDiscriminator
from pti_ldm_vae_amd import ops

size, batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 2
dev = torch.device("cuda:0")
torch.manual_seed(size)
ref = Oracle()
with torch.no_grad():
    for p in ref.parameters():
        p.mul_(5.0)
torch.manual_seed(size + 1)
x = torch.randn(batch, 1, size, size) * 0.8
ys = []
def _hook(m, i, o):
    o.retain_grad()
    ys.append(o)


hooks = [blk.conv.register_forward_hook(_hook) for blk in ref.children()]
xg = x.clone().requires_grad_(True)
logits = ref(xg)[-1]
(0.1 * pal(logits, True, False)).backward()
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
net = PatchDiscriminator(); net.load_state_dict(ref.state_dict()); net = net.to(dev)
eng = net.engine()
ctx = eng.forward(x.to(dev), save=True)
for i, (y, yo) in enumerate(zip(ctx.y, ys)):
    yh = y[..., :yo.shape[1]].float().permute(0, 3, 1, 2).cpu()
    print(f"fwd y{i}: rel {rel(yh, yo.detach()):.3e}  shape {tuple(yo.shape)}")
gl, d = eng.lsgan(ctx, target_is_real=True, weight=0.1)
# replay backward layer by layer
dy = d.view(ctx.y[-1].shape)
L = eng.layers
for i in range(len(L) - 1, -1, -1):
    go = ys[i].grad
    print(f"bwd dy{i}: rel {rel(dy[..., :go.shape[1]].float().permute(0, 3, 1, 2).cpu(), go):.3e}")
    lay, P = L[i], ctx.P[i]
    dP = torch.empty_like(P)
    ops.conv_mfma(dy, eng.wpt[i], None, dP, cout=lay["k"], ksize=1)
    if i == 0:
        d_img = torch.zeros(batch, 1, size, size, device=dev)
        ops.pd_col2im_image(dP, d_img)
        print(f"bwd dx: rel {rel(d_img.cpu(), xg.grad):.3e}")
        break
    # reference dP: gradient w.r.t. the activated input of conv i, unfolded
    yp, tp = ctx.y[i - 1], ctx.t[i - 1]
    g = torch.empty_like(yp)
    g, sums = ops.pd_col2im(dP, yp, tp, g, stride=lay["stride"], slope=0.2)
    # oracle g: dL/dxhat of block i-1 = grad wrt y_{i-1} is after IN backward; recompute from oracle tensors
    yo = ys[i - 1].detach()
    if tp is not None:
        mean = yo.mean((2, 3), keepdim=True); var = yo.var((2, 3), unbiased=False, keepdim=True)
        xhat = ((yo - mean) / (var + 1e-5).sqrt()).requires_grad_(True)
    else:
        xhat = yo.clone().requires_grad_(True)
    a = torch.nn.functional.leaky_relu(xhat, 0.2)
    blk = list(ref.children())[i]
    out = blk.conv(a)
    out.backward(ys[i].grad)
    print(f"    g{i-1} (dL/dxhat): rel {rel(g.float().permute(0, 3, 1, 2).cpu(), xhat.grad):.3e}")
    if tp is not None:
        tm = torch.stack([mean.flatten(1), (var + 1e-5).rsqrt().flatten(1)], -1)
        print(f"    norm table: mean abs err {float((tp.cpu()[..., 0] - tm[..., 0]).abs().max()):.3e}  rstd rel {rel(tp.cpu()[..., 1], tm[..., 1]):.3e}")
        so = torch.stack([xhat.grad.sum((2, 3)), (xhat.grad * xhat.detach()).sum((2, 3))], -1)
        print(f"    sums: rel {rel(sums.cpu(), so):.3e}")
        dy = ops.pd_in_bwd_apply(g, yp, tp, sums)
    else:
        dy = g
