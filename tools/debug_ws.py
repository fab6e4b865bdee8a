import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd import ops
dev = torch.device("cuda:0")
c, groups, eps = 128, 16, 1e-6
torch.manual_seed(0)
def run(n, hw, pro, res, ostats, save, ws, cap=0):
    torch.manual_seed(1)
    xd = (torch.randn(n, hw, hw, c, device=dev) * 1.3 + 0.2).half()
    rsd = torch.randn(n, hw, hw, c, device=dev).half() if res else None
    wp = ops.pack_conv_weight((torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5), 3, ops.PTI_CONV_S1, f16=True)
    bias, gamma, beta = torch.randn(c, device=dev) * 0.1, 1 + 0.2 * torch.randn(c, device=dev), 0.1 * torch.randn(c, device=dev)
    st = ops.gn_stats(xd, groups) if pro else None
    os.environ["PTI_CONV_WS"] = "1" if ws else "0"
    if cap: os.environ["PTI_CONV_WS_MAX_WGS"] = str(cap)
    else: os.environ.pop("PTI_CONV_WS_MAX_WGS", None)
    y = torch.full((n, hw, hw, c), float("nan"), dtype=torch.float16, device=dev)
    ost = torch.zeros(n, 16, 2, dtype=torch.int64, device=dev) if ostats else None
    act = torch.full((n, hw, hw, c), float("nan"), dtype=torch.bfloat16, device=dev) if save else None
    ops.conv_mfma(xd, wp, bias, y, cout=c, ksize=3, prologue=pro, in_stats=st, gamma=gamma if pro else None, beta=beta if pro else None,
                  groups=groups, eps=eps, residual=rsd, out_stats=ost, out_groups=16, act_out=act)
    torch.cuda.synchronize()
    return y, ost, act
for (n, hw, pro, res, ostats, save, cap) in [(32, 64, 2, True, True, True, 0), (32, 64, 2, False, False, False, 0), (32, 64, 0, True, False, False, 0),
                                             (32, 64, 0, False, False, False, 0), (32, 64, 2, False, False, True, 0), (8, 64, 2, True, True, True, 0),
                                             (8, 64, 2, True, True, True, 64), (32, 32, 2, True, True, True, 0), (32, 64, 2, True, True, True, 256)]:
    y0, o0, a0 = run(n, hw, pro, res, ostats, save, False)
    y1, o1, a1 = run(n, hw, pro, res, ostats, save, True, cap)
    y2, _, _ = run(n, hw, pro, res, ostats, save, True, cap)
    d = (y0.float() - y1.float()).abs()
    bad = (d > 0).nonzero()
    msg = f"n{n} hw{hw} pro{pro} res{res} st{ostats} save{save} cap{cap}: y diff count {bad.shape[0]} max {d.max().item():.3e} ws run-to-run equal {torch.equal(y1, y2)}"
    if a0 is not None: msg += f" | act diff {(a0.float()-a1.float()).abs().gt(0).sum().item()}"
    if o0 is not None: msg += f" | stats equal {torch.equal(o0, o1)}"
    print(msg)
    if bad.shape[0]:
        nn, yy, xx, cc = bad[:, 0], bad[:, 1], bad[:, 2], bad[:, 3]
        tiles_x, tiles_y = hw // 16, hw // 8
        t = (nn * tiles_y + yy // 8) * tiles_x + xx // 16
        grid = min(cap if cap else 256, n * tiles_x * tiles_y)
        print("   samples", sorted(set(nn.tolist()))[:12], " tile round (t // grid):", sorted(set((t // grid).tolist())), " rows in tile", sorted(set((yy % 8).tolist())),
              " cols in tile", sorted(set((xx % 16).tolist()))[:16], " #channels", len(set(cc.tolist())))
