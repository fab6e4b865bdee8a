import os, sys, torch
sys.path.insert(0, os.getcwd())
from pti_ldm_vae_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for cin, cout, hw in [(32, 32, 256), (32, 32, 128), (64, 64, 128)]:
    x = torch.randn(32, hw, hw, cin, device=dev).bfloat16(); dy = torch.randn(32, hw, hw, cout, device=dev).bfloat16()
    dw, db = torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev)
    t = timeit(lambda: ops.conv_wgrad_mfma(x, dy, dw, db))
    nb = 2.0 * (x.numel() + dy.numel())
    print(f"diag={os.environ.get('PTI_WGRAD_V4_DIAG','0')} {cin}->{cout}@{hw}: {t:.1f} us  {nb / t / 1e6:.2f} TB/s algorithmic  {2.0*32*hw*hw*cin*cout*9/t/1e6:.0f} TF/s", flush=True)
