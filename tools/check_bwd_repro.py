#!/usr/bin/env python3
"""Which parameter gradients differ between two identical forward+backward passes?  (bitwise comparison of the
gradient arena, per parameter; run with PTI_WGRAD_STREAM=0 and =1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd.models import VAEModel  # noqa: E402
from pti_ldm_vae_amd.trainer import VAETrainer  # noqa: E402

SMALL = dict(spatial_dims=2, in_channels=1, out_channels=1, latent_channels=4, channels=[32, 64], num_res_blocks=1,
             norm_num_groups=16, norm_eps=1e-6, attention_levels=[False, False], with_encoder_nonlocal_attn=True,
             with_decoder_nonlocal_attn=True)
A = dict(SMALL, channels=[32, 64, 128, 128], num_res_blocks=2, attention_levels=[False] * 4)   # BASELINE.json configs[1]
dev = torch.device("cuda:0")
for name, cfg, b, size, down in (("small", SMALL, 4, 128, 2), ("A", A, 32, 256, 8)):
    torch.manual_seed(0)
    m = VAEModel.from_config(cfg).to(dev)
    tr = VAETrainer(m, lr=1e-3)
    tr.opt.step = lambda **kw: None
    x = torch.randn(b, 1, size, size, device=dev)
    eps = torch.randn(b, 4, size // down, size // down, device=dev)
    net = m.autoencoder
    grads = []
    for _ in range(6):
        tr.step(x, eps)
        torch.cuda.synchronize()
        grads.append(net.grad_arena.clone())
    bad = {}
    for g in grads[1:]:
        for pname, (o, n, _) in net._slots.items():
            if not torch.equal(g[o:o + n], grads[0][o:o + n]):
                d = (g[o:o + n] - grads[0][o:o + n]).abs().max().item()
                bad[pname] = max(bad.get(pname, 0.0), d / (grads[0][o:o + n].abs().max().item() + 1e-30))
    print(f"[{name}] PTI_WGRAD_STREAM={os.environ.get('PTI_WGRAD_STREAM', '1')}: {len(bad)} of {len(net._slots)} parameter gradients differ")
    for k, v in bad.items():
        print(f"   {k}: rel max diff {v:.3e}")
