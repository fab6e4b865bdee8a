#!/usr/bin/env python3
"""Per-step wall times of the native training step (batch 32, 256^2, config A): spots stalls inside a timed region."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pti_ldm_vae_amd.models import VAEModel  # noqa: E402
from pti_ldm_vae_amd.trainer import VAETrainer  # noqa: E402

A = dict(spatial_dims=2, in_channels=1, out_channels=1, latent_channels=4, channels=[32, 64, 128, 128], num_res_blocks=2,
         norm_num_groups=16, norm_eps=1e-6, attention_levels=[False] * 4, with_encoder_nonlocal_attn=True,
         with_decoder_nonlocal_attn=True)
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = VAEModel.from_config(A).to(dev)
tr = VAETrainer(m, lr=1e-4)
x = torch.randn(32, 1, 256, 256, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
torch.cuda.synchronize()
ev[0].record()
for i in range(n):
    t0 = time.perf_counter()
    tr.step(x)
    host.append((time.perf_counter() - t0) * 1e3)
    ev[i + 1].record()
torch.cuda.synchronize()
gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("gpu ms/step:", " ".join(f"{t:.1f}" for t in gpu))
print("host enqueue ms/step:", " ".join(f"{t:.1f}" for t in host))
