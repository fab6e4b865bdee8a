#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one training step (no sync) vs the GPU to run it?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch
from pti_ldm_vae_amd.models import VAEModel
from pti_ldm_vae_amd.trainer import VAETrainer
from pti_ldm_vae_amd.utils import read_config
dev = torch.device("cuda:0")
cfg = read_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "vae_dente_no_adv.json"))
for batch in (32, 4):
    torch.manual_seed(42)
    model = VAEModel.from_config(cfg["autoencoder_def"]).to(dev)
    tr = VAETrainer(model, lr=2.5e-5)
    x = synthetic_batch(batch, 1, 256, dev, 42)
    for _ in range(3):
        tr.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"batch {batch}: host enqueue {1e3 * (t1 - t0) / 10:.2f} ms/step, wall {1e3 * (t2 - t0) / 10:.2f} ms/step", flush=True)
