#!/usr/bin/env python3
"""SURVEY.md §8(d) run C5: encoder-only no_grad inference throughput (VAEModel.encode_deterministic, the call the
latent-regression head makes, models/autoencoder.py:127-140) on synthetic 256x256 images.
usage: python tools/bench_encode.py [batch=32] [config=config/vae_dente_no_adv.json]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_batch  # noqa: E402
from pti_ldm_vae_amd.models import VAEModel  # noqa: E402
from pti_ldm_vae_amd.utils import read_config  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    cfg = read_config(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "config", "vae_dente_no_adv.json"))
    d = cfg["autoencoder_def"]
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    model = VAEModel.from_config(d).to(dev).eval()
    x = synthetic_batch(batch, d["in_channels"], 256, dev, seed=42)
    with torch.no_grad():
        for _ in range(3):
            model.encode_deterministic(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            z = model.encode_deterministic(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"encode_deterministic: batch {batch} 256x256 -> {tuple(z.shape)}: {dt * 1e3:.3f} ms, {batch / dt:.0f} images/s")


if __name__ == "__main__":
    main()
