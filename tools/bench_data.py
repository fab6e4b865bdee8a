#!/usr/bin/env python3
"""Throughput of the input pipeline (TIFF decode -> pinned buffer -> H2D -> pti_preprocess_batch) next to the CPU
restatement of the reference's per-sample transform chain.  Writes synthetic float32 TIFFs to a temp directory first.
usage: python tools/bench_data.py [n_images=512] [source_size=512] [patch=256] [batch=32] [workers=8]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.data_pipeline import preprocess  # noqa: E402  (CPU baseline leg only)
from pti_ldm_vae_amd.data import DeviceImageLoader, list_tif_paths, read_tiff, write_tiff  # noqa: E402


def main():
    n, src, patch, batch, workers = (int(v) for v in (sys.argv[1:] + ["512", "512", "256", "32", "8"][len(sys.argv) - 1:])[:5])
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        yy, xx = np.mgrid[0:src, 0:src]
        mask = ((xx - src / 2) / (0.4 * src)) ** 2 + ((yy - src / 2) / (0.32 * src)) ** 2 <= 1.0
        for i in range(n):
            write_tiff(os.path.join(d, f"{i:05d}.tif"), (rng.standard_normal((src, src)).astype(np.float32) * 300 + 900) * mask)
        paths = list_tif_paths(d)
        dev = torch.device("cuda:0")
        ld = DeviceImageLoader(paths, batch, (patch, patch), dev, shuffle=True, seed=1, num_workers=workers)
        for epoch in range(2):          # epoch 0 warms the page cache and the allocator
            ld.set_epoch(epoch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            seen = 0
            for b in ld:
                seen += b.shape[0]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"device pipeline: {seen / dt:9.1f} img/s  ({n} x {src}x{src} f32 TIFF -> {patch}x{patch}, batch {batch}, "
              f"{workers} decode threads, {src * src * 4 * seen / dt / 1e9:.2f} GB/s of pixels)")
        t0 = time.perf_counter()
        k = min(n, 64)
        for p in paths[:k]:
            preprocess(read_tiff(p), (patch, patch))
        dt = time.perf_counter() - t0
        print(f"CPU restatement (1 thread, decode + Resize(area) + LocalNormalizeByMask): {k / dt:9.1f} img/s")


if __name__ == "__main__":
    main()
