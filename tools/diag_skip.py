#!/usr/bin/env python3
"""Timing-only diagnostics that DROP launches of the training step -- results are WRONG by construction.

They used to be environment knobs inside ``pti_ldm_vae_amd/ops.py``; they now live here, outside the product, and are
installed by monkeypatching the launch wrappers of the already-imported ``ops`` module.  Nothing in
the package, ``bench.py`` or ``train_vae.py`` reads a ``PTI_DIAG_*`` variable any more, and both refuse to run when one
is set.

    python tools/diag_skip.py --skip wgrad            # the main stream's work with the GPU to itself
    python tools/diag_skip.py --skip gnb_apply_zero   # gn_bwd_apply replaced by a memset of dx
    python tools/diag_skip.py --skip finalize         # no pti_gn_sums_finalize launches

Prints ms per step (config A, batch 32, 256x256) for the patched step, with a banner on stderr.  Use for A/B against
``python bench.py`` on the SAME box only.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def install(skip: set[str]) -> None:
    from pti_ldm_vae_amd import _lib as L
    from pti_ldm_vae_amd import ops
    print(f"[diag_skip] WRONG-RESULT TIMING DIAGNOSTIC: dropping {sorted(skip)} launches", file=sys.stderr, flush=True)
    if "wgrad" in skip:
        ops.wgrad_direct = lambda wide, narrow, dw, **kw: dw
        ops.conv_wgrad_mfma = lambda x, dy, dw, dbias, **kw: dw
        ops.conv_wgrad_mfma_batched = lambda jobs, workspace=None, accumulate=True: None
    if "finalize" in skip:
        L.lib().pti_gn_sums_finalize = lambda *a: 0
    if "gnb_apply" in skip or "gnb_apply_zero" in skip:
        zero = "gnb_apply_zero" in skip

        def gn_bwd_apply(x, dy, dx, *a, **kw):
            if zero:
                dx.zero_()
            return dx
        ops.gn_bwd_apply = gn_bwd_apply


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip", action="append", required=True,
                    choices=["wgrad", "finalize", "gnb_apply", "gnb_apply_zero"])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    install(set(args.skip))
    import bench
    from pti_ldm_vae_amd.models import VAEModel
    from pti_ldm_vae_amd.trainer import VAETrainer
    from pti_ldm_vae_amd.utils import read_config
    cfg = read_config(os.path.join(ROOT, "config", "vae_dente_no_adv.json"))
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    model = VAEModel.from_config(cfg["autoencoder_def"]).to(dev)
    tr = cfg["autoencoder_train"]
    trainer = VAETrainer(model, lr=tr["lr"], recon_loss=tr["recon_loss"], kl_weight=tr["kl_weight"])
    x = bench.synthetic_batch(args.batch, 1, 256, dev, 42)
    for _ in range(args.warmup):
        trainer.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(x)
    torch.cuda.synchronize()
    print(f"skip={','.join(sorted(args.skip))} ms_per_step={(time.perf_counter() - t0) / args.steps * 1e3:.3f} (results wrong)")


if __name__ == "__main__":
    main()
