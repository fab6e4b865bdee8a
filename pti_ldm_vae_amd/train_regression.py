#!/usr/bin/env python3
"""Regression-head training on frozen VAE latents for MI355X -- the counterpart of the reference's
``reg_scripts/train_regression.py`` (BASELINE config 5, ``config/reg_edente_from_dente.json``).

Same CLI (``-c --batch-size --lr --max-epochs --num-workers --cache-rate --seed --subset-size --resume-checkpoint``),
same config schema (``data / vae / targets / regressor_def / regression_train``), same outputs under
``<run_dir>/trained_weights``: ``head_last.pth`` every epoch, ``head_best.pth`` on validation improvement,
``target_norm_stats.json`` when ``target_norm`` is ``"standard"`` (train_regression.py:113-136,138-244).
The encoder forward runs on the HIP engine under ``no_grad``; images come through the device input pipeline
(``pti_ldm_vae_amd.data.create_regression_dataloaders``).  ``--random-init-vae`` builds the VAE of
``vae.config_file`` with seeded random weights when no checkpoint exists (throughput runs; never silently).
W&B logging is out of scope: one JSON line per epoch goes to ``<run_dir>/metrics.jsonl`` with the reference's keys.
"""
from __future__ import annotations

import argparse
import json
import os
from pathlib import Path

import torch

from .data import create_regression_dataloaders
from .models import VAEModel
from .utils import regression_utils as R
from .utils.config import load_vae_config

NORM_STATS_FILENAME = "target_norm_stats.json"


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train a regression head on frozen VAE latents (HIP encoder).")
    p.add_argument("-c", "--config-file", required=True)
    p.add_argument("--batch-size", type=int)
    p.add_argument("--lr", type=float)
    p.add_argument("--max-epochs", type=int)
    p.add_argument("--num-workers", type=int)
    p.add_argument("--cache-rate", type=float)
    p.add_argument("--seed", type=int)
    p.add_argument("--subset-size", type=int)
    p.add_argument("--resume-checkpoint", type=str)
    p.add_argument("--random-init-vae", action="store_true",
                   help="seeded random VAE weights instead of vae.checkpoint (throughput / smoke runs)")
    return p.parse_args(argv)


def apply_overrides(config, args):
    """train_regression.py:52-89: CLI values replace data / train config entries."""
    data_cfg = R.extract_regression_data_config(config)
    train_cfg = R.extract_regression_train_config(config)
    for key, val in (("num_workers", args.num_workers), ("cache_rate", args.cache_rate), ("seed", args.seed),
                     ("subset_size", args.subset_size)):
        if val is not None:
            data_cfg[key] = val
    for key, val in (("batch_size", args.batch_size), ("lr", args.lr), ("max_epochs", args.max_epochs)):
        if val is not None:
            train_cfg[key] = val
    config["data"], config["regression_train"] = data_cfg, train_cfg
    return data_cfg, train_cfg


def maybe_build_normalizer(train_loader, targets, weights_dir: Path, mode: str):
    """train_regression.py:113-136: ``"none"`` -> no normaliser; ``"standard"`` -> mean/std of the training targets,
    saved next to the weights."""
    if mode == "none":
        return None
    if mode != "standard":
        raise ValueError(f"Unsupported target_norm '{mode}'. Use 'none' or 'standard'.")
    normalizer = R.compute_target_normalizer(train_loader.stacked_targets())
    R.save_target_normalizer(weights_dir / NORM_STATS_FILENAME, normalizer, targets)
    return normalizer


def main(argv=None):
    args = parse_args(argv)
    with open(args.config_file, encoding="utf-8") as fh:
        config = json.load(fh)
    data_cfg, train_cfg = apply_overrides(config, args)
    run_dir = Path(config.get("run_dir") or Path("runs") / Path(args.config_file).stem)
    weights_dir = run_dir / "trained_weights"
    weights_dir.mkdir(parents=True, exist_ok=True)
    if not torch.cuda.is_available():
        raise RuntimeError("train_regression: the HIP encoder needs an MI355X (no CPU fallback)")
    device = torch.device("cuda:0")
    seed = data_cfg.get("seed")
    if seed is not None:
        torch.manual_seed(int(seed))
    targets = list(config["targets"])
    if args.random_init_vae:
        print("[WARN] --random-init-vae: the VAE encoder has seeded random weights, not vae.checkpoint")
        vae = VAEModel.from_config(load_vae_config(config["vae"]["config_file"]).autoencoder_def).to(device).eval()
        model, latent_dim = R.build_regression_model(vae, config, targets, device)
    else:
        model, latent_dim = R.build_regression_model_from_config(config, targets, device)
    n_head = sum(p.numel() for p in model.regressor.parameters())
    print(f"latent_dim {latent_dim} | targets {targets} | head parameters {n_head} | frozen VAE parameters "
          f"{sum(p.numel() for p in model.vae.parameters())}")
    train_loader, val_loader, train_paths, val_paths = create_regression_dataloaders(
        data_cfg["data_base_dir"], data_cfg["attributes_path"], targets, train_cfg["batch_size"],
        tuple(data_cfg["patch_size"]), train_split=float(data_cfg.get("train_split", 0.9)),
        num_workers=int(data_cfg.get("num_workers", 4)), seed=seed, subset_size=data_cfg.get("subset_size"),
        val_dir=data_cfg.get("val_dir"), cache_rate=float(data_cfg.get("cache_rate", 0.0)),
        data_source=data_cfg.get("data_source", "edente"), normalize_attributes=data_cfg.get("normalize_attributes"),
        device=device)
    normalizer = maybe_build_normalizer(train_loader, targets, weights_dir, train_cfg.get("target_norm", "none"))
    loss_fn, loss_key = R.build_loss_fn(train_cfg.get("loss", "mse")), R.regression_loss_key(train_cfg)
    optimizer = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=train_cfg["lr"],
                                 weight_decay=train_cfg.get("weight_decay", 0.0))
    if args.resume_checkpoint is not None:
        R.load_regression_checkpoint(Path(args.resume_checkpoint), model, targets)
    max_epochs, val_interval = train_cfg["max_epochs"], train_cfg.get("val_interval", 1)
    best_val, best_path = float("inf"), None
    with open(run_dir / "metrics.jsonl", "a") as log:
        for epoch in range(1, max_epochs + 1):
            train_loader.set_epoch(epoch)
            train_loss = R.train_one_epoch(model, train_loader, optimizer, loss_fn, device, normalizer)
            rec = {"epoch": epoch, f"train/{loss_key}": train_loss}
            if epoch % val_interval == 0 or epoch == max_epochs:
                val_loss, metrics = R.validate_one_epoch(model, val_loader, loss_fn, device, targets, normalizer)
                best_val, best_path = R.maybe_save_best_regression_checkpoint(weights_dir, model, targets, epoch, val_loss,
                                                                               best_val, best_path)
                rec.update({f"val/{loss_key}": val_loss, f"val/best_{loss_key}": best_val,
                            **{f"val/{k}": v for k, v in metrics.items()}})
                print(f"[Epoch {epoch:03d}/{max_epochs:03d}] train_{loss_key}={train_loss:.4f} val_{loss_key}={val_loss:.4f}")
            else:
                print(f"[Epoch {epoch:03d}/{max_epochs:03d}] train_{loss_key}={train_loss:.4f}")
            R.save_last_regression_checkpoint(weights_dir, model, targets, epoch)
            log.write(json.dumps(rec) + "\n")
            log.flush()
    print(f"Training complete: {len(train_paths)} train / {len(val_paths)} val images; weights in {weights_dir}")


if __name__ == "__main__":
    main()
