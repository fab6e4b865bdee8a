"""Regression on frozen VAE latents (BASELINE config 5) -- the loop around the encoder-only hot path.

Same functions, arguments and file formats as the reference's ``src/pti_ldm_vae/utils/regression_utils.py``:
config normalisation (:17-128), ``TargetNormalizer`` (:239-306), ``build_loss_fn`` (:309-315), ``train_one_epoch``
(:318-347), ``validate_one_epoch`` (:350-388), head checkpoints ``head_last.pth`` / ``head_best.pth`` with keys
``regressor_state_dict / targets / epoch / latent_dim`` (:391-477) and ``build_regression_model_from_config``
(:480-515).  W&B helpers are out of scope.

What runs where: the encoder forward (17.7 GFLOP per 256x256 image) is the HIP engine under ``no_grad``
(``VAEModel.encode_deterministic``); the MLP head (0.002 GFLOP per image) and its Adam are a few tiny torch ops.
MI355X-first differences: the epoch loss is accumulated on the device (ONE host sync per epoch instead of one
``.item()`` per step, so the host keeps enqueuing encoder launches), validation predictions are gathered on the device
and moved once; checkpoints are read with ``weights_only=True``.
"""
from __future__ import annotations

import json
from collections.abc import Callable
from pathlib import Path
from typing import Any

import torch
from torch import nn

from ..models import LatentRegressor, VAELatentRegressor
from .metrics import compute_regression_metrics
from .vae_loader import load_vae_config, load_vae_model


# ---- config normalisation (both schemas the reference accepts) ---------------------------------------------------------
def extract_regression_data_config(config: dict[str, Any]) -> dict[str, Any]:
    data, legacy = dict(config.get("data", {})), config.get("train", {})
    defaults = {"data_base_dir": config.get("data_base_dir"), "attributes_path": config.get("attributes_path"),
                "data_source": config.get("data_source", "edente"), "train_split": config.get("train_split", 0.9),
                "val_dir": config.get("val_dir"), "patch_size": config.get("patch_size"),
                "cache_rate": config.get("cache_rate", legacy.get("cache_rate", 0.0)),
                "num_workers": config.get("num_workers", legacy.get("num_workers", 4)),
                "seed": config.get("seed", legacy.get("seed")),
                "subset_size": config.get("subset_size", legacy.get("subset_size")),
                "normalize_attributes": config.get("normalize_attributes")}
    for k, v in defaults.items():
        data.setdefault(k, v)
    missing = [f for f in ("data_base_dir", "attributes_path", "patch_size") if data.get(f) is None]
    if missing:
        raise KeyError(f"Missing required data config fields: {missing}")
    return data


def extract_regression_train_config(config: dict[str, Any]) -> dict[str, Any]:
    train = dict(config.get("regression_train") or config.get("train") or {})
    missing = [f for f in ("batch_size", "lr", "max_epochs") if train.get(f) is None]
    if missing:
        raise KeyError(f"Missing required training config fields: {missing}")
    for k, v in (("val_interval", 1), ("target_norm", "none"), ("loss", "mse"), ("weight_decay", 0.0)):
        train.setdefault(k, v)
    return train


def extract_regression_eval_config(config: dict[str, Any], data_cfg: dict[str, Any] | None = None) -> dict[str, Any]:
    base = data_cfg or extract_regression_data_config(config)
    ev = dict(config.get("evaluation", {}))
    for k, d in (("data_base_dir", None), ("attributes_path", None), ("data_source", "edente"), ("patch_size", None),
                 ("num_workers", 4), ("normalize_attributes", None)):
        ev.setdefault(k, base.get(k, d))
    missing = [f for f in ("data_base_dir", "attributes_path", "patch_size") if ev.get(f) is None]
    if missing:
        raise KeyError(f"Missing required evaluation config fields: {missing}")
    return ev


def extract_regressor_def_config(config: dict[str, Any]) -> dict[str, Any]:
    reg = dict(config.get("regressor_def") or config.get("regressor") or {})
    for k, v in (("hidden_dims", []), ("dropout", 0.0), ("activation", "relu")):
        reg.setdefault(k, v)
    return reg


def regression_loss_key(train_cfg: dict[str, Any]) -> str:
    return "loss_huber" if str(train_cfg.get("loss", "mse")).lower() in {"smooth_l1", "huber"} else "loss_mse"


# ---- target normalisation ---------------------------------------------------------------------------------------------
class TargetNormalizer:
    """Standard scaling of the target vectors; a zero std is replaced by 1 (reference :242-253)."""

    def __init__(self, mean: torch.Tensor, std: torch.Tensor):
        if mean.shape != std.shape:
            raise ValueError("Mean and std must share the same shape.")
        self.mean, self.std = mean, torch.where(std == 0, torch.ones_like(std), std)

    def normalize(self, targets: torch.Tensor) -> torch.Tensor:
        return (targets - self.mean.to(targets.device)) / self.std.to(targets.device)

    def denormalize(self, values: torch.Tensor) -> torch.Tensor:
        return values * self.std.to(values.device) + self.mean.to(values.device)

    def to_dict(self, target_names: list[str]) -> dict:
        return {"target_names": target_names, "mean": self.mean.tolist(), "std": self.std.tolist()}

    @classmethod
    def from_dict(cls, data: dict, target_names: list[str]) -> "TargetNormalizer":
        stored = data.get("target_names", [])
        if stored and list(stored) != target_names:
            raise ValueError(f"Target order mismatch: expected {target_names}, found {stored}")
        return cls(torch.tensor(data["mean"], dtype=torch.float32), torch.tensor(data["std"], dtype=torch.float32))


def compute_target_normalizer(targets: torch.Tensor) -> TargetNormalizer:
    return TargetNormalizer(targets.mean(dim=0), targets.std(dim=0, unbiased=False))


def save_target_normalizer(path: Path, normalizer: TargetNormalizer, target_names: list[str]) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    with path.open("w", encoding="utf-8") as fh:
        json.dump(normalizer.to_dict(target_names), fh, indent=2)


def load_target_normalizer(path: Path, target_names: list[str]) -> TargetNormalizer:
    with path.open(encoding="utf-8") as fh:
        return TargetNormalizer.from_dict(json.load(fh), target_names)


def build_loss_fn(loss_name: str) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    name = loss_name.lower()
    if name in {"mse", "mse_loss"}:
        return nn.MSELoss()
    if name in {"smooth_l1", "huber"}:
        return nn.SmoothL1Loss()
    raise ValueError(f"Unsupported loss '{loss_name}'. Use 'mse' or 'smooth_l1'.")


# ---- the loop -----------------------------------------------------------------------------------------------------------
def train_one_epoch(model: nn.Module, dataloader, optimizer, loss_fn, device: torch.device,
                    normalizer: TargetNormalizer | None) -> float:
    """Reference :318-347: frozen-encoder forward -> head -> loss -> backward -> optimizer step, mean loss of the epoch.
    ``dataloader`` yields ``(images, targets)``; the loss is summed on the device, one sync at the end."""
    model.train()
    total = torch.zeros((), device=device)
    n = 0
    for images, targets in dataloader:
        images, targets = images.to(device, non_blocking=True), targets.to(device, non_blocking=True)
        want = normalizer.normalize(targets) if normalizer is not None else targets
        optimizer.zero_grad()
        loss = loss_fn(model(images), want)
        loss.backward()
        optimizer.step()
        total += loss.detach()
        n += 1
    if n == 0:
        raise RuntimeError("Training dataloader produced zero batches.")
    return float(total.item()) / n


def validate_one_epoch(model: nn.Module, dataloader, loss_fn, device: torch.device, target_names: list[str],
                       normalizer: TargetNormalizer | None) -> tuple[float, dict[str, float]]:
    """Reference :350-388: mean loss on (normalised) targets; metrics on de-normalised predictions vs raw targets."""
    model.eval()
    total = torch.zeros((), device=device)
    n = 0
    preds, tgts = [], []
    with torch.no_grad():
        for images, targets in dataloader:
            images, targets = images.to(device, non_blocking=True), targets.to(device, non_blocking=True)
            want = normalizer.normalize(targets) if normalizer is not None else targets
            out = model(images)
            total += loss_fn(out, want)
            n += 1
            preds.append(normalizer.denormalize(out) if normalizer is not None else out)
            tgts.append(targets)
    if n == 0:
        raise RuntimeError("Validation dataloader produced zero batches.")
    metrics = compute_regression_metrics(torch.cat(preds).cpu(), torch.cat(tgts).cpu(), target_names)
    return float(total.item()) / n, metrics


# ---- head checkpoints ---------------------------------------------------------------------------------------------------
def save_regression_checkpoint(path: Path, model: nn.Module, targets: list[str], epoch: int | None = None) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save({"regressor_state_dict": {k: v.detach().cpu() for k, v in model.regressor.state_dict().items()},
                "targets": targets, "epoch": epoch, "latent_dim": getattr(model, "latent_dim", None)}, path)


def load_regression_checkpoint(path: Path, model: nn.Module, expected_targets: list[str]) -> dict[str, Any]:
    ck = torch.load(path, map_location="cpu", weights_only=True)
    stored = ck.get("targets")
    if stored and list(stored) != list(expected_targets):
        raise ValueError(f"Target mismatch: expected {expected_targets}, found {stored}.")
    model.regressor.load_state_dict(ck["regressor_state_dict"])
    return ck


def save_last_regression_checkpoint(weights_dir: Path, model: nn.Module, targets: list[str], epoch: int) -> Path:
    path = weights_dir / "head_last.pth"
    save_regression_checkpoint(path, model, targets, epoch)
    return path


def maybe_save_best_regression_checkpoint(weights_dir: Path, model: nn.Module, targets: list[str], epoch: int,
                                          val_loss: float, best_val_loss: float,
                                          best_path: Path | None = None) -> tuple[float, Path]:
    path = best_path or weights_dir / "head_best.pth"
    if val_loss < best_val_loss:
        save_regression_checkpoint(path, model, targets, epoch)
        return val_loss, path
    return best_val_loss, path


def build_regression_model(vae, config: dict[str, Any], targets: list[str], device: torch.device):
    """Head + wrapper around an already-built (frozen) ``VAEModel``."""
    reg = extract_regressor_def_config(config)
    patch = tuple(extract_regression_data_config(config)["patch_size"])
    latent_dim = VAELatentRegressor.infer_flat_dim_from_patch(vae, patch, device)
    head = LatentRegressor(in_features=latent_dim, hidden_dims=reg.get("hidden_dims", []), output_dim=len(targets),
                           dropout=float(reg.get("dropout", 0.0)), activation=reg.get("activation", "relu"))
    return VAELatentRegressor(vae=vae, regressor=head, latent_dim=latent_dim).to(device), latent_dim


def build_regression_model_from_config(config: dict[str, Any], targets: list[str], device: torch.device):
    """Reference :480-515: VAE from ``config["vae"]`` (config file + checkpoint), frozen; head from ``regressor_def``."""
    vae_cfg = load_vae_config(config["vae"]["config_file"])
    vae = load_vae_model(vae_cfg, config["vae"]["checkpoint"], device)
    return build_regression_model(vae, config, targets, device)
