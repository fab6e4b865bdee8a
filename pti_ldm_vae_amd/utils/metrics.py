"""Regression metrics (reference ``src/pti_ldm_vae/utils/metrics.py:6-40``; pinned by SURVEY.md 8c KA6)."""
from __future__ import annotations

from typing import Any

import torch


def compute_regression_metrics(predictions: torch.Tensor, targets: torch.Tensor, target_names: list[str]) -> dict[str, Any]:
    """MAE / MSE per target column and their means over the columns."""
    if predictions.shape != targets.shape:
        raise ValueError(f"Shape mismatch between predictions {predictions.shape} and targets {targets.shape}.")
    err = predictions - targets
    mae_t, mse_t = err.abs().mean(dim=0), (err * err).mean(dim=0)
    out: dict[str, Any] = {"mae": float(mae_t.mean()), "mse": float(mse_t.mean())}
    for i, name in enumerate(target_names):
        out[f"mae_{name}"], out[f"mse_{name}"] = float(mae_t[i]), float(mse_t[i])
    return out
