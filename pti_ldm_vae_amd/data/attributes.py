"""Per-image attribute files of the AR-VAE branch, joined to the image list.

Behaviour follows the reference's ``src/pti_ldm_vae/data/dataloaders.py``: ``_load_attribute_json`` (:120-139),
``_select_attribute_sources`` (:142-153), ``_normalize_attributes`` (:156-172), ``_filter_attributes_for_paths``
(:175-221) and ``collate_with_attributes`` (:108-117) -- same lookups (file BASENAME inside the JSON of the source the
path names: ``"edente"`` is tested before ``"dente"``, which it contains), same exception types.  The reference
shuffles (path, attributes) pairs together (:471-475); looking attributes up by basename AFTER the path split gives the
same pairing because ``random.shuffle`` draws the same permutation for the same list length.
"""
from __future__ import annotations

import json
import os
from typing import Any

import torch


def load_attribute_json(attribute_file: str) -> dict[str, dict[str, float]]:
    if not os.path.exists(attribute_file):
        raise FileNotFoundError(f"Attribute file not found: {attribute_file}")
    with open(attribute_file, encoding="utf-8") as fh:
        try:
            return json.load(fh)
        except json.JSONDecodeError as exc:
            raise ValueError(f"Invalid attribute JSON: {attribute_file}") from exc


def select_attribute_sources(attribute_file: str | dict[str, str], data_source: str) -> dict[str, dict]:
    """One JSON path (it then describes ``data_source``) or a {source: path} mapping."""
    if isinstance(attribute_file, str):
        return {data_source: load_attribute_json(attribute_file)}
    if isinstance(attribute_file, dict):
        return {src: load_attribute_json(path) for src, path in attribute_file.items()}
    raise ValueError("regularized_attributes.attribute_file must be a string or mapping from source to file.")


def normalize_attributes(attributes: dict[str, float], normalize_cfg: dict[str, Any] | None) -> dict[str, float]:
    if not normalize_cfg or not normalize_cfg.get("enabled", False):
        return attributes
    divisor = float(normalize_cfg.get("divisor", 1.0))
    if divisor == 0:
        raise ValueError("Normalization divisor must be non-zero.")
    return {k: float(v) / divisor for k, v in attributes.items()}


def source_of_path(path: str) -> str:
    if "edente" in path:
        return "edente"
    if "dente" in path:
        return "dente"
    raise ValueError(f"Cannot identify data source from path: {path}")


def filter_attributes_for_paths(paths: list[str], attribute_sources: dict[str, dict], attribute_latent_mapping: dict[str, Any],
                                normalize_cfg: dict[str, Any] | None) -> list[dict[str, float]]:
    """The mapped attributes of every image, in ``paths`` order."""
    wanted = list(attribute_latent_mapping)
    out = []
    for path in paths:
        base = os.path.basename(path)
        src = source_of_path(path)
        entry = attribute_sources.get(src, {}).get(base)
        if entry is None:
            raise FileNotFoundError(f"Attribute entry missing for {base} in source {src}")
        missing = {k for k in wanted if k not in entry}
        if missing:
            raise KeyError(f"Missing attributes for {base}: {missing}")
        out.append(normalize_attributes({k: float(entry[k]) for k in wanted}, normalize_cfg))
    return out


def attribute_mapping(regularized_attributes: dict | None) -> dict[str, Any]:
    """``attribute_latent_mapping`` without the ``_comment``-style keys (train_vae.py:376-377)."""
    raw = (regularized_attributes or {}).get("attribute_latent_mapping", {})
    return {k: v for k, v in raw.items() if not str(k).startswith("_")}


def attributes_for_paths(paths: list[str], regularized_attributes: dict | None, data_source: str) -> list[dict[str, float]]:
    """What ``create_vae_dataloaders(ar_vae_enabled=True, ...)`` attaches to the image list (dataloaders.py:432-465)."""
    if regularized_attributes is None:
        raise ValueError("AR-VAE enabled but regularized_attributes config is missing.")
    mapping = attribute_mapping(regularized_attributes)
    if not mapping:
        raise ValueError("attribute_latent_mapping must be provided when AR-VAE is enabled.")
    sources = select_attribute_sources(regularized_attributes.get("attribute_file"), data_source)
    if data_source != "both" and sources.get(data_source) is None:
        raise ValueError(f"No attribute mapping found for source {data_source}")
    return filter_attributes_for_paths(paths, sources, mapping, regularized_attributes.get("normalize_attributes"))


def collate_with_attributes(batch: list[tuple[torch.Tensor, dict[str, float]]]) -> tuple[torch.Tensor, dict[str, torch.Tensor]]:
    """[(image, {name: value})] -> (stacked images, {name: float32 [b]})."""
    images = torch.stack([img for img, _ in batch], dim=0)
    names = batch[0][1].keys()
    return images, {k: torch.tensor([float(a[k]) for _, a in batch], dtype=torch.float32) for k in names}
