"""JSON run configuration with the subset of ``monai.bundle.ConfigParser`` semantics the reference
relies on (``vae_scripts/train_vae.py:100-124``, ``src/pti_ldm_vae/utils/vae_loader.py:11-24``):

* a string value that is EXACTLY ``"@id"`` is replaced by the value of the top-level key ``id``
  (nested ids with ``::`` or ``#`` separators are followed);
* anything else — including the dotted ``"@regularized_attributes.gamma"`` forms in the shipped
  configs (``config/vae_dente_no_adv.json:113,116``), which MONAI does not resolve either — stays a
  literal string and is handled by the script-side fall-backs (``resolve_bool``, gamma fall-back);
* ``_comment``-style keys are inert data.
"""
from __future__ import annotations

import json
import re
from types import SimpleNamespace
from typing import Any

_REF = re.compile(r"@(\w+(?:(?:::|#)\w+)*)")


def _lookup(root: dict, ref_id: str):
    node: Any = root
    for part in re.split(r"::|#", ref_id):
        if isinstance(node, list):
            node = node[int(part)]
        else:
            if part not in node:
                raise KeyError(f"can not find expected ID '{ref_id}' in the references.")
            node = node[part]
    return node


def _resolve(node, root, depth=0):
    if depth > 32:
        raise ValueError("circular @reference in config")
    if isinstance(node, dict):
        return {k: _resolve(v, root, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    if isinstance(node, str):
        m = _REF.fullmatch(node)
        if m:
            return _resolve(_lookup(root, m.group(1)), root, depth + 1)
    return node


def parse_config(raw: dict) -> dict:
    """Resolve ``"@id"`` references of an already-loaded config dict."""
    return _resolve(raw, raw)


def read_config(path: str) -> dict:
    with open(path, encoding="utf-8") as f:
        return parse_config(json.load(f))


def load_vae_config(config_file: str) -> SimpleNamespace:
    """Reference ``utils/vae_loader.py:11-24``."""
    return SimpleNamespace(**read_config(config_file))


def resolve_bool(value: Any) -> bool:
    """Reference ``train_vae.py:246-259``: strings like "false"/"true"; unknown strings -> False."""
    if isinstance(value, bool):
        return value
    if isinstance(value, str):
        return value.strip().lower() in {"true", "1", "yes", "y"}
    if value is None:
        return False
    return bool(value)


def resolve_ar_settings(autoencoder_train: dict, regularized_attributes: dict | None):
    """AR-VAE enable flag / gamma with the reference's string fall-backs (train_vae.py:776-792)."""
    ra = regularized_attributes or {}
    enabled = resolve_bool(autoencoder_train.get("ar_vae_enabled", False)) or resolve_bool(ra.get("enabled", False))
    raw = autoencoder_train.get("ar_vae_weight", ra.get("gamma", 0.0))
    if isinstance(raw, str):
        try:
            gamma = float(raw)
        except ValueError:
            gamma = float(ra.get("gamma", 0.0))
    else:
        gamma = float(raw)
    return enabled, gamma, ra.get("pairwise", "all"), ra.get("subset_pairs")
