"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference loss functions.

Follows ``src/pti_ldm_vae/models/losses.py`` of the reference:
  * compute_kl_loss      losses.py:4-30
  * compute_total_loss   losses.py:33-66
  * compute_ar_vae_loss  losses.py:69-166  (pair order of :132, tie mask of :147-159)

Pinned: ``oracle/make_golden.py`` ran the reference file itself (imported by path,
in the build container only) and stored its outputs in
``tests/golden/losses_golden.json``; ``tests/test_oracle.py`` checks this
restatement against those vectors and against SURVEY.md §8c KA1-KA4.
"""
from __future__ import annotations

import random

import torch


def kl_loss(z_mu: torch.Tensor, z_third: torch.Tensor, *, input_is_logvar: bool = True) -> torch.Tensor:
    """losses.py:25-30.  ``z_third`` is whatever the model's third output is; the
    reference call site (train_vae.py:394) passes MONAI's sigma with the default
    ``input_is_logvar=True``."""
    if not input_is_logvar:
        z_third = torch.log(z_third.pow(2) + 1e-8)
    dims = list(range(1, z_third.dim()))
    kl = -0.5 * torch.sum(1 + z_third - z_mu.pow(2) - torch.exp(z_third), dim=dims)
    return kl.mean()


def total_loss(recons, kl, perceptual, adv_gen, ar, *, kl_weight, perceptual_weight, adv_weight,
               ar_gamma, ar_vae_enabled):
    """losses.py:62-66."""
    total = recons + kl_weight * kl + perceptual_weight * perceptual + adv_weight * adv_gen
    if ar_vae_enabled:
        total = total + ar_gamma * ar
    return total


def ar_vae_loss(latent_vectors, attributes, attribute_latent_mapping, pairwise_mode, subset_pairs,
                delta_global):
    """losses.py:89-166, with the O(b^2) Python pair list of :132 kept as the definition."""
    if latent_vectors.dim() == 4:
        latent_vectors = latent_vectors.mean(dim=(2, 3))
    elif latent_vectors.dim() != 2:
        raise ValueError(f"Expected latent shape [B, C] or [B, C, H, W], got {latent_vectors.shape}")
    b, latent_dim = latent_vectors.shape
    if pairwise_mode not in {"all", "subset"}:
        raise ValueError(f"pairwise must be 'all' or 'subset', got {pairwise_mode}")
    if pairwise_mode == "subset" and (subset_pairs is None or subset_pairs <= 0):
        raise ValueError("subset_pairs must be a positive integer when pairwise='subset'")
    total = torch.tensor(0.0)
    per_attr, counts, deltas = {}, {}, {}
    for name, mapping in attribute_latent_mapping.items():
        ch = int(mapping["latent_channel"])
        if ch >= latent_dim:
            raise ValueError(f"Latent channel {ch} for attribute {name} exceeds latent size {latent_dim}")
        a = attributes.get(name)
        if a is None:
            raise KeyError(f"Missing attribute values for {name} in batch.")
        delta = mapping.get("delta")
        if delta is None and delta_global and delta_global.get("enabled", False):
            delta = delta_global.get("value")
        if delta is None:
            raise ValueError(f"Delta not provided for {name} and no delta_global fallback.")
        z = latent_vectors[:, ch]
        pairs = [(i, j) for i in range(b) for j in range(b) if i != j]
        if pairwise_mode == "subset":
            pairs = random.sample(pairs, min(len(pairs), int(subset_pairs)))
        if not pairs:
            per_attr[name], counts[name], deltas[name] = torch.tensor(0.0), 0, float(delta)
            continue
        ii = torch.tensor([p[0] for p in pairs])
        jj = torch.tensor([p[1] for p in pairs])
        order = torch.sign(a[jj] - a[ii])
        mask = order != 0
        if not torch.any(mask):
            per_attr[name], counts[name], deltas[name] = torch.tensor(0.0), 0, float(delta)
            continue
        pred = torch.tanh(float(delta) * (z[jj] - z[ii])[mask])
        la = torch.mean((pred - order[mask]) ** 2)
        per_attr[name], counts[name], deltas[name] = la, int(mask.sum().item()), float(delta)
        total = total + la
    return total, per_attr, counts, deltas


def train_step_losses(model, images, eps, *, recon_loss="l1", kl_weight=1e-3,
                      third_output="sigma"):
    """One reference training step's loss graph, train_vae.py:385-430, with the
    perceptual and adversarial terms omitted (unavailable offline / inactive at
    epoch 0).  Returns (loss_g, recon, kl, (reconstruction, z_mu, z_third))."""
    reconstruction, z_mu, z_sigma = model(images, eps)
    z_third = z_sigma if third_output == "sigma" else 2.0 * torch.log(z_sigma)
    if recon_loss == "l2":
        recons = torch.nn.functional.mse_loss(reconstruction, images)
    else:
        recons = torch.nn.functional.l1_loss(reconstruction, images)
    kl = kl_loss(z_mu, z_third)
    zero = torch.tensor(0.0)
    loss_g = total_loss(recons, kl, zero, zero, zero, kl_weight=kl_weight, perceptual_weight=0.0,
                        adv_weight=0.0, ar_gamma=0.0, ar_vae_enabled=False)
    return loss_g, recons, kl, (reconstruction, z_mu, z_third)
