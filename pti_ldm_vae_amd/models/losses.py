"""Loss step of the VAE training hot path — same API as the reference's
``src/pti_ldm_vae/models/losses.py`` (``compute_kl_loss`` :4-30, ``compute_total_loss`` :33-66,
``compute_ar_vae_loss`` :69-166), re-implemented for the GPU:

* ``compute_kl_loss`` / ``compute_total_loss``: same formulas, device-agnostic tensor code.
* ``compute_ar_vae_loss``: the reference's O(b^2) Python pair list + one ``.item()`` sync per
  attribute becomes one broadcasted [b,b] difference per attribute and a single host sync for all
  pair counts; the pair set (i != j, ties dropped) and the ``"subset"`` sampling (Python
  ``random.sample`` over the same ordered pair list) are identical.
* ``fused_recon_kl_loss``: ``recon + kl_weight * kl`` with both terms, their mean reductions and the
  gradient seeds computed by ONE HIP kernel (``pti_vae_loss``) for the native training loop.
"""
from __future__ import annotations

import random

import torch


def compute_kl_loss(z_mu: torch.Tensor, z_logvar: torch.Tensor, *, input_is_logvar: bool = True) -> torch.Tensor:
    """KL(q||N(0,1)) summed over latent dims, averaged over the batch (reference losses.py:25-30)."""
    s = z_logvar if input_is_logvar else torch.log(z_logvar.pow(2) + 1e-8)
    dims = list(range(1, s.dim()))
    return (-0.5 * torch.sum(1 + s - z_mu.pow(2) - torch.exp(s), dim=dims)).mean()


def compute_total_loss(recons_loss, kl_loss, perceptual_loss, adv_gen_loss, ar_loss, *, kl_weight: float,
                       perceptual_weight: float, adv_weight: float, ar_gamma: float, ar_vae_enabled: bool):
    """Reference losses.py:62-66."""
    total = recons_loss + kl_weight * kl_loss + perceptual_weight * perceptual_loss + adv_weight * adv_gen_loss
    return total + ar_gamma * ar_loss if ar_vae_enabled else total


def compute_ar_vae_loss(latent_vectors, attributes, attribute_latent_mapping, pairwise_mode, subset_pairs,
                        delta_global):
    """Attribute-regularised VAE loss (reference losses.py:69-166), vectorised."""
    if latent_vectors.dim() == 4:
        latent_vectors = latent_vectors.mean(dim=(2, 3))
    elif latent_vectors.dim() != 2:
        raise ValueError(f"Expected latent shape [B, C] or [B, C, H, W], got {latent_vectors.shape}")
    b, latent_dim = latent_vectors.shape
    if pairwise_mode not in {"all", "subset"}:
        raise ValueError(f"pairwise must be 'all' or 'subset', got {pairwise_mode}")
    if pairwise_mode == "subset" and (subset_pairs is None or subset_pairs <= 0):
        raise ValueError("subset_pairs must be a positive integer when pairwise='subset'")
    dev = latent_vectors.device
    total = torch.zeros((), device=dev)
    per_attr, deltas, count_t = {}, {}, {}
    for name, mapping in attribute_latent_mapping.items():
        ch = int(mapping["latent_channel"])
        if ch >= latent_dim:
            raise ValueError(f"Latent channel {ch} for attribute {name} exceeds latent size {latent_dim}")
        a = attributes.get(name)
        if a is None:
            raise KeyError(f"Missing attribute values for {name} in batch.")
        a = a.to(dev)
        delta = mapping.get("delta")
        if delta is None and delta_global and delta_global.get("enabled", False):
            delta = delta_global.get("value")
        if delta is None:
            raise ValueError(f"Delta not provided for {name} and no delta_global fallback.")
        deltas[name] = float(delta)
        z = latent_vectors[:, ch]
        order = torch.sign(a[None, :] - a[:, None])          # [i, j] = sign(a_j - a_i); diagonal is 0
        sel = order != 0
        if pairwise_mode == "subset":
            pairs = [(i, j) for i in range(b) for j in range(b) if i != j]
            pairs = random.sample(pairs, min(len(pairs), int(subset_pairs)))
            keep = torch.zeros(b, b, dtype=torch.bool, device=dev)
            if pairs:
                idx = torch.tensor(pairs, device=dev)
                keep[idx[:, 0], idx[:, 1]] = True
            sel = sel & keep
        pred = torch.tanh(float(delta) * (z[None, :] - z[:, None]))
        cnt = sel.sum()
        la = (((pred - order) ** 2) * sel).sum() / cnt.clamp(min=1)
        per_attr[name], count_t[name] = la, cnt
        total = total + la
    counts = {}
    if count_t:
        vals = torch.stack(list(count_t.values())).tolist()    # the one host sync
        counts = {k: int(v) for k, v in zip(count_t, vals)}
    return total, per_attr, counts, deltas


class _FusedReconKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, recon, images, z_mu, z_third, l2, third_mode, kl_weight):
        from .. import ops
        out = torch.zeros(2, dtype=torch.float32, device=recon.device)
        d_recon, d_mu, d_third = torch.empty_like(recon), torch.empty_like(z_mu), torch.empty_like(z_third)
        ops.vae_loss(recon.contiguous(), images.contiguous(), z_mu.contiguous(), z_third.contiguous(), out, d_recon, d_mu,
                     d_third, l2=l2, third_mode=third_mode, kl_weight=kl_weight)
        ctx.save_for_backward(d_recon, d_mu, d_third)
        total = out[0] + kl_weight * out[1]
        return total, out[0], out[1]

    @staticmethod
    def backward(ctx, g_total, g_recon, g_kl):
        d_recon, d_mu, d_third = ctx.saved_tensors
        # the seeds are d(total)/d(.) already; only the total is a differentiable output
        return d_recon * g_total, None, d_mu * g_total, d_third * g_total, None, None, None


def fused_recon_kl_loss(reconstruction, images, z_mu, z_third, *, recon_loss="l1", kl_weight=1e-3,
                        input_is_logvar=True):
    """-> (recon + kl_weight*kl, recon, kl) with one HIP kernel; gradients flow through the first
    output only (the other two are detached views for logging).  Mirrors train_vae.py:393-394,419-430
    with perceptual / adversarial / AR weights contributing zero."""
    total, r, k = _FusedReconKL.apply(reconstruction, images, z_mu, z_third, recon_loss == "l2",
                                      0 if input_is_logvar else 1, float(kl_weight))
    return total, r.detach(), k.detach()
