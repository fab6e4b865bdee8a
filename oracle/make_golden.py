"""TEST INFRASTRUCTURE ONLY — generates tests/golden/*.  Run in the BUILD container:

    python oracle/make_golden.py

Part 1 (pinned by the reference itself): imports the reference's
``src/pti_ldm_vae/models/losses.py`` and ``src/pti_ldm_vae/utils/losses.py`` BY FILE
PATH from /root/reference (their only import is torch), feeds them seeded inputs and
stores inputs' seeds + outputs in ``tests/golden/losses_golden.json``.  The reference
sources are never copied; only input/output numbers are stored.

Part 2 (parity unpinned — MONAI absent): runs this repo's own CPU fp32 oracle
(oracle/autoencoderkl.py) on seeded weights/inputs/eps and stores mu / sigma /
reconstruction / loss scalars / per-parameter gradient norms in
``tests/golden/model_golden_*.npz`` so later refactors of the oracle and the HIP path
are checked against a frozen vector.

Part 3 (parity unpinned, same reason): the oracle PatchDiscriminator -> ``tests/golden/disc_golden.npz``.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def losses_golden():
    ref = _load_by_path("ref_losses", os.path.join(REF, "src/pti_ldm_vae/models/losses.py"))
    refu = _load_by_path("ref_ulosses", os.path.join(REF, "src/pti_ldm_vae/utils/losses.py"))
    out = {"source": "reference src/pti_ldm_vae/models/losses.py imported by path", "cases": []}

    # KL cases (SURVEY.md §8c KA1/KA2 + two more shapes)
    for seed, shape, scale in [(1234, (2, 4, 32, 32), 0.1), (5, (3, 10, 8, 8), 0.5), (9, (4, 4), 1.0)]:
        g = torch.Generator().manual_seed(seed)
        mu = torch.randn(*shape, generator=g)
        t = torch.randn(*shape, generator=g) * scale
        sig = torch.exp(0.5 * t)
        out["cases"].append({
            "kind": "kl", "seed": seed, "shape": list(shape), "scale": scale,
            "kl_logvar": float(ref.compute_kl_loss(mu, t)),
            "kl_sigma_flag": float(ref.compute_kl_loss(mu, sig, input_is_logvar=False)),
            "kl_sigma_as_logvar": float(ref.compute_kl_loss(mu, sig)),
        })
    # total loss (KA3)
    args = dict(kl_weight=1e-3, perceptual_weight=1.0, adv_weight=3.0, ar_gamma=0.5)
    vals = [0.25, 133.0, 0.5, 0.0, 0.7]
    tens = [torch.tensor(v) for v in vals]
    out["cases"].append({
        "kind": "total", "vals": vals, "args": args,
        "enabled": float(ref.compute_total_loss(*tens, ar_vae_enabled=True, **args)),
        "disabled": float(ref.compute_total_loss(*tens, ar_vae_enabled=False, **args)),
    })
    # AR-VAE loss (KA4 + a b=8 case with 6 attributes like ar_vae_dente_kl1e3.json:80-88)
    g = torch.Generator().manual_seed(7)
    z = torch.randn(4, 10, 8, 8, generator=g)
    attrs = {"height_0": torch.tensor([1., 3., 2., 3.]), "width_0": torch.tensor([5., 5., 5., 5.]),
             "width_1": torch.tensor([.1, .4, .2, .3])}
    mapping = {"height_0": {"latent_channel": 0, "delta": 1.0}, "width_0": {"latent_channel": 1, "delta": 1.0},
               "width_1": {"latent_channel": 2}}
    tot, per, cnt, dl = ref.compute_ar_vae_loss(z, attrs, mapping, "all", None, {"enabled": True, "value": 2.0})
    out["cases"].append({"kind": "ar", "seed": 7, "zshape": [4, 10, 8, 8],
                         "attrs": {k: v.tolist() for k, v in attrs.items()}, "mapping": mapping,
                         "delta_global": {"enabled": True, "value": 2.0}, "total": float(tot),
                         "per_attr": {k: float(v) for k, v in per.items()}, "pairs": cnt, "deltas": dl})
    g = torch.Generator().manual_seed(11)
    z = torch.randn(8, 10, generator=g)
    names = ["height_0", "width_0", "width_1", "width_2", "width_3", "width_4"]
    attrs = {n: torch.rand(8, generator=g) for n in names}
    attrs["width_2"][3] = attrs["width_2"][5]  # one tie
    mapping = {n: {"latent_channel": i, "delta": 1.0} for i, n in enumerate(names)}
    tot, per, cnt, dl = ref.compute_ar_vae_loss(z, attrs, mapping, "all", None, {"enabled": True, "value": 1.0})
    out["cases"].append({"kind": "ar", "seed": 11, "zshape": [8, 10],
                         "attrs": {k: v.tolist() for k, v in attrs.items()}, "mapping": mapping,
                         "delta_global": {"enabled": True, "value": 1.0}, "total": float(tot),
                         "per_attr": {k: float(v) for k, v in per.items()}, "pairs": cnt, "deltas": dl})
    # ensure_three_channels (KA5)
    x = torch.arange(8.).reshape(2, 1, 2, 2)
    y = refu.ensure_three_channels(x)
    out["cases"].append({"kind": "three", "in_shape": list(x.shape), "out_shape": list(y.shape),
                         "equal_channels": bool((y[:, 0] == y[:, 1]).all() and (y[:, 1] == y[:, 2]).all())})
    with open(os.path.join(GOLD, "losses_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote losses_golden.json", len(out["cases"]), "cases")


def model_golden():
    from oracle.autoencoderkl import CONFIG_A, CONFIG_AR, build_oracle, synthetic_images
    from oracle.losses import train_step_losses

    torch.set_num_threads(8)
    for tag, cfg, batch, size in [("A64", CONFIG_A, 2, 64), ("AR64", CONFIG_AR, 1, 64)]:
        model = build_oracle(cfg, seed=42)
        x = synthetic_images(batch, cfg["in_channels"], size, seed=42)
        lat = size // (2 ** (len(cfg["channels"]) - 1))
        eps = torch.randn(batch, cfg["latent_channels"], lat, lat, generator=torch.Generator().manual_seed(43))
        loss, recons, kl, (rec, mu, sig) = train_step_losses(model, x, eps)
        loss.backward()
        names = [n for n, _ in model.named_parameters()]
        gnorm = np.array([float(p.grad.norm()) for _, p in model.named_parameters()], dtype=np.float64)
        with torch.no_grad():
            det = model.reconstruct(x)
        np.savez_compressed(
            os.path.join(GOLD, f"model_golden_{tag}.npz"),
            mu=mu.detach().numpy(), sigma=sig.detach().numpy(), recon=rec.detach().numpy(),
            recon_det=det.numpy(), loss=np.float64(loss.item()), recons=np.float64(recons.item()),
            kl=np.float64(kl.item()), grad_norms=gnorm, param_names=np.array(names),
            n_params=np.int64(sum(p.numel() for p in model.parameters())),
            x_sum=np.float64(x.double().sum().item()), eps_sum=np.float64(eps.double().sum().item()))
        print(tag, "loss", loss.item(), "recons", recons.item(), "kl", kl.item(),
              "params", sum(p.numel() for p in model.parameters()))


def discriminator_golden():
    """Part 3 (parity unpinned -- MONAI absent): the oracle PatchDiscriminator (oracle/patch_discriminator.py) on a seeded
    state (weights at 5x the initialisation scale, as the GPU tests use) and a seeded 96x96 input: logits, the three
    least-squares losses, gradient of the generator term w.r.t. the input, per-parameter gradient norms of the
    discriminator loss -> tests/golden/disc_golden.npz (state and input are regenerated from the seeds by the tests)."""
    from oracle.patch_discriminator import PatchDiscriminator, patch_adversarial_loss as pal
    torch.manual_seed(2024)
    ref = PatchDiscriminator()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(5.0)
    g = torch.Generator().manual_seed(2025)
    x = torch.randn(2, 1, 96, 96, generator=g) * 0.8
    real = torch.randn(2, 1, 96, 96, generator=g) * 0.8 + 0.2
    xg = x.clone().requires_grad_(True)
    logits = ref(xg)[-1]
    gen = pal(logits, True, False)
    dx, = torch.autograd.grad(gen, xg)
    ref.zero_grad(set_to_none=True)
    lf, lr = pal(ref(x)[-1], False, True), pal(ref(real)[-1], True, True)
    (0.5 * (lf + lr)).backward()
    np.savez_compressed(os.path.join(GOLD, "disc_golden.npz"), logits=logits.detach().numpy(), gen=float(gen), fake=float(lf),
                        real=float(lr), dx=dx.numpy(),
                        **{"gnorm_" + n.replace(".", "_"): float(p.grad.norm()) for n, p in ref.named_parameters()})
    print("wrote disc_golden.npz")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if os.path.isdir(REF):
        losses_golden()
    else:
        print("reference absent: losses_golden.json not regenerated")
    model_golden()
    discriminator_golden()
