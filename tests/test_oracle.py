"""CPU tests of the oracle (test infrastructure) against the committed golden vectors.

* losses: oracle/losses.py vs tests/golden/losses_golden.json, which was produced by importing the
  REFERENCE's src/pti_ldm_vae/models/losses.py by path (oracle/make_golden.py) -> pinned;
  also the known answers KA1-KA5 of SURVEY.md §8c.
* encoder/decoder: oracle/autoencoderkl.py vs its own frozen outputs (model_golden_*.npz) and the
  structural facts of SURVEY.md Appendix A (parameter totals, key names) -> "parity unpinned" with
  respect to MONAI, regression-pinned with respect to this repo.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle.autoencoderkl import CONFIG_A, CONFIG_AR, AutoencoderKLOracle, build_oracle, synthetic_images
from oracle.losses import ar_vae_loss, kl_loss, total_loss, train_step_losses

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases(kind):
    with open(os.path.join(GOLD, "losses_golden.json")) as f:
        return [c for c in json.load(f)["cases"] if c["kind"] == kind]


@pytest.mark.parametrize("case", _cases("kl"), ids=lambda c: f"seed{c['seed']}")
def test_kl_matches_reference(case):
    g = torch.Generator().manual_seed(case["seed"])
    mu = torch.randn(*case["shape"], generator=g)
    t = torch.randn(*case["shape"], generator=g) * case["scale"]
    sig = torch.exp(0.5 * t)
    assert float(kl_loss(mu, t)) == pytest.approx(case["kl_logvar"], rel=1e-6)
    assert float(kl_loss(mu, sig, input_is_logvar=False)) == pytest.approx(case["kl_sigma_flag"], rel=1e-6)
    assert float(kl_loss(mu, sig)) == pytest.approx(case["kl_sigma_as_logvar"], rel=1e-6)


def test_known_answers_survey_8c():
    g = torch.Generator().manual_seed(1234)
    mu = torch.randn(2, 4, 32, 32, generator=g)
    t = torch.randn(2, 4, 32, 32, generator=g) * 0.1
    assert float(kl_loss(mu, t)) == pytest.approx(2086.13623046875, rel=1e-6)          # KA1
    assert float(kl_loss(mu, torch.exp(0.5 * t))) == pytest.approx(3560.2001953125, rel=1e-6)  # KA2
    tl = total_loss(*[torch.tensor(v) for v in (0.25, 133.0, 0.5, 0.0, 0.7)], kl_weight=1e-3, perceptual_weight=1.0,
                    adv_weight=3.0, ar_gamma=0.5, ar_vae_enabled=True)
    assert float(tl) == pytest.approx(1.2330000400543213, rel=1e-6)                      # KA3


def test_total_loss_matches_reference():
    c = _cases("total")[0]
    t = [torch.tensor(v) for v in c["vals"]]
    assert float(total_loss(*t, ar_vae_enabled=True, **c["args"])) == pytest.approx(c["enabled"], rel=1e-6)
    assert float(total_loss(*t, ar_vae_enabled=False, **c["args"])) == pytest.approx(c["disabled"], rel=1e-6)


@pytest.mark.parametrize("case", _cases("ar"), ids=lambda c: f"seed{c['seed']}")
def test_ar_loss_matches_reference(case):
    g = torch.Generator().manual_seed(case["seed"])
    z = torch.randn(*case["zshape"], generator=g)
    attrs = {k: torch.tensor(v) for k, v in case["attrs"].items()}
    tot, per, cnt, dl = ar_vae_loss(z, attrs, case["mapping"], "all", None, case["delta_global"])
    assert float(tot) == pytest.approx(case["total"], rel=1e-5)
    for k in case["per_attr"]:
        assert float(per[k]) == pytest.approx(case["per_attr"][k], rel=1e-5, abs=1e-7)
        assert cnt[k] == case["pairs"][k]
        assert dl[k] == case["deltas"][k]


def test_ar_loss_errors():
    z = torch.zeros(2, 4)
    with pytest.raises(ValueError):
        ar_vae_loss(z, {"a": torch.zeros(2)}, {"a": {"latent_channel": 7, "delta": 1.0}}, "all", None, None)
    with pytest.raises(KeyError):
        ar_vae_loss(z, {}, {"a": {"latent_channel": 0, "delta": 1.0}}, "all", None, None)
    with pytest.raises(ValueError):
        ar_vae_loss(z, {"a": torch.zeros(2)}, {"a": {"latent_channel": 0}}, "all", None, None)
    with pytest.raises(ValueError):
        ar_vae_loss(z, {}, {}, "bogus", None, None)
    with pytest.raises(ValueError):
        ar_vae_loss(torch.zeros(2, 3, 4), {}, {}, "all", None, None)


@pytest.mark.parametrize("cfg,total", [(CONFIG_A, 4_562_593), (CONFIG_AR, 12_324_885)])
def test_parameter_totals(cfg, total):
    m = AutoencoderKLOracle(**cfg)
    assert sum(p.numel() for p in m.parameters()) == total


def test_state_dict_key_map_appendix_a3():
    sd = AutoencoderKLOracle(**CONFIG_A).state_dict()
    expect = {
        "encoder.blocks.0.conv.weight": (32, 1, 3, 3),
        "encoder.blocks.1.norm1.weight": (32,),
        "encoder.blocks.1.conv1.conv.weight": (32, 32, 3, 3),
        "encoder.blocks.3.conv.conv.weight": (32, 32, 3, 3),          # AEKLDownsample: double "conv"
        "encoder.blocks.4.nin_shortcut.conv.weight": (64, 32, 1, 1),
        "encoder.blocks.13.norm.weight": (128,),
        "encoder.blocks.13.attn.to_q.weight": (128, 128),
        "encoder.blocks.13.attn.out_proj.bias": (128,),
        "encoder.blocks.15.weight": (128,),                           # bare final GroupNorm
        "encoder.blocks.16.conv.weight": (4, 128, 3, 3),
        "decoder.blocks.0.conv.weight": (128, 4, 3, 3),
        "decoder.blocks.6.postconv.conv.weight": (128, 128, 3, 3),
        "decoder.blocks.10.nin_shortcut.conv.bias": (64,),
        "decoder.blocks.16.conv.weight": (1, 32, 3, 3),
        "quant_conv_mu.conv.weight": (4, 4, 1, 1),
        "quant_conv_log_sigma.conv.bias": (4,),
        "post_quant_conv.conv.weight": (4, 4, 1, 1),
    }
    for k, shp in expect.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == shp, (k, sd[k].shape)
    assert "encoder.blocks.1.nin_shortcut.conv.weight" not in sd   # Identity has no params
    assert not any(k.startswith("autoencoder.") for k in sd)


def test_ctor_validation():
    with pytest.raises(ValueError):
        AutoencoderKLOracle(**{**CONFIG_A, "norm_num_groups": 24})
    with pytest.raises(ValueError):
        AutoencoderKLOracle(**{**CONFIG_A, "attention_levels": [False, False]})


@pytest.mark.parametrize("tag,cfg,batch", [("A64", CONFIG_A, 2), ("AR64", CONFIG_AR, 1)])
def test_oracle_matches_frozen_golden(tag, cfg, batch):
    torch.set_num_threads(4)
    g = np.load(os.path.join(GOLD, f"model_golden_{tag}.npz"))
    model = build_oracle(cfg, seed=42)
    assert sum(p.numel() for p in model.parameters()) == int(g["n_params"])
    x = synthetic_images(batch, 1, 64, seed=42)
    assert float(x.double().sum()) == pytest.approx(float(g["x_sum"]), rel=1e-9)
    lat = 64 // (2 ** (len(cfg["channels"]) - 1))
    eps = torch.randn(batch, cfg["latent_channels"], lat, lat, generator=torch.Generator().manual_seed(43))
    loss, recons, kl, (rec, mu, sig) = train_step_losses(model, x, eps)
    loss.backward()
    assert (sig > 0).all()                       # third output is sigma
    np.testing.assert_allclose(mu.detach().numpy(), g["mu"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sig.detach().numpy(), g["sigma"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rec.detach().numpy(), g["recon"], rtol=1e-3, atol=2e-5)
    assert float(loss) == pytest.approx(float(g["loss"]), rel=1e-5)
    assert float(kl) == pytest.approx(float(g["kl"]), rel=1e-5)
    gn = np.array([float(p.grad.norm()) for _, p in model.named_parameters()])
    np.testing.assert_allclose(gn, g["grad_norms"], rtol=2e-3, atol=1e-7)
    with torch.no_grad():
        np.testing.assert_allclose(model.reconstruct(x).numpy(), g["recon_det"], rtol=1e-3, atol=2e-5)
        rec0, mu0, _ = model(x, torch.zeros_like(eps))
        np.testing.assert_allclose(rec0.numpy(), g["recon_det"], rtol=1e-3, atol=2e-5)  # eps=0 == deterministic
